// pika_amd/csrc/rnnt_loss.hip -- RNN-T loss for gfx950 (MI355X), hand-written HIP.
//
// Replaces the third-party `warp_rnnt` CUDA loss the reference calls at
//   /root/reference/trainer/train_transducer_bmuf_otfaug.py:58,97-99
// (SURVEY.md 8a row 10).  This is a from-scratch CDNA4 design, not a hipify:
//
//   gather   : one thread per lattice cell pulls the TWO log-probs the cell needs
//              (blank, next label) out of the (B,T,U1,V) tensor into two compact
//              planes stored SKEWED: element (t,u) lives at [t+u][u], so every
//              anti-diagonal of the lattice is one contiguous 256-byte row.
//   alpha/beta: one 64-lane wavefront per utterance and direction walks the
//              anti-diagonals; lane = u, the neighbour term moves one lane with a
//              single DPP wave-shift (no LDS, no barrier), log-probs for the next
//              8 diagonals are prefetched into registers while the current 8 are
//              consumed (the recurrence is latency-bound, T+U-1 dependent steps).
//              U1 > 64 falls back to one workgroup of pad64(U1) threads with an
//              LDS exchange per diagonal.
//   grad     : the HBM-bound part -- one streaming pass that writes the dense
//              (B,T,U1,V) gradient exactly once: a wave owns whole V-rows, 16-byte
//              stores, the two non-zeros of a row are blended into the zero
//              stream in registers (no memset + scatter, no read of log_probs).
//
// Algorithmic HBM bytes per utterance (T=1000,U1=51,V=5000): 1.020 GB gradient
// write + ~1.2 MB lattice traffic; see DESIGN.md.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pika_rnnt.h"

namespace {

constexpr float NEG = -1.0e30f;  // "log zero": finite so NEG+NEG and NEG-NEG never make NaN
constexpr int WAVE = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

typedef float v4f __attribute__((ext_vector_type(4)));

// Width (in lanes) of one skewed lattice row = threads of the alpha/beta workgroup:
// the smallest instantiated wave count that covers U1 label columns.
inline int lattice_width(int U1) {
    static const int kWaves[] = {1, 2, 3, 4, 6, 8, 12, 16};
    for (int nw : kWaves)
        if (nw * 64 >= U1) return nw * 64;
    return 0;
}

struct Lattice {
    float *lpb;    // [B][D][Wp] blank log-prob of cell (t,u) at row t+u, col u
    float *lpe;    // [B][D][Wp] log-prob of emitting y_{u+1} from cell (t,u)
    float *alpha;  // [B][D][Wp]
    float *beta;   // [B][D][Wp]
    float *ll;     // [B] log-likelihood from beta[0,0]
    float *ll_a;   // [B] log-likelihood from the alpha side (diagnostic)
    int Wp, D;
};

inline size_t plane_elems(int B, int T, int U1) {
    return (size_t)B * (size_t)(T + U1 - 1) * (size_t)lattice_width(U1);
}

inline Lattice carve(void *ws, int B, int T, int U1) {
    Lattice L;
    const size_t n = plane_elems(B, T, U1);
    float *p = static_cast<float *>(ws);
    L.lpb = p;
    L.lpe = p + n;
    L.alpha = p + 2 * n;
    L.beta = p + 3 * n;
    L.ll = p + 4 * n;
    L.ll_a = L.ll + B;
    L.Wp = lattice_width(U1);
    L.D = T + U1 - 1;
    return L;
}

__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// log(exp(x)+exp(y)) on the transcendental pipe (v_exp_f32 / v_log_f32).
__device__ inline float lae(float x, float y) {
    const float m = fmaxf(x, y);
    const float d = fminf(x, y) - m;  // <= 0
    return m + LN2 * __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(d * LOG2E));
}

// lane i <- lane i-1 across the whole 64-lane wave (DPP wave_shr:1); lane 0 <- fill.
__device__ inline float wave_shr1(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v),
                                                      0x138, 0xf, 0xf, false));
}
// lane i <- lane i+1 (DPP wave_shl:1); lane 63 <- fill.
__device__ inline float wave_shl1(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v),
                                                      0x130, 0xf, 0xf, false));
}

// ---------------------------------------------------------------------------------------------
// gather: (B,T,U1,V) -> two skewed planes
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rnnt_gather_kernel(
    const float *__restrict__ lp, const int *__restrict__ labels, const int *__restrict__ Tn_,
    const int *__restrict__ Un_, int B, int T, int U1, int V, int blank, float *__restrict__ lpb,
    float *__restrict__ lpe, int Wp, int D) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * T * U1;
    if (idx >= total) return;
    const int u = (int)(idx % U1);
    const int t = (int)((idx / U1) % T);
    const int b = (int)(idx / ((size_t)U1 * T));
    const int Tn = clampi(Tn_[b], 1, T), Un = clampi(Un_[b], 0, U1 - 1);
    if (t >= Tn || u > Un) return;
    const float *row = lp + idx * (size_t)V;
    const float vb = fmaxf(row[blank], NEG);
    float ve = NEG;
    if (u < Un) {
        const int y = labels[(size_t)b * (U1 - 1) + u];
        if (y >= 0 && y < V) ve = fmaxf(row[y], NEG);
    }
    const size_t o = ((size_t)b * D + (t + u)) * Wp + u;
    lpb[o] = vb;
    lpe[o] = ve;
}

// ---------------------------------------------------------------------------------------------
// alpha / beta recurrences.  grid = (2, B): blockIdx.x 0 = alpha, 1 = beta.
// block = NW*64 threads = pad64(U1); thread u owns lattice column u.
// ---------------------------------------------------------------------------------------------
template <int NW>
struct Shift {
    // NW == 1: pure DPP.  NW > 1: DPP inside a wave + LDS hand-off at wave edges.
    float *edge;  // [2][NW+1]
    __device__ inline float up(float v, int step) const {  // thread u <- thread u-1
        if constexpr (NW == 1) {
            return wave_shr1(v, NEG);
        } else {
            const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
            float *e = edge + (step & 1) * (NW + 1);
            if (l == 63) e[w + 1] = v;
            __syncthreads();
            return wave_shr1(v, e[w]);  // e[0] preset to NEG
        }
    }
    __device__ inline float down(float v, int step) const {  // thread u <- thread u+1
        if constexpr (NW == 1) {
            return wave_shl1(v, NEG);
        } else {
            const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
            float *e = edge + (step & 1) * (NW + 1);
            if (l == 0) e[w] = v;
            __syncthreads();
            return wave_shl1(v, e[w + 1]);  // e[NW] preset to NEG
        }
    }
};

constexpr int UNR = 8;  // diagonals prefetched per register batch

template <int NW>
__global__ __launch_bounds__(NW * 64) void rnnt_alpha_beta_kernel(
    const float *__restrict__ lpb_, const float *__restrict__ lpe_, float *__restrict__ alpha_,
    float *__restrict__ beta_, const int *__restrict__ Tn_, const int *__restrict__ Un_,
    float *__restrict__ ll_, float *__restrict__ ll_a_, float *__restrict__ costs, int T, int U1,
    int Wp, int D) {
    __shared__ float edge_buf[2 * (NW + 1)];
    Shift<NW> sh{edge_buf};
    if constexpr (NW > 1) {
        if (threadIdx.x < 2 * (NW + 1)) edge_buf[threadIdx.x] = NEG;
        __syncthreads();
    }
    const int b = blockIdx.y;
    const int u = threadIdx.x;
    const int Tn = clampi(Tn_[b], 1, T), Un = clampi(Un_[b], 0, U1 - 1);
    const int dend = Tn - 1 + Un;  // last diagonal of the (Tn, Un+1) sub-lattice
    const size_t base = (size_t)b * D * Wp + u;
    const float *lpb = lpb_ + base;
    const float *lpe = lpe_ + base;

    // cell (d-u, u) is inside the sub-lattice
    auto inside = [&](int d) { const int t = d - u; return t >= 0 && t < Tn && u <= Un; };

    float pbv[2][UNR], pev[2][UNR];

    if (blockIdx.x == 0) {
        // ----- alpha: A_d[u] = lae(A_{d-1}[u] + lpb_{d-1}[u], A_{d-1}[u-1] + lpe_{d-1}[u-1]) -----
        float *alpha = alpha_ + base;
        float a = (u == 0) ? 0.0f : NEG;
        alpha[0] = a;
        auto load = [&](int buf, int d0) {  // rows d0-1 .. d0+UNR-2
#pragma unroll
            for (int i = 0; i < UNR; ++i) {
                const int r = min(d0 + i - 1, D - 1);
                const bool ok = inside(r);
                const float vb = lpb[(size_t)r * Wp], ve = lpe[(size_t)r * Wp];
                pbv[buf][i] = ok ? vb : 0.0f;  // masked: scratch outside the sub-lattice is garbage
                pev[buf][i] = ok ? ve : 0.0f;
            }
        };
        auto steps = [&](int buf, int d0) {
#pragma unroll
            for (int i = 0; i < UNR; ++i) {
                const int d = d0 + i;
                if (d <= dend) {  // workgroup-uniform
                    const float x = a + pbv[buf][i];
                    const float y = sh.up(a + pev[buf][i], d);
                    a = inside(d) ? lae(x, y) : NEG;
                    alpha[(size_t)d * Wp] = a;
                }
            }
        };
        load(0, 1);
        for (int d0 = 1; d0 <= dend; d0 += 2 * UNR) {
            load(1, d0 + UNR);
            steps(0, d0);
            load(0, d0 + 2 * UNR);
            steps(1, d0 + UNR);
        }
        if (u == Un) ll_a_[b] = a + fmaxf(lpb[(size_t)dend * Wp], NEG);
    } else {
        // ----- beta: B_d[u] = lae(B_{d+1}[u] + lpb_d[u], B_{d+1}[u+1] + lpe_d[u]) -----
        float *beta = beta_ + base;
        float bt = NEG;
        auto load = [&](int buf, int d0) {  // rows d0, d0-1, ..., d0-UNR+1
#pragma unroll
            for (int i = 0; i < UNR; ++i) {
                const int r = max(d0 - i, 0);
                const bool ok = inside(r);
                const float vb = lpb[(size_t)r * Wp], ve = lpe[(size_t)r * Wp];
                pbv[buf][i] = ok ? vb : 0.0f;
                pev[buf][i] = ok ? ve : 0.0f;
            }
        };
        auto steps = [&](int buf, int d0) {
#pragma unroll
            for (int i = 0; i < UNR; ++i) {
                const int d = d0 - i;
                if (d >= 0) {
                    const int t = d - u;
                    const float dn = sh.down(bt, d);
                    const float x = bt + pbv[buf][i];
                    const float y = dn + pev[buf][i];
                    float nb = lae(x, y);
                    if (t == Tn - 1 && u == Un) nb = pbv[buf][i];  // terminal blank
                    bt = inside(d) ? nb : NEG;
                    beta[(size_t)d * Wp] = bt;
                }
            }
        };
        load(0, dend);
        for (int d0 = dend; d0 >= 0; d0 -= 2 * UNR) {
            load(1, d0 - UNR);
            steps(0, d0);
            load(0, d0 - 2 * UNR);
            steps(1, d0 - UNR);
        }
        if (u == 0) {
            ll_[b] = bt;
            costs[b] = -bt;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// grad: dense (B,T,U1,V) gradient, written once.  A wave takes RPT consecutive V-rows per task:
// lanes 0..RPT-1 compute the two non-zeros of "their" row with vector ops, then the wave streams
// each row with 16-byte stores, the row's scalars broadcast by v_readlane.
// ---------------------------------------------------------------------------------------------
constexpr int RPT = 16;

template <bool VEC4, bool NT>
__global__ __launch_bounds__(256) void rnnt_grad_kernel(
    const int *__restrict__ labels, const int *__restrict__ Tn_, const int *__restrict__ Un_,
    int B, int T, int U1, int V, int blank, const float *__restrict__ grad_costs,
    const float *__restrict__ lpb, const float *__restrict__ lpe, const float *__restrict__ alpha,
    const float *__restrict__ beta, const float *__restrict__ ll, int Wp, int D,
    float *__restrict__ grads) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const long nrows = (long)B * T * U1;
    const long ntasks = (nrows + RPT - 1) / RPT;

    for (long task = wave; task < ntasks; task += nwaves) {
        const long row0 = task * RPT;
        float gb = 0.0f, ge = 0.0f;
        int ye = -1;
        const long row = row0 + lane;
        if (lane < RPT && row < nrows) {
            const int u = (int)(row % U1);
            const int t = (int)((row / U1) % T);
            const int b = (int)(row / ((long)U1 * T));
            const int Tn = clampi(Tn_[b], 1, T), Un = clampi(Un_[b], 0, U1 - 1);
            if (t < Tn && u <= Un) {
                const size_t o = ((size_t)b * D + (t + u)) * Wp + u;
                const float a = alpha[o], l = ll[b];
                const float sc = grad_costs ? grad_costs[b] : 1.0f;
                if (t < Tn - 1)
                    gb = -sc * __expf(a + beta[o + Wp] + lpb[o] - l);
                else if (u == Un)
                    gb = -sc * __expf(a + lpb[o] - l);
                if (u < Un) {
                    const int y = labels[(size_t)b * (U1 - 1) + u];
                    if (y >= 0 && y < V) {
                        ye = y;
                        ge = -sc * __expf(a + beta[o + Wp + 1] + lpe[o] - l);
                    }
                }
            }
        }
        const int nr = (int)min((long)RPT, nrows - row0);
        for (int r = 0; r < nr; ++r) {
            const float sgb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gb), r));
            const float sge = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ge), r));
            const int sye = __builtin_amdgcn_readlane(ye, r);
            float *rowp = grads + (size_t)(row0 + r) * V;
            if constexpr (VEC4) {
                const int V4 = V >> 2;
                const int qb = blank >> 2, cb = blank & 3;
                const int qe = sye >> 2, ce = sye & 3;  // sye = -1 -> qe = -1: never matches
                v4f *p = reinterpret_cast<v4f *>(rowp);
                // the row's two non-zeros as whole 16-byte groups (scalar work, once per row)
                v4f vb4 = {cb == 0 ? sgb : 0.f, cb == 1 ? sgb : 0.f, cb == 2 ? sgb : 0.f,
                           cb == 3 ? sgb : 0.f};
                v4f ve4 = {ce == 0 ? sge : 0.f, ce == 1 ? sge : 0.f, ce == 2 ? sge : 0.f,
                           ce == 3 ? sge : 0.f};
                if (qe == qb) {  // same group: label wins on a clash (oracle order)
                    ve4.x = ce == 0 ? sge : vb4.x; ve4.y = ce == 1 ? sge : vb4.y;
                    ve4.z = ce == 2 ? sge : vb4.z; ve4.w = ce == 3 ? sge : vb4.w;
                }
#pragma unroll 4
                for (int q = lane; q < V4; q += WAVE) {
                    v4f v = {0.f, 0.f, 0.f, 0.f};
                    if (q == qb) v = vb4;
                    if (q == qe) v = ve4;
                    if constexpr (NT)
                        __builtin_nontemporal_store(v, p + q);
                    else
                        p[q] = v;
                }
            } else {
                for (int j = lane; j < V; j += WAVE) {
                    float v = 0.0f;
                    if (j == blank) v = sgb;
                    if (j == sye) v = sge;
                    rowp[j] = v;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void rnnt_export_kernel(
    const float *__restrict__ alpha, const float *__restrict__ beta, const int *__restrict__ Tn_,
    const int *__restrict__ Un_, int B, int T, int U1, int Wp, int D, float *__restrict__ out_a,
    float *__restrict__ out_b) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * T * U1) return;
    const int u = (int)(idx % U1);
    const int t = (int)((idx / U1) % T);
    const int b = (int)(idx / ((size_t)U1 * T));
    const size_t o = ((size_t)b * D + (t + u)) * Wp + u;
    const int Tn = clampi(Tn_[b], 1, T), Un = clampi(Un_[b], 0, U1 - 1);
    const bool ok = t < Tn && u <= Un;
    if (out_a) out_a[idx] = ok ? alpha[o] : NEG;
    if (out_b) out_b[idx] = ok ? beta[o] : NEG;
}

int check_dims(int B, int T, int U1, int V, int blank) {
    if (B <= 0 || T <= 0 || U1 <= 0 || V <= 0 || blank < 0 || blank >= V) return PIKA_EINVAL;
    if (U1 > 1024) return PIKA_ETOOBIG;
    return PIKA_OK;
}

template <int NW>
void launch_ab(const Lattice &L, const int *Tn, const int *Un, float *costs, int B, int T, int U1,
               hipStream_t s) {
    hipLaunchKernelGGL((rnnt_alpha_beta_kernel<NW>), dim3(2, B), dim3(NW * 64), 0, s, L.lpb, L.lpe,
                       L.alpha, L.beta, Tn, Un, L.ll, L.ll_a, costs, T, U1, L.Wp, L.D);
}

int grad_variant() {  // tuning hook (A/B on hardware): PIKA_RNNT_GRAD_NT=0 disables nontemporal stores
    static const int v = [] {
        const char *e = getenv("PIKA_RNNT_GRAD_NT");
        return e ? atoi(e) : 1;
    }();
    return v;
}

int grad_blocks() {
    static const int v = [] {
        const char *e = getenv("PIKA_RNNT_GRAD_BLOCKS");
        return e ? atoi(e) : 2048;  // 256 CUs x 8 workgroups of 4 waves
    }();
    return v;
}

}  // namespace

extern "C" {

int pika_amd_abi_version(void) { return 1; }

size_t pika_rnnt_workspace_bytes(int B, int T, int U1) {
    if (B <= 0 || T <= 0 || U1 <= 0 || U1 > 1024) return 0;
    return (4 * plane_elems(B, T, U1) + 2 * (size_t)B) * sizeof(float);
}

int pika_rnnt_loss_forward(const float *log_probs, const int *labels, const int *frames_lengths,
                           const int *labels_lengths, int B, int T, int U1, int V, int blank,
                           float *costs, void *workspace, void *stream) {
    if (int rc = check_dims(B, T, U1, V, blank)) return rc;
    if (!log_probs || !frames_lengths || !labels_lengths || !costs || !workspace) return PIKA_EINVAL;
    if (U1 > 1 && !labels) return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Lattice L = carve(workspace, B, T, U1);
    const size_t cells = (size_t)B * T * U1;
    hipLaunchKernelGGL(rnnt_gather_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s,
                       log_probs, labels, frames_lengths, labels_lengths, B, T, U1, V, blank, L.lpb,
                       L.lpe, L.Wp, L.D);
    switch (L.Wp / 64) {
        case 1: launch_ab<1>(L, frames_lengths, labels_lengths, costs, B, T, U1, s); break;
        case 2: launch_ab<2>(L, frames_lengths, labels_lengths, costs, B, T, U1, s); break;
        case 3: launch_ab<3>(L, frames_lengths, labels_lengths, costs, B, T, U1, s); break;
        case 4: launch_ab<4>(L, frames_lengths, labels_lengths, costs, B, T, U1, s); break;
        case 6: launch_ab<6>(L, frames_lengths, labels_lengths, costs, B, T, U1, s); break;
        case 8: launch_ab<8>(L, frames_lengths, labels_lengths, costs, B, T, U1, s); break;
        case 12: launch_ab<12>(L, frames_lengths, labels_lengths, costs, B, T, U1, s); break;
        case 16: launch_ab<16>(L, frames_lengths, labels_lengths, costs, B, T, U1, s); break;
        default: return PIKA_ETOOBIG;
    }
    return (int)hipGetLastError();
}

int pika_rnnt_loss_backward(const int *labels, const int *frames_lengths, const int *labels_lengths,
                            int B, int T, int U1, int V, int blank, const float *grad_costs,
                            const void *workspace, float *grads, void *stream) {
    if (int rc = check_dims(B, T, U1, V, blank)) return rc;
    if (!frames_lengths || !labels_lengths || !workspace || !grads) return PIKA_EINVAL;
    if (U1 > 1 && !labels) return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Lattice L = carve(const_cast<void *>(workspace), B, T, U1);
    const bool vec4 = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(grads) & 15) == 0);
    const long ntasks = ((long)B * T * U1 + RPT - 1) / RPT;
    const int blocks = (int)min((long)grad_blocks(), (ntasks + 3) / 4);
#define PIKA_GRAD(VEC, NT)                                                                        \
    hipLaunchKernelGGL((rnnt_grad_kernel<VEC, NT>), dim3(blocks), dim3(256), 0, s, labels,       \
                       frames_lengths, labels_lengths, B, T, U1, V, blank, grad_costs, L.lpb,     \
                       L.lpe, L.alpha, L.beta, L.ll, L.Wp, L.D, grads)
    if (vec4) {
        if (grad_variant()) PIKA_GRAD(true, true); else PIKA_GRAD(true, false);
    } else {
        PIKA_GRAD(false, false);
    }
#undef PIKA_GRAD
    return (int)hipGetLastError();
}

int pika_rnnt_loss_fwd_bwd(const float *log_probs, const int *labels, const int *frames_lengths,
                           const int *labels_lengths, int B, int T, int U1, int V, int blank,
                           float *costs, float *grads, void *workspace, void *stream) {
    if (int rc = pika_rnnt_loss_forward(log_probs, labels, frames_lengths, labels_lengths, B, T, U1,
                                        V, blank, costs, workspace, stream))
        return rc;
    return pika_rnnt_loss_backward(labels, frames_lengths, labels_lengths, B, T, U1, V, blank,
                                   nullptr, workspace, grads, stream);
}

int pika_rnnt_export_lattice(const void *workspace, const int *frames_lengths,
                             const int *labels_lengths, int B, int T, int U1, float *alphas,
                             float *betas, void *stream) {
    if (B <= 0 || T <= 0 || U1 <= 0 || U1 > 1024 || !workspace || !frames_lengths || !labels_lengths)
        return PIKA_EINVAL;
    const Lattice L = carve(const_cast<void *>(workspace), B, T, U1);
    const size_t cells = (size_t)B * T * U1;
    hipLaunchKernelGGL(rnnt_export_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), L.alpha, L.beta, frames_lengths,
                       labels_lengths, B, T, U1, L.Wp, L.D, alphas, betas);
    return (int)hipGetLastError();
}

}  // extern "C"
