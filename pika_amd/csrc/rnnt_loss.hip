// pika_amd/csrc/rnnt_loss.hip -- RNN-T loss for gfx950 (MI355X), hand-written HIP.
//
// Replaces the third-party `warp_rnnt` CUDA loss the reference calls at
//   /root/reference/trainer/train_transducer_bmuf_otfaug.py:58,97-99
// (SURVEY.md 8a row 10).  From-scratch CDNA4 design, not a hipify:
//
//   gather    : one thread per lattice cell pulls the TWO log-probs the cell needs (blank, next
//               label) out of the (B,T,U1,V) tensor into two compact planes stored SKEWED:
//               cell (t,u) lives at [t+u][u], so every anti-diagonal is one contiguous row.
//   alpha/beta: one 64-lane wavefront per utterance and direction walks the anti-diagonals;
//               lane = u, the neighbour term moves one lane with a single DPP wave-shift (no
//               LDS, no barrier); log-probs of the next 8 diagonals are prefetched into
//               registers while the current 8 are consumed (the recurrence is latency-bound).
//               Every 16 diagonals the wave subtracts its running maximum and accumulates it
//               in fp64 ("offsets"), so the fp32 lattice values stay O(100) however long the
//               utterance is -- the absolute error of a plain fp32 log-space lattice grows
//               with |alpha| ~ T (ulp(9000) = 1e-3 at the benchmark shape).
//               U1 > 64: one workgroup of up to 16 waves, LDS hand-off at the wave edges.
//   rowmeta   : per lattice cell, the two non-zero gradient values (fp64 offset arithmetic,
//               scaled by autograd's grad_output) + the label index, 16 bytes per cell.
//   grad      : THE HBM-bound kernel -- one flat, address-ordered streaming pass that writes the
//               dense (B,T,U1,V) gradient exactly once with 16-byte stores in 1 KiB-aligned
//               wave transactions; each 16-byte group looks up its row's rowmeta and blends
//               the non-zeros in registers (no memset + scatter, log_probs is not re-read).
//               Measured on MI355X: row-structured writers (a wave or a workgroup per 20 kB
//               V-row) reach only 3.5-4.7 TB/s because rows start on non-128-byte boundaries;
//               the flat form runs at the hipMemsetAsync rate (5.7-5.9 TB/s).
//
// Algorithmic HBM bytes per utterance (T=1000,U1=51,V=5000): 1.020 GB gradient write + ~1.2 MB
// lattice traffic; see DESIGN.md.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pika_rnnt.h"
#include "pika_internal.h"

namespace {

constexpr float NEG = -1.0e30f;  // "log zero": finite, so NEG+NEG / NEG-NEG never make NaN
constexpr float NEG_HALF = -0.5e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int UNR = 8;     // diagonals prefetched per register batch
constexpr int RENORM = 16; // diagonals between renormalisations (multiple of UNR)

typedef float v4f __attribute__((ext_vector_type(4)));

// Width (in lanes) of one skewed lattice row = threads of the alpha/beta workgroup: the
// smallest instantiated wave count that covers U1 label columns.
inline int lattice_width(int U1) {
    static const int kWaves[] = {1, 2, 3, 4, 6, 8, 12, 16};
    for (int nw : kWaves)
        if (nw * 64 >= U1) return nw * 64;
    return 0;
}

struct RowMeta {  // 16 bytes per lattice cell
    float gb;     // gradient at [.., blank]
    float ge;     // gradient at [.., ye]
    int ye;       // next label, -1 if the cell emits none
    int pad;
};

struct Lattice {
    float *lpb;    // [B][D][Wp] blank log-prob of cell (t,u) at row t+u, col u
    float *lpe;    // [B][D][Wp] log-prob of emitting y_{u+1} from cell (t,u)
    float *alpha;  // [B][D][Wp] alpha minus off_a[b][row]
    float *beta;   // [B][D][Wp] beta  minus off_b[b][row]
    double *off_a; // [B][D]
    double *off_b; // [B][D]
    double *ll;    // [B] log-likelihood, beta side
    double *ll_a;  // [B] log-likelihood, alpha side (diagnostic)
    RowMeta *meta; // [B*T*U1]
    int Wp, D;
};

inline size_t plane_elems(int B, int T, int U1) {
    return (size_t)B * (size_t)(T + U1 - 1) * (size_t)lattice_width(U1);
}

inline Lattice carve(void *ws, int B, int T, int U1) {
    Lattice L;
    const size_t n = plane_elems(B, T, U1);
    L.Wp = lattice_width(U1);
    L.D = T + U1 - 1;
    float *p = static_cast<float *>(ws);
    L.lpb = p;
    L.lpe = p + n;
    L.alpha = p + 2 * n;
    L.beta = p + 3 * n;
    double *q = reinterpret_cast<double *>(p + 4 * n);  // n is a multiple of 64: 8-byte aligned
    L.off_a = q;
    L.off_b = q + (size_t)B * L.D;
    L.ll = q + 2 * (size_t)B * L.D;
    L.ll_a = L.ll + B;
    L.meta = reinterpret_cast<RowMeta *>(L.ll_a + B);  // 2BD+2B doubles: 16-byte aligned
    return L;
}

inline size_t workspace_bytes(int B, int T, int U1) {
    const size_t D = (size_t)(T + U1 - 1);
    return 4 * plane_elems(B, T, U1) * sizeof(float) +
           (2 * (size_t)B * D + 2 * (size_t)B) * sizeof(double) +
           (size_t)B * T * U1 * sizeof(RowMeta);
}

__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// log(exp(x)+exp(y)) on the transcendental pipe (v_exp_f32 / v_log_f32).
__device__ inline float lae(float x, float y) {
    const float m = fmaxf(x, y);
    const float d = fminf(x, y) - m;  // <= 0
    return m + LN2 * __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(d * LOG2E));
}

template <int CTRL, int ROW_MASK = 0xf>
__device__ inline float dpp(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL,
                                                      ROW_MASK, 0xf, false));
}
// lane i <- lane i-1 across the whole 64-lane wave (DPP wave_shr:1); lane 0 <- fill.
__device__ inline float wave_shr1(float v, float fill) { return dpp<0x138>(v, fill); }
// lane i <- lane i+1 (DPP wave_shl:1); lane 63 <- fill.
__device__ inline float wave_shl1(float v, float fill) { return dpp<0x130>(v, fill); }

// max over the 64 lanes, broadcast (row_shr 1/2/4/8 + row_bcast 15/31, then readlane 63).
__device__ inline float wave_max(float v) {
    v = fmaxf(v, dpp<0x111>(v, NEG));
    v = fmaxf(v, dpp<0x112>(v, NEG));
    v = fmaxf(v, dpp<0x114>(v, NEG));
    v = fmaxf(v, dpp<0x118>(v, NEG));
    v = fmaxf(v, dpp<0x142, 0xa>(v, NEG));
    v = fmaxf(v, dpp<0x143, 0xc>(v, NEG));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
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
// block = NW*64 threads = lattice_width(U1); thread u owns lattice column u.
// ---------------------------------------------------------------------------------------------
template <int NW>
struct Xchg {
    // NW == 1: pure DPP.  NW > 1: DPP inside a wave + LDS hand-off at wave edges.
    float *edge;  // [2][NW+1], edge[.][0] and edge[.][NW] stay NEG
    float *red;   // [2][NW]
    __device__ inline float up(float v, int step) const {  // thread u <- thread u-1
        if constexpr (NW == 1) {
            return wave_shr1(v, NEG);
        } else {
            const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
            float *e = edge + (step & 1) * (NW + 1);
            if (l == 63 && w + 1 < NW) e[w + 1] = v;
            __syncthreads();
            return wave_shr1(v, e[w]);
        }
    }
    __device__ inline float down(float v, int step) const {  // thread u <- thread u+1
        if constexpr (NW == 1) {
            return wave_shl1(v, NEG);
        } else {
            const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
            float *e = edge + (step & 1) * (NW + 1);
            if (l == 0 && w > 0) e[w] = v;
            __syncthreads();
            return wave_shl1(v, e[w + 1]);
        }
    }
    __device__ inline float max_all(float v, int step) const {  // workgroup-wide max
        float m = wave_max(v);
        if constexpr (NW > 1) {
            float *r = red + ((step / RENORM) & 1) * NW;
            if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = m;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < NW; ++w) m = fmaxf(m, r[w]);
        }
        return m;
    }
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void rnnt_alpha_beta_kernel(
    const float *__restrict__ lpb_, const float *__restrict__ lpe_, float *__restrict__ alpha_,
    float *__restrict__ beta_, double *__restrict__ off_a_, double *__restrict__ off_b_,
    const int *__restrict__ Tn_, const int *__restrict__ Un_, double *__restrict__ ll_,
    double *__restrict__ ll_a_, float *__restrict__ costs, int T, int U1, int Wp, int D) {
    __shared__ float lds[2 * (NW + 1) + 2 * NW];
    Xchg<NW> xc{lds, lds + 2 * (NW + 1)};
    if constexpr (NW > 1) {
        if (threadIdx.x < 2 * (NW + 1) + 2 * NW) lds[threadIdx.x] = NEG;
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
    double off = 0.0;  // sum of subtracted maxima (identical in every thread)

    if (blockIdx.x == 0) {
        // ----- alpha: A_d[u] = lae(A_{d-1}[u] + lpb_{d-1}[u], A_{d-1}[u-1] + lpe_{d-1}[u-1]) -----
        float *alpha = alpha_ + base;
        double *offs = off_a_ + (size_t)b * D;
        float a = (u == 0) ? 0.0f : NEG;
        alpha[0] = a;
        if (u == 0) offs[0] = 0.0;
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
                    const float y = xc.up(a + pev[buf][i], d);
                    a = inside(d) ? lae(x, y) : NEG;
                    if ((d0 + i) % RENORM == 0) {  // d0 = 1 (mod UNR): compile-time per i
                        const float m = xc.max_all(a, d);
                        a = a > NEG_HALF ? a - m : NEG;
                        off += (double)m;
                    }
                    alpha[(size_t)d * Wp] = a;
                    if (u == 0) offs[d] = off;
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
        if (u == Un) ll_a_[b] = (double)a + off + (double)fmaxf(lpb[(size_t)dend * Wp], NEG);
    } else {
        // ----- beta: B_d[u] = lae(B_{d+1}[u] + lpb_d[u], B_{d+1}[u+1] + lpe_d[u]) -----
        float *beta = beta_ + base;
        double *offs = off_b_ + (size_t)b * D;
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
        int k = 0;  // steps taken, for the renormalisation cadence
        auto steps = [&](int buf, int d0) {
#pragma unroll
            for (int i = 0; i < UNR; ++i) {
                const int d = d0 - i;
                if (d >= 0) {
                    const int t = d - u;
                    const float dn = xc.down(bt, d);
                    const float x = bt + pbv[buf][i];
                    const float y = dn + pev[buf][i];
                    float nb = lae(x, y);
                    if (t == Tn - 1 && u == Un) nb = pbv[buf][i];  // terminal blank
                    bt = inside(d) ? nb : NEG;
                    if (i == UNR - 1 && (k & (RENORM / UNR - 1)) == RENORM / UNR - 1) {
                        const float m = xc.max_all(bt, k * UNR);
                        bt = bt > NEG_HALF ? bt - m : NEG;
                        off += (double)m;
                    }
                    beta[(size_t)d * Wp] = bt;
                    if (u == 0) offs[d] = off;
                }
            }
            ++k;
        };
        load(0, dend);
        for (int d0 = dend; d0 >= 0; d0 -= 2 * UNR) {
            load(1, d0 - UNR);
            steps(0, d0);
            load(0, d0 - 2 * UNR);
            steps(1, d0 - UNR);
        }
        if (u == 0) {
            const double ll = (double)bt + off;
            ll_[b] = ll;
            costs[b] = (float)(-ll);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// rowmeta: the two non-zeros of every V-row of the gradient
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rnnt_rowmeta_kernel(
    const int *__restrict__ labels, const int *__restrict__ Tn_, const int *__restrict__ Un_,
    int B, int T, int U1, int V, const float *__restrict__ grad_costs,
    const float *__restrict__ lpb, const float *__restrict__ lpe, const float *__restrict__ alpha,
    const float *__restrict__ beta, const double *__restrict__ off_a,
    const double *__restrict__ off_b, const double *__restrict__ ll, int Wp, int D,
    RowMeta *__restrict__ meta) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nrows = (long)B * T * U1;
    if (row >= nrows) return;
    float gb = 0.f, ge = 0.f;
    int ye = -1;
    const int u = (int)(row % U1);
    const int t = (int)((row / U1) % T);
    const int b = (int)(row / ((long)U1 * T));
    const int Tn = clampi(Tn_[b], 1, T), Un = clampi(Un_[b], 0, U1 - 1);
    if (t < Tn && u <= Un) {
        const int d = t + u;
        const size_t o = ((size_t)b * D + d) * Wp + u;
        const float a = alpha[o];
        const float sc = grad_costs ? grad_costs[b] : 1.0f;
        const double base = off_a[(size_t)b * D + d] - ll[b];
        // exponent = alpha + beta' + lp - ll, with the large parts cancelled in fp64
        const float k1 = (d + 1 < D) ? (float)(base + off_b[(size_t)b * D + d + 1]) : 0.0f;
        if (t < Tn - 1)
            gb = -sc * __expf(k1 + (a + beta[o + Wp] + lpb[o]));
        else if (u == Un)
            gb = -sc * __expf((float)base + (a + lpb[o]));
        if (u < Un) {
            const int y = labels[(size_t)b * (U1 - 1) + u];
            if (y >= 0 && y < V) {
                ye = y;
                ge = -sc * __expf(k1 + (a + beta[o + Wp + 1] + lpe[o]));
            }
        }
    }
    meta[row] = RowMeta{gb, ge, ye, 0};
}

// ---------------------------------------------------------------------------------------------
// grad: flat streaming writer.  Thread -> PT 16-byte groups, 256 groups apart, so each wave
// instruction stores one 1 KiB-aligned contiguous KiB and workgroups are dispatched in address
// order.  A wave instruction (64 groups) spans at most two V-rows when V/4 >= 64, so both rows'
// metadata arrive through the scalar cache (s_load) and the per-lane work is two compares.
// Hardware A/B (tools/grad_sweep.hip, MI355X, 32.64 GB tensor): PT=2 + scalar metadata 4.65 ms
// (7.0 TB/s) vs per-lane metadata PT=4 5.3 ms, hipMemsetAsync 5.1-5.5 ms, row-structured
// writers 7-9 ms.
// ---------------------------------------------------------------------------------------------
__device__ inline v4f blend_row(int q, int qb, int cb, float gb, float ge, int ye) {
    const int qe = ye >> 2, ce = ye & 3;  // ye = -1 -> qe = -1: never matches
    v4f v = {0.f, 0.f, 0.f, 0.f};
    if (q == qb) {
        v.x = cb == 0 ? gb : 0.f; v.y = cb == 1 ? gb : 0.f;
        v.z = cb == 2 ? gb : 0.f; v.w = cb == 3 ? gb : 0.f;
    }
    if (q == qe) {  // label wins if it equals blank (oracle order)
        v.x = ce == 0 ? ge : v.x; v.y = ce == 1 ? ge : v.y;
        v.z = ce == 2 ? ge : v.z; v.w = ce == 3 ? ge : v.w;
    }
    return v;
}

template <bool SCALAR_META, int PT>
__global__ __launch_bounds__(256) void rnnt_grad_kernel(const RowMeta *__restrict__ meta,
                                                        size_t n4, size_t nrows, int V4, int blank,
                                                        v4f *__restrict__ grads) {
    const int qb = blank >> 2, cb = blank & 3;
    if constexpr (SCALAR_META) {  // requires V4 >= 64
        const int lane = threadIdx.x & 63;
        // first group of this wave's first store (wave-uniform, made provably so)
        size_t i0 = (size_t)blockIdx.x * (256 * PT) + (threadIdx.x & ~63);
        i0 = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(i0 >> 32)) << 32) |
             (unsigned)__builtin_amdgcn_readfirstlane((int)i0);
        size_t row0 = i0 / (unsigned)V4;
        int q0 = (int)(i0 - row0 * V4);
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const size_t i = i0 + lane;
            const size_t r0 = row0 < nrows ? row0 : nrows - 1;
            const size_t r1 = row0 + 1 < nrows ? row0 + 1 : nrows - 1;
            const RowMeta m0 = meta[r0], m1 = meta[r1];  // scalar loads
            int q = q0 + lane;
            const bool second = q >= V4;
            q = second ? q - V4 : q;
            const v4f v = blend_row(q, qb, cb, second ? m1.gb : m0.gb, second ? m1.ge : m0.ge,
                                    second ? m1.ye : m0.ye);
            if (i < n4) grads[i] = v;
            i0 += 256;
            q0 += 256;
            while (q0 >= V4) { q0 -= V4; ++row0; }
        }
    } else {
        size_t i = (size_t)blockIdx.x * (256 * PT) + threadIdx.x;
        size_t row = i / (unsigned)V4;
        int q = (int)(i - row * V4);
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            if (i < n4) {
                const RowMeta m = meta[row];
                grads[i] = blend_row(q, qb, cb, m.gb, m.ge, m.ye);
            }
            i += 256;
            q += 256;
            while (q >= V4) { q -= V4; ++row; }
        }
    }
}

// generic path (V % 4 != 0 or unaligned base): one float per thread-iteration
__global__ __launch_bounds__(256) void rnnt_grad_scalar_kernel(const RowMeta *__restrict__ meta,
                                                               size_t n, int V, int blank,
                                                               float *__restrict__ grads) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t row = i / (unsigned)V;
        const int j = (int)(i - row * V);
        float v = 0.f;
        const RowMeta m = meta[row];
        if (j == blank) v = m.gb;
        if (j == m.ye) v = m.ge;
        grads[i] = v;
    }
}

__global__ __launch_bounds__(256) void rnnt_export_kernel(
    const float *__restrict__ alpha, const float *__restrict__ beta, const double *__restrict__ off_a,
    const double *__restrict__ off_b, const int *__restrict__ Tn_, const int *__restrict__ Un_,
    int B, int T, int U1, int Wp, int D, float *__restrict__ out_a, float *__restrict__ out_b) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * T * U1) return;
    const int u = (int)(idx % U1);
    const int t = (int)((idx / U1) % T);
    const int b = (int)(idx / ((size_t)U1 * T));
    const size_t o = ((size_t)b * D + (t + u)) * Wp + u;
    const int Tn = clampi(Tn_[b], 1, T), Un = clampi(Un_[b], 0, U1 - 1);
    const bool ok = t < Tn && u <= Un;
    if (out_a) out_a[idx] = ok ? (float)((double)alpha[o] + off_a[(size_t)b * D + t + u]) : NEG;
    if (out_b) out_b[idx] = ok ? (float)((double)beta[o] + off_b[(size_t)b * D + t + u]) : NEG;
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
                       L.alpha, L.beta, L.off_a, L.off_b, Tn, Un, L.ll, L.ll_a, costs, T, U1, L.Wp,
                       L.D);
}

// d(logits) of log_softmax(scale * logits) under the RNN-T gradient WITHOUT reading the dense gradient:
// row r of it has at most two non-zeros, kept in meta[r] by the backward call; with s = gb + ge
//   out[r, v] = scale * ((v == blank) * gb + (v == ye) * ge - exp(lp[r, v]) * s)      (bf16, zero-padded)
// One wavefront per row (V <= 64*4*CQ, V % 4 == 0), 16-byte loads, 8-byte stores.  CQ = 20 (V <= 5120: the 80 row
// registers the benchmarked V = 5000 needs) or 32 (V <= 8192: the shipped recipes' own vocabulary is 6268,
// egs/train_transducer_bmuf_otfaug.sh:37).
constexpr int CQ_MAX = 32, V_MAX = 64 * 4 * CQ_MAX;
#define PIKA_CQ(extent, CALL) do { if ((extent) <= 64 * 4 * 20) { constexpr int CQ = 20; CALL; } else { constexpr int CQ = 32; CALL; } } while (0)
// Rows a wave of the column-summing d(logits) kernels walks before its column sums go out (one atomicAdd per column and
// workgroup): as many as still leave two workgroups per CU.  (It was 64 from 2^18 rows on and 4 below: the recipes' own
// lattice -- 8 utterances, 97 920 rows -- issued 38 M atomics per pass and ran at 1.5 TB/s, 1.62 ms; now 0.5.)
inline int rows_per_wave(long long rows) {
    const long long r = rows / (4 * 2 * 256);
    return (int)(r < 4 ? 4 : (r > 64 ? 64 : r));
}
typedef __bf16 cbf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// COLSUM: every wave walks RPW consecutive rows and keeps the column sums of what it writes in registers
// (d(bias) of the layer that produced the logits); the four waves of a workgroup combine through LDS and issue one
// atomicAdd per column.
// The logits arrive as fp32 (TI = float) or as the fp16 matrix of pika_gemm_bf16_nt_lse_f16 (TI = _Float16, row pitch
// ld_in): half the bytes of the pass's input.
typedef _Float16 ch16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ch16x8 __attribute__((ext_vector_type(8)));
__device__ inline f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ inline f32x4 ld4(const _Float16 *p) { return __builtin_convertvector(*reinterpret_cast<const ch16x4 *>(p), f32x4); }

template <bool COLSUM, typename TI, int CQ>
__global__ __launch_bounds__(256) void rnnt_dlogits_compact_kernel(const TI *__restrict__ lp,
                                                                   const RowMeta *__restrict__ meta,
                                                                   __bf16 *__restrict__ out, long long rows,
                                                                   int V, long long ld_out, int blank,
                                                                   float scale, int rpw,
                                                                   float *__restrict__ colsum,
                                                                   const float *__restrict__ lse,
                                                                   long long ld_in = 0,
                                                                   const float *__restrict__ gathered = nullptr,
                                                                   const int *__restrict__ g_labels = nullptr,
                                                                   int g_blank = 0, int T = 1, int U1 = 1) {
    if (ld_in == 0) ld_in = V;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c4 = V >> 2, o4 = (int)(ld_out >> 2);
    const long long r0 = ((long long)blockIdx.x * 4 + wave) * rpw;
    f32x4 cs[COLSUM ? CQ : 1];
    if constexpr (COLSUM) {
#pragma unroll
        for (int q = 0; q < CQ; ++q) cs[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long long r = r0; r < r0 + rpw && r < rows; ++r) {
        const RowMeta m = meta[r];
        const float s = m.gb + m.ge;
        const float l = lse ? lse[r] : 0.f;   // `lp` holds raw logits: log-prob = logit - log-sum-exp of its row
        const TI *lrow = lp + r * ld_in;
        cbf16x4 *orow = reinterpret_cast<cbf16x4 *>(out + r * ld_out);
        // rows outside an utterance's sub-lattice carry no gradient and, when `lp` holds raw logits, no defined
        // log-sum-exp: their exp() must not reach the output (inf * 0).  The row is fetched regardless -- making
        // the loads wait for the metadata would serialise two memory latencies per row.
        const bool live = (m.gb != 0.f) || (m.ge != 0.f);
        // (the gathered pair of a row belongs to column g_blank and to the label the PRODUCT was given for it: used where they
        // are the loss' blank and label)
        const bool use_gb = gathered && g_blank == blank;
        bool use_ge = false;
        if (gathered && m.ye >= 0) {
            const int u = (int)(r % U1);
            const long long b_ = r / ((long long)T * U1);
            use_ge = u < U1 - 1 && g_labels[b_ * (U1 - 1) + u] == m.ye;
        }
        f32x4 v[CQ];
#pragma unroll
        for (int q = 0; q < CQ; ++q)
            if (lane + q * 64 < c4) v[q] = ld4(lrow + 4 * (lane + q * 64));
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const int i = lane + q * 64;
            if (i < c4) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                if (live) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int col = 4 * i + e;
                        float g = col == blank ? m.gb : 0.f;
                        if (col == m.ye) g += m.ge;
                        float x = v[q][e];
                        // the two columns whose gradient carries the loss' own terms: their logits in fp32 (the forward
                        // product's epilogue kept them) instead of the fp16 copy -- 2^-11 of |logit| is percents of a softmax
                        if (use_gb && col == blank) x = gathered[2 * r];
                        if (use_ge && col == m.ye) x = gathered[2 * r + 1];
                        o[e] = scale * (g - __expf(x - l) * s);
                    }
                }
                orow[i] = __builtin_convertvector(o, cbf16x4);
                if constexpr (COLSUM) cs[q] += o;
            } else if (i < o4) {
                orow[i] = cbf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            }
        }
    }
    if constexpr (COLSUM) {
        __shared__ f32x4 red[4][64];
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            if (q * 64 >= c4) break;          // uniform
            __syncthreads();
            red[wave][lane] = cs[q];
            __syncthreads();
            const int i = lane + q * 64;
            if (wave == 0 && i < c4) {
                const f32x4 t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
                atomicAdd(colsum + 4 * i + 0, t.x); atomicAdd(colsum + 4 * i + 1, t.y);
                atomicAdd(colsum + 4 * i + 2, t.z); atomicAdd(colsum + 4 * i + 3, t.w);
            }
        }
    }
}


// The same on 8-column granules (V % 8 == 0, ld_out % 8 == 0, V <= 512 * NIT): a lane loads two adjacent 16-byte
// groups and stores ONE 16-byte bf16 granule, nothing of the row is staged in registers (the row log-sum-exp is
// known: every element is one fma + exp2 + mul), and the two special columns of a row are patched under
// wave-uniform tests instead of two compares per element -- a third of the instructions of the kernel above, 120
// registers instead of 289 (one wave per SIMD before), so the loads of several rows are in flight per CU.
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
template <int NIT, typename TI = float>
__global__ __launch_bounds__(256) void rnnt_dlogits_compact8_kernel(const TI *__restrict__ lp,
                                                                    const RowMeta *__restrict__ meta,
                                                                    __bf16 *__restrict__ out, long long rows,
                                                                    int V, long long ld_out, int blank,
                                                                    float scale, int rpw,
                                                                    float *__restrict__ colsum,
                                                                    const float *__restrict__ lse,
                                                                    long long ld_in = 0,
                                                                    const float *__restrict__ gathered = nullptr,
                                                                    const int *__restrict__ g_labels = nullptr,
                                                                    int g_blank = 0, int T = 1, int U1 = 1) {
    if (ld_in == 0) ld_in = V;
    constexpr float LOG2E = 1.4426950408889634f;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long r0 = ((long long)blockIdx.x * 4 + wave) * rpw;
    float cs[NIT][8];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[it][e] = 0.f;
    const int gb_lane = (blank >> 3) & 63;                       // the lane that stores the granule of the blank column
    float cs_blank = 0.f;
    for (long long r = r0; r < r0 + rpw && r < rows; ++r) {
        const RowMeta m = meta[r];
        const float ssum = m.gb + m.ge;
        const bool live = (m.gb != 0.f) || (m.ge != 0.f);       // wave-uniform; dead rows are zeros (see above)
        const float nl2 = lse ? -lse[r] * LOG2E : 0.f;
        const float k = -scale * ssum;
        const TI *lrow = lp + r * ld_in;
        __bf16 *orow = out + r * ld_out;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (lane + 64 * it) * 8;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.f;
            if (live) {
                const TI *src = lrow + (c < V ? c : 0);          // always a valid address; the store is guarded
                f32x4 a, b2;
                if constexpr (sizeof(TI) == 2) {                 // one 16-byte load carries the eight columns
                    const ch16x8 h = *reinterpret_cast<const ch16x8 *>(src);
                    a = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                    b2 = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
                } else {
                    a = ld4(src);
                    b2 = ld4(src + 4);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = k * __builtin_amdgcn_exp2f(__builtin_fmaf(a[e], LOG2E, nl2));
                    o[4 + e] = k * __builtin_amdgcn_exp2f(__builtin_fmaf(b2[e], LOG2E, nl2));
                }
            }
            if (c < V) {
                if (c + 8 > V) {            // V % 8 == 4: the last granule holds four columns and four of the pitch's padding
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = c + e < V ? o[e] : 0.f;
                }
                cbf16x8 w;
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = (__bf16)o[e];
                *reinterpret_cast<cbf16x8 *>(orow + c) = w;
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[it][e] += o[e];
            } else if (c < ld_out) {
                cbf16x8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
                *reinterpret_cast<cbf16x8 *>(orow + c) = z;
            }
        }
        // the (at most two) columns that carry an extra term: the lane that stored their granule stores the element
        // again (same thread, same address: program order), their share of the column sums goes in separately
        if (live) {
            // (their logits in fp32 where the forward product's epilogue kept them: `gathered`; the fp16 copy is 2^-11 of
            // |logit| off, percents of a softmax value at |logit| ~ 30 -- and these two entries carry the gradient's own terms.
            // The column sums took the bulk value above: they get the difference.)
            if (lane == gb_lane && m.gb != 0.f) {
                const float vb = k * __builtin_amdgcn_exp2f(__builtin_fmaf((float)lrow[blank], LOG2E, nl2));
                const float v = (gathered && g_blank == blank) ? k * __builtin_amdgcn_exp2f(__builtin_fmaf(gathered[2 * r], LOG2E, nl2)) : vb;
                orow[blank] = (__bf16)(v + scale * m.gb);
                cs_blank += scale * m.gb + (v - vb);
            }
            const int ye = m.ye;
            if (ye >= 0 && m.ge != 0.f && lane == ((ye >> 3) & 63)) {
                const float vb = k * __builtin_amdgcn_exp2f(__builtin_fmaf((float)lrow[ye], LOG2E, nl2));
                // (the gathered label logit belongs to the label the PRODUCT was given for this row: used where it is the loss')
                const int u = (int)(r % U1);
                const bool mine = gathered && u < U1 - 1 && g_labels[(r / ((long long)T * U1)) * (U1 - 1) + u] == ye;
                const float v = mine ? k * __builtin_amdgcn_exp2f(__builtin_fmaf(gathered[2 * r + 1], LOG2E, nl2)) : vb;
                float add = scale * m.ge;
                if (ye == blank && m.gb != 0.f) add += scale * m.gb;         // never in practice (labels > blank)
                orow[ye] = (__bf16)(v + add);
                atomicAdd(colsum + ye, scale * m.ge + (v - vb));
            }
        }
    }
    __shared__ float red[4][64][9];      // pitch 9: the 8 floats of a lane never share a bank with its neighbour's
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const bool any = it * 512 < V;    // uniform
        __syncthreads();
        if (any) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave][lane][e] = cs[it][e];
        }
        __syncthreads();
        const int c = (lane + 64 * it) * 8;
        if (any && wave == 0 && c < V) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c + e < V)
                    atomicAdd(colsum + c + e, (red[0][lane][e] + red[1][lane][e]) + (red[2][lane][e] + red[3][lane][e]));
        }
    }
    if (cs_blank != 0.f) atomicAdd(colsum + blank, cs_blank);
}

int run_alpha_beta(const Lattice &L, const int *frames_lengths, const int *labels_lengths, float *costs, int B,
                   int T, int U1, hipStream_t s) {
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

// ---------------------------------------------------------------------------------------------
// Fused boundary logits -> (costs, d loss / d logits)  (SURVEY 8d M1'): the log-softmax of the joint output
// is never materialised.  Pass 1 reads every V-row of the logits once (one wavefront per row, row in
// registers): log-sum-exp -> lse[row], and the two log-probs the lattice needs go straight into the skewed
// planes.  After alpha/beta and the row metadata, pass 2 re-reads the logits and writes
//   grad[r, v] = (v == blank) * gb + (v == ye) * ge - exp(logits[r, v] - lse[r]) * (gb + ge).
// 3 x B*T*U1*V*4 bytes in total instead of 6 x for log_softmax + loss + log_softmax backward.
__device__ inline float fwave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ inline float fwave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int CQ>
__global__ __launch_bounds__(256) void rnnt_lse_gather_kernel(
    const float *__restrict__ logits, const int *__restrict__ labels, const int *__restrict__ Tn_,
    const int *__restrict__ Un_, long long rows, int T, int U1, int V, int blank, float *__restrict__ lse,
    float *__restrict__ lpb, float *__restrict__ lpe, int Wp, int D) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63, c4 = V >> 2;
    const int u = (int)(r % U1);
    const int t = (int)((r / U1) % T);
    const int b = (int)(r / ((long long)U1 * T));
    const int Tn = clampi(Tn_[b], 1, T), Un = clampi(Un_[b], 0, U1 - 1);
    if (t >= Tn || u > Un) {            // outside the sub-lattice: no gradient, nothing to gather
        if (lane == 0) lse[r] = 0.f;
        return;
    }
    const float *row = logits + r * V;
    const f32x4 *row4 = reinterpret_cast<const f32x4 *>(row);
    // The row's label is requested FIRST and its two gathered logits right behind the row's own loads: the tail of a row
    // (label -> row[label], row[blank] -> three stores on one lane) was a chain of dependent round trips as long as the
    // streaming part, with the whole wave's registers parked behind it (7.4 ms for one 32.6 GB read = 4.4 TB/s).
    int y = -1;
    if (u < Un) y = labels[(size_t)b * (U1 - 1) + u];
    f32x4 v[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q)
        if (lane + q * 64 < c4) v[q] = row4[lane + q * 64];
    const bool y_ok = y >= 0 && y < V;
    const float xb = row[blank], xy = y_ok ? row[y] : 0.f;       // (wave-uniform addresses: one request each, L2 hits)
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < CQ; ++q)
        if (lane + q * 64 < c4) m = fmaxf(m, fmaxf(fmaxf(v[q].x, v[q].y), fmaxf(v[q].z, v[q].w)));
    m = fwave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < CQ; ++q)
        if (lane + q * 64 < c4)
            sum += (__expf(v[q].x - m) + __expf(v[q].y - m)) + (__expf(v[q].z - m) + __expf(v[q].w - m));
    const float l = m + __logf(fwave_sum(sum));
    if (lane == 0) {
        lse[r] = l;
        const size_t o = ((size_t)b * D + (t + u)) * Wp + u;
        lpb[o] = fmaxf(xb - l, NEG);
        lpe[o] = y_ok ? fmaxf(xy - l, NEG) : NEG;
    }
}

// The same outputs from the per-row partial (max, sum exp) pairs the joint's output GEMM emitted in its epilogue
// (pika_gemm_bf16_nt_lse): 16 lanes per row merge n_part pairs, then the row's two needed logits are fetched (two
// 64-byte sectors per lattice cell instead of the whole 20 KB row).
// GATHERED: the logits are the fp16 matrix `logits16` (pitch ld16) of pika_gemm_bf16_nt_lse_f16, whose epilogue left
// the blank column g_blank and the column of g_labels[b][u] of every row in fp32 (gathered[2 r], gathered[2 r + 1]): a
// cell whose label / blank IS that one takes the fp32 value, any other one the fp16 logit.
template <bool GATHERED = false>
__global__ __launch_bounds__(256) void rnnt_lse_merge_gather_kernel(
    const float *__restrict__ logits, const float *__restrict__ pmax, const float *__restrict__ psum, int n_part,
    const int *__restrict__ labels, const int *__restrict__ Tn_, const int *__restrict__ Un_, long long rows, int T,
    int U1, int V, int blank, float *__restrict__ lse, float *__restrict__ lpb, float *__restrict__ lpe, int Wp, int D,
    const _Float16 *__restrict__ logits16 = nullptr, long long ld16 = 0, const float *__restrict__ gathered = nullptr,
    const int *__restrict__ g_labels = nullptr, int g_blank = 0) {
    const long long r = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    const bool in = r < rows;
    int u = 0, t = 0, b = 0, Tn = 1, Un = 0;
    if (in) {
        u = (int)(r % U1);
        t = (int)((r / U1) % T);
        b = (int)(r / ((long long)U1 * T));
        Tn = clampi(Tn_[b], 1, T);
        Un = clampi(Un_[b], 0, U1 - 1);
    }
    const bool live = in && t < Tn && u <= Un;
    float m = -INFINITY;
    if (live)
        for (int q = l16; q < n_part; q += 16) m = fmaxf(m, pmax[r * n_part + q]);
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    if (live)
        for (int q = l16; q < n_part; q += 16) {
            const float pm = pmax[r * n_part + q];
            if (pm > -INFINITY) s += psum[r * n_part + q] * __expf(pm - m);
        }
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (!in || l16) return;
    if (!live) { lse[r] = 0.f; return; }
    const float l = m + __logf(s);
    lse[r] = l;
    float ve = NEG, vb;
    if constexpr (GATHERED) {
        const _Float16 *row = logits16 + r * ld16;
        if (u < Un) {
            const int y = labels[(size_t)b * (U1 - 1) + u];
            if (y >= 0 && y < V) {
                const float x = (g_labels && g_labels[(size_t)b * (U1 - 1) + u] == y) ? gathered[2 * r + 1] : (float)row[y];
                ve = fmaxf(x - l, NEG);
            }
        }
        vb = (blank == g_blank ? gathered[2 * r] : (float)row[blank]) - l;
    } else {
        const float *row = logits + r * V;
        if (u < Un) {
            const int y = labels[(size_t)b * (U1 - 1) + u];
            if (y >= 0 && y < V) ve = fmaxf(row[y] - l, NEG);
        }
        vb = row[blank] - l;
    }
    const size_t o = ((size_t)b * D + (t + u)) * Wp + u;
    lpb[o] = fmaxf(vb, NEG);
    lpe[o] = ve;
}

template <typename TO, int CQ>
__global__ __launch_bounds__(256) void rnnt_dlogits_fused_kernel(const float *__restrict__ logits,
                                                                 const float *__restrict__ lse,
                                                                 const RowMeta *__restrict__ meta,
                                                                 TO *__restrict__ out, long long rows, int V,
                                                                 long long ld_out, int blank) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63, c4 = V >> 2, o4 = (int)(ld_out >> 2);
    const RowMeta m = meta[r];
    const float s = m.gb + m.ge, l = lse[r];
    const bool live = (m.gb != 0.f) || (m.ge != 0.f);
    const f32x4 *lrow = reinterpret_cast<const f32x4 *>(logits + r * V);
    f32x4 v[CQ];
    if (live) {
#pragma unroll
        for (int q = 0; q < CQ; ++q)
            if (lane + q * 64 < c4) v[q] = lrow[lane + q * 64];
    }
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        const int i = lane + q * 64;
        if (i >= o4) continue;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (live && i < c4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = 4 * i + e;
                float g = col == blank ? m.gb : 0.f;
                if (col == m.ye) g += m.ge;
                o[e] = g - __expf(v[q][e] - l) * s;
            }
        }
        if constexpr (sizeof(TO) == 4) reinterpret_cast<f32x4 *>(out + r * ld_out)[i] = o;
        else reinterpret_cast<cbf16x4 *>(out + r * ld_out)[i] = __builtin_convertvector(o, cbf16x4);
    }
}

}  // namespace

extern "C" {

int pika_amd_abi_version(void) { return 22; }

size_t pika_rnnt_workspace_bytes(int B, int T, int U1) {
    if (B <= 0 || T <= 0 || U1 <= 0 || U1 > 1024) return 0;
    return workspace_bytes(B, T, U1);
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
    return run_alpha_beta(L, frames_lengths, labels_lengths, costs, B, T, U1, s);
}

namespace {
// the streaming pass: dense (B,T,U1,V) gradient from the row metadata in the workspace
int write_dense_grads(const Lattice &L, int B, int T, int U1, int V, int blank, float *grads, hipStream_t s) {
    const size_t nrows = (size_t)B * T * U1;
    const size_t n = nrows * (size_t)V;
    const bool vec4 = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(grads) & 15) == 0);
    if (vec4) {
        const size_t n4 = n / 4;
        const int V4 = V / 4;
        if (V4 >= 64) {
            const size_t blocks = (n4 + 511) / 512;
            if (blocks > 0x7fffffffu) return PIKA_ETOOBIG;
            hipLaunchKernelGGL((rnnt_grad_kernel<true, 2>), dim3((unsigned)blocks), dim3(256), 0, s,
                               L.meta, n4, nrows, V4, blank, reinterpret_cast<v4f *>(grads));
        } else {
            const size_t blocks = (n4 + 1023) / 1024;
            if (blocks > 0x7fffffffu) return PIKA_ETOOBIG;
            hipLaunchKernelGGL((rnnt_grad_kernel<false, 4>), dim3((unsigned)blocks), dim3(256), 0, s,
                               L.meta, n4, nrows, V4, blank, reinterpret_cast<v4f *>(grads));
        }
    } else {
        const size_t want = (n + 255) / 256;
        hipLaunchKernelGGL(rnnt_grad_scalar_kernel, dim3((unsigned)(want < 65536 ? want : 65536)),
                           dim3(256), 0, s, L.meta, n, V, blank, grads);
    }
    return (int)hipGetLastError();
}
}  // namespace

int pika_rnnt_loss_backward(const int *labels, const int *frames_lengths, const int *labels_lengths,
                            int B, int T, int U1, int V, int blank, const float *grad_costs,
                            const void *workspace, float *grads, void *stream) {
    if (int rc = check_dims(B, T, U1, V, blank)) return rc;
    if (!frames_lengths || !labels_lengths || !workspace) return PIKA_EINVAL;
    if (U1 > 1 && !labels) return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Lattice L = carve(const_cast<void *>(workspace), B, T, U1);
    const size_t nrows = (size_t)B * T * U1;
    hipLaunchKernelGGL(rnnt_rowmeta_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, s,
                       labels, frames_lengths, labels_lengths, B, T, U1, V, grad_costs, L.lpb, L.lpe,
                       L.alpha, L.beta, L.off_a, L.off_b, L.ll, L.Wp, L.D, L.meta);
    if (!grads) return (int)hipGetLastError();   // row metadata only (pika_rnnt.h)
    return write_dense_grads(L, B, T, U1, V, blank, grads, s);
}

int pika_rnnt_loss_dense_grads(const void *workspace, int B, int T, int U1, int V, int blank, float *grads,
                               void *stream) {
    if (int rc = check_dims(B, T, U1, V, blank)) return rc;
    if (!workspace || !grads) return PIKA_EINVAL;
    const Lattice L = carve(const_cast<void *>(workspace), B, T, U1);
    return write_dense_grads(L, B, T, U1, V, blank, grads, static_cast<hipStream_t>(stream));
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
                       static_cast<hipStream_t>(stream), L.alpha, L.beta, L.off_a, L.off_b,
                       frames_lengths, labels_lengths, B, T, U1, L.Wp, L.D, alphas, betas);
    return (int)hipGetLastError();
}

int pika_rnnt_dlogits_compact_bf16(const float *log_probs, const float *lse, const void *workspace, int B, int T,
                                   int U1, int V, int blank, void *out, long long ld_out, float scale,
                                   float *colsum, void *stream) {
    if (!log_probs || !workspace || !out || B <= 0 || T <= 0 || U1 <= 0 || U1 > 1024 || V <= 0 || blank < 0 ||
        blank >= V)
        return PIKA_EINVAL;
    if ((V & 3) || V > V_MAX || ld_out < V || (ld_out & 3) || ld_out > V_MAX ||
        (reinterpret_cast<uintptr_t>(log_probs) & 15) || (reinterpret_cast<uintptr_t>(out) & 7))
        return PIKA_EINVAL;
    const Lattice L = carve(const_cast<void *>(workspace), B, T, U1);
    const long long rows = (long long)B * T * U1;
    if (rows > 0x7fffffffLL) return PIKA_ETOOBIG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (colsum) {
        hipError_t e = hipMemsetAsync(colsum, 0, (size_t)V * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
        static const int rpw_env = [] { const char *e = pika_knob("PIKA_DLOGITS_RPW"); return e ? atoi(e) : 0; }();   // A/B
        // tools/dlogits_bench.py at the config-2 lattice: 16 rows per wave 2.43 ms, 32: 2.35, 64: 2.31, 128: 2.31, 256: 2.57;
        // the 4- and 8-column kernels tie (2.35 ms): the pass is the mixed read / write HBM stream at 5.0-5.1 TB/s
        const int rpw = rpw_env > 0 ? rpw_env : rows_per_wave(rows);
        const long long per_block = 4LL * rpw;
        static const bool wide_off = pika_knob("PIKA_DLOGITS_NARROW") != nullptr;     // A/B: the 4-column kernel
        if (!wide_off && !(V & 7) && !(ld_out & 7) && V > 512 * 9 && !(reinterpret_cast<uintptr_t>(out) & 15)) {
#define PIKA_C8(NIT) hipLaunchKernelGGL((rnnt_dlogits_compact8_kernel<NIT, float>), dim3((unsigned)((rows + per_block - 1) / per_block)), \
                                        dim3(256), 0, s, log_probs, L.meta, static_cast<__bf16 *>(out), rows, V, ld_out, blank, scale,  \
                                        rpw, colsum, lse, 0LL)
            if (ld_out <= 512 * 10) PIKA_C8(10); else if (ld_out <= 512 * 13) PIKA_C8(13); else PIKA_C8(16);
#undef PIKA_C8
            return (int)hipGetLastError();
        }
        PIKA_CQ(ld_out, hipLaunchKernelGGL((rnnt_dlogits_compact_kernel<true, float, CQ>),
                                           dim3((unsigned)((rows + per_block - 1) / per_block)), dim3(256), 0, s, log_probs, L.meta,
                                           static_cast<__bf16 *>(out), rows, V, ld_out, blank, scale, rpw, colsum, lse, 0LL));
    } else {
        PIKA_CQ(ld_out, hipLaunchKernelGGL((rnnt_dlogits_compact_kernel<false, float, CQ>), dim3((unsigned)((rows + 3) / 4)),
                                           dim3(256), 0, s, log_probs, L.meta, static_cast<__bf16 *>(out), rows, V, ld_out, blank,
                                           scale, 1, static_cast<float *>(nullptr), lse, 0LL));
    }
    return (int)hipGetLastError();
}

int pika_rnnt_fused_forward(const float *logits, const int *labels, const int *frames_lengths,
                            const int *labels_lengths, int B, int T, int U1, int V, int blank, float *costs,
                            float *lse, void *workspace, void *stream) {
    if (int rc = check_dims(B, T, U1, V, blank)) return rc;
    if (!logits || !frames_lengths || !labels_lengths || !costs || !lse || !workspace) return PIKA_EINVAL;
    if (U1 > 1 && !labels) return PIKA_EINVAL;
    if ((V & 3) || V > V_MAX || (reinterpret_cast<uintptr_t>(logits) & 15)) return PIKA_EINVAL;
    const long long rows = (long long)B * T * U1;
    if (rows > 0x7fffffffLL) return PIKA_ETOOBIG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Lattice L = carve(workspace, B, T, U1);
    PIKA_CQ(V, hipLaunchKernelGGL(rnnt_lse_gather_kernel<CQ>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, labels,
                                  frames_lengths, labels_lengths, rows, T, U1, V, blank, lse, L.lpb, L.lpe, L.Wp, L.D));
    return run_alpha_beta(L, frames_lengths, labels_lengths, costs, B, T, U1, s);
}

int pika_rnnt_fused_forward_partials(const float *logits, const float *pmax, const float *psum, int n_part,
                                     const int *labels, const int *frames_lengths, const int *labels_lengths, int B,
                                     int T, int U1, int V, int blank, float *costs, float *lse, void *workspace,
                                     void *stream) {
    if (int rc = check_dims(B, T, U1, V, blank)) return rc;
    if (!logits || !pmax || !psum || n_part <= 0 || !frames_lengths || !labels_lengths || !costs || !lse || !workspace)
        return PIKA_EINVAL;
    if (U1 > 1 && !labels) return PIKA_EINVAL;
    const long long rows = (long long)B * T * U1;
    if (rows > 0x7fffffffLL) return PIKA_ETOOBIG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Lattice L = carve(workspace, B, T, U1);
    hipLaunchKernelGGL(rnnt_lse_merge_gather_kernel<false>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, logits,
                       pmax, psum, n_part, labels, frames_lengths, labels_lengths, rows, T, U1, V, blank, lse, L.lpb, L.lpe,
                       L.Wp, L.D, static_cast<const _Float16 *>(nullptr), 0LL, static_cast<const float *>(nullptr),
                       static_cast<const int *>(nullptr), 0);
    return run_alpha_beta(L, frames_lengths, labels_lengths, costs, B, T, U1, s);
}

int pika_rnnt_fused_forward_gathered(const void *logits16, long long ld16, const float *gathered, const int *g_labels,
                                     int g_blank, const float *pmax, const float *psum, int n_part, const int *labels,
                                     const int *frames_lengths, const int *labels_lengths, int B, int T, int U1, int V,
                                     int blank, float *costs, float *lse, void *workspace, void *stream) {
    if (int rc = check_dims(B, T, U1, V, blank)) return rc;
    if (!logits16 || ld16 < V || !gathered || !pmax || !psum || n_part <= 0 || !frames_lengths || !labels_lengths || !costs ||
        !lse || !workspace)
        return PIKA_EINVAL;
    if (U1 > 1 && !labels) return PIKA_EINVAL;
    const long long rows = (long long)B * T * U1;
    if (rows > 0x7fffffffLL) return PIKA_ETOOBIG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Lattice L = carve(workspace, B, T, U1);
    hipLaunchKernelGGL(rnnt_lse_merge_gather_kernel<true>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s,
                       static_cast<const float *>(nullptr), pmax, psum, n_part, labels, frames_lengths, labels_lengths, rows,
                       T, U1, V, blank, lse, L.lpb, L.lpe, L.Wp, L.D, static_cast<const _Float16 *>(logits16), ld16, gathered,
                       g_labels, g_blank);
    return run_alpha_beta(L, frames_lengths, labels_lengths, costs, B, T, U1, s);
}

int pika_rnnt_dlogits_compact_bf16_f16in(const void *logits16, long long ld_in, const float *lse, const void *workspace,
                                         int B, int T, int U1, int V, int blank, void *out, long long ld_out, float scale,
                                         float *colsum, const float *gathered, const int *g_labels, int g_blank,
                                         void *stream) {
    if (gathered && !g_labels) return PIKA_EINVAL;
    if (!logits16 || !lse || !workspace || !out || B <= 0 || T <= 0 || U1 <= 0 || U1 > 1024 || V <= 0 || blank < 0 ||
        blank >= V)
        return PIKA_EINVAL;
    if ((V & 3) || V > V_MAX || ld_out < V || (ld_out & 3) || ld_out > V_MAX || ld_in < V || (ld_in & 3) ||
        (reinterpret_cast<uintptr_t>(logits16) & 7) || (reinterpret_cast<uintptr_t>(out) & 7))
        return PIKA_EINVAL;
    const Lattice L = carve(const_cast<void *>(workspace), B, T, U1);
    const long long rows = (long long)B * T * U1;
    if (rows > 0x7fffffffLL) return PIKA_ETOOBIG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const _Float16 *x = static_cast<const _Float16 *>(logits16);
    if (colsum) {
        hipError_t e = hipMemsetAsync(colsum, 0, (size_t)V * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
        const int rpw = rows_per_wave(rows);
        const long long per_block = 4LL * rpw;
        // (V % 8 == 4 -- the shipped recipes' 6268 -- rides on a 16-bit pitch of whole granules: the last one is masked)
        if (!(ld_out & 7) && !(ld_in & 7) && ld_in >= ((V + 7) & ~7) && V > 512 * 9 && !(reinterpret_cast<uintptr_t>(out) & 15) &&
            !(reinterpret_cast<uintptr_t>(logits16) & 15)) {
#define PIKA_C8(NIT) hipLaunchKernelGGL((rnnt_dlogits_compact8_kernel<NIT, _Float16>), dim3((unsigned)((rows + per_block - 1) / per_block)), \
                                        dim3(256), 0, s, x, L.meta, static_cast<__bf16 *>(out), rows, V, ld_out, blank, scale, rpw, colsum,  \
                                        lse, ld_in, gathered, g_labels, g_blank, T, U1)
            if (ld_out <= 512 * 10) PIKA_C8(10); else if (ld_out <= 512 * 13) PIKA_C8(13); else PIKA_C8(16);
#undef PIKA_C8
            return (int)hipGetLastError();
        }
        PIKA_CQ(ld_out, hipLaunchKernelGGL((rnnt_dlogits_compact_kernel<true, _Float16, CQ>),
                                           dim3((unsigned)((rows + per_block - 1) / per_block)), dim3(256), 0, s, x, L.meta,
                                           static_cast<__bf16 *>(out), rows, V, ld_out, blank, scale, rpw, colsum, lse, ld_in,
                                           gathered, g_labels, g_blank, T, U1));
    } else {
        PIKA_CQ(ld_out, hipLaunchKernelGGL((rnnt_dlogits_compact_kernel<false, _Float16, CQ>), dim3((unsigned)((rows + 3) / 4)),
                                           dim3(256), 0, s, x, L.meta, static_cast<__bf16 *>(out), rows, V, ld_out, blank, scale, 1,
                                           static_cast<float *>(nullptr), lse, ld_in, gathered, g_labels, g_blank, T, U1));
    }
    return (int)hipGetLastError();
}

int pika_rnnt_fused_backward(const float *logits, const float *lse, const int *labels,
                             const int *frames_lengths, const int *labels_lengths, int B, int T, int U1, int V,
                             int blank, const float *grad_costs, const void *workspace, void *grad_logits,
                             int out_dtype, long long ld_out, void *stream) {
    if (int rc = check_dims(B, T, U1, V, blank)) return rc;
    if (!logits || !lse || !frames_lengths || !labels_lengths || !workspace || !grad_logits) return PIKA_EINVAL;
    if (U1 > 1 && !labels) return PIKA_EINVAL;
    if ((V & 3) || V > V_MAX || ld_out < V || (ld_out & 3) || ld_out > V_MAX ||
        (reinterpret_cast<uintptr_t>(logits) & 15))
        return PIKA_EINVAL;
    if (out_dtype != 0 && out_dtype != 1) return PIKA_EINVAL;   // PIKA_F32 / PIKA_BF16 (pika_gemm.h)
    if (reinterpret_cast<uintptr_t>(grad_logits) & (out_dtype == 0 ? 15 : 7)) return PIKA_EINVAL;
    const long long rows = (long long)B * T * U1;
    if (rows > 0x7fffffffLL) return PIKA_ETOOBIG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Lattice L = carve(const_cast<void *>(workspace), B, T, U1);
    hipLaunchKernelGGL(rnnt_rowmeta_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s,
                       labels, frames_lengths, labels_lengths, B, T, U1, V, grad_costs, L.lpb, L.lpe,
                       L.alpha, L.beta, L.off_a, L.off_b, L.ll, L.Wp, L.D, L.meta);
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (out_dtype == 0)
        PIKA_CQ(ld_out, hipLaunchKernelGGL((rnnt_dlogits_fused_kernel<float, CQ>), grid, dim3(256), 0, s, logits, lse, L.meta,
                                           static_cast<float *>(grad_logits), rows, V, ld_out, blank));
    else
        PIKA_CQ(ld_out, hipLaunchKernelGGL((rnnt_dlogits_fused_kernel<__bf16, CQ>), grid, dim3(256), 0, s, logits, lse, L.meta,
                                           static_cast<__bf16 *>(grad_logits), rows, V, ld_out, blank));
    return (int)hipGetLastError();
}

}  // extern "C"
