// pika_amd/csrc/joint.hip -- joint-network elementwise / row kernels for gfx950
// (include/pika_joint.h; reference trainer/model/transducer.py:98-111).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pika_gemm.h"
#include "pika_joint.h"
#include "pika_rnnt.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// tanh(z1) and sigmoid(zg) of one element with TWO exponentials and ONE reciprocal (the kernels below are bound by
// the transcendental pipe: 400 M elements per pass at config 2):
//   e = exp(2 z1), f = exp(-zg), r = 1 / ((e + 1)(1 + f))
//   sigmoid = (e + 1) r,   1 - tanh = 2 (1 + f) r =: q,   1 - sigmoid = f (e + 1) r =: ns
// q and ns carry no cancellation, so the backward factors (1 - tanh^2) = q (2 - q) and sigmoid (1 - sigmoid) keep their
// relative accuracy in the saturated tails.  The clamps keep the product below FLT_MAX; tanh(+-15) is +-1 in fp32
// and sigmoid(-50) = 2e-22.
struct GateVals { float th, sg, q, ns; };
__device__ inline GateVals gate_vals(float z1, float zg) {
    z1 = fminf(fmaxf(z1, -15.f), 15.f);
    zg = fmaxf(zg, -50.f);
    const float e1 = __expf(2.0f * z1) + 1.0f, f = __expf(-zg), f1 = 1.0f + f;
    const float r = __builtin_amdgcn_rcpf(e1 * f1);
    GateVals v;
    v.sg = e1 * r;
    v.q = 2.0f * f1 * r;
    v.th = 1.0f - v.q;
    v.ns = f * v.sg;
    return v;
}

__device__ inline f32x4 gate4(f32x4 z1, f32x4 zg) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const GateVals v = gate_vals(z1[i], zg[i]);
        r[i] = v.th * v.sg;
    }
    return r;
}

// grid (T, B), block H/4 threads (<= 1024): thread owns 4 channels, loops over u.
template <typename TOUT>
__global__ void gate_fwd_kernel(const float *__restrict__ e1, const float *__restrict__ p1,
                                const float *__restrict__ eg, const float *__restrict__ pg,
                                TOUT *__restrict__ h, int T, int U, int H) {
    const int b = blockIdx.y, t = blockIdx.x;
    const int H4 = H >> 2;
    for (int c = threadIdx.x; c < H4; c += blockDim.x) {
        const f32x4 a1 = reinterpret_cast<const f32x4 *>(e1 + ((size_t)b * T + t) * H)[c];
        const f32x4 ag = reinterpret_cast<const f32x4 *>(eg + ((size_t)b * T + t) * H)[c];
        for (int u = 0; u < U; ++u) {
            const f32x4 b1 = reinterpret_cast<const f32x4 *>(p1 + ((size_t)b * U + u) * H)[c];
            const f32x4 bg = reinterpret_cast<const f32x4 *>(pg + ((size_t)b * U + u) * H)[c];
            const f32x4 r = gate4(a1 + b1, ag + bg);
            TOUT *dst = h + (((size_t)b * T + t) * U + u) * H + 4 * c;
            if constexpr (sizeof(TOUT) == 4)
                *reinterpret_cast<f32x4 *>(dst) = r;
            else
                *reinterpret_cast<bf16x4 *>(dst) = __builtin_convertvector(r, bf16x4);
        }
    }
}

__device__ inline void gate_grads(float z1, float zg, float dh, float &d1, float &dg) {
    const GateVals v = gate_vals(z1, zg);
    const float ds = dh * v.sg;
    d1 = ds * v.q * (2.0f - v.q);      // sigmoid * (1 - tanh^2)
    dg = ds * v.th * v.ns;             // tanh * sigmoid * (1 - sigmoid)
}

// REDUCE_U: grid (T,B): sums over u -> de1/deg[b,t,:].  else grid (U,B): sums over t -> dp1/dpg.
template <typename TD>
__device__ inline f32x4 load_dh4(const TD *p) {
    if constexpr (sizeof(TD) == 4) {
        return *reinterpret_cast<const f32x4 *>(p);
    } else {
        const bf16x4 v = *reinterpret_cast<const bf16x4 *>(p);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
}

template <bool REDUCE_U, typename TD>
__global__ void gate_bwd_kernel(const TD *__restrict__ dh, const float *__restrict__ e1,
                                const float *__restrict__ p1, const float *__restrict__ eg,
                                const float *__restrict__ pg, float *__restrict__ o1,
                                float *__restrict__ og, int T, int U, int H) {
    const int b = blockIdx.y, fixed = blockIdx.x;
    const int H4 = H >> 2;
    const int n = REDUCE_U ? U : T;
    for (int c = threadIdx.x; c < H4; c += blockDim.x) {
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, sg = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < n; ++i) {
            const int t = REDUCE_U ? fixed : i, u = REDUCE_U ? i : fixed;
            const f32x4 z1 = reinterpret_cast<const f32x4 *>(e1 + ((size_t)b * T + t) * H)[c] +
                             reinterpret_cast<const f32x4 *>(p1 + ((size_t)b * U + u) * H)[c];
            const f32x4 zg = reinterpret_cast<const f32x4 *>(eg + ((size_t)b * T + t) * H)[c] +
                             reinterpret_cast<const f32x4 *>(pg + ((size_t)b * U + u) * H)[c];
            const f32x4 d = load_dh4<TD>(dh + (((size_t)b * T + t) * U + u) * H + 4 * c);
            float a, g;
            gate_grads(z1.x, zg.x, d.x, a, g); s1.x += a; sg.x += g;
            gate_grads(z1.y, zg.y, d.y, a, g); s1.y += a; sg.y += g;
            gate_grads(z1.z, zg.z, d.z, a, g); s1.z += a; sg.z += g;
            gate_grads(z1.w, zg.w, d.w, a, g); s1.w += a; sg.w += g;
        }
        const size_t o = ((size_t)b * (REDUCE_U ? T : U) + fixed) * H;
        reinterpret_cast<f32x4 *>(o1 + o)[c] = s1;
        reinterpret_cast<f32x4 *>(og + o)[c] = sg;
    }
}

// ---- row kernels: one 256-thread workgroup per row, row held in registers (cols <= 8192) ----
constexpr int ROW_THREADS = 256;
constexpr int MAXQ = 8;  // float4 per thread

__device__ inline float block_reduce(float v, bool is_max, float *lds) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, w) : v + w;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[wave] = v;
    __syncthreads();
    v = lds[0];
#pragma unroll
    for (int w = 1; w < ROW_THREADS / 64; ++w) v = is_max ? fmaxf(v, lds[w]) : v + lds[w];
    __syncthreads();
    return v;
}

__global__ __launch_bounds__(ROW_THREADS) void log_softmax_kernel(float *__restrict__ x, int cols,
                                                                  long long ld, float scale) {
    __shared__ float lds[ROW_THREADS / 64];
    float *row = x + (long long)blockIdx.x * ld;
    const bool vec = ((cols & 3) == 0) && ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    if (vec && cols <= ROW_THREADS * 4 * MAXQ) {
        const int c4 = cols >> 2;
        f32x4 v[MAXQ];
        float m = -INFINITY;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int i = threadIdx.x + q * ROW_THREADS;
            if (i < c4) {
                v[q] = reinterpret_cast<const f32x4 *>(row)[i] * scale;
                m = fmaxf(m, fmaxf(fmaxf(v[q].x, v[q].y), fmaxf(v[q].z, v[q].w)));
            }
        }
        m = block_reduce(m, true, lds);
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int i = threadIdx.x + q * ROW_THREADS;
            if (i < c4) s += __expf(v[q].x - m) + __expf(v[q].y - m) + __expf(v[q].z - m) + __expf(v[q].w - m);
        }
        s = block_reduce(s, false, lds);
        const float lse = m + __logf(s);
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int i = threadIdx.x + q * ROW_THREADS;
            if (i < c4) reinterpret_cast<f32x4 *>(row)[i] = v[q] - lse;
        }
    } else {
        float m = -INFINITY;
        for (int i = threadIdx.x; i < cols; i += ROW_THREADS) m = fmaxf(m, row[i] * scale);
        m = block_reduce(m, true, lds);
        float s = 0.f;
        for (int i = threadIdx.x; i < cols; i += ROW_THREADS) s += __expf(row[i] * scale - m);
        s = block_reduce(s, false, lds);
        const float lse = m + __logf(s);
        for (int i = threadIdx.x; i < cols; i += ROW_THREADS) row[i] = row[i] * scale - lse;
    }
}

__global__ __launch_bounds__(ROW_THREADS) void log_softmax_bwd_kernel(const float *__restrict__ lp,
                                                                      float *__restrict__ g, int cols,
                                                                      long long ld, float scale) {
    __shared__ float lds[ROW_THREADS / 64];
    const float *lrow = lp + (long long)blockIdx.x * ld;
    float *grow = g + (long long)blockIdx.x * ld;
    const bool vec = ((cols & 3) == 0) && ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(g) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(lp) & 15) == 0);
    if (vec && cols <= ROW_THREADS * 4 * MAXQ) {
        const int c4 = cols >> 2;
        f32x4 v[MAXQ];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int i = threadIdx.x + q * ROW_THREADS;
            if (i < c4) {
                v[q] = reinterpret_cast<const f32x4 *>(grow)[i];
                s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
            }
        }
        s = block_reduce(s, false, lds);
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int i = threadIdx.x + q * ROW_THREADS;
            if (i < c4) {
                const f32x4 l = reinterpret_cast<const f32x4 *>(lrow)[i];
                f32x4 r;
                r.x = scale * (v[q].x - __expf(l.x) * s);
                r.y = scale * (v[q].y - __expf(l.y) * s);
                r.z = scale * (v[q].z - __expf(l.z) * s);
                r.w = scale * (v[q].w - __expf(l.w) * s);
                reinterpret_cast<f32x4 *>(grow)[i] = r;
            }
        }
    } else {
        float s = 0.f;
        for (int i = threadIdx.x; i < cols; i += ROW_THREADS) s += grow[i];
        s = block_reduce(s, false, lds);
        for (int i = threadIdx.x; i < cols; i += ROW_THREADS)
            grow[i] = scale * (grow[i] - __expf(lrow[i]) * s);
    }
}


// ---- wave-per-row variants: no workgroup barriers, 4 rows in flight per workgroup; row held in
// registers (cols <= 64 lanes * 4 * WQ floats) -------------------------------------------------
// WQ float4 per lane: 20 -> up to 5120 columns (the benchmarked V = 5000), 32 -> up to 8192 (the shipped recipes' V = 6268)
constexpr int WQ_MAX = 32;
#define PIKA_WQ(extent, CALL) do { if ((extent) <= 64 * 4 * 20) { constexpr int WQ = 20; CALL; } else { constexpr int WQ = 32; CALL; } } while (0)

__device__ inline float wave_red(float v, bool is_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, w) : v + w;
    }
    return v;
}

template <int WQ>
__global__ __launch_bounds__(256) void log_softmax_wave_kernel(float *__restrict__ x, long long rows,
                                                               int cols, long long ld, float scale) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63, c4 = cols >> 2;
    f32x4 *row = reinterpret_cast<f32x4 *>(x + r * ld);
    f32x4 v[WQ];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
        const int i = lane + q * 64;
        if (i < c4) {
            v[q] = row[i] * scale;
            m = fmaxf(m, fmaxf(fmaxf(v[q].x, v[q].y), fmaxf(v[q].z, v[q].w)));
        }
    }
    m = wave_red(m, true);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < WQ; ++q)
        if (lane + q * 64 < c4)
            s += __expf(v[q].x - m) + __expf(v[q].y - m) + __expf(v[q].z - m) + __expf(v[q].w - m);
    s = wave_red(s, false);
    const float lse = m + __logf(s);
#pragma unroll
    for (int q = 0; q < WQ; ++q)
        if (lane + q * 64 < c4) row[lane + q * 64] = v[q] - lse;
}

template <int WQ>
__global__ __launch_bounds__(256) void log_softmax_bwd_wave_kernel(const float *__restrict__ lp,
                                                                   float *__restrict__ g, long long rows,
                                                                   int cols, long long ld, float scale) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63, c4 = cols >> 2;
    const f32x4 *lrow = reinterpret_cast<const f32x4 *>(lp + r * ld);
    f32x4 *grow = reinterpret_cast<f32x4 *>(g + r * ld);
    f32x4 v[WQ];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < WQ; ++q)
        if (lane + q * 64 < c4) {
            v[q] = grow[lane + q * 64];
            s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
        }
    s = wave_red(s, false);
#pragma unroll
    for (int q = 0; q < WQ; ++q)
        if (lane + q * 64 < c4) {
            const f32x4 l = lrow[lane + q * 64];
            f32x4 o;
            o.x = scale * (v[q].x - __expf(l.x) * s);
            o.y = scale * (v[q].y - __expf(l.y) * s);
            o.z = scale * (v[q].z - __expf(l.z) * s);
            o.w = scale * (v[q].w - __expf(l.w) * s);
            grow[lane + q * 64] = o;
        }
}

template <int WQ>
__global__ __launch_bounds__(256) void log_softmax_bwd_bf16_kernel(const float *__restrict__ lp,
                                                                   const float *__restrict__ g,
                                                                   __bf16 *__restrict__ out, long long rows,
                                                                   int cols, long long ld, long long ld_out,
                                                                   float scale) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63, c4 = cols >> 2, o4 = (int)(ld_out >> 2);
    const f32x4 *lrow = reinterpret_cast<const f32x4 *>(lp + r * ld);
    const f32x4 *grow = reinterpret_cast<const f32x4 *>(g + r * ld);
    bf16x4 *orow = reinterpret_cast<bf16x4 *>(out + r * ld_out);
    f32x4 v[WQ];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < WQ; ++q)
        if (lane + q * 64 < c4) {
            v[q] = grow[lane + q * 64];
            s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
        }
    s = wave_red(s, false);
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
        const int i = lane + q * 64;
        if (i < c4) {
            const f32x4 l = lrow[i];
            f32x4 o;
            o.x = scale * (v[q].x - __expf(l.x) * s);
            o.y = scale * (v[q].y - __expf(l.y) * s);
            o.z = scale * (v[q].z - __expf(l.z) * s);
            o.w = scale * (v[q].w - __expf(l.w) * s);
            orow[i] = __builtin_convertvector(o, bf16x4);
        } else if (i < o4) {
            orow[i] = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        }
    }
}

inline bool wave_row_ok(const void *a, const void *b, int cols, long long ld) {
    return (cols & 3) == 0 && (ld & 3) == 0 && cols <= 64 * 4 * WQ_MAX &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

__global__ __launch_bounds__(ROW_THREADS) void mbr_risk_grad_kernel(float *__restrict__ lp,
                                                                    const int *__restrict__ sym,
                                                                    const float *__restrict__ val,
                                                                    int cols, long long ld, float scale) {
    float *row = lp + (long long)blockIdx.x * ld;
    const float v = val[blockIdx.x] * scale;
    const int s = sym[blockIdx.x];
    for (int i = threadIdx.x; i < cols; i += ROW_THREADS)
        row[i] = v == 0.f ? 0.f : v * ((i == s ? 1.f : 0.f) - __expf(row[i]));
}

}  // namespace

extern "C" {

int pika_joint_gate_fwd(const float *e1, const float *p1, const float *eg, const float *pg, void *h,
                        int out_dtype, int B, int T, int U, int H, void *stream) {
    if (!e1 || !p1 || !eg || !pg || !h || B <= 0 || T <= 0 || U <= 0 || H <= 0 || (H & 3))
        return PIKA_EINVAL;
    if (B > 65535) return PIKA_ETOOBIG;
    const int threads = min(1024, ((H / 4 + 63) / 64) * 64);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_dtype == PIKA_F32)
        hipLaunchKernelGGL(gate_fwd_kernel<float>, dim3(T, B), dim3(threads), 0, s, e1, p1, eg, pg,
                           static_cast<float *>(h), T, U, H);
    else if (out_dtype == PIKA_BF16)
        hipLaunchKernelGGL(gate_fwd_kernel<__bf16>, dim3(T, B), dim3(threads), 0, s, e1, p1, eg, pg,
                           static_cast<__bf16 *>(h), T, U, H);
    else
        return PIKA_EINVAL;
    return (int)hipGetLastError();
}

int pika_joint_gate_bwd(const void *dh, int dh_dtype, const float *e1, const float *p1, const float *eg,
                        const float *pg, float *de1, float *dp1, float *deg, float *dpg, int B, int T,
                        int U, int H, void *stream) {
    if (!dh || !e1 || !p1 || !eg || !pg || !de1 || !dp1 || !deg || !dpg || B <= 0 || T <= 0 ||
        U <= 0 || H <= 0 || (H & 3))
        return PIKA_EINVAL;
    if (B > 65535) return PIKA_ETOOBIG;
    const int threads = min(1024, ((H / 4 + 63) / 64) * 64);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dh_dtype == PIKA_F32) {
        const float *d = static_cast<const float *>(dh);
        hipLaunchKernelGGL((gate_bwd_kernel<true, float>), dim3(T, B), dim3(threads), 0, s, d, e1, p1, eg, pg,
                           de1, deg, T, U, H);
        hipLaunchKernelGGL((gate_bwd_kernel<false, float>), dim3(U, B), dim3(threads), 0, s, d, e1, p1, eg, pg,
                           dp1, dpg, T, U, H);
    } else if (dh_dtype == PIKA_BF16) {
        const __bf16 *d = static_cast<const __bf16 *>(dh);
        hipLaunchKernelGGL((gate_bwd_kernel<true, __bf16>), dim3(T, B), dim3(threads), 0, s, d, e1, p1, eg, pg,
                           de1, deg, T, U, H);
        hipLaunchKernelGGL((gate_bwd_kernel<false, __bf16>), dim3(U, B), dim3(threads), 0, s, d, e1, p1, eg, pg,
                           dp1, dpg, T, U, H);
    } else {
        return PIKA_EINVAL;
    }
    return (int)hipGetLastError();
}

int pika_log_softmax_rows(float *x, long long rows, int cols, long long ld, float scale,
                          void *stream) {
    if (!x || rows <= 0 || cols <= 0 || ld < cols) return PIKA_EINVAL;
    if (rows > 0x7fffffffLL) return PIKA_ETOOBIG;
    if (wave_row_ok(x, x, cols, ld))
        PIKA_WQ(cols, hipLaunchKernelGGL(log_softmax_wave_kernel<WQ>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                                         static_cast<hipStream_t>(stream), x, rows, cols, ld, scale));
    else
        hipLaunchKernelGGL(log_softmax_kernel, dim3((unsigned)rows), dim3(ROW_THREADS), 0,
                           static_cast<hipStream_t>(stream), x, cols, ld, scale);
    return (int)hipGetLastError();
}

int pika_log_softmax_bwd_rows(const float *lp, float *g, long long rows, int cols, long long ld,
                              float scale, void *stream) {
    if (!lp || !g || rows <= 0 || cols <= 0 || ld < cols) return PIKA_EINVAL;
    if (rows > 0x7fffffffLL) return PIKA_ETOOBIG;
    if (wave_row_ok(lp, g, cols, ld))
        PIKA_WQ(cols, hipLaunchKernelGGL(log_softmax_bwd_wave_kernel<WQ>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                                         static_cast<hipStream_t>(stream), lp, g, rows, cols, ld, scale));
    else
        hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((unsigned)rows), dim3(ROW_THREADS), 0,
                           static_cast<hipStream_t>(stream), lp, g, cols, ld, scale);
    return (int)hipGetLastError();
}

int pika_mbr_risk_grad_rows(float *lp, const int *sym, const float *val, long long rows, int cols,
                            long long ld, float scale, void *stream) {
    if (!lp || !sym || !val || rows <= 0 || cols <= 0 || ld < cols) return PIKA_EINVAL;
    if (rows > 0x7fffffffLL) return PIKA_ETOOBIG;
    hipLaunchKernelGGL(mbr_risk_grad_kernel, dim3((unsigned)rows), dim3(ROW_THREADS), 0,
                       static_cast<hipStream_t>(stream), lp, sym, val, cols, ld, scale);
    return (int)hipGetLastError();
}

int pika_log_softmax_bwd_rows_bf16(const float *lp, const float *g, void *out, long long rows,
                                   int cols, long long ld, long long ld_out, float scale,
                                   void *stream) {
    if (!lp || !g || !out || rows <= 0 || cols <= 0 || ld < cols || ld_out < cols) return PIKA_EINVAL;
    if (!wave_row_ok(lp, g, cols, ld) || (ld_out & 3) || ld_out > 64 * 4 * WQ_MAX ||
        (reinterpret_cast<uintptr_t>(out) & 7))
        return PIKA_EINVAL;
    if (rows > 0x7fffffffLL) return PIKA_ETOOBIG;
    PIKA_WQ(ld_out, hipLaunchKernelGGL(log_softmax_bwd_bf16_kernel<WQ>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                                       static_cast<hipStream_t>(stream), lp, g, static_cast<__bf16 *>(out), rows, cols,
                                       ld, ld_out, scale));
    return (int)hipGetLastError();
}

}  // extern "C"

// host-side Levenshtein distances (include/pika_joint.h): two rolling rows per pair
extern "C" int pika_edit_distances(const int *seqs, const long long *a_off, const int *a_len, const long long *b_off,
                                   const int *b_len, int n_pairs, int *out) {
    if (!seqs || !a_off || !a_len || !b_off || !b_len || !out || n_pairs < 0) return PIKA_EINVAL;
    int cap = 0;
    for (int i = 0; i < n_pairs; ++i) {
        if (a_len[i] < 0 || b_len[i] < 0) return PIKA_EINVAL;
        if (b_len[i] > cap) cap = b_len[i];
    }
    int *row = new int[2 * (size_t)(cap + 1)];
    for (int i = 0; i < n_pairs; ++i) {
        const int *a = seqs + a_off[i], *b = seqs + b_off[i];
        const int na = a_len[i], nb = b_len[i];
        int *prev = row, *cur = row + cap + 1;
        for (int j = 0; j <= nb; ++j) prev[j] = j;
        for (int x = 1; x <= na; ++x) {
            cur[0] = x;
            for (int j = 1; j <= nb; ++j) {
                const int sub = prev[j - 1] + (a[x - 1] != b[j - 1]);
                const int del = prev[j] + 1, ins = cur[j - 1] + 1;
                cur[j] = sub < del ? (sub < ins ? sub : ins) : (del < ins ? del : ins);
            }
            int *t = prev; prev = cur; cur = t;
        }
        out[i] = prev[nb];
    }
    delete[] row;
    return PIKA_OK;
}
