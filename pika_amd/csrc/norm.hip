// pika_amd/csrc/norm.hip -- BatchNorm (training mode) over (rows, C) for gfx950 (include/pika_norm.h).
// Reductions: grid (C/64 column slabs, row chunks), 256 threads = 4 row-groups x 64 columns (lanes
// along the channel axis: 256-byte coalesced rows), fp32 partial sums per thread, one fp64
// atomicAdd per column per block.  Elementwise passes: 16-byte accesses, channel parameters from
// L2.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "pika_gemm.h"
#include "pika_norm.h"
#include "pika_rnnt.h"
#include "pika_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// rows per reduction workgroup (PIKA_BN_RPB overrides for A/B runs): fewer rows = more workgroups in flight on a pass
// that is all memory round trips, but more fp64 atomics per column
// tools/bn_bench.py at 31808 x 1024: statistics 25 / 32 / 48 / 66 us for 512 / 256 / 128 / 64 rows, backward (two input
// streams, longer chains) 107 / 91 / 101 / 121 us
inline int bn_rows_per_block(int mode) {
    static const int v = [] { const char *e = pika_knob("PIKA_BN_RPB"); const int x = e ? atoi(e) : 0; return x >= 16 ? x : 0; }();
    return v ? v : (mode ? 256 : 512);
}

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ inline f32x4 ld4(const T *p) {
    if constexpr (sizeof(T) == 4) {
        return *reinterpret_cast<const f32x4 *>(p);
    } else {
        const bf16x4_t v = *reinterpret_cast<const bf16x4_t *>(p);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
}
template <typename T>
__device__ inline void st4(T *p, f32x4 v) {
    if constexpr (sizeof(T) == 4) *reinterpret_cast<f32x4 *>(p) = v;
    else *reinterpret_cast<bf16x4_t *>(p) = __builtin_convertvector(v, bf16x4_t);
}

// Rows that count (pika_bn_valid_t, include/pika_norm.h): row r = b * rpb + t is valid iff t < (*t_valid - sub) / div.
// rpb == 0: every row is valid.  Passed by value to the kernels; the valid length is read from the device word, so a
// launch sequence captured once serves every length that fits its padded shape.
struct BnValid {
    const int *t_valid;
    int rpb, sub, div;
    __device__ inline int len() const {
        if (!rpb) return 0x7fffffff;
        const int v = (*t_valid - sub) / div;
        return v < 0 ? 0 : (v > rpb ? rpb : v);
    }
    __device__ inline bool ok(long long r, int vlen) const { return !rpb || (int)(r % rpb) < vlen; }
};

// MODE 0: (x, x^2).  MODE 1: (dy, dy*xhat).  Every lane owns 4 consecutive channels (16-byte loads), a wave
// covers 256 channels of a row, the 4 waves of a block take rows r, r+1, r+2, r+3; four row-steps in flight.
// `a` is f32 or bf16 (an incoming gradient may be bf16), x always f32.
template <int MODE, typename TA>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const TA *__restrict__ a,
                                                        const float *__restrict__ x,
                                                        const float *__restrict__ mean,
                                                        const float *__restrict__ rstd,
                                                        long long rows, int C, int rows_per_block,
                                                        double *__restrict__ out, BnValid V) {
    __shared__ f32x4 p0[4][64], p1[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int vlen = V.len();
    const int c = (blockIdx.x * 64 + lane) * 4;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = mu;
        if (MODE) { mu = *reinterpret_cast<const f32x4 *>(mean + c); rs = *reinterpret_cast<const f32x4 *>(rstd + c); }
        long long r = r0 + g;
        for (; r + 12 < r1; r += 16) {
            f32x4 v[4], w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = ld4(a + (r + 4 * q) * C + c);
                if (MODE) w[q] = *reinterpret_cast<const f32x4 *>(x + (r + 4 * q) * C + c);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!V.ok(r + 4 * q, vlen)) continue;       // rows beyond the valid length do not count
                s0 += v[q];
                s1 += MODE ? v[q] * ((w[q] - mu) * rs) : v[q] * v[q];
            }
        }
        for (; r < r1; r += 4) {
            if (!V.ok(r, vlen)) continue;
            const f32x4 v = ld4(a + r * C + c);
            s0 += v;
            if (MODE) s1 += v * ((*reinterpret_cast<const f32x4 *>(x + r * C + c) - mu) * rs);
            else s1 += v * v;
        }
    }
    p0[g][lane] = s0;
    p1[g][lane] = s1;
    __syncthreads();
    if (g == 0 && c < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(out + c + e, (double)p0[0][lane][e] + (double)p0[1][lane][e] + (double)p0[2][lane][e] + (double)p0[3][lane][e]);
            atomicAdd(out + C + c + e, (double)p1[0][lane][e] + (double)p1[1][lane][e] + (double)p1[2][lane][e] + (double)p1[3][lane][e]);
        }
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double *__restrict__ stats,
                                                          long long rows, int C, float eps,
                                                          float momentum, float *running_mean,
                                                          float *running_var, float *save_mean,
                                                          float *save_rstd, BnValid V) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    if (V.rpb) rows = (rows / V.rpb) * (long long)V.len();      // the rows that were summed
    if (rows < 1) rows = 1;
    const double mean = stats[c] / (double)rows;
    double var = stats[C + c] / (double)rows - mean * mean;
    if (var < 0) var = 0;
    save_mean[c] = (float)mean;
    save_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

template <typename TO>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, long long n4,
                                                       int C4, const float *__restrict__ mean,
                                                       const float *__restrict__ rstd,
                                                       const float *__restrict__ gamma,
                                                       const float *__restrict__ beta,
                                                       TO *__restrict__ y, TO *__restrict__ y_lo, BnValid V) {
    const long long stride = (long long)gridDim.x * 256;
    const int vlen = V.len();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int c = (int)(i % C4);
        const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
        const f32x4 m = reinterpret_cast<const f32x4 *>(mean)[c], r = reinterpret_cast<const f32x4 *>(rstd)[c];
        const f32x4 g = reinterpret_cast<const f32x4 *>(gamma)[c], b = reinterpret_cast<const f32x4 *>(beta)[c];
        f32x4 o = (v - m) * r * g + b;
        if (!V.ok(i / C4, vlen)) o = f32x4{0.f, 0.f, 0.f, 0.f};      // padding rows: zeros (finite, and the same every time)
        st4(y + 4 * i, o);
        if constexpr (sizeof(TO) == 2) {     // two-term output: the second plane holds what the bf16 rounding dropped
            if (y_lo) st4(y_lo + 4 * i, o - __builtin_convertvector(__builtin_convertvector(o, bf16x4_t), f32x4));
        }
    }
}

template <typename TD, typename TX>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const TD *__restrict__ dy,
                                                           const float *__restrict__ x, long long n4,
                                                           int C4, long long rows,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ rstd,
                                                           const float *__restrict__ gamma,
                                                           const double *__restrict__ sums,
                                                           TX *__restrict__ dx, int relu_mask, BnValid V) {
    const long long stride = (long long)gridDim.x * 256;
    const int vlen = V.len();
    if (V.rpb) rows = (rows / V.rpb) * (long long)vlen;
    const float inv = 1.0f / (float)(rows < 1 ? 1 : rows);
    const int C = C4 * 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int c = (int)(i % C4);
        const f32x4 d = ld4(dy + 4 * i), v = reinterpret_cast<const f32x4 *>(x)[i];
        const f32x4 m = reinterpret_cast<const f32x4 *>(mean)[c], r = reinterpret_cast<const f32x4 *>(rstd)[c];
        const f32x4 g = reinterpret_cast<const f32x4 *>(gamma)[c];
        f32x4 sd, sx;
#pragma unroll
        for (int e = 0; e < 4; ++e) { sd[e] = (float)sums[4 * c + e]; sx[e] = (float)sums[C + 4 * c + e]; }
        const f32x4 xhat = (v - m) * r;
        f32x4 o = g * r * (d - sd * inv - xhat * (sx * inv));
        if (relu_mask) {
            o.x = v.x > 0.f ? o.x : 0.f; o.y = v.y > 0.f ? o.y : 0.f;
            o.z = v.z > 0.f ? o.z : 0.f; o.w = v.w > 0.f ? o.w : 0.f;
        }
        if (!V.ok(i / C4, vlen)) o = f32x4{0.f, 0.f, 0.f, 0.f};      // no gradient into padding rows
        st4(dx + 4 * i, o);
    }
}

__global__ __launch_bounds__(256) void bn_param_grad_kernel(const double *__restrict__ sums, int C,
                                                            float *__restrict__ dgamma,
                                                            float *__restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    dbeta[c] = (float)sums[c];
    dgamma[c] = (float)sums[C + c];
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dimension (C <= 2048, C % 4 == 0): one wavefront per row, the row lives
// in registers (LNQ float4 per lane), statistics by xor-shuffles.  Output / incoming gradient may
// be bf16: the normalised activation only feeds MFMA products (q/k/v and feed-forward GEMMs), so it
// is rounded once here instead of in a separate pass.
constexpr int LNQ_MAX = 8;   // float4 per lane: C <= 64 * 4 * 8.  The kernels are instantiated for 2 / 4 / 8 (C <= 512 /
                           // 1024 / 2048): at C = 1024 the row arrays of the backward take 80 registers instead of 160 --
                           // four waves per SIMD instead of two on kernels that are chains of row-sized memory round trips
typedef float ln_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 ln_b4 __attribute__((ext_vector_type(4)));

__device__ inline float ln_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename TO, int LNQ>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, TO *__restrict__ y,
                                                     TO *__restrict__ y_lo,
                                                     float *__restrict__ mean, float *__restrict__ rstd,
                                                     long long rows, int C, float eps) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63, c4 = C >> 2;
    const ln_f4 *xr = reinterpret_cast<const ln_f4 *>(x + r * C);
    ln_f4 v[LNQ];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < LNQ; ++q)
        if (lane + q * 64 < c4) {
            v[q] = xr[lane + q * 64];
            s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
        }
    const float mu = ln_wave_sum(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < LNQ; ++q)
        if (lane + q * 64 < c4) {
            const ln_f4 d = v[q] - mu;
            ss += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
    const float rs = rsqrtf(ln_wave_sum(ss) / (float)C + eps);
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
#pragma unroll
    for (int q = 0; q < LNQ; ++q) {
        const int i = lane + q * 64;
        if (i < c4) {
            const ln_f4 g = reinterpret_cast<const ln_f4 *>(gamma)[i], b = reinterpret_cast<const ln_f4 *>(beta)[i];
            const ln_f4 o = (v[q] - mu) * rs * g + b;
            if constexpr (sizeof(TO) == 4) {
                reinterpret_cast<ln_f4 *>(y + r * C)[i] = o;
            } else {
                const ln_b4 hi = __builtin_convertvector(o, ln_b4);
                reinterpret_cast<ln_b4 *>(y + r * C)[i] = hi;
                if (y_lo)   // two-term output: the second plane holds what the bf16 rounding dropped
                    reinterpret_cast<ln_b4 *>(y_lo + r * C)[i] = __builtin_convertvector(o - __builtin_convertvector(hi, ln_f4), ln_b4);
            }
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  dgamma += dy * xhat, dbeta += dy
// (per-lane column partials over the block's rows, one atomicAdd per column per block).
// PART: the block's column sums go to partials[block][gamma | beta][C] instead (ln_param_grad_kernel adds them up in block
// order): 497 workgroups x 1024 atomics onto the same 1024 words were half of this kernel's time at 31808 x 512.
template <typename TD, int LNQ, bool PART>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TD *__restrict__ dy, const float *__restrict__ x,
                                                     const float *__restrict__ gamma, const float *__restrict__ mean,
                                                     const float *__restrict__ rstd, float *__restrict__ dx,
                                                     float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                     long long rows, int C, int rows_per_block,
                                                     const float *__restrict__ add) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c4 = C >> 2;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    ln_f4 gm[LNQ], ag[LNQ], ab[LNQ];
#pragma unroll
    for (int q = 0; q < LNQ; ++q) {
        ag[q] = ab[q] = ln_f4{0.f, 0.f, 0.f, 0.f};
        if (lane + q * 64 < c4) gm[q] = reinterpret_cast<const ln_f4 *>(gamma)[lane + q * 64];
    }
    for (long long r = r0 + wave; r < r1; r += 4) {
        const float mu = mean[r], rs = rstd[r];
        ln_f4 xh[LNQ], g[LNQ];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < LNQ; ++q) {
            const int i = lane + q * 64;
            if (i < c4) {
                ln_f4 d;
                if constexpr (sizeof(TD) == 4) {
                    d = reinterpret_cast<const ln_f4 *>(dy + r * C)[i];
                } else {
                    const ln_b4 t = reinterpret_cast<const ln_b4 *>(dy + r * C)[i];
                    d = ln_f4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
                }
                xh[q] = (reinterpret_cast<const ln_f4 *>(x + r * C)[i] - mu) * rs;
                g[q] = d * gm[q];
                ag[q] += d * xh[q];
                ab[q] += d;
                s1 += (g[q].x + g[q].y) + (g[q].z + g[q].w);
                const ln_f4 gx = g[q] * xh[q];
                s2 += (gx.x + gx.y) + (gx.z + gx.w);
            }
        }
        const float m1 = ln_wave_sum(s1) / (float)C, m2 = ln_wave_sum(s2) / (float)C;
#pragma unroll
        for (int q = 0; q < LNQ; ++q) {
            const int i = lane + q * 64;
            if (i < c4) {
                ln_f4 v = (g[q] - m1 - xh[q] * m2) * rs;
                if (add) v += reinterpret_cast<const ln_f4 *>(add + r * C)[i];      // the gradient of the skip connection around the LN
                reinterpret_cast<ln_f4 *>(dx + r * C)[i] = v;
            }
        }
    }
    __shared__ ln_f4 red[2][4][64];
#pragma unroll
    for (int q = 0; q < LNQ; ++q) {
        const bool any = q * 64 < c4;     // uniform
        __syncthreads();
        if (any) {
            red[0][wave][lane] = ag[q];
            red[1][wave][lane] = ab[q];
        }
        __syncthreads();
        const int i = lane + q * 64;
        if (any && wave < 2 && i < c4) {
            const ln_f4 t = red[wave][0][lane] + red[wave][1][lane] + red[wave][2][lane] + red[wave][3][lane];
            if constexpr (PART) {       // (dgamma = the partials here)
                reinterpret_cast<ln_f4 *>(dgamma + ((long long)blockIdx.x * 2 + wave) * C)[i] = t;
            } else {
                float *dst = (wave == 0 ? dgamma : dbeta) + 4 * i;
                atomicAdd(dst + 0, t.x); atomicAdd(dst + 1, t.y); atomicAdd(dst + 2, t.z); atomicAdd(dst + 3, t.w);
            }
        }
    }
}

// dgamma / dbeta from the per-block partials of ln_bwd_kernel<PART>: 16 groups of 64 lanes take the blocks round-robin, the
// groups' sums meet in LDS in group order -- the same value whatever the timing.  grid (ceil(C/4/64), 2), 1024 threads.
__global__ __launch_bounds__(1024) void ln_param_grad_kernel(const float *__restrict__ partials, int nb, int C,
                                                             float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ ln_f4 red[16][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6, which = blockIdx.y, i = blockIdx.x * 64 + lane, c4 = C >> 2;
    ln_f4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = ln_f4{0.f, 0.f, 0.f, 0.f};
    if (i < c4) {
        int b = rg;
        for (; b + 48 < nb; b += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                a[u] += reinterpret_cast<const ln_f4 *>(partials + ((long long)(b + 16 * u) * 2 + which) * C)[i];
        }
        for (; b < nb; b += 16) a[0] += reinterpret_cast<const ln_f4 *>(partials + ((long long)b * 2 + which) * C)[i];
    }
    red[rg][lane] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (rg == 0 && i < c4) {
        ln_f4 t = red[0][lane];
#pragma unroll
        for (int g = 1; g < 16; ++g) t += red[g][lane];
        reinterpret_cast<ln_f4 *>(which == 0 ? dgamma : dbeta)[i] = t;
    }
}

inline dim3 red_grid(long long rows, int C, int mode) {
    const int rpb = bn_rows_per_block(mode);
    return dim3((C / 4 + 63) / 64, (unsigned)((rows + rpb - 1) / rpb));
}
inline int ew_grid(long long n4) { return (int)((n4 + 1023) / 1024 < 4096 ? (n4 + 1023) / 1024 : 4096); }

inline bool bn_valid(const pika_bn_valid_t *v, long long rows, BnValid &V) {
    V = BnValid{nullptr, 0, 0, 1};
    if (!v) return true;
    if (!v->t_valid || v->rows_per_batch <= 0 || v->div <= 0 || rows % v->rows_per_batch) return false;
    V = BnValid{v->t_valid, v->rows_per_batch, v->sub, v->div};
    return true;
}

template <typename TD, typename TX>
int bn_backward_impl(const TD *dy, const float *x, long long rows, int C, const float *gamma,
                            const float *save_mean, const float *save_rstd, double *sums, TX *dx,
                            float *dgamma, float *dbeta, int relu_mask, hipStream_t s, BnValid V) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((bn_reduce_kernel<1, TD>), red_grid(rows, C, 1), dim3(256), 0, s, dy, x, save_mean,
                       save_rstd, rows, C, bn_rows_per_block(1), sums, V);
    const long long n4 = rows * C / 4;
    hipLaunchKernelGGL((bn_bwd_apply_kernel<TD, TX>), dim3(ew_grid(n4)), dim3(256), 0, s, dy, x, n4, C / 4, rows,
                       save_mean, save_rstd, gamma, sums, dx, relu_mask, V);
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, C, dgamma,
                       dbeta);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int pika_bn_stats(const float *x, long long rows, int C, double *stats, const pika_bn_valid_t *valid, void *stream) {
    if (!x || !stats || rows <= 0 || C <= 0 || (C & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return PIKA_EINVAL;
    BnValid V;
    if (!bn_valid(valid, rows, V)) return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * 2 * C, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((bn_reduce_kernel<0, float>), red_grid(rows, C, 0), dim3(256), 0, s, x, x, nullptr, nullptr,
                       rows, C, bn_rows_per_block(0), stats, V);
    return (int)hipGetLastError();
}

int pika_bn_apply(const float *x, long long rows, int C, const double *stats, const float *gamma,
                  const float *beta, float eps, float momentum, float *running_mean,
                  float *running_var, float *save_mean, float *save_rstd, void *y, int y_dtype,
                  void *y_lo, const pika_bn_valid_t *valid, void *stream) {
    if (!x || !stats || !gamma || !beta || !save_mean || !save_rstd || !y || rows <= 0 || C <= 0 || (C & 3))
        return PIKA_EINVAL;
    BnValid V;
    if (!bn_valid(valid, rows, V)) return PIKA_EINVAL;
    if (y_dtype != PIKA_F32 && y_dtype != PIKA_BF16) return PIKA_EINVAL;
    if (y_lo && (y_dtype != PIKA_BF16 || (reinterpret_cast<uintptr_t>(y_lo) & 7))) return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, stats, rows, C, eps,
                       momentum, running_mean, running_var, save_mean, save_rstd, V);
    const long long n4 = rows * C / 4;
    if (y_dtype == PIKA_F32)
        hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(ew_grid(n4)), dim3(256), 0, s, x, n4, C / 4, save_mean,
                           save_rstd, gamma, beta, static_cast<float *>(y), static_cast<float *>(nullptr), V);
    else
        hipLaunchKernelGGL(bn_apply_kernel<__bf16>, dim3(ew_grid(n4)), dim3(256), 0, s, x, n4, C / 4, save_mean,
                           save_rstd, gamma, beta, static_cast<__bf16 *>(y), static_cast<__bf16 *>(y_lo), V);
    return (int)hipGetLastError();
}

int pika_bn_backward(const void *dy, int dy_dtype, const float *x, long long rows, int C, const float *gamma,
                     const float *save_mean, const float *save_rstd, double *sums, void *dx, int dx_dtype,
                     float *dgamma, float *dbeta, int relu_mask, const pika_bn_valid_t *valid, void *stream) {
    if (!dy || !x || !gamma || !save_mean || !save_rstd || !sums || !dx || !dgamma || !dbeta ||
        rows <= 0 || C <= 0 || (C & 3))
        return PIKA_EINVAL;
    BnValid V;
    if (!bn_valid(valid, rows, V)) return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float *dyf = static_cast<const float *>(dy);
    const __bf16 *dyb = static_cast<const __bf16 *>(dy);
    float *dxf = static_cast<float *>(dx);
    __bf16 *dxb = static_cast<__bf16 *>(dx);
#define BNB(DY, DX) return bn_backward_impl(DY, x, rows, C, gamma, save_mean, save_rstd, sums, DX, dgamma, dbeta, relu_mask, s, V)
    if (dy_dtype == PIKA_F32 && dx_dtype == PIKA_F32) BNB(dyf, dxf);
    if (dy_dtype == PIKA_F32 && dx_dtype == PIKA_BF16) BNB(dyf, dxb);
    if (dy_dtype == PIKA_BF16 && dx_dtype == PIKA_F32) BNB(dyb, dxf);
    if (dy_dtype == PIKA_BF16 && dx_dtype == PIKA_BF16) BNB(dyb, dxb);
#undef BNB
    return PIKA_EINVAL;
}

int pika_layer_norm_fwd(const float *x, long long rows, int C, const float *gamma, const float *beta,
                        float eps, void *y, int y_dtype, void *y_lo, float *mean, float *rstd, void *stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || C <= 0) return PIKA_EINVAL;
    if (y_lo && (y_dtype != PIKA_BF16 || (reinterpret_cast<uintptr_t>(y_lo) & 7))) return PIKA_EINVAL;
    if ((C & 3) || C > 64 * 4 * LNQ_MAX || rows > 0x7fffffffLL * 4) return PIKA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15)
        return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((rows + 3) / 4));
#define PIKA_LN_FWD(TO, Q) hipLaunchKernelGGL((ln_fwd_kernel<TO, Q>), grid, dim3(256), 0, s, x, gamma, beta, static_cast<TO *>(y), static_cast<TO *>(y_lo), mean, rstd, rows, C, eps)
#define PIKA_LN_FWD_Q(TO) do { if (C <= 512) PIKA_LN_FWD(TO, 2); else if (C <= 1024) PIKA_LN_FWD(TO, 4); else PIKA_LN_FWD(TO, 8); } while (0)
    if (y_dtype == PIKA_F32 && !(reinterpret_cast<uintptr_t>(y) & 15))
        PIKA_LN_FWD_Q(float);
    else if (y_dtype == PIKA_BF16 && !(reinterpret_cast<uintptr_t>(y) & 7))
        PIKA_LN_FWD_Q(__bf16);
    else
        return PIKA_EINVAL;
#undef PIKA_LN_FWD_Q
#undef PIKA_LN_FWD
    return (int)hipGetLastError();
}

static int ln_bwd_rows_per_block(long long rows) {
    static const int rpb_env = [] { const char *e = pika_knob("PIKA_LN_BWD_RPB"); return e ? atoi(e) : 0; }();   // A/B runs
    return rpb_env > 0 ? rpb_env : (rows >= 16384 ? 64 : rows >= 4096 ? 32 : 8);
}

long long pika_layer_norm_bwd_partial_floats(long long rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    const int rpb = ln_bwd_rows_per_block(rows);
    return (rows + rpb - 1) / rpb * 2 * C;
}

int pika_layer_norm_bwd(const void *dy, int dy_dtype, const float *x, long long rows, int C, const float *gamma,
                        const float *mean, const float *rstd, float *dx, float *dgamma, float *dbeta,
                        float *partials, const float *dx_add, void *stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows <= 0 || C <= 0) return PIKA_EINVAL;
    if ((C & 3) || C > 64 * 4 * LNQ_MAX) return PIKA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(dx) |
         reinterpret_cast<uintptr_t>(dgamma) | reinterpret_cast<uintptr_t>(dbeta) | reinterpret_cast<uintptr_t>(partials) |
         reinterpret_cast<uintptr_t>(dx_add)) & 15)
        return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!partials) {
        hipError_t e = hipMemsetAsync(dgamma, 0, (size_t)C * sizeof(float), s);
        if (e == hipSuccess) e = hipMemsetAsync(dbeta, 0, (size_t)C * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    // rows per workgroup (the form without `partials`): every workgroup ends with 2*C column-sum atomics onto the same addresses, so few rows per
    // workgroup are bound by that contention and many by the row chain (measured at 31808 x 1024, tools/ln_bwd_bench.py:
    // 8: 483 us, 16: 299, 32: 231, 64: 197, 128: 215, 256: 286 per backward)
    const int rpb = ln_bwd_rows_per_block(rows);
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb));
#define PIKA_LN_BWD(TD, Q) do { if (partials) hipLaunchKernelGGL((ln_bwd_kernel<TD, Q, true>), grid, dim3(256), 0, s, static_cast<const TD *>(dy), x, gamma, mean, rstd, dx, partials, nullptr, rows, C, rpb, dx_add); \
                                else hipLaunchKernelGGL((ln_bwd_kernel<TD, Q, false>), grid, dim3(256), 0, s, static_cast<const TD *>(dy), x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, rpb, dx_add); } while (0)
#define PIKA_LN_BWD_Q(TD) do { if (C <= 512) PIKA_LN_BWD(TD, 2); else if (C <= 1024) PIKA_LN_BWD(TD, 4); else PIKA_LN_BWD(TD, 8); } while (0)
    if (dy_dtype == PIKA_F32 && !(reinterpret_cast<uintptr_t>(dy) & 15))
        PIKA_LN_BWD_Q(float);
    else if (dy_dtype == PIKA_BF16 && !(reinterpret_cast<uintptr_t>(dy) & 7))
        PIKA_LN_BWD_Q(__bf16);
    else
        return PIKA_EINVAL;
#undef PIKA_LN_BWD_Q
#undef PIKA_LN_BWD
    if (partials)
        hipLaunchKernelGGL(ln_param_grad_kernel, dim3((unsigned)((C / 4 + 63) / 64), 2), dim3(1024), 0, s, partials, (int)grid.x, C,
                           dgamma, dbeta);
    return (int)hipGetLastError();
}

}  // extern "C"
