// pika_amd/csrc/norm.hip -- BatchNorm (training mode) over (rows, C) for gfx950 (include/pika_norm.h).
// Reductions: grid (C/64 column slabs, row chunks), 256 threads = 4 row-groups x 64 columns (lanes
// along the channel axis: 256-byte coalesced rows), fp32 partial sums per thread, one fp64
// atomicAdd per column per block.  Elementwise passes: 16-byte accesses, channel parameters from
// L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pika_norm.h"
#include "pika_rnnt.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS_PER_BLOCK = 512;

// MODE 0: (x, x^2).  MODE 1: (dy, dy*xhat)
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float *__restrict__ a,
                                                        const float *__restrict__ x,
                                                        const float *__restrict__ mean,
                                                        const float *__restrict__ rstd,
                                                        long long rows, int C,
                                                        double *__restrict__ out) {
    __shared__ float p0[4][64], p1[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.y * ROWS_PER_BLOCK;
    const long long r1 = min(rows, r0 + ROWS_PER_BLOCK);
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        const float mu = MODE ? mean[c] : 0.f, rs = MODE ? rstd[c] : 0.f;
        for (long long r = r0 + g; r < r1; r += 4) {
            const float v = a[r * C + c];
            if (MODE == 0) { s0 += v; s1 += v * v; }
            else { s0 += v; s1 += v * ((x[r * C + c] - mu) * rs); }
        }
    }
    p0[g][threadIdx.x & 63] = s0;
    p1[g][threadIdx.x & 63] = s1;
    __syncthreads();
    if (g == 0 && c < C) {
        const int l = threadIdx.x;
        atomicAdd(out + c, (double)p0[0][l] + (double)p0[1][l] + (double)p0[2][l] + (double)p0[3][l]);
        atomicAdd(out + C + c, (double)p1[0][l] + (double)p1[1][l] + (double)p1[2][l] + (double)p1[3][l]);
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double *__restrict__ stats,
                                                          long long rows, int C, float eps,
                                                          float momentum, float *running_mean,
                                                          float *running_var, float *save_mean,
                                                          float *save_rstd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
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

__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, long long n4,
                                                       int C4, const float *__restrict__ mean,
                                                       const float *__restrict__ rstd,
                                                       const float *__restrict__ gamma,
                                                       const float *__restrict__ beta,
                                                       float *__restrict__ y) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int c = (int)(i % C4);
        const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
        const f32x4 m = reinterpret_cast<const f32x4 *>(mean)[c], r = reinterpret_cast<const f32x4 *>(rstd)[c];
        const f32x4 g = reinterpret_cast<const f32x4 *>(gamma)[c], b = reinterpret_cast<const f32x4 *>(beta)[c];
        reinterpret_cast<f32x4 *>(y)[i] = (v - m) * r * g + b;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ dy,
                                                           const float *__restrict__ x, long long n4,
                                                           int C4, long long rows,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ rstd,
                                                           const float *__restrict__ gamma,
                                                           const double *__restrict__ sums,
                                                           float *__restrict__ dx, int relu_mask) {
    const long long stride = (long long)gridDim.x * 256;
    const float inv = 1.0f / (float)rows;
    const int C = C4 * 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int c = (int)(i % C4);
        const f32x4 d = reinterpret_cast<const f32x4 *>(dy)[i], v = reinterpret_cast<const f32x4 *>(x)[i];
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
        reinterpret_cast<f32x4 *>(dx)[i] = o;
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

inline dim3 red_grid(long long rows, int C) {
    return dim3((C + 63) / 64, (unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
}
inline int ew_grid(long long n4) { return (int)((n4 + 1023) / 1024 < 4096 ? (n4 + 1023) / 1024 : 4096); }

}  // namespace

extern "C" {

int pika_bn_stats(const float *x, long long rows, int C, double *stats, void *stream) {
    if (!x || !stats || rows <= 0 || C <= 0) return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * 2 * C, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bn_reduce_kernel<0>, red_grid(rows, C), dim3(256), 0, s, x, x, nullptr, nullptr,
                       rows, C, stats);
    return (int)hipGetLastError();
}

int pika_bn_apply(const float *x, long long rows, int C, const double *stats, const float *gamma,
                  const float *beta, float eps, float momentum, float *running_mean,
                  float *running_var, float *save_mean, float *save_rstd, float *y, void *stream) {
    if (!x || !stats || !gamma || !beta || !save_mean || !save_rstd || !y || rows <= 0 || C <= 0 || (C & 3))
        return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, stats, rows, C, eps,
                       momentum, running_mean, running_var, save_mean, save_rstd);
    const long long n4 = rows * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, s, x, n4, C / 4, save_mean,
                       save_rstd, gamma, beta, y);
    return (int)hipGetLastError();
}

int pika_bn_backward(const float *dy, const float *x, long long rows, int C, const float *gamma,
                     const float *save_mean, const float *save_rstd, double *sums, float *dx,
                     float *dgamma, float *dbeta, int relu_mask, void *stream) {
    if (!dy || !x || !gamma || !save_mean || !save_rstd || !sums || !dx || !dgamma || !dbeta ||
        rows <= 0 || C <= 0 || (C & 3))
        return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bn_reduce_kernel<1>, red_grid(rows, C), dim3(256), 0, s, dy, x, save_mean,
                       save_rstd, rows, C, sums);
    const long long n4 = rows * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, s, dy, x, n4, C / 4, rows,
                       save_mean, save_rstd, gamma, sums, dx, relu_mask);
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, C, dgamma,
                       dbeta);
    return (int)hipGetLastError();
}

}  // extern "C"
