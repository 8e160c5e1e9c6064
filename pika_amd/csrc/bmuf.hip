// pika_amd/csrc/bmuf.hip -- fused BMUF vector kernels for gfx950.
// Reference math: /root/reference/trainer/bmuf.py:83-98.  HBM-bound streaming kernels over
// the ~90 M-float flat parameter vector (361 MB): 16-byte accesses, grid-stride, 2048 blocks.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pika_bmuf.h"
#include "pika_rnnt.h"  // PIKA_EINVAL

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int BLOCKS = 2048;  // 256 CUs x 8

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <bool VEC>
__global__ __launch_bounds__(256) void bmuf_delta_kernel(const float *__restrict__ g,
                                                         const float *__restrict__ l,
                                                         float *__restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (VEC) {
        const size_t n4 = n >> 2;
        for (size_t k = i; k < n4; k += stride)
            reinterpret_cast<v4f *>(d)[k] =
                reinterpret_cast<const v4f *>(g)[k] - reinterpret_cast<const v4f *>(l)[k];
        for (size_t k = (n4 << 2) + i; k < n; k += stride) d[k] = g[k] - l[k];
    } else {
        for (; i < n; i += stride) d[i] = g[i] - l[i];
    }
}

__global__ __launch_bounds__(256) void bmuf_nan_kernel(const float *__restrict__ d, size_t n,
                                                       int *__restrict__ flag) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        bad |= (d[i] != d[i]);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// Same operation order and roundings as bmuf.py:93-96 run as separate PyTorch ops: FMA
// contraction is switched off so every product is rounded before it is added.
template <typename V>
__device__ inline void upd(V d, V &dp, V &g, V &l, float world, float bm, float blr) {
#pragma clang fp contract(off)
    const V avg = d / world;      // bmuf.py:93 `delta / float(world_size)`: a division, exact for every world size
    const V t1 = dp * bm;
    const V t2 = avg * (blr * (1.0f - bm));
    dp = t1 + t2;
    const V t3 = dp * (1.0f + bm);
    g = g - t3;
    l = g;
}

template <bool VEC>
__global__ __launch_bounds__(256) void bmuf_update_kernel(const float *__restrict__ delta,
                                                          float *__restrict__ dprev,
                                                          float *__restrict__ g,
                                                          float *__restrict__ l, size_t n,
                                                          float world, float bm, float blr,
                                                          const int *__restrict__ skip_flag) {
    if (skip_flag && *skip_flag) return;     // a NaN in the summed delta: leave every vector as it is (bmuf.py:89-90)
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (VEC) {
        const size_t n4 = n >> 2;
        for (size_t k = i; k < n4; k += stride) {
            const v4f d = reinterpret_cast<const v4f *>(delta)[k];
            v4f p = reinterpret_cast<v4f *>(dprev)[k], gg = reinterpret_cast<v4f *>(g)[k], ll;
            upd(d, p, gg, ll, world, bm, blr);
            reinterpret_cast<v4f *>(dprev)[k] = p;
            reinterpret_cast<v4f *>(g)[k] = gg;
            reinterpret_cast<v4f *>(l)[k] = ll;
        }
        for (size_t k = (n4 << 2) + i; k < n; k += stride)
            upd(delta[k], dprev[k], g[k], l[k], world, bm, blr);
    } else {
        for (size_t k = i; k < n; k += stride) upd(delta[k], dprev[k], g[k], l[k], world, bm, blr);
    }
}

// ---- BMUF-Adam: the block update of Adam's two moments (Chen et al. 2020; /root/reference/trainer/bmuf.py:291-313) ------
// x = the summed moment of all ranks (IN: the all-reduced optimizer state; OUT: the block moment, i.e. what the optimizer
// continues from), blk = the block moment kept across blocks:
//   blk = (c1 * blk + c2 * (x / world)) / c3,   c1 = beta^tau (beta^(rho bm) - 1), c2 = 1 - beta^tau beta^(rho bm), c3 = 1 - beta^tau
// every product rounded before it is added (the reference's sequence of separate torch ops).
__global__ __launch_bounds__(256) void bmuf_adam_moments_kernel(float *__restrict__ x1, float *__restrict__ b1,
                                                                float *__restrict__ x2, float *__restrict__ b2, size_t n,
                                                                float world, float c1a, float c2a, float c3a, float c1b,
                                                                float c2b, float c3b, const int *__restrict__ skip_flag) {
#pragma clang fp contract(off)
    if (skip_flag && *skip_flag) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float m = x1[i] / world, v = x2[i] / world;      // bmuf.py:276
        float a = c1a * b1[i];
        a = a + c2a * m;
        a = a / c3a;
        float q = c1b * b2[i];
        q = q + c2b * v;
        q = q / c3b;
        b1[i] = a; x1[i] = a;
        b2[i] = q; x2[i] = q;
    }
}

inline int grid_for(size_t n) {
    const size_t want = (n + 1023) / 1024;
    return (int)(want < 1 ? 1 : (want > BLOCKS ? BLOCKS : want));
}

}  // namespace

extern "C" {

int pika_bmuf_delta(const float *global, const float *local, float *delta, size_t n, void *stream) {
    if (!global || !local || !delta) return PIKA_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (aligned16(global) && aligned16(local) && aligned16(delta))
        hipLaunchKernelGGL(bmuf_delta_kernel<true>, dim3(grid_for(n)), dim3(256), 0, s, global, local, delta, n);
    else
        hipLaunchKernelGGL(bmuf_delta_kernel<false>, dim3(grid_for(n)), dim3(256), 0, s, global, local, delta, n);
    return (int)hipGetLastError();
}

int pika_bmuf_nan_flag(const float *delta, size_t n, int *flag, void *stream) {
    if (!delta || !flag) return PIKA_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(bmuf_nan_kernel, dim3(grid_for(n)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), delta, n, flag);
    return (int)hipGetLastError();
}

int pika_bmuf_update(const float *delta, float *delta_prev, float *global, float *local, size_t n,
                     float world, float block_momentum, float block_lr, const int *skip_flag, void *stream) {
    if (!delta || !delta_prev || !global || !local) return PIKA_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (aligned16(delta) && aligned16(delta_prev) && aligned16(global) && aligned16(local))
        hipLaunchKernelGGL(bmuf_update_kernel<true>, dim3(grid_for(n)), dim3(256), 0, s, delta,
                           delta_prev, global, local, n, world, block_momentum, block_lr, skip_flag);
    else
        hipLaunchKernelGGL(bmuf_update_kernel<false>, dim3(grid_for(n)), dim3(256), 0, s, delta,
                           delta_prev, global, local, n, world, block_momentum, block_lr, skip_flag);
    return (int)hipGetLastError();
}

int pika_bmuf_adam_moments(float *sum_avg, float *blk_avg, float *sum_sq, float *blk_sq, size_t n, float world,
                           float c1_avg, float c2_avg, float c3_avg, float c1_sq, float c2_sq, float c3_sq,
                           const int *skip_flag, void *stream) {
    if (!sum_avg || !blk_avg || !sum_sq || !blk_sq || c3_avg == 0.f || c3_sq == 0.f) return PIKA_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(bmuf_adam_moments_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), sum_avg,
                       blk_avg, sum_sq, blk_sq, n, world, c1_avg, c2_avg, c3_avg, c1_sq, c2_sq, c3_sq, skip_flag);
    return (int)hipGetLastError();
}

}  // extern "C"
