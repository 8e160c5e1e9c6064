// pika_amd/csrc/optim.hip -- inf-norm gradient clipping + Nesterov SGD over all parameter tensors in three
// launches (include/pika_optim.h).  HBM-bound streaming: one workgroup per chunk of <= 16 K elements of one tensor,
// 16-byte accesses where the tensor's base address allows (torch allocations are 256-byte aligned; views of the BMUF
// flat vector start at arbitrary element offsets, so the head of a chunk may be scalar).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pika_optim.h"
#include "pika_rnnt.h"  // PIKA_EINVAL

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

struct Chunk {
    long long off;
    int len, tensor;
};

__device__ inline Chunk get_chunk(const int *ct, const long long *co, const int *cl) {
    const int c = blockIdx.x;
    return Chunk{co[c], cl[c], ct[c]};
}

// f(ptr + i) for i in [0, n): scalar head until 16-byte alignment, float4 body, scalar tail
template <typename F4, typename F1>
__device__ inline void for_each(uintptr_t addr, int n, F4 f4, F1 f1) {
    const int tid = threadIdx.x;
    int head = (int)(((16 - (addr & 15)) & 15) >> 2);
    if (head > n) head = n;
    if (tid < head) f1(tid);
    const int body = (n - head) >> 2;
    for (int k = tid; k < body; k += 256) f4(head + 4 * k);
    const int tail0 = head + 4 * body;
    if (tid < n - tail0) f1(tail0 + tid);
}

__global__ __launch_bounds__(256) void absmax_kernel(const float *const *__restrict__ gp, const int *__restrict__ ct,
                                                     const long long *__restrict__ co, const int *__restrict__ cl,
                                                     float *__restrict__ out) {
    const Chunk c = get_chunk(ct, co, cl);
    const float *g = gp[c.tensor] + c.off;
    float m = 0.f;
    bool nan = false;
    for_each(reinterpret_cast<uintptr_t>(g), c.len,
             [&](int i) {
                 const v4f v = *reinterpret_cast<const v4f *>(g + i);
                 m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                 nan |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
             },
             [&](int i) { const float v = g[i]; m = fmaxf(m, fabsf(v)); nan |= v != v; });
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    nan = __any(nan);
    // non-negative floats order like their bit patterns; NaN (0x7fc00000) is above every finite value and +inf.
    // ONE atomic per workgroup, and only when it would raise the word: every wave of ~3700 workgroups hitting the one
    // address was a queue of ~15 000 device-scope atomics at ~12 ns each -- 0.19 of this kernel's 0.27 ms over 240 MB
    __shared__ unsigned wave_key[4];
    if ((threadIdx.x & 63) == 0) wave_key[threadIdx.x >> 6] = nan ? 0x7fc00000u : __float_as_uint(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned key = max(max(wave_key[0], wave_key[1]), max(wave_key[2], wave_key[3]));
        if (key > __hip_atomic_load(reinterpret_cast<unsigned *>(out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(reinterpret_cast<unsigned *>(out), key);
    }
}

__global__ __launch_bounds__(256) void scale_kernel(float *const *__restrict__ gp, const int *__restrict__ ct,
                                                    const long long *__restrict__ co, const int *__restrict__ cl,
                                                    const float *__restrict__ total_norm, float max_norm) {
    const float coef = max_norm / (*total_norm + 1e-6f);
    if (!(coef < 1.0f) && coef == coef) return;     // (a NaN norm scales by NaN, as torch does)
    const Chunk c = get_chunk(ct, co, cl);
    float *g = gp[c.tensor] + c.off;
    for_each(reinterpret_cast<uintptr_t>(g), c.len,
             [&](int i) { *reinterpret_cast<v4f *>(g + i) = *reinterpret_cast<const v4f *>(g + i) * coef; },
             [&](int i) { g[i] *= coef; });
}

__global__ __launch_bounds__(256) void sgd_kernel(float *const *__restrict__ pp, const float *const *__restrict__ gp,
                                                  float *const *__restrict__ bp, const int *__restrict__ ct,
                                                  const long long *__restrict__ co, const int *__restrict__ cl,
                                                  float lr, float momentum, int first) {
#pragma clang fp contract(off)                        // round every product like the separate torch ops do
    const Chunk c = get_chunk(ct, co, cl);
    float *p = pp[c.tensor] + c.off;
    const float *g = gp[c.tensor] + c.off;
    float *b = bp[c.tensor] + c.off;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const bool same = ((reinterpret_cast<uintptr_t>(g) ^ a) & 15) == 0 && ((reinterpret_cast<uintptr_t>(b) ^ a) & 15) == 0;
    if (same) {
        for_each(a, c.len,
                 [&](int i) {
                     const v4f gv = *reinterpret_cast<const v4f *>(g + i);
                     v4f bv = first ? gv : *reinterpret_cast<const v4f *>(b + i) * momentum + gv;
                     *reinterpret_cast<v4f *>(b + i) = bv;
                     const v4f d = gv + bv * momentum;
                     *reinterpret_cast<v4f *>(p + i) = *reinterpret_cast<const v4f *>(p + i) - d * lr;
                 },
                 [&](int i) {
                     const float gv = g[i];
                     const float bv = first ? gv : b[i] * momentum + gv;
                     b[i] = bv;
                     p[i] = p[i] - (gv + bv * momentum) * lr;
                 });
    } else {
        for (int i = threadIdx.x; i < c.len; i += 256) {
            const float gv = g[i];
            const float bv = first ? gv : b[i] * momentum + gv;
            b[i] = bv;
            p[i] = p[i] - (gv + bv * momentum) * lr;
        }
    }
}

int check(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

}  // namespace

extern "C" {

int pika_multi_absmax(const float *const *grad_ptrs, const int *chunk_tensor, const long long *chunk_off,
                      const int *chunk_len, int n_chunks, float *out, void *stream) {
    if (!grad_ptrs || !chunk_tensor || !chunk_off || !chunk_len || !out || n_chunks <= 0) return PIKA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(absmax_kernel, dim3(n_chunks), dim3(256), 0, st, grad_ptrs, chunk_tensor, chunk_off, chunk_len, out);
    return check(hipGetLastError());
}

int pika_multi_scale_by_clip(float *const *grad_ptrs, const int *chunk_tensor, const long long *chunk_off,
                             const int *chunk_len, int n_chunks, const float *total_norm, float max_norm,
                             void *stream) {
    if (!grad_ptrs || !chunk_tensor || !chunk_off || !chunk_len || !total_norm || n_chunks <= 0) return PIKA_EINVAL;
    hipLaunchKernelGGL(scale_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, grad_ptrs, chunk_tensor, chunk_off,
                       chunk_len, total_norm, max_norm);
    return check(hipGetLastError());
}

int pika_multi_sgd_nesterov(float *const *param_ptrs, const float *const *grad_ptrs, float *const *buf_ptrs,
                            const int *chunk_tensor, const long long *chunk_off, const int *chunk_len, int n_chunks,
                            float lr, float momentum, int first, void *stream) {
    if (!param_ptrs || !grad_ptrs || !buf_ptrs || !chunk_tensor || !chunk_off || !chunk_len || n_chunks <= 0)
        return PIKA_EINVAL;
    hipLaunchKernelGGL(sgd_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, param_ptrs, grad_ptrs, buf_ptrs,
                       chunk_tensor, chunk_off, chunk_len, lr, momentum, first);
    return check(hipGetLastError());
}

}  // extern "C"
