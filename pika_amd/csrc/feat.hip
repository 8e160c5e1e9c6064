// pika_amd/csrc/feat.hip -- CMVN and SpecAugment kernels for gfx950.
// Reference: trainer/train_transducer_bmuf_otfaug.py:86-93, utils/spec_augment.py:10-20.
#include <hip/hip_runtime.h>

#include "pika_feat.h"
#include "pika_rnnt.h"  // PIKA_EINVAL

namespace {

// One workgroup per (utterance, 64-column slab): 16 row-groups x 64 columns = 1024 threads.
// Lanes run along the feature axis (coalesced 256-byte rows), row-groups stride over time.
// Pass 1 accumulates per-column sums (fp32, like torch.mean), LDS tree over the 16 groups;
// pass 2 re-reads the slab (L2-resident: 64 cols x T x 4 B) and writes the normalised values.
constexpr int COLS = 64, GROUPS = 16;

__global__ __launch_bounds__(COLS * GROUPS) void cmvn_kernel(float *__restrict__ x, int T, int F,
                                                            const float *__restrict__ offset,
                                                            const float *__restrict__ scale,
                                                            int cmn) {
    __shared__ float part[GROUPS][COLS];
    const int b = blockIdx.y;
    const int c = threadIdx.x & (COLS - 1), g = threadIdx.x / COLS;
    const int f = blockIdx.x * COLS + c;
    float *xb = x + (size_t)b * T * F;
    float mean = 0.f;
    if (cmn) {
        float s = 0.f;
        if (f < F)
            for (int t = g; t < T; t += GROUPS) s += xb[(size_t)t * F + f];
        part[g][c] = s;
        __syncthreads();
        for (int h = GROUPS / 2; h > 0; h >>= 1) {
            if (g < h) part[g][c] += part[g + h][c];
            __syncthreads();
        }
        mean = part[0][c] / (float)T;
    }
    if (f >= F) return;
    const float o = offset ? offset[f] : 0.f, sc = scale ? scale[f] : 1.f;
    for (int t = g; t < T; t += GROUPS) {
        const size_t i = (size_t)t * F + f;
        xb[i] = ((xb[i] - mean) + o) * sc;
    }
}

// grid (ceil(T/rows), B), block 256: each thread-row sweeps one (b,t) feature row.
__global__ __launch_bounds__(256) void specaug_kernel(float *__restrict__ x, int T, int F, int f0,
                                                      int fs, int t0, int ts) {
    const int b = blockIdx.y;
    const int rows_per_block = 256 / 64;
    const int t = blockIdx.x * rows_per_block + (threadIdx.x >> 6);
    if (t >= T) return;
    const int lane = threadIdx.x & 63;
    float *row = x + ((size_t)b * T + t) * F;
    if (t >= t0 && t < t0 + ts) {
        for (int f = lane; f < F; f += 64) row[f] = 0.f;
    } else {
        for (int f = f0 + lane; f < f0 + fs; f += 64) row[f] = 0.f;
    }
}

}  // namespace

extern "C" {

int pika_cmvn_apply(float *x, int B, int T, int F, const float *offset, const float *scale, int cmn,
                    void *stream) {
    if (!x || B <= 0 || T <= 0 || F <= 0) return PIKA_EINVAL;
    hipLaunchKernelGGL(cmvn_kernel, dim3((F + COLS - 1) / COLS, B), dim3(COLS * GROUPS), 0,
                       static_cast<hipStream_t>(stream), x, T, F, offset, scale, cmn);
    return (int)hipGetLastError();
}

int pika_specaug_apply(float *x, int B, int T, int F, int f0, int fs, int t0, int ts, void *stream) {
    if (!x || B <= 0 || T <= 0 || F <= 0 || f0 < 0 || fs < 0 || t0 < 0 || ts < 0 || f0 + fs > F ||
        t0 + ts > T)
        return PIKA_EINVAL;
    if (fs == 0 && ts == 0) return 0;
    hipLaunchKernelGGL(specaug_kernel, dim3((T + 3) / 4, B), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, T, F, f0, fs, t0, ts);
    return (int)hipGetLastError();
}

}  // extern "C"
