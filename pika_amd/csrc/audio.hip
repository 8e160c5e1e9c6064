// pika_amd/csrc/audio.hip -- loader front end on gfx950 (include/pika_audio.h).
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "pika_audio.h"
#include "pika_rnnt.h"

namespace {

// ---------------- speed / volume perturbation ----------------------------------------------
__device__ inline double resampled(const short *__restrict__ p, long long n, long long m, long long i) {
    if (m == n) return (double)((float)p[i] * (1.0f / 32768.0f));
    // np.linspace(0, n, m)[i] = i * (n / (m-1)); np.interp clamps beyond the last sample
    const double step = m > 1 ? (double)n / (double)(m - 1) : 0.0;
    const double x = (i == m - 1 && m > 1) ? (double)n : (double)i * step;
    const long long j = (long long)x;
    if (j >= n - 1) return (double)((float)p[n - 1] * (1.0f / 32768.0f));
    const double f0 = (double)((float)p[j] * (1.0f / 32768.0f));
    const double f1 = (double)((float)p[j + 1] * (1.0f / 32768.0f));
    return (f1 - f0) * (x - (double)j) + f0;
}

__global__ __launch_bounds__(256) void perturb_sumsq_kernel(const short *__restrict__ pcm,
                                                            const long long *__restrict__ in_off,
                                                            const long long *__restrict__ out_off,
                                                            double *__restrict__ sumsq) {
    __shared__ double part[4];
    const int b = blockIdx.y;
    const long long n = in_off[b + 1] - in_off[b], m = out_off[b + 1] - out_off[b];
    const short *p = pcm + in_off[b];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long long)gridDim.x * 256) {
        const double v = resampled(p, n, m, i);
        s += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sumsq + b, part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void perturb_apply_kernel(const short *__restrict__ pcm,
                                                            const long long *__restrict__ in_off,
                                                            const long long *__restrict__ out_off,
                                                            const double *__restrict__ target_db,
                                                            const double *__restrict__ sumsq,
                                                            float *__restrict__ out) {
    const int b = blockIdx.y;
    const long long n = in_off[b + 1] - in_off[b], m = out_off[b + 1] - out_off[b];
    if (m <= 0) return;
    const short *p = pcm + in_off[b];
    const double ms = fmax(1e-20, sumsq[b] / (double)m);
    const double rms_db = 10.0 * log10(ms);
    const double gain = pow(10.0, fmin(300.0, target_db[b] - rms_db) / 20.0);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long long)gridDim.x * 256) {
        double v;
        if (m == n) {  // unchanged speed: the reference stays in float32 (in-place *= on f32)
            const float f = (float)p[i] * (1.0f / 32768.0f);
            v = (double)((f * (float)gain) * 32768.0f);
        } else {
            v = resampled(p, n, m, i) * gain * 32768.0;
        }
        v = fmin(fmax(v, -32768.0), 32767.0);
        out[out_off[b] + i] = (float)trunc(v);  // astype(int16) truncates toward zero
    }
}

// ---------------- fbank ---------------------------------------------------------------------
__device__ inline unsigned long long mix64(unsigned long long z) {  // splitmix64 finaliser
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__device__ inline float gauss(unsigned long long seed, unsigned long long idx) {
    const unsigned long long r = mix64(seed ^ mix64(idx));
    const float u1 = ((float)(unsigned)(r >> 40) + 1.0f) * (1.0f / 16777217.0f);
    const float u2 = (float)(unsigned)((r >> 8) & 0xffffff) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
}

// one 256-thread workgroup per frame; nfft <= 1024
__global__ __launch_bounds__(256) void fbank_kernel(
    const float *__restrict__ wave, const long long *__restrict__ wave_off,
    const long long *__restrict__ frame_off, int B, int frame_len, int frame_shift, int nfft,
    int log2n, float preemph, float dither, unsigned long long seed, int num_bins,
    const int *__restrict__ mel_lo, const int *__restrict__ mel_cnt, const int *__restrict__ mel_ptr,
    const float *__restrict__ mel_w, float *__restrict__ feats) {
    __shared__ float re[1024], im[1024];
    __shared__ float red[4];
    const long long frame = blockIdx.x;
    // utterance of this frame: B is small, linear search
    int b = 0;
    while (b + 1 < B && frame >= frame_off[b + 1]) ++b;
    const long long f = frame - frame_off[b];
    const float *src = wave + wave_off[b] + f * frame_shift;
    const int tid = threadIdx.x;
    // load (+dither), mean
    float s = 0.f;
    for (int i = tid; i < nfft; i += 256) {
        float v = 0.f;
        if (i < frame_len) {
            v = src[i];
            if (dither != 0.f) v += dither * gauss(seed, (unsigned long long)(frame * frame_len + i));
            s += v;
        }
        re[i] = v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)frame_len;
    // DC removal + pre-emphasis + Hamming, written bit-reversed into the FFT buffer
    float tmp[4];
    for (int k = 0, i = tid; i < nfft; i += 256, ++k) {
        float v = 0.f;
        if (i < frame_len) {
            const float cur = re[i] - mean;
            const float prev = re[i > 0 ? i - 1 : 0] - mean;
            const float w = 0.54f - 0.46f * __cosf(6.283185307179586f * (float)i / (float)(frame_len - 1));
            v = (cur - preemph * prev) * w;
        }
        tmp[k] = v;
    }
    __syncthreads();
    for (int k = 0, i = tid; i < nfft; i += 256, ++k) {
        const int r = (int)(__brev((unsigned)i) >> (32 - log2n));
        re[r] = tmp[k];
        im[r] = 0.f;
    }
    __syncthreads();
    // radix-2 DIT
    for (int st = 1; st <= log2n; ++st) {
        const int half = 1 << (st - 1);
        for (int t = tid; t < nfft / 2; t += 256) {
            const int grp = t / half, pos = t - grp * half;
            const int i0 = grp * 2 * half + pos, i1 = i0 + half;
            float sn, cs;
            sincospif(-(float)pos / (float)half, &sn, &cs);
            const float xr = re[i1] * cs - im[i1] * sn, xi = re[i1] * sn + im[i1] * cs;
            const float ar = re[i0], ai = im[i0];
            re[i0] = ar + xr; im[i0] = ai + xi;
            re[i1] = ar - xr; im[i1] = ai - xi;
        }
        __syncthreads();
    }
    // power spectrum into re[0..nfft/2)
    for (int i = tid; i < nfft / 2; i += 256) re[i] = re[i] * re[i] + im[i] * im[i];
    __syncthreads();
    for (int m = tid; m < num_bins; m += 256) {
        float e = 0.f;
        const int lo = mel_lo[m], cnt = mel_cnt[m], ptr = mel_ptr[m];
        for (int j = 0; j < cnt; ++j) e += mel_w[ptr + j] * re[lo + j];
        feats[frame * num_bins + m] = __logf(fmaxf(e, FLT_EPSILON));
    }
}

__global__ __launch_bounds__(256) void splice_pad_kernel(const float *__restrict__ feats,
                                                         const long long *__restrict__ frame_off,
                                                         int dim, int lctx, int rctx, int stride,
                                                         int t_max, float *__restrict__ out) {
    const int b = blockIdx.y, t = blockIdx.x;
    const long long n = frame_off[b + 1] - frame_off[b];
    if (n <= 0) return;
    const int len = (int)((n + stride - 1) / stride);
    const int ts = (t < len ? t : len - 1) * stride;
    const int width = dim * (lctx + 1 + rctx);
    float *dst = out + ((long long)b * t_max + t) * width;
    for (int i = threadIdx.x; i < width; i += 256) {
        const int j = i / dim - lctx, c = i % dim;
        long long r = ts + j;
        r = r < 0 ? 0 : (r > n - 1 ? n - 1 : r);
        dst[i] = feats[(frame_off[b] + r) * dim + c];
    }
}

// ---------------------------------------------------------------------------------------------
// Noise / reverberation augmentation on float samples (reference loader/audio.py:174-193, 426-513): sums of
// squares for the RMS, y += a*x, y *= a, and the "same"-mode convolution with a room impulse response.
__global__ __launch_bounds__(256) void aug_sumsq_kernel(const float *__restrict__ x, long long n,
                                                        double *__restrict__ out) {
    __shared__ double part[4];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double v = (double)x[i];
        s += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void aug_axpy_kernel(float *__restrict__ y, const float *__restrict__ x,
                                                       long long n, float a, float b) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = b * y[i] + (x ? a * x[i] : 0.f);
}

// out[i] = sum_k h[k] * x[i + (m-1)/2 - k]  (scipy.signal.fftconvolve(x, h, "same") for len(x) >= len(h)):
// 256 outputs per workgroup, the impulse response streamed through LDS in chunks of 1024 taps together with
// the 1279 input samples the chunk touches; fp64 accumulation (the reference convolves through a double FFT
// of float samples: ~1e-7 relative either way, this keeps the direct sum out of the error budget).
constexpr int CONV_TAPS = 1024;
__global__ __launch_bounds__(256) void aug_convolve_same_kernel(const float *__restrict__ x, long long n,
                                                                const float *__restrict__ h, int m,
                                                                float *__restrict__ out) {
    __shared__ float hs[CONV_TAPS], xs[CONV_TAPS + 256];
    const long long i0 = (long long)blockIdx.x * 256, i = i0 + threadIdx.x;
    const long long c = (m - 1) / 2;
    double acc = 0.0;
    for (int k0 = 0; k0 < m; k0 += CONV_TAPS) {
        const int kc = min(CONV_TAPS, m - k0);
        __syncthreads();
        for (int k = threadIdx.x; k < kc; k += 256) hs[k] = h[k0 + k];
        // inputs needed: index j = i + c - k for i in [i0, i0+256), k in [k0, k0+kc): j in [i0+c-k0-kc+1, i0+c-k0+255]
        const long long jlo = i0 + c - k0 - kc + 1;
        for (int q = threadIdx.x; q < kc + 255; q += 256) {
            const long long j = jlo + q;
            xs[q] = (j >= 0 && j < n) ? x[j] : 0.f;
        }
        __syncthreads();
        // j - jlo = (i - i0) + (kc - 1) - (k - k0)
        const int base = threadIdx.x + kc - 1;
        for (int k = 0; k < kc; ++k) acc += (double)hs[k] * (double)xs[base - k];
    }
    if (i < n) out[i] = (float)acc;
}

}  // namespace

extern "C" {

int pika_audio_perturb(const short *pcm, const long long *in_off, const long long *out_off,
                       const double *target_db, int B, long long max_out, float *out, double *sumsq,
                       void *stream) {
    if (!pcm || !in_off || !out_off || !target_db || !out || !sumsq || B <= 0 || max_out <= 0)
        return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(sumsq, 0, sizeof(double) * B, s);
    if (e != hipSuccess) return (int)e;
    const int gx = (int)((max_out + 256 * 8 - 1) / (256 * 8));
    hipLaunchKernelGGL(perturb_sumsq_kernel, dim3(gx, B), dim3(256), 0, s, pcm, in_off, out_off, sumsq);
    hipLaunchKernelGGL(perturb_apply_kernel, dim3(gx, B), dim3(256), 0, s, pcm, in_off, out_off,
                       target_db, sumsq, out);
    return (int)hipGetLastError();
}

int pika_fbank(const float *wave, const long long *wave_off, const long long *frame_off, int B,
               long long total_frames, int frame_len, int frame_shift, int nfft, float preemph,
               float dither, unsigned long long dither_seed, int num_bins, const int *mel_lo,
               const int *mel_cnt, const int *mel_ptr, const float *mel_w, float *feats,
               void *stream) {
    if (!wave || !wave_off || !frame_off || !mel_lo || !mel_cnt || !mel_ptr || !mel_w || !feats ||
        B <= 0 || frame_len <= 1 || frame_shift <= 0 || num_bins <= 0)
        return PIKA_EINVAL;
    int log2n = 0;
    while ((1 << log2n) < nfft) ++log2n;
    if ((1 << log2n) != nfft || nfft > 1024 || nfft < frame_len || nfft < 2) return PIKA_EINVAL;
    if (total_frames <= 0) return 0;
    if (total_frames > 0x7fffffffLL) return PIKA_ETOOBIG;
    hipLaunchKernelGGL(fbank_kernel, dim3((unsigned)total_frames), dim3(256), 0,
                       static_cast<hipStream_t>(stream), wave, wave_off, frame_off, B, frame_len,
                       frame_shift, nfft, log2n, preemph, dither, dither_seed, num_bins, mel_lo,
                       mel_cnt, mel_ptr, mel_w, feats);
    return (int)hipGetLastError();
}

int pika_splice_pad(const float *feats, const long long *frame_off, int B, int dim, int lctx,
                    int rctx, int stride, int t_max, float *out, void *stream) {
    if (!feats || !frame_off || !out || B <= 0 || dim <= 0 || lctx < 0 || rctx < 0 || stride <= 0 ||
        t_max <= 0)
        return PIKA_EINVAL;
    hipLaunchKernelGGL(splice_pad_kernel, dim3(t_max, B), dim3(256), 0,
                       static_cast<hipStream_t>(stream), feats, frame_off, dim, lctx, rctx, stride,
                       t_max, out);
    return (int)hipGetLastError();
}

int pika_audio_sumsq(const float *x, long long n, double *out, void *stream) {
    if (!x || !out || n <= 0) return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), s);
    if (e != hipSuccess) return (int)e;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(aug_sumsq_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, s, x, n, out);
    return (int)hipGetLastError();
}

int pika_audio_axpby(float *y, const float *x, long long n, float a, float b, void *stream) {
    if (!y || n <= 0) return PIKA_EINVAL;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(aug_axpy_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), y, x, n, a, b);
    return (int)hipGetLastError();
}

int pika_audio_convolve_same(const float *x, long long n, const float *h, int m, float *out, void *stream) {
    if (!x || !h || !out || n <= 0 || m <= 0 || m > n || x == out) return PIKA_EINVAL;
    if ((n + 255) / 256 > 0x7fffffffLL) return PIKA_ETOOBIG;
    hipLaunchKernelGGL(aug_convolve_same_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, n, h, m, out);
    return (int)hipGetLastError();
}

}  // extern "C"
