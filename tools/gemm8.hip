// tools/gemm8.hip -- prototype: 256x256x64 bf16 NT GEMM, 8 waves (2 x 4), 128x64 outputs per wave,
// direct global->LDS staging (K-tile double buffer, 128 KB), two wave groups running half a phase
// apart ("ping-pong": while waves 0-3 issue ds_reads / global_load_lds, waves 4-7 -- their SIMD
// neighbours -- run 16 MFMAs, and vice versa).   C[M,N] f32 = A[M,K] bf16 * B[N,K]^T bf16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include <string.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int BM = 256, BN = 256, BK = 64, THREADS = 512;
constexpr int T_BYTES = 256 * 128, BUF = 2 * T_BYTES;

typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef const __attribute__((address_space(1))) unsigned int glb_u32;

#ifndef VARIANT
#define VARIANT 0
#endif
#define BAR() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

#if VARIANT == 0
#define BAR_L() BAR()
#define BAR_M() BAR()
#elif VARIANT == 1      // no barrier after the MFMA cluster
#define BAR_L() BAR()
#define BAR_M() __builtin_amdgcn_sched_barrier(0)
#elif VARIANT == 2      // no barrier after the load segment
#define BAR_L() __builtin_amdgcn_sched_barrier(0)
#define BAR_M() BAR()
#else                   // hazards only: one barrier per K-tile
#define BAR_L() __builtin_amdgcn_sched_barrier(0)
#define BAR_M() __builtin_amdgcn_sched_barrier(0)
#endif
__device__ inline bf16x8 ldsv(const unsigned char *p) { return *reinterpret_cast<const bf16x8 *>(p); }

__global__ __launch_bounds__(THREADS) void gemm8(const __bf16 *__restrict__ A, const __bf16 *__restrict__ B,
                                                 float *__restrict__ C, int M, int N, int K,
                                                 long long lda, long long ldb, long long ldc,
                                                 const float *__restrict__ bias) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int nx = gridDim.x, ntiles = nx * gridDim.y;
    int tile = blockIdx.y * nx + blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (tile / nx) * BM, n0 = (tile % nx) * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;

    // staging: operand tile = 32 pieces of 8 rows x 128 B; wave w owns pieces 4w..4w+3 of A and of B
    const char *pa[4], *pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3), g = (lane & 7) ^ (r & 7);
        int ra = m0 + r, rb = n0 + r;
        ra = ra < M ? ra : M - 1;
        rb = rb < N ? rb : N - 1;
        pa[i] = reinterpret_cast<const char *>(A + (long long)ra * lda) + g * 16;
        pb[i] = reinterpret_cast<const char *>(B + (long long)rb * ldb) + g * 16;
    }
    const int piece0 = wave * 4 * 1024;
    auto gl = [&](const char *p, unsigned char *dst) {
        __builtin_amdgcn_global_load_lds((glb_u32 *)p, (lds_u32 *)dst, 16, 0, 0);
    };

    // fragment addressing: row & 7 == lane & 7 for every fragment row, so the swizzle is per lane
    const int sw0 = (((lane >> 4)) ^ (lane & 7)) << 4, sw1 = sw0 ^ 64;
    const int aoff = (wr * 128 + (lane & 15)) * 128, boff = T_BYTES + (wc * 64 + (lane & 15)) * 128;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = K / BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        gl(pa[i], smem + piece0 + i * 1024);
        gl(pb[i], smem + T_BYTES + piece0 + i * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BAR();
    if (wr == 1) BAR();   // group 1 runs one segment behind group 0

    bf16x8 fa[4][2], fb[4][2];
    for (int t = 0; t < nt; ++t) {
        const unsigned char *cur = smem + (t & 1) * BUF;
        unsigned char *nxt = smem + ((t + 1) & 1) * BUF + piece0;
        const bool pf = t + 1 < nt;
        const long long kadv = (long long)(t + 1) * (BK * 2);
        // ---- phase 0: quadrant (m-half 0, n-half 0)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fb[j][0] = ldsv(cur + boff + j * 2048 + sw0);
            fb[j][1] = ldsv(cur + boff + j * 2048 + sw1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i][0] = ldsv(cur + aoff + i * 2048 + sw0);
            fa[i][1] = ldsv(cur + aoff + i * 2048 + sw1);
        }
        if (pf) {
            gl(pa[0] + kadv, nxt);
            gl(pa[1] + kadv, nxt + 1024);
            gl(pa[2] + kadv, nxt + 2048);
        }
        BAR_L();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa[i][kk], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        BAR_M();
        // ---- phase 1: (m-half 0, n-half 1)
#pragma unroll
        for (int j = 2; j < 4; ++j) {
            fb[j][0] = ldsv(cur + boff + j * 2048 + sw0);
            fb[j][1] = ldsv(cur + boff + j * 2048 + sw1);
        }
        if (pf) {
            gl(pa[3] + kadv, nxt + 3072);
            gl(pb[0] + kadv, nxt + T_BYTES);
            gl(pb[1] + kadv, nxt + T_BYTES + 1024);
            gl(pb[2] + kadv, nxt + T_BYTES + 2048);
            gl(pb[3] + kadv, nxt + T_BYTES + 3072);
        }
        BAR_L();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 2; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa[i][kk], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        BAR_M();
        // ---- phase 2: (m-half 1, n-half 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i][0] = ldsv(cur + aoff + (4 + i) * 2048 + sw0);
            fa[i][1] = ldsv(cur + aoff + (4 + i) * 2048 + sw1);
        }
        BAR_L();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 2; j < 4; ++j)
                    acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa[i][kk], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        BAR_M();
        // ---- phase 3: (m-half 1, n-half 0); the prefetched tile must have landed before the
        // barrier that lets the other group start reading it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa[i][kk], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        BAR_M();
    }
    if (wr == 0) BAR();

#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
            if (n + 3 < N) {
                f32x4 v = acc[i][j];
                if (bias) v += *reinterpret_cast<const f32x4 *>(bias + n);
                *reinterpret_cast<f32x4 *>(C + (long long)m * ldc + n) = v;
            } else {
                for (int e = 0; e < 4; ++e) if (n + e < N) C[(long long)m * ldc + n + e] = acc[i][j][e] + (bias ? bias[n + e] : 0.f);
            }
        }
    }
}


// ---- variant B: k-major phases, fragments of phase p+1 fetched at the start of phase p's MFMA cluster ----
#ifndef BBARS
#define BBARS 1
#endif
#if BBARS
#define BB() BAR()
#else
#define BB() __builtin_amdgcn_sched_barrier(0)
#endif

#define MF16(MH, AR, BR)                                                                            \
    do {                                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                           \
                acc[(MH) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BR[j], AR[i], acc[(MH) * 4 + i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                              \
    } while (0)

__global__ __launch_bounds__(THREADS) void gemm8b(const __bf16 *__restrict__ A, const __bf16 *__restrict__ B,
                                                  float *__restrict__ C, int M, int N, int K,
                                                  long long lda, long long ldb, long long ldc,
                                                  const float *__restrict__ bias) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int nx = gridDim.x, ntiles = nx * gridDim.y;
    int tile = blockIdx.y * nx + blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (tile / nx) * BM, n0 = (tile % nx) * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
    const char *pa[4], *pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3), g = (lane & 7) ^ (r & 7);
        int ra = m0 + r, rb = n0 + r;
        ra = ra < M ? ra : M - 1;
        rb = rb < N ? rb : N - 1;
        pa[i] = reinterpret_cast<const char *>(A + (long long)ra * lda) + g * 16;
        pb[i] = reinterpret_cast<const char *>(B + (long long)rb * ldb) + g * 16;
    }
    const int piece0 = wave * 4 * 1024;
    auto gl = [&](const char *p, unsigned char *dst) {
        __builtin_amdgcn_global_load_lds((glb_u32 *)p, (lds_u32 *)dst, 16, 0, 0);
    };
    const int sw0 = (((lane >> 4)) ^ (lane & 7)) << 4, sw1 = sw0 ^ 64;
    const int aoff = (wr * 128 + (lane & 15)) * 128, boff = T_BYTES + (wc * 64 + (lane & 15)) * 128;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nt = K / BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        gl(pa[i], smem + piece0 + i * 1024);
        gl(pb[i], smem + T_BYTES + piece0 + i * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BAR();
    bf16x8 a0[4], a1[4], b0[4], b1[4];
    // fragments of phase 0 of tile 0
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] = ldsv(smem + aoff + i * 2048 + sw0);
#pragma unroll
    for (int j = 0; j < 4; ++j) b0[j] = ldsv(smem + boff + j * 2048 + sw0);
    if (wr == 1) BAR();

    for (int t = 0; t < nt; ++t) {
        const unsigned char *cur = smem + (t & 1) * BUF;
        const unsigned char *nx_ = smem + ((t + 1) & 1) * BUF;
        unsigned char *nxt = smem + ((t + 1) & 1) * BUF + piece0;
        const bool pf = t + 1 < nt;
        const long long kadv = (long long)(t + 1) * (BK * 2);
        // ---- slot L0: stage the next K-tile
        if (pf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) gl(pa[i] + kadv, nxt + i * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) gl(pb[i] + kadv, nxt + T_BYTES + i * 1024);
        }
        BB();
        // ---- P0 = (m-half 0, k 0..31); fetch A(m1, k0)
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[i] = ldsv(cur + aoff + (4 + i) * 2048 + sw0);
        MF16(0, a0, b0);
        BB();
        BB();
        // ---- P1 = (m1, k0); fetch A(m0, k1), B(k1)
#pragma unroll
        for (int i = 0; i < 4; ++i) a0[i] = ldsv(cur + aoff + i * 2048 + sw1);
#pragma unroll
        for (int j = 0; j < 4; ++j) b1[j] = ldsv(cur + boff + j * 2048 + sw1);
        MF16(1, a1, b0);
        BB();
        // ---- slot L2: the staged tile must have landed before the barrier after which anyone reads it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BAR();
        // ---- P2 = (m0, k1); fetch A(m1, k1)
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[i] = ldsv(cur + aoff + (4 + i) * 2048 + sw1);
        MF16(0, a0, b1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // last reads of this buffer retire before the barrier
        BAR();
        BB();
        // ---- P3 = (m1, k1); fetch the next tile's A(m0, k0), B(k0)
        if (pf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a0[i] = ldsv(nx_ + aoff + i * 2048 + sw0);
#pragma unroll
            for (int j = 0; j < 4; ++j) b0[j] = ldsv(nx_ + boff + j * 2048 + sw0);
        }
        MF16(1, a1, b1);
        BB();
    }
    if (wr == 0) BAR();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
            if (n + 3 < N) {
                f32x4 v = acc[i][j];
                if (bias) v += *reinterpret_cast<const f32x4 *>(bias + n);
                *reinterpret_cast<f32x4 *>(C + (long long)m * ldc + n) = v;
            } else {
                for (int e = 0; e < 4; ++e) if (n + e < N) C[(long long)m * ldc + n + e] = acc[i][j][e] + (bias ? bias[n + e] : 0.f);
            }
        }
    }
}

#ifdef USE_B
#define KERNEL gemm8b
#else
#define KERNEL gemm8
#endif

__global__ void fill_random(unsigned short *p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const float f = ((x & 0xffffff) / 8388608.0f) - 1.0f;   // uniform [-1,1)
        unsigned u; memcpy(&u, &f, 4);
        p[i] = (unsigned short)(u >> 16);
    }
}

static float bf(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fff + ((u >> 16) & 1); u &= 0xffff0000; float r; memcpy(&r, &u, 4); return r; }

int main() {
    CK(hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
    struct T { int M, N, K; } tests[] = {{300, 200, 192}, {256, 256, 64}, {1000, 520, 1024}, {513, 257, 128}};
    for (auto tc : tests) {
        const int M = tc.M, N = tc.N, K = tc.K;
        std::vector<float> a((size_t)M * K), b((size_t)N * K);
        std::vector<unsigned short> ah(a.size()), bh(b.size());
        auto tobits = [](float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); };
        for (size_t i = 0; i < a.size(); ++i) { a[i] = bf(sinf(i * 0.37f) + 0.001f * (i % 97)); ah[i] = tobits(a[i]); }
        for (size_t i = 0; i < b.size(); ++i) { b[i] = bf(cosf(i * 0.11f) * (1 + (i % 13) * 0.1f)); bh[i] = tobits(b[i]); }
        __bf16 *dA, *dB; float *dC;
        CK(hipMalloc(&dA, a.size() * 2)); CK(hipMalloc(&dB, b.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 4));
        CK(hipMemcpy(dA, ah.data(), a.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, bh.data(), b.size() * 2, hipMemcpyHostToDevice));
        double worst = 0;
        for (int rep = 0; rep < 3; ++rep) {   // repeat: a race shows up as run-to-run differences
            CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
            hipLaunchKernelGGL(KERNEL, dim3((N + BN - 1) / BN, (M + BM - 1) / BM), dim3(THREADS), 2 * BUF, 0, dA, dB, dC, M, N, K, (long long)K, (long long)K, (long long)N, (const float *)nullptr);
            CK(hipDeviceSynchronize());
            std::vector<float> c((size_t)M * N);
            CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
            for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
                double s = 0; for (int k = 0; k < K; ++k) s += (double)a[(size_t)m * K + k] * b[(size_t)n * K + k];
                const double e = fabs(s - c[(size_t)m * N + n]);
                worst = fmax(worst, e == e ? e : 1e30);
            }
        }
        printf("correctness %dx%dx%d: max abs err %.3e %s\n", M, N, K, worst, worst < 2e-3 ? "OK" : "WRONG");
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    struct S { const char *name; int M, N, K; } shapes[] = {{"fc2_fwd", 391680, 5000, 1024}, {"fc2_dh ", 391680, 1024, 5056}, {"tdnn_fwd", 31616, 1024, 3072}, {"tdnn_dx", 31616, 3072, 1024}, {"ffn1", 31616, 4096, 1024}, {"sq4096", 4096, 4096, 4096}, {"sq8192", 8192, 8192, 8192}};
    for (auto s : shapes) {
        __bf16 *dA, *dB; float *dC;
        CK(hipMalloc(&dA, (size_t)s.M * s.K * 2)); CK(hipMalloc(&dB, (size_t)s.N * s.K * 2)); CK(hipMalloc(&dC, (size_t)s.M * s.N * 4));
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned short *)dA, (size_t)s.M * s.K, 1u);
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned short *)dB, (size_t)s.N * s.K, 2u);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        dim3 grid((s.N + BN - 1) / BN, (s.M + BM - 1) / BM);
        for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(KERNEL, grid, dim3(THREADS), 2 * BUF, 0, dA, dB, dC, s.M, s.N, s.K, (long long)s.K, (long long)s.K, (long long)s.N, (const float *)nullptr);
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(KERNEL, grid, dim3(THREADS), 2 * BUF, 0, dA, dB, dC, s.M, s.N, s.K, (long long)s.K, (long long)s.K, (long long)s.N, (const float *)nullptr);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%s %dx%dx%d  %.3f ms  %.1f TFLOP/s\n", s.name, s.M, s.N, s.K, ms, 2.0 * s.M * s.N * s.K / ms / 1e9);
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
