// tools/glds_gemm.hip -- prototype of the bf16 x bf16 NT GEMM with direct global->LDS loads
// (global_load_lds_dwordx4) and a source-side XOR swizzle.  Verifies against a host reference on a
// small shape, then times the joint-network shapes.   C[M,N] f32 = A[M,K] bf16 * B[N,K]^T bf16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include <string.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int BM = 256, BN = 128, BK = 64, THREADS = 512;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, BUF = A_BYTES + B_BYTES;

typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef const __attribute__((address_space(1))) unsigned int glb_u32;

template <int ROWS>
__device__ inline void stage(const __bf16 *__restrict__ src, long long ld, int r0, int nrows, int k0,
                             unsigned char *lds_tile) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < ROWS / 64; ++j) {
        const int rbase = j * 64 + wave * 8;              // 8 rows per wave instruction
        const int r = rbase + (lane >> 3), g = lane & 7;
        int gr = r0 + r;
        gr = gr < nrows ? gr : nrows - 1;                 // clamp: garbage rows are never stored
        const __bf16 *p = src + (long long)gr * ld + k0 + ((g ^ (r & 7)) << 3);
        __builtin_amdgcn_global_load_lds((glb_u32 *)p, (lds_u32 *)(lds_tile + rbase * 128), 16, 0, 0);
    }
}

__device__ inline bf16x8 frag(const unsigned char *tile, int r, int G) {
    return *reinterpret_cast<const bf16x8 *>(tile + r * 128 + ((G ^ (r & 7)) << 4));
}

__global__ __launch_bounds__(THREADS) void gemm_glds(const __bf16 *__restrict__ A, const __bf16 *__restrict__ B,
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = K / BK;
    stage<BM>(A, lda, m0, M, 0, smem);
    stage<BN>(B, ldb, n0, N, 0, smem + A_BYTES);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kb = 0; kb < nk; ++kb) {
        const unsigned char *cur = smem + (kb & 1) * BUF;
        unsigned char *nxt = smem + ((kb + 1) & 1) * BUF;
        if (kb + 1 < nk) {
            stage<BM>(A, lda, m0, M, (kb + 1) * BK, nxt);
            stage<BN>(B, ldb, n0, N, (kb + 1) * BK, nxt + A_BYTES);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = frag(cur, wm * 64 + i * 16 + (lane & 15), kk * 4 + (lane >> 4));
                fb[i] = frag(cur + A_BYTES, wn * 64 + i * 16 + (lane & 15), kk * 4 + (lane >> 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
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

static float bf(float x) __attribute__((unused));
static float bf(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fff + ((u >> 16) & 1); u &= 0xffff0000; float r; memcpy(&r, &u, 4); return r; }

int main() {
    hipFuncSetAttribute((const void *)gemm_glds, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
    // ---- correctness on a ragged shape ----
    {
        const int M = 300, N = 200, K = 192;
        std::vector<float> a(M * K), b(N * K);
        std::vector<unsigned short> ah(M * K), bh(N * K);
        auto tobits = [](float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); };
        for (int i = 0; i < M * K; ++i) { a[i] = bf(sinf(i * 0.37f) + 0.001f * (i % 97)); ah[i] = tobits(a[i]); }
        for (int i = 0; i < N * K; ++i) { b[i] = bf(cosf(i * 0.11f) * (1 + (i % 13) * 0.1f)); bh[i] = tobits(b[i]); }
        __bf16 *dA, *dB; float *dC;
        CK(hipMalloc(&dA, M * K * 2)); CK(hipMalloc(&dB, N * K * 2)); CK(hipMalloc(&dC, M * N * 4));
        CK(hipMemcpy(dA, ah.data(), M * K * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, bh.data(), N * K * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(gemm_glds, dim3((N + BN - 1) / BN, (M + BM - 1) / BM), dim3(THREADS), 2 * BUF, 0, dA, dB, dC, M, N, K, (long long)K, (long long)K, (long long)N, (const float *)nullptr);
        CK(hipDeviceSynchronize());
        std::vector<float> c(M * N);
        CK(hipMemcpy(c.data(), dC, M * N * 4, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
            double s = 0; for (int k = 0; k < K; ++k) s += (double)a[m * K + k] * b[n * K + k];
            maxerr = fmax(maxerr, fabs(s - c[m * N + n]));
        }
        printf("correctness 300x200x192: max abs err %.3e %s\n", maxerr, maxerr < 1e-3 ? "OK" : "WRONG");
    }
    // ---- timing ----
    struct S { const char *name; int M, N, K; } shapes[] = {{"fc2_fwd", 391680, 5000, 1024}, {"fc2_dh ", 391680, 1024, 5056}, {"tdnn_fwd", 32000, 1024, 3072}};
    for (auto s : shapes) {
        __bf16 *dA, *dB; float *dC;
        CK(hipMalloc(&dA, (size_t)s.M * s.K * 2)); CK(hipMalloc(&dB, (size_t)s.N * s.K * 2)); CK(hipMalloc(&dC, (size_t)s.M * s.N * 4));
        CK(hipMemset(dA, 0x3c, (size_t)s.M * s.K * 2)); CK(hipMemset(dB, 0x3c, (size_t)s.N * s.K * 2));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        dim3 grid((s.N + BN - 1) / BN, (s.M + BM - 1) / BM);
        for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(gemm_glds, grid, dim3(THREADS), 2 * BUF, 0, dA, dB, dC, s.M, s.N, s.K, (long long)s.K, (long long)s.K, (long long)s.N, (const float *)nullptr);
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(gemm_glds, grid, dim3(THREADS), 2 * BUF, 0, dA, dB, dC, s.M, s.N, s.K, (long long)s.K, (long long)s.K, (long long)s.N, (const float *)nullptr);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%s %dx%dx%d  %.3f ms  %.1f TFLOP/s\n", s.name, s.M, s.N, s.K, ms, 2.0 * s.M * s.N * s.K / ms / 1e9);
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
