// pika_amd/csrc/gemm_glds.hip -- bf16 x bf16 NT GEMM with direct global->LDS loads for gfx950.
//
// C[M,N] f32 = A[M,K] bf16 * B[N,K]^T bf16 (+ bias[n]).  Used where both operands already live in
// HBM as bf16 (the joint network: hidden h, bf16 weight copies, bf16 d(logits)).  256x128x64 tile,
// 8 waves; every K-step each wave issues 6 global_load_lds_dwordx4 (1 KiB each) straight into the
// other LDS buffer -- no VGPR staging, no conversion -- and the 16-byte granule index is XOR-ed
// with (row & 7) on the SOURCE address (LDS image stays lane-linear, as the instruction
// requires) and again on the fragment read, which makes ds_read_b128 at most 2-way conflicted.
// Measured (tools/glds_gemm.hip, MI355X): 640-770 TFLOP/s on the joint shapes vs 460-500 for the
// register-staged kernel of gemm.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pika_gemm.h"
#include "pika_rnnt.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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


}  // namespace

extern "C" int pika_gemm_bf16_nt(const void *A, long long lda, const void *B, long long ldb, float *C,
                                 long long ldc, int M, int N, int K, const float *bias, void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return PIKA_EINVAL;
    if ((K % BK) || (lda & 7) || (ldb & 7) || (ldc & 3) || ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) |
                                                             reinterpret_cast<uintptr_t>(C)) & 15))
        return PIKA_EINVAL;
    if ((M + BM - 1) / BM > 65535) return PIKA_ETOOBIG;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_glds),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_glds, dim3((N + BN - 1) / BN, (M + BM - 1) / BM), dim3(THREADS), 2 * BUF,
                       static_cast<hipStream_t>(stream), static_cast<const __bf16 *>(A),
                       static_cast<const __bf16 *>(B), C, M, N, K, lda, ldb, ldc, bias);
    return (int)hipGetLastError();
}
