// Time stamps inside the persistent direct-to-LDS GEMM (gemm_pp built with PIKA_PP_TRACE): where does a tile's time go?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/pp_trace.hip -o tools/_pp_trace && tools/_pp_trace [M N K [epi]]
#define PIKA_PP_TRACE
#include "../pika_amd/csrc/gemm_glds.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill_bf16(__bf16 *p, long long n, unsigned seed) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 13;
        p[i] = (__bf16)(((int)(h & 0xffff) - 32768) / 32768.f);
    }
}
int main(int argc, char **argv) {
    const int M = argc > 3 ? atoi(argv[1]) : 196608, N = argc > 3 ? atoi(argv[2]) : 5120, K = argc > 3 ? atoi(argv[3]) : 1024;
    const int epi = argc > 4 ? atoi(argv[4]) : 0;
    __bf16 *A, *B; float *C; unsigned long long *tr;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMalloc(&tr, 8 * 2 * 16 * 24 * 8)); CK(hipMemset(tr, 0, 8 * 2 * 16 * 24 * 8));
    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, A, (long long)M * K, 1u);
    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, B, (long long)N * K, 2u);
    PPArgs P{};
    P.A = A; P.B = B; P.C = C; P.bias = nullptr; P.a_batch = 0; P.a_row = K; P.a_tap = 0; P.ldb = K; P.ldc = N;
    P.a_rpb = M; P.a_C = K; P.a_tin = 1; P.a_t0 = 0; P.a_tstep = 0; P.a_dtap = 0; P.a_bounds = 0; P.M = M; P.N = N; P.K = K;
    float *res = nullptr, *bias = nullptr;
    if (epi == 3) {
        CK(hipMalloc(&res, (size_t)M * N * 4)); CK(hipMemset(res, 0, (size_t)M * N * 4));
        CK(hipMalloc(&bias, (size_t)N * 4)); CK(hipMemset(bias, 0, (size_t)N * 4));
        P.res = res; P.ld_res = N; P.bias = bias; P.thr = 6554; P.scale = 65536.f / (65536.f - 6554.f); P.seed = 7;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) {
        P.trace = it == 2 ? tr : nullptr;
        CK(hipEventRecord(e0));
        int rc = epi == 3 ? launch_pp_epi<3>(P, 0) : launch_pp_epi<0>(P, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rc %d  %.3f ms  %.1f TFLOP/s\n", rc, ms, 2.0 * M * N * K / ms / 1e9);
    }
    std::vector<unsigned long long> h(8 * 2 * 16 * 24);
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    for (int wg : {1}) for (int g = 0; g < 2; ++g) {
        const unsigned long long *q0 = &h[((wg * 2 + g) * 16) * 24];
        int last = 15; while (last > 0 && !q0[last * 24]) --last;
        const double tick_us = last ? 0.01 * (double)(q0[last * 24 + 1] - q0[1]) / (double)(q0[last * 24 + 0] - q0[0]) : 0.0005;   // realtime = 100 MHz
        printf("wg %d group %d  (memtime tick = %.4f us)\n", wg, g, tick_us);
        for (int t = 0; t < 16; ++t) {
            const unsigned long long *q = q0 + t * 24;
            if (!q[0]) break;
            const double nxt = t < 15 ? (double)(q[24] - q[0]) * tick_us : 0.0;
            printf("  tile %2d: k-loop %6.2f us  epilogue issue %5.2f us  tile total %6.2f us\n",
                   t, (q[4] - q[0]) * tick_us, (q[5] - q[4]) * tick_us, nxt);
            if (t == 3 || t == 4) {
                printf("           K-tile 8, cycles from its start to the exit of each of its 8 barriers:");
                for (int e = 0; e < 8; ++e) printf(" %5lld", (long long)(q[8 + e] - q[6]));
                printf("\n");
            }
        }
    }
    return 0;
}
