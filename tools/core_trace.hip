// Time stamps inside the register-staged product loop of the search step's launches (Core<> of decode_step.hip built with
// PIKA_CORE_TRACE): where does a workgroup's time go?  s_memrealtime, 100 MHz (10 ns ticks).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ipika_amd/csrc tools/core_trace.hip -o tools/_core_trace && tools/_core_trace [rows]
#define PIKA_CORE_TRACE
#include "../pika_amd/csrc/decode_step.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill_f32(float *p, long long n, unsigned seed, float amp) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 13;
        p[i] = amp * (((int)(h & 0xffff) - 32768) / 32768.f);
    }
}
static void dump(const char *what, int steps) {
    std::vector<unsigned long long> h(8 * 40 * 8);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_core_trace), h.size() * 8);
    for (int wg : {0, 3}) {
        const unsigned long long *q = &h[wg * 40 * 8];
        const unsigned long long t0 = q[39 * 8 + 0];
        printf("%s, workgroup %d: prologue loads+stage %.2f us, barrier %.2f us\n", what, wg * 8, (q[39 * 8 + 1] - t0) * 0.01,
               (q[39 * 8 + 2] - q[39 * 8 + 1]) * 0.01);
        printf("  step: issue loads | lds reads + mfma | stage A (waits for the loads) | barrier   (us; at = start of step since t0)\n");
        for (int s = 0; s < steps && s < 39; ++s) {
            const unsigned long long *r = q + s * 8;
            if (!r[0]) break;
            printf("  %2d: %5.2f | %5.2f | %5.2f | %5.2f     at %6.2f\n", s, (r[1] - r[0]) * 0.01, (r[2] - r[1]) * 0.01,
                   r[3] ? (r[3] - r[2]) * 0.01 : 0.0, (r[4] - (r[3] ? r[3] : r[2])) * 0.01, (r[0] - t0) * 0.01);
        }
        printf("  product loop ends at %.2f us", (q[(steps - 1) * 8 + 4] - t0) * 0.01);
        if (q[38 * 8 + 2]) printf("; logits -> LDS slab %.2f us, row statistics + top-K %.2f us, workgroup ends at %.2f us",
                                  (q[38 * 8 + 1] - q[38 * 8 + 0]) * 0.01, (q[38 * 8 + 2] - q[38 * 8 + 1]) * 0.01, (q[38 * 8 + 2] - t0) * 0.01);
        printf("\n");
        if (q[37 * 8 + 4]) printf("  first group of 4 rows: slab reads + lane max %.2f | wave max %.2f | exp + wave sum %.2f | bisection %.2f | stores %.2f us\n",
                                  (q[37 * 8 + 0] - q[38 * 8 + 1]) * 0.01, (q[37 * 8 + 1] - q[37 * 8 + 0]) * 0.01, (q[37 * 8 + 2] - q[37 * 8 + 1]) * 0.01,
                                  (q[37 * 8 + 3] - q[37 * 8 + 2]) * 0.01, (q[37 * 8 + 4] - q[37 * 8 + 3]) * 0.01);
    }
    std::vector<unsigned long long> z(8 * 40 * 8, 0);
    hipMemcpyToSymbol(HIP_SYMBOL(g_core_trace), z.data(), z.size() * 8);
}
int main(int argc, char **argv) {
    int rows = argc > 1 ? atoi(argv[1]) : 1024; const int V = 5000, K = 1024, topk = 16;
    float *h, *W, *bias, *pmax, *psum, *C; void *packed4, *pcand;
    CK(hipMalloc(&h, (size_t)rows * K * 4)); CK(hipMalloc(&W, (size_t)V * K * 4)); CK(hipMalloc(&bias, V * 4));
    CK(hipMalloc(&C, (size_t)rows * K * 4));
    hipLaunchKernelGGL(fill_f32, dim3(1024), dim3(256), 0, 0, h, (long long)rows * K, 1u, 1.f);
    hipLaunchKernelGGL(fill_f32, dim3(1024), dim3(256), 0, 0, W, (long long)V * K, 2u, 0.05f);
    hipLaunchKernelGGL(fill_f32, dim3(16), dim3(256), 0, 0, bias, (long long)V, 3u, 1.f);
    CK(hipMalloc(&packed4, pika_dpack_bytes(V, K, 4)));
    if (pika_dpack_weight(W, K, V, K, 4, 0, packed4, nullptr)) return 2;
    const int splits = pika_dfc2_splits(V);
    CK(hipMalloc(&pmax, (size_t)rows * splits * 4)); CK(hipMalloc(&psum, (size_t)rows * splits * 4));
    CK(hipMalloc(&pcand, (size_t)rows * splits * topk * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        int rc = pika_dfc2_topk(h, K, packed4, bias, rows, V, K, 4, 0.8f, topk, pmax, psum, pcand, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("dfc2_topk rows %d: rc %d  %.1f us\n", rows, rc, ms * 1e3);
    }
    dump("dfc2_topk (16 steps of 64 columns)", 16);
    // a 1024-wide projection of the prediction network: dgemm, N = K = 1024
    pika_dgemm_t g{};
    g.A = h; g.lda = K; g.W = packed4; g.C = C; g.ldc = K; g.M = rows; g.N = 1024; g.K = K; g.terms = 4;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        int rc = pika_dgemm(&g, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("dgemm %d x 1024 x 1024: rc %d  %.1f us\n", rows, rc, ms * 1e3);
    }
    dump("dgemm", rows > 256 ? 16 : 8);
    // a gate product of the LAS scoring pass: N = 4096, K = 2048 (dgemm_wide_kernel; PIKA_DGEMM_WIDE / _WD pick the variant)
    {
        const int N2 = 4096, K2 = 2048;
        float *A2, *W2, *C2; void *p2;
        CK(hipMalloc(&A2, (size_t)rows * K2 * 4)); CK(hipMalloc(&W2, (size_t)N2 * K2 * 4)); CK(hipMalloc(&C2, (size_t)rows * N2 * 4));
        hipLaunchKernelGGL(fill_f32, dim3(1024), dim3(256), 0, 0, A2, (long long)rows * K2, 5u, 1.f);
        hipLaunchKernelGGL(fill_f32, dim3(1024), dim3(256), 0, 0, W2, (long long)N2 * K2, 6u, 0.03f);
        CK(hipMalloc(&p2, pika_dpack_bytes(N2, K2, 4)));
        if (pika_dpack_weight(W2, K2, N2, K2, 4, 0, p2, nullptr)) return 2;
        pika_dgemm_t w{};
        w.A = A2; w.lda = K2; w.W = p2; w.C = C2; w.ldc = N2; w.M = rows; w.N = N2; w.K = K2; w.terms = 4;
        if (argc > 2) {      // launch sized for n_max rows, `rows` of them in use (a device word, as in the LAS token loop)
            int *md; CK(hipMalloc(&md, 4)); CK(hipMemcpy(md, &rows, 4, hipMemcpyHostToDevice));
            w.M = atoi(argv[2]); w.m_dev = md;
            CK(hipFree(C2)); CK(hipMalloc(&C2, (size_t)w.M * N2 * 4)); w.C = C2;
        }
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            int rc = pika_dgemm(&w, nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("dgemm %d x 4096 x 2048: rc %d  %.1f us\n", rows, rc, ms * 1e3);
        }
        dump("dgemm wide", 32);
    }
    return 0;
}
