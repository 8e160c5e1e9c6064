// Per-CU store throughput of the GEMM epilogue's access pattern: a 256 x 256 fp32 tile written by 8 waves (2 x 4),
// each store instruction covering R rows x (1024 / R) contiguous bytes.  R = 16 is what the MFMA accumulator layout
// gives directly; R = 8 / 4 need a cross-lane exchange first.
// hipcc --offload-arch=gfx950 -O3 tools/write_bw.hip -o tools/_write_bw && tools/_write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int R>
__global__ __launch_bounds__(512) void wr_tile(float *p, long long ld, int reps, int slots) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr_ = wave >> 2, wc = wave & 3;
    constexpr int LPR = 64 / R;                 // lanes per row
    const int row = lane / LPR, col = (lane % LPR) * 4;
    f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (int r = 0; r < reps; ++r) {
        float *base = p + ((long long)blockIdx.x * slots + (r % slots)) * 256 * ld + wc * 64;
        // wave covers 128 rows x 64 columns = 32 KB = 32 instructions
        for (int k = 0; k < 32; ++k) {
            const int seg = k * 64 + lane;                // 16-byte piece index within the wave's block when R == 4
            int rr, cc;
            if (R == 4) { rr = k * 4 + row; cc = col; }
            else if (R == 8) { rr = (k >> 1) * 8 + row; cc = (k & 1) * 32 + col; }
            else { rr = (k >> 2) * 16 + row; cc = (k & 3) * 16 + col; }
            (void)seg;
            *reinterpret_cast<f32x4 *>(base + (long long)(wr_ * 128 + rr) * ld + cc) = v;
        }
    }
}
template <int R>
int run(void *buf, long long ld, int G, hipEvent_t e0, hipEvent_t e1) {
    const int rr = 16, slots = G <= 64 ? 16 : 4;   // few workgroups: every tile is new memory (no L2 write hits)
    hipLaunchKernelGGL(wr_tile<R>, dim3(G), dim3(512), 0, 0, (float *)buf, ld, 1, slots);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(wr_tile<R>, dim3(G), dim3(512), 0, 0, (float *)buf, ld, rr, slots);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("rows/instr %2d  ld %5lld G %4d: per-WG %6.1f GB/s  (%.2f us per 256 KB tile)\n", R, ld, G, 262144.0 * rr / ms / 1e6, ms * 1e3 / rr);
    return 0;
}
int main() {
    const long long total = 6LL << 30;   // 256 WGs x 4 tile slots x 256 rows x 5120 floats x 4 B = 5.4 GB
    void *buf; CK(hipMalloc(&buf, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (long long ld : {5120LL, 5000LL}) for (int G : {8, 32, 256}) {
        if (run<16>(buf, ld, G, e0, e1) || run<8>(buf, ld, G, e0, e1) || run<4>(buf, ld, G, e0, e1)) return 1;
    }
    return 0;
}
