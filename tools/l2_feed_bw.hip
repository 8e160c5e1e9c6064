// How fast can the CUs pull L2-resident operand tiles?  The ping-pong GEMM (gemm_glds.hip) moves 64 KB per K-tile per CU;
// this measures the ceiling of that transport alone, for the three ways of moving it:
//   dma : global_load_lds_dwordx4, 1 KB per wave instruction straight into LDS (what gemm_pp issues)
//   reg : global_load_dwordx4 into registers (no LDS write)
//   regs: global_load_dwordx4 + ds_write_b128
// Every workgroup (512 threads, one per CU with 128 KB of LDS) streams its own `region` bytes again and again (regions of
// all workgroups together fit the L2s), 8 pieces per wave per round in flight.
// hipcc --offload-arch=gfx950 -O3 tools/l2_feed_bw.hip -o tools/_l2_feed_bw && tools/_l2_feed_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef const __attribute__((address_space(1))) unsigned int glb_u32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void feed(const char *src, long long region, int rounds, unsigned *sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char *base = src + (long long)blockIdx.x * region;
    const int pieces = (int)(region >> 10);                 // 1 KB pieces in the region
    u32x4 acc = {0, 0, 0, 0};
    int p = wave * 8;
    for (int r = 0; r < rounds; ++r) {
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const char *q = base + (long long)((p + i) % pieces) * 1024 + lane * 16;
            unsigned char *dst = smem + ((wave * 8 + i) * 1024) + (r & 1) * 65536;
            if (MODE == 0) __builtin_amdgcn_global_load_lds((glb_u32 *)q, (lds_u32 *)dst, 16, 0, 0);
            else v[i] = *reinterpret_cast<const u32x4 *>(q);
        }
        if (MODE == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 2) *reinterpret_cast<u32x4 *>(smem + ((wave * 8 + i) * 1024) + (r & 1) * 65536 + lane * 16) = v[i];
                else acc ^= v[i];
            }
        }
        p += 64;
    }
    if (MODE == 2 || MODE == 0) { __syncthreads(); acc = *reinterpret_cast<u32x4 *>(smem + threadIdx.x * 16); }
    if (acc.x == 0x12345678u) sink[0] = acc.y;               // keep the loads alive
}

template <int MODE>
int run(const char *name, const char *buf, long long region, int G, unsigned *sink, hipEvent_t e0, hipEvent_t e1) {
    const int rounds = 2000;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(feed<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipLaunchKernelGGL(feed<MODE>, dim3(G), dim3(512), 131072, 0, buf, region, 50, sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(feed<MODE>, dim3(G), dim3(512), 131072, 0, buf, region, rounds, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 65536.0 * rounds * G;
    printf("%-5s region %5lld KB  G %4d: %7.2f TB/s chip  %6.1f GB/s per CU  (%.0f ns per 64 KB round)\n", name, region >> 10, G,
           bytes / ms / 1e9, bytes / G / ms / 1e6, ms * 1e6 / rounds);
    return 0;
}

int main() {
    char *buf; unsigned *sink;
    CK(hipMalloc(&buf, 256LL << 20)); CK(hipMemset(buf, 1, 256LL << 20)); CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (long long region : {65536LL, 262144LL, 1048576LL})      // per workgroup: 16 / 64 / 256 MB over 256 workgroups
        for (int G : {256}) {
            if (run<0>("dma", buf, region, G, sink, e0, e1) || run<1>("reg", buf, region, G, sink, e0, e1) ||
                run<2>("regs", buf, region, G, sink, e0, e1)) return 1;
        }
    return 0;
}
