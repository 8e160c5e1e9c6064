// Prototype: C[M,N] f32 = A[M,K] bf16 . B[N,K]^T bf16 with ONE wave per SIMD.
//   256 x 256 x 32 tile, 256 threads = 4 waves (2 x 2), 128 x 128 outputs per wave (256 accumulator registers),
//   4 LDS stages of 32 KB filled by global_load_lds (prefetch distance 4 K-tiles), ONE barrier per K-tile, the
//   fragment reads of K-tile t+1 and the LDS-DMA pieces of K-tile t+4 interleaved with the 64 MFMAs of K-tile t.
// Why: tools/l2_feed_bw.hip shows the L2 can feed a CU its 64 KB per 64-deep K-tile in 0.58 us (28 TB/s chip) against
// 1.1 us of MFMA issue, yet the 8-wave ping-pong kernel needs 1.68 us: its segments are latency chains (8 pieces in
// flight per CU).  A wave that owns its SIMD keeps 24 pieces in flight and never waits at a mid-tile barrier.
// hipcc --offload-arch=gfx950 -O3 tools/gemm4w.hip -o tools/_gemm4w && tools/_gemm4w
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef const __attribute__((address_space(1))) unsigned int glb_u32;

constexpr int BK = 32, STAGES = 4, STAGE_BYTES = 2 * 256 * BK * 2;   // A + B tile of one K-tile: 32 KB

#ifndef READ_GROUPS
#define READ_GROUPS 8
#endif

__global__ __launch_bounds__(256, 1) void gemm4w(const __bf16 *__restrict__ A, const __bf16 *__restrict__ B,
                                                 float *__restrict__ C, int M, int N, int K, long long lda,
                                                 long long ldb, long long ldc) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nx = (N + 255) / 256, ntiles = nx * ((M + 255) / 256);
    int tile = blockIdx.x;
    {   // XCD-aware: workgroups that share blockIdx % 8 (one XCD) take a contiguous run of tiles
        const int q = ntiles >> 3, r = ntiles & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // bands of 4 tile rows, column by column inside a band
    const int band = tile / (4 * nx), in = tile - band * (4 * nx);
    const int rows = min(4, ntiles / nx - band * 4);
    const int tn = in / rows, tm = band * 4 + (in - tn * rows);
    const int m0 = tm * 256, n0 = tn * 256;

    // LDS-DMA sources.  A K-tile is 32 pieces of 1 KB (16 rows x 64 B): pieces 0..15 = A rows, 16..31 = B rows; wave w
    // issues pieces w*8 .. w*8+7 (waves 0,1: A; waves 2,3: B).  Within a piece, lane -> row lane >> 2, LDS slot lane & 3,
    // source granule (lane & 3) ^ (-(row >> 2) & 3): the LDS image stays lane-linear, the fragment reads undo the swizzle.
    // (ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: with 64-byte rows the 16 lanes
    // of a group hit 16 distinct 16-byte slots of a 256-byte window iff s(0), s(3), 1^s(1), 1^s(2) are distinct for the
    // row-quad swizzle s: s = [0, 3, 2, 1].)
    const bool is_b = wave >= 2;
    const __bf16 *src = is_b ? B : A;
    const long long ld = is_b ? ldb : lda;
    const int lim = is_b ? N : M, o0 = is_b ? n0 : m0;
    unsigned off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int prow = ((wave & 1) * 8 + i) * 16 + (lane >> 2);      // row inside the 256-row operand tile
        int gr = o0 + prow;
        gr = gr < lim ? gr : lim - 1;
        const int g = (lane & 3) ^ ((0 - (prow >> 2)) & 3);
        off[i] = (unsigned)(((long long)(gr - o0) * ld + g * 8) * 2);
    }
    const char *sbase = reinterpret_cast<const char *>(src + (long long)o0 * ld);
    const int dst0 = (is_b ? 256 * BK * 2 : 0) + (wave & 1) * 8 * 1024;
    const int nt = K / BK;
    // kptr = source of the K-tile being fetched, a SCALAR pointer (laundered so that the compiler does not fold it into
    // loop-invariant 64-bit per-lane pointers: a piece is then `global_load_lds v_off32, s[base]`)
    const char *kptr = sbase;
    auto set_ktile = [&](int kt) {
        const int ks = kt < nt ? kt : nt - 1;          // pieces "beyond the end" re-fetch the last K-tile
        unsigned long long v = reinterpret_cast<unsigned long long>(sbase + (long long)ks * (BK * 2));
        unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        asm volatile("" : "+s"(lo), "+s"(hi));
        kptr = reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
    };
    auto issue = [&](int i, int kt) {   // piece i of this wave for K-tile kt (set_ktile(kt) first)
        __builtin_amdgcn_global_load_lds((glb_u32 *)(kptr + off[i]),
                                         (lds_u32 *)(smem + (kt & (STAGES - 1)) * STAGE_BYTES + dst0 + i * 1024), 16, 0, 0);
    };

    // v_mfma_f32_32x32x16_bf16 fragments (the 16x16x32 form runs at 4/5 of its rate on gfx950): fragment f = 2*blk + ks
    // holds rows blk*32 + (lane & 31) of the wave's 128, k = ks*16 + (lane >> 5)*8 .. +7, i.e. granule g = 2*ks + (lane>>5)
    // of the 64-byte row: byte r*64 + ((g ^ s(r>>2)) << 4)
    int aoff[8], boff[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        const int blk = f >> 1, ks = f & 1, g = 2 * ks + (lane >> 5);
        const int ra = wm * 128 + blk * 32 + (lane & 31), rb = wn * 128 + blk * 32 + (lane & 31);
        aoff[f] = ra * 64 + ((g ^ ((0 - (ra >> 2)) & 3)) << 4);
        boff[f] = 256 * BK * 2 + rb * 64 + ((g ^ ((0 - (rb >> 2)) & 3)) << 4);
    }

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // prologue: K-tiles 0 .. STAGES-1 in flight
#pragma unroll
    for (int s = 0; s < STAGES; ++s) {
        set_ktile(s);
#pragma unroll
        for (int i = 0; i < 8; ++i) issue(i, s);
    }
    bf16x8 fa[8], fb[2][8];
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");     // K-tile 0 of this wave has landed
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fa[i] = *reinterpret_cast<const bf16x8 *>(smem + aoff[i]);
        fb[0][i] = *reinterpret_cast<const bf16x8 *>(smem + boff[i]);
    }

#define BAR4() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    // One K-tile.  CUR: which fragment set holds K-tile t; MORE: K-tile t+1 exists (fetch its fragments); FETCH: K-tile
    // t+STAGES exists (issue its pieces into the buffer K-tile t's fragments were read from one iteration ago).
    auto ktile = [&](int t, auto cur_tag, auto more_tag, auto fetch_tag, auto wait_tag) {
        constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
        constexpr bool more = decltype(more_tag)::value, fetch = decltype(fetch_tag)::value;
        constexpr int wait = decltype(wait_tag)::value;
        // K-tile t+1 must be complete in LDS before its fragments are fetched below: this wave's pieces by count (those of
        // K-tiles t+2, t+3 may stay in flight), everybody's by the barrier -- which also says that every wave is done
        // with the fragment reads of K-tile t (issued one iteration ago), so its buffer can take K-tile t+STAGES
        if constexpr (wait == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if constexpr (wait == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef NO_BAR
        BAR4();
#endif
        const unsigned char *nb = smem + ((t + 1) & (STAGES - 1)) * STAGE_BYTES;
        if constexpr (fetch) set_ktile(t + STAGES);
        // 32 MFMAs of 32x32x16 (32 cycles of matrix pipe each): row block i x column block j x k-step ks; one memory
        // instruction in the shadow of every MFMA but the first of a row block: the A fragments of the row block that has just
        // finished (in place), the next B set, the LDS-DMA pieces of K-tile t+STAGES
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
                                 : "+a"(acc[i][j]) : "v"(fb[cur][2 * j + ks]), "v"(fa[2 * i + ks]));
                    const int slot = j * 2 + ks;      // 0..7 within the row block
                    __builtin_amdgcn_sched_barrier(0);
                    if (slot == 1 || slot == 2) {     // A fragments of row block i-1, K-tile t+1
                        if constexpr (more) { if (i >= 1) fa[2 * (i - 1) + slot - 1] = *reinterpret_cast<const bf16x8 *>(nb + aoff[2 * (i - 1) + slot - 1]); }
                    } else if (slot == 3 || slot == 4) {   // two B fragments of the next set per row block
                        if constexpr (more) fb[nxt][2 * i + slot - 3] = *reinterpret_cast<const bf16x8 *>(nb + boff[2 * i + slot - 3]);
                    } else if (slot == 5 || slot == 6) {   // two pieces per row block
                        if constexpr (fetch) issue(2 * i + slot - 5, t + STAGES);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        if constexpr (more) {
            fa[6] = *reinterpret_cast<const bf16x8 *>(nb + aoff[6]);
            fa[7] = *reinterpret_cast<const bf16x8 *>(nb + aoff[7]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using W16 = std::integral_constant<int, 16>;
    // ONE loop body for every K-tile: pieces "beyond the end" re-fetch the last K-tile into buffers nobody reads again
    // (issue() clamps), so the wait count and the interleave pattern never change -- two code variants (fragment set 0 / 1)
    int t = 0;
#ifdef NO_DSR
    using MORE = std::false_type;
#else
    using MORE = std::true_type;
#endif
#ifdef NO_DMA
    using FETCH = std::false_type;
    using WT = T0;
#else
    using FETCH = std::true_type;
    using WT = W16;
#endif
    for (; t + 1 < nt; t += 2) {
        ktile(t, T0{}, MORE{}, FETCH{}, WT{});
        ktile(t + 1, T1{}, MORE{}, FETCH{}, WT{});
    }
    if (t < nt) ktile(t, T0{}, MORE{}, FETCH{}, WT{});
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + i * 32 + (lane & 31);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
                const int n = n0 + wn * 128 + j * 32 + 8 * b4 + 4 * (lane >> 5);
                const f32x4 v = {acc[i][j][4 * b4], acc[i][j][4 * b4 + 1], acc[i][j][4 * b4 + 2], acc[i][j][4 * b4 + 3]};
                if (n + 3 < N) *reinterpret_cast<f32x4 *>(C + (long long)m * ldc + n) = v;
                else for (int e = 0; e < 4; ++e) if (n + e < N) C[(long long)m * ldc + n + e] = v[e];
            }
    }
}

static float bf(float x) { return (float)(__bf16)x; }

int run(int M, int N, int K, bool check, hipEvent_t e0, hipEvent_t e1) {
    std::vector<__bf16> ha((size_t)M * K), hb((size_t)N * K);
    unsigned s = 12345u + M + N;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto &v : ha) v = (__bf16)rnd();
    for (auto &v : hb) v = (__bf16)(rnd() * 0.05f);
    __bf16 *A, *B; float *C;
    CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&B, hb.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm4w), hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * STAGE_BYTES));
    auto launch = [&]() { hipLaunchKernelGGL(gemm4w, dim3(tiles), dim3(256), STAGES * STAGE_BYTES, 0, A, B, C, M, N, K, (long long)K, (long long)K, (long long)N); };
    launch();
    CK(hipDeviceSynchronize());
    if (check) {
        std::vector<float> hc((size_t)M * N);
        CK(hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int m = 0; m < M; m += (M > 600 ? 37 : 1))
            for (int n = 0; n < N; n += (N > 600 ? 41 : 1)) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)(float)ha[(size_t)m * K + k] * (double)(float)hb[(size_t)n * K + k];
                worst = fmax(worst, fabs(ref - hc[(size_t)m * N + n]));
            }
        printf("correctness %dx%dx%d: max abs err %.3e %s\n", M, N, K, worst, worst < 1e-3 ? "OK" : "FAIL");
    }
    const int n = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= n;
    printf("gemm4w %6d x %5d x %5d  %8.3f ms  %7.1f TFLOP/s\n", M, N, K, ms, 2.0 * M * N * K / ms / 1e9);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
    (void)bf;
    return 0;
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (run(300, 520, 192, true, e0, e1)) return 1;
    if (run(1000, 776, 1024, true, e0, e1)) return 1;
    if (run(8192, 8192, 8192, false, e0, e1)) return 1;
    if (run(31616, 1024, 4096, false, e0, e1)) return 1;
    if (run(31616, 4096, 1024, false, e0, e1)) return 1;
    if (run(391680, 1024, 5056, false, e0, e1)) return 1;
    return 0;
}
