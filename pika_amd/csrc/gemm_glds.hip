// pika_amd/csrc/gemm_glds.hip -- bf16 x bf16 GEMMs with direct global->LDS loads (LDS-DMA) for gfx950.
//
// Used wherever both operands already live in HBM as bf16 (the joint network: hidden h, bf16 weight copies, bf16
// d(logits); every encoder / transformer product of the bf16 arithmetic mode).  No VGPR staging, no conversion: every
// wave issues global_load_lds_dwordx4 pieces (1 KiB each) straight into the K-tile buffers; the 16-byte granule index
// is XOR-ed with (row & 7) on the SOURCE address (the LDS image must stay lane-linear, as the instruction requires)
// and again on the fragment read, which keeps ds_read_b128 conflict-free.
//   gemm_pp<EPI, BOUNDS>  C = A B^T (+ fused epilogues), 256x256x64 tile, two wave groups in ping-pong, PERSISTENT
//                         workgroups, never-drained two-K-tile fetch cursor: 850-1320 TFLOP/s on random data
//   gemm_pp_tn<FULL>      C = A^T B (weight gradients), same schedule, transposed fragment reads: 700-1000 TFLOP/s
//   gemm_glds             256x128 tile, lockstep, 2 barriers per K-step, for outputs narrower than 128 columns
// Measurements and the time-stamp study behind the schedule: profiles/r1_gemm_pp_schedule.txt, tools/pp_trace.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "pika_gemm.h"
#include "pika_rnnt.h"
#include "pika_internal.h"

#define PIKA_NOT_APPLICABLE (-100)

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



// ---------------------------------------------------------------------------------------------
// 256 x 256 x 64 "ping-pong" kernel: 8 waves (2 x 4), 128 x 64 outputs per wave (128 accumulator
// VGPRs), K-tile double buffer of 2 x 64 KB filled by global_load_lds.  Waves 0-3 and waves 4-7
// (the two waves of every SIMD) run ONE barrier apart: every K-tile is four quadrant phases
// {ds_read fragments + two 1 KB load pieces | 16 MFMAs}, and while one group is in its load segment
// the other group owns the matrix pipe.  Workgroups are persistent (one per CU); a fetch cursor runs
// two K-tiles ahead of the MFMAs and across output-tile boundaries, loads are never drained (counted
// vmcnt, see the K loop), tiles are walked XCD-aware in bands of PP_GM tile rows.  Per CU and K-tile:
// 64 KB through LDS-DMA (about 30 cycles per 1 KB piece: the resource that bounds the loop) and
// 192 KB of conflict-free ds_read_b128 for 2176 cycles of MFMA issue.
// A may be a time-delay view (pika_operand_t, C % 64 == 0): a K-tile never straddles a tap, so it
// only changes the per-tile source offset; rows whose source time leaves the signal read a zero page
// (BOUNDS instantiation).  tests/test_gemm_schedule_model.py checks the buffer protocol.
constexpr int PP_T = 256 * 128, PP_BUF = 2 * PP_T;
#ifndef PP_GM
#define PP_GM 4
#endif

#ifdef PIKA_PP_TRACE
#define PP_STAMP(n) do { if (tr_on && t == 8) tr_ph[n] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PP_STAMP(n) do { } while (0)
#endif
#define PP_BAR() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __attribute__((aligned(512))) const unsigned int pp_zero_page[128] = {0};

struct PPArgs {
    const __bf16 *A, *B;
    float *C;
    const float *bias;
    long long a_batch, a_row, a_tap;   // element strides of A: per batch, per output row, per tap
    long long ldb, ldc;
    int a_rpb, a_C;                    // rows per batch, channels per tap
    int a_tin, a_t0, a_tstep, a_dtap;  // source time of (row t, tap) = t * a_tstep + tap * a_dtap - a_t0, zero outside [0, a_tin)
    int a_bounds;                      // 0: every (row, tap) is in range (no checks in the loop)
    // two-term A operand (pika_operand_t.seg): the a_C = 3 * a_seg reduction columns of a tap are the segments
    // [hi | lo | hi] of a_seg columns each; "lo" columns are read a_lo elements behind the same "hi" column
    int a_seg;
    long long a_lo;
    int M, N, K, relu;
    // fused epilogues (EPI 1, 2): bf16 output, dropout / auxiliary mask
    __bf16 *out16;
    __bf16 *out16_lo;                  // EPI 1, optional: second plane, bf16(v - (float)bf16(v)) of what out16 received
    const __bf16 *aux;
    long long ldo16, ld_aux;
    float scale;                       // EPI 1: 1/(1-p) for kept values; EPI 2: factor for unmasked values
    unsigned seed, thr;                // EPI 1, 3: drop where hash16 < thr (thr = 0: no dropout)
    const unsigned *salt;              // optional device word added to seed (pika_set_dropout_salt; filled in by launch_pp_epi)
    const float *res;                  // EPI 3: fp32 residual added after the dropout
    long long ld_res;
    // EPI 0, optional: per output row and 64-column block b = (tile column) * 4 + (wave column), the partial
    // log-sum-exp statistics of the row: lse_pm[m * lse_np + b] = max over the block, lse_ps[...] = sum exp(x - max)
    float *lse_pm, *lse_ps;
    int lse_np;
    // EPI 4 (the joint's logits over the RNN-T lattice, 16-bit): out16 receives the tile as IEEE fp16 (pitch ldo16), the
    // lse statistics above come from the fp32 accumulators as in EPI 0, and the TWO columns of a lattice row the loss reads
    // -- blank and the row's label -- also leave as fp32: g_out[2 m] = C[m][g_blank], g_out[2 m + 1] = C[m][label(m)] with
    // label(m) = g_labels[b (g_U1 - 1) + u] for row m = (b g_T + t) g_U1 + u, u < g_U1 - 1 (no label in the last column)
    const int *g_labels;
    float *g_out;
    int g_T, g_U1, g_blank;
    int nx, ntiles;                    // output tiles per row of tiles / in total (filled in by launch_pp_epi)
    int gm;                            // tile rows per band of the XCD-aware walk (filled in by launch_pp_epi)
    int f16;                           // the 16-bit operands are fp16 (EPI 0 only: PIKA_GEMM_F16_OPERANDS)
#ifdef PIKA_PP_TRACE
    unsigned long long *trace;         // tools/pp_trace.hip: [wg < 8][group 2][tile < 16][24] time stamps
#endif
};

// counter-based dropout decisions for the four consecutive columns n..n+3 (n % 4 == 0) of row m
__device__ inline unsigned pp_mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ inline void pp_keep4(unsigned seed, unsigned m, unsigned n, unsigned thr, bool keep[4]) {
    const unsigned h = pp_mix32(pp_mix32(seed + m * 0x9E3779B9u) ^ ((n >> 2) * 0x85ebca6bu));
    unsigned h2 = (h ^ (h >> 15)) * 0x2c1b3c6du;
    h2 ^= h2 >> 12;
    keep[0] = (h & 0xffffu) >= thr; keep[1] = (h >> 16) >= thr;
    keep[2] = (h2 & 0xffffu) >= thr; keep[3] = (h2 >> 16) >= thr;
}

__device__ inline bf16x8 ldsv(const unsigned char *p) { return *reinterpret_cast<const bf16x8 *>(p); }

// 16-bit epilogues, column-interior tiles: a lane holds, per 16-column block j of its row, the 4 consecutive columns
// (lane >> 4) * 4.. -- an 8-byte store per block, 32-byte runs per row and instruction, and the store tail of a tile is
// bound by the NUMBER of store instructions (MI355X guide T21).  For a pair of blocks (j, j + 1) one
// v_permlane16_swap per dword (odd 16-lane rows of the first operand <-> even rows of the second) leaves every lane with 8
// consecutive columns: lane group g = lane >> 4 holds columns (g >> 1) * 8.. of block j + (g & 1) -- one 16-byte store
// per pair instead of two 8-byte ones.  x, y: the lane's packed 4 x 16 bit of blocks j and j + 1.
// x (op) the values of lanes l ^ 16 / l ^ 32 without the LDS crossbar (__shfl_xor = ds_bpermute: a ~60-cycle round trip):
// a permlane swap of a register with its own copy leaves {own, partner's} in the two results -- which is which depends on
// the lane, the (commutative) combination does not.
__device__ inline float pp_max_x16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float pp_max_x32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float pp_sum_x16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float pp_sum_x32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ inline u32x4 pp_pair16(u32x2 x, u32x2 y) {
    const auto r0 = __builtin_amdgcn_permlane16_swap(x[0], y[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(x[1], y[1], false, false);
    return u32x4{r0[0], r1[0], r0[1], r1[1]};
}

// EPI 0: C f32 = act(A B^T + bias).   EPI 1: out16 bf16 = dropout(act(A B^T + bias)).
// EPI 2: out16 bf16 = scale * (A B^T) where aux > 0, else 0  (ReLU + dropout backward in one mask).
// EPI 3: C f32 = dropout(A B^T + bias) + res  (projection + residual dropout + residual add).
// EPI 4: out16 FP16 = A B^T + bias, with the lse statistics of EPI 0 and two gathered fp32 columns per row (see PPArgs).
// BOUNDS: A is a padded time-delay view (rows whose source time leaves the signal read the zero page)
template <int EPI, bool BOUNDS, bool F16 = false>
__global__ __launch_bounds__(512) void gemm_pp(PPArgs P) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int M = P.M, N = P.N;
    const int nx = P.nx, ntiles = P.ntiles, stride = gridDim.x;
    // wave index as a scalar: everything derived from it (LDS destinations = M0, wave-group tests) stays in SGPRs
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wr = wave >> 2, wc = wave & 3;
    // launch slot v -> tile: slots that share v % 8 (one XCD, one L2) get a contiguous run of tiles.  A workgroup
    // walks the slots blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x % 8 == 0 or a single pass), so all its
    // tiles stay on its own XCD's run
    auto tile_of = [&](int v) {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, idx = v >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    };
    // ... and the run is walked in bands of PP_GM tile rows, column by column inside a band: the ~32 tiles an XCD
    // works on at any moment form a PP_GM x (32 / PP_GM) block that shares PP_GM + 32 / PP_GM operand panels in its
    // L2 instead of the 1 + 32 of a row-major walk (fewer L2 misses = shorter load latency = faster LDS fill)
    const int gm = P.gm;                // rows of a band (PP_GM; PIKA_GEMM_PP_GM for A/B runs)
    auto tile_mn = [&](int tile, int &tm, int &tn) {
        const int band = tile / (gm * nx), in = tile - band * (gm * nx);
        const int rows = min(gm, ntiles / nx - band * gm);
        tn = in / rows;
        tm = band * gm + (in - tn * rows);
    };

    // Sources: a scalar 64-bit base per operand and tile plus a 32-bit byte offset per lane and piece (the host
    // checks they fit), so a load is `global_load_lds v_off, s[base]` with no 64-bit vector arithmetic in the loop.
    const char *sa;            // A: first row of the cursor's tile
    unsigned ao[4], bo[4];     // byte offsets of this lane's 16 bytes of piece i from sa / P.B
    int ts[4];                 // source time of tap 0 for this lane's row of piece i
    int n0;
    const int g16 = ((lane & 7) ^ (lane >> 3)) * 16;   // this lane's 16-byte granule of a 128-byte row, XOR-swizzled by row & 7
    auto setup = [&](int v) {
        int tm, tn;
        tile_mn(tile_of(v), tm, tn);
        const int m0 = tm * 256;
        n0 = tn * 256;
        const int ab0 = m0 / P.a_rpb, at0 = m0 - ab0 * P.a_rpb;
        // rows of a later batch may lie BELOW the tile's first row when the batches of a padded time-delay view
        // overlap in row space: bias the base so that every offset is >= 0
        long long below = 0;
        if (min(m0 + 255, M - 1) / P.a_rpb > ab0) below = max(0LL, (long long)at0 * P.a_row - P.a_batch);
        sa = reinterpret_cast<const char *>(P.A + (long long)ab0 * P.a_batch + (long long)at0 * P.a_row - below);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // A piece i of a wave: 8 rows of the 64-row block i (blocks 0/1 = the two m-halves of wave group 0,
            // 2/3 = of group 1), so the pieces a quadrant phase reads are the same pieces for every wave
            int ra = m0 + i * 64 + wave * 8 + (lane >> 3);
            ra = ra < M ? ra : M - 1;
            const int ab = ra / P.a_rpb, at = ra - ab * P.a_rpb;
            ts[i] = at * P.a_tstep - P.a_t0;
            ao[i] = (unsigned)(((long long)(ab - ab0) * P.a_batch + (long long)(at - at0) * P.a_row + below) * 2) + g16;
            // B piece i of a wave: rows n0 + (wave * 4 + i) * 8 .. + 7 (clamped to the last row at the edge)
            int rb = n0 + (wave * 4 + i) * 8 + (lane >> 3);
            rb = rb < N ? rb : N - 1;
            bo[i] = (unsigned)((long long)rb * P.ldb * 2) + g16;
        }
    };
    const char *zp = reinterpret_cast<const char *>(pp_zero_page) + (lane & 7) * 16;
    constexpr bool bounds = BOUNDS;
    const int piece0 = wave * 4 * 1024;
    auto gl = [&](const char *p, unsigned char *dst) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_global_load_lds((glb_u32 *)p, (lds_u32 *)dst, 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    const int sw0 = (((lane >> 4)) ^ (lane & 7)) << 4, sw1 = sw0 ^ 64;
    const int aoff = (wr * 128 + (lane & 15)) * 128, boff = PP_T + (wc * 64 + (lane & 15)) * 128;
    const int nt = P.K / 64;

    // The fetch cursor: the K-tile whose eight 1 KB pieces (A0..A3, B0..B3 per wave) are being issued.  It runs
    // one to two K-tiles ahead of the MFMAs and walks straight from the last K-tile of an output tile into the
    // first K-tile of the workgroup's next output tile.  pa/pb/ts always belong to the cursor's tile.
    // x_src = where the 64 reduction columns at x_c0 of the tap lie in memory, in elements from the tap's first
    // column: x_c0 itself, or for a two-term operand the column inside its [hi | lo | hi] segment (+ a_lo for "lo")
    int x_slot = blockIdx.x, x_kt = 0, x_tap = 0, x_c0 = 0;
    long long x_src = 0;
    bool x_ok = true;
    setup(x_slot);
    auto next_cols = [&]() {
        x_c0 += 64;
        x_src += 64;
        if (x_c0 == P.a_C) { x_c0 = 0; x_src = 0; ++x_tap; }
        else if (P.a_seg) {
            if (x_c0 == P.a_seg) x_src = P.a_lo;
            else if (x_c0 == 2 * P.a_seg) x_src = 0;
        }
    };
    auto advance = [&]() {
        if (++x_kt < nt) {
            next_cols();
        } else {
            x_kt = 0; x_tap = 0; x_c0 = 0; x_src = 0;
            x_slot += stride;
            x_ok = x_slot < ntiles;
            if (x_ok) setup(x_slot);
        }
    };
    // `dst` = the K-tile buffer.  With `bounds`, rows whose source time is outside the signal read a zero page.
    auto issue_a = [&](int i, unsigned char *dst) {
        const char *base = sa + ((long long)x_tap * P.a_tap + x_src) * 2;
        if (!bounds) {
            gl(base + ao[i], dst + i * 8192 + wave * 1024);
        } else {
            gl((unsigned)(ts[i] + x_tap * P.a_dtap) < (unsigned)P.a_tin ? base + ao[i] : zp, dst + i * 8192 + wave * 1024);
        }
    };
    auto issue_b = [&](int i, unsigned char *dst) {
        gl(reinterpret_cast<const char *>(P.B) + (long long)x_kt * 128 + bo[i], dst + PP_T + piece0 + i * 1024);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: K-tile 0 in full and the first half (A0 A2 B0 B1) of the K-tile behind it; the cursor stays on that one
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        issue_a(i, smem);
        issue_b(i, smem);
    }
    advance();
    if (x_ok) {
        issue_a(0, smem + PP_BUF);
        issue_a(2, smem + PP_BUF);
        issue_b(0, smem + PP_BUF);
        issue_b(1, smem + PP_BUF);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PP_ABL     // tools/r6_pp_ablate.sh: timing-only builds (results are wrong): bit 0 = fragment reads only in the first
    //               K-tile, bit 1 = no LDS-DMA piece behind the prologue
    if (PP_ABL & 2) x_ok = false;
#define PP_RD ((PP_ABL & 1) == 0 || gt == 0)
#else
#define PP_RD true
#endif
    PP_BAR();
    if (wr == 1) PP_BAR();   // group 1 runs one segment behind group 0

    bf16x8 fa[4][2], fb[4][2];
    auto quadrant = [&](int ia, int j0) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if constexpr (F16) {       // the operands hold fp16 bit patterns (PIKA_GEMM_F16_OPERANDS)
                        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                        acc[ia + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                            __builtin_bit_cast(h8, fb[j0 + j][kk]), __builtin_bit_cast(h8, fa[i][kk]), acc[ia + i][j0 + j], 0, 0, 0);
                    } else {
                        acc[ia + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j0 + j][kk], fa[i][kk], acc[ia + i][j0 + j], 0, 0, 0);
                    }
        __builtin_amdgcn_s_setprio(0);
    };
    int gt = 0, cur_slot = blockIdx.x;   // K-tiles consumed so far (parity = LDS buffer); the tile being accumulated
#ifdef PIKA_PP_TRACE
    int tr_tile = 0;
    unsigned long long tr_t0 = 0, tr_rt0 = 0, tr_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_k8 = 0;
    const bool tr_on = P.trace && blockIdx.x < 8 && (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 3) == 0;   // wave-uniform: stamps stay in SGPRs
#endif
    for (;;) {
#ifdef PIKA_PP_TRACE
    if (tr_on) { tr_t0 = __builtin_amdgcn_s_memtime(); tr_rt0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    // One K-tile.  `sw` = the cursor may step into the workgroup's next output tile in this iteration (the last two
    // K-tiles of an output tile): kept out of the steady-state loop body, where the lane offsets are loop invariants
    auto ktile = [&]([[maybe_unused]] int t, auto sw) {   // t: only the time-stamp build looks at it
        // Load segments (ds_read fragments, two pieces of the fetch cursor's K-tile) alternate with MFMA segments; while
        // one wave group is in a load segment the other owns the matrix pipe.  The CU takes about one 1 KB piece per
        // 30-35 cycles whoever issues it, so all four load segments carry exactly two pieces per wave, and a K-tile
        // is fetched over four consecutive segments that straddle two iterations, into the buffer read two K-tiles
        // earlier, each pair as soon as BOTH wave groups have retired their reads of the rows it overwrites:
        //   phase 2 of iteration t:   A0 A2 of K-tile t+2   (read in phase 0 of t by group 0 / 1)
        //   phase 3 of iteration t:   B0 B1                 (B is last read in phase 1)
        //   phase 0 of iteration t+1: B2 B3
        //   phase 1 of iteration t+1: A1 A3                 (m-half 1 of group 0 / 1, read in phase 2 of t)
        // Loads are never drained: phase 3 waits until all but the last four (A1 A3 of K-tile t+1, A0 A2 of t+2) have
        // landed -- what phases 0 and 1 of the next iteration read -- and phase 1 until all but the last six, which
        // retires A1 A3 of THIS K-tile one barrier before phase 2 reads them.  Loads return in order, so the counts
        // hold whatever the stores of an epilogue in between do.
        const unsigned char *cur = smem + (gt & 1) * PP_BUF;
        unsigned char *nxt = smem + ((gt + 1) & 1) * PP_BUF, *nn = smem + (gt & 1) * PP_BUF;
#ifdef PIKA_PP_TRACE
        if (tr_on && t == 8) tr_k8 = __builtin_amdgcn_s_memtime();
#endif
        // ---- phase 0: quadrant (m-half 0, n-half 0)
        if (PP_RD) {
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
        }
        if (x_ok) {
            issue_b(2, nxt);
            issue_b(3, nxt);
        }
        PP_BAR();
        PP_STAMP(0);
        quadrant(0, 0);
        PP_BAR();
        PP_STAMP(1);
        // ---- phase 1: (m-half 0, n-half 1)
        if (PP_RD) {
#pragma unroll
        for (int j = 2; j < 4; ++j) {
            fb[j][0] = ldsv(cur + boff + j * 2048 + sw0);
            fb[j][1] = ldsv(cur + boff + j * 2048 + sw1);
        }
        }
        if (x_ok) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            issue_a(1, nxt);
            issue_a(3, nxt);
            if constexpr (decltype(sw)::value) {
                advance();   // (into the workgroup's next output tile behind the last K-tile of this one)
            } else {
                ++x_kt;
                next_cols();
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_BAR();
        PP_STAMP(2);
        quadrant(0, 2);
        PP_BAR();
        PP_STAMP(3);
        // ---- phase 2: (m-half 1, n-half 1)
        if (PP_RD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i][0] = ldsv(cur + aoff + (4 + i) * 2048 + sw0);
            fa[i][1] = ldsv(cur + aoff + (4 + i) * 2048 + sw1);
        }
        }
        if (x_ok) {
            issue_a(0, nn);
            issue_a(2, nn);
        }
        PP_BAR();
        PP_STAMP(4);
        quadrant(4, 2);
        PP_BAR();
        PP_STAMP(5);
        // ---- phase 3: (m-half 1, n-half 0)
        if (x_ok) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            issue_b(0, nn);
            issue_b(1, nn);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_BAR();
        PP_STAMP(6);
        quadrant(4, 0);
        PP_BAR();
        PP_STAMP(7);
    };
    {
        int t = 0;
        for (; t < nt - 2; ++t, ++gt) ktile(t, std::false_type{});
        for (; t < nt; ++t, ++gt) ktile(t, std::true_type{});
    }
#ifdef PIKA_PP_TRACE
    unsigned long long tr_k = 0;
    if (tr_on) tr_k = __builtin_amdgcn_s_memtime();
#endif

    // The two wave groups write their halves of the tile at the same time: group 0 lets group 1 catch up here and
    // runs ahead again behind the epilogue (staggered, each group's epilogue would sit inside the other's barrier wait)
    if (wr == 0) PP_BAR();
    const float *bias = P.bias;
    [[maybe_unused]] const unsigned seed = P.seed + ((P.thr && P.salt) ? *P.salt : 0u);
    float *C = P.C;
    int etm, etn;
    tile_mn(tile_of(cur_slot), etm, etn);
    const int em0 = etm * 256, en0 = etn * 256;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    // residual / mask operands of a column-interior tile are fetched one or two 16-row blocks at a time, ahead
    // of the arithmetic and the stores: one at a time, every load pays its full latency behind the previous store
    const bool wide = en0 + 256 <= N;
    // ... and a 16-bit epilogue on such a tile stores 16 bytes per lane and block pair (pp_pair16) when rows are 16-byte aligned
    [[maybe_unused]] const bool wide16 = wide && !(P.ldo16 & 7) && !(reinterpret_cast<uintptr_t>(P.out16) & 15) &&
                                         !(reinterpret_cast<uintptr_t>(P.out16_lo) & 15);
    constexpr int EB = EPI == 3 ? 1 : 2;   // 16-row blocks fetched ahead (sixteen registers either way)
    [[maybe_unused]] int g_u = 0, g_t = 0, g_b = 0;      // EPI 4: lattice position of the lane's row (see g_lab)
#pragma unroll
    for (int ip = 0; ip < 8; ip += EB) {
        f32x4 rv[EB][4];
        bf16x4 av[EB][4];
        if constexpr (EPI == 2 || EPI == 3) {
            if (wide) {
#pragma unroll
                for (int di = 0; di < EB; ++di) {
                    const int mr = min(em0 + wr * 128 + (ip + di) * 16 + (lane & 15), M - 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = en0 + wc * 64 + j * 16 + (lane >> 4) * 4;
                        if constexpr (EPI == 3) rv[di][j] = *reinterpret_cast<const f32x4 *>(P.res + (long long)mr * P.ld_res + n);
                        else av[di][j] = *reinterpret_cast<const bf16x4 *>(P.aux + (long long)mr * P.ld_aux + n);
                    }
                }
            }
        }
#pragma unroll
    for (int di = 0; di < EB; ++di) {
        const int i = ip + di;
        const int m = em0 + wr * 128 + i * 16 + (lane & 15);
        if (m >= M) continue;
        [[maybe_unused]] f32x4 lv[4];
        [[maybe_unused]] u32x2 pk[4], pk_lo[4];      // 16-bit epilogues on a column-interior tile: packed values per block
        if constexpr (EPI == 0 || EPI == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) lv[j] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
        [[maybe_unused]] int g_lab = -1;
        if constexpr (EPI == 4) {
            if (P.g_out) {
                // (b, t, u) of the lane's row: two divisions per TILE, then carried from one 16-row block to the next
                if (i == 0) {
                    const int bt = m / P.g_U1;
                    g_u = m - bt * P.g_U1;
                    g_b = bt / P.g_T;
                    g_t = bt - g_b * P.g_T;
                }
                if (g_u < P.g_U1 - 1) g_lab = P.g_labels[(long long)g_b * (P.g_U1 - 1) + g_u];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = en0 + wc * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= N) continue;
            f32x4 v = acc[i][j];
            if constexpr (EPI != 2) {
                if (bias) {
                    if (n + 3 < N) v += *reinterpret_cast<const f32x4 *>(bias + n);
                    else for (int e = 0; e < 4; ++e) if (n + e < N) v[e] += bias[n + e];
                }
                if (P.relu) v = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
            }
            if constexpr (EPI == 3) {
                if (P.thr) {
                    bool keep[4];
                    pp_keep4(seed, (unsigned)m, (unsigned)n, P.thr, keep);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * P.scale : 0.f;
                }
                const float *rp = P.res + (long long)m * P.ld_res + n;
                if (wide) v += rv[di][j];
                else if (n + 3 < N) v += *reinterpret_cast<const f32x4 *>(rp);
                else for (int e = 0; e < 4; ++e) if (n + e < N) v[e] += rp[e];
            }
            if constexpr (EPI == 0 || EPI == 4) {
                if (n + 3 < N) lv[j] = v;
                else for (int e = 0; e < 4; ++e) if (n + e < N) lv[j][e] = v[e];
            }
            if constexpr (EPI == 4) {
                typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                f32x4 c = v;        // fp16 saturates at +-65504: never an infinity in the stored logits
#pragma unroll
                for (int e = 0; e < 4; ++e) c[e] = fminf(fmaxf(c[e], -65504.f), 65504.f);
                const f16x4 h4 = __builtin_convertvector(c, f16x4);
                _Float16 *op = reinterpret_cast<_Float16 *>(P.out16) + (long long)m * P.ldo16 + n;
                if (wide16) pk[j] = __builtin_bit_cast(u32x2, h4);          // stored pairwise behind the j loop
                else if (n + 3 < N) *reinterpret_cast<f16x4 *>(op) = h4;
                else for (int e = 0; e < 4; ++e) if (n + e < N) op[e] = h4[e];
            } else if constexpr (EPI == 0 || EPI == 3) {
                if (n + 3 < N) {
                    *reinterpret_cast<f32x4 *>(C + (long long)m * P.ldc + n) = v;
                } else {
                    for (int e = 0; e < 4; ++e) if (n + e < N) C[(long long)m * P.ldc + n + e] = v[e];
                }
            } else {
                if constexpr (EPI == 1) {
                    if (P.thr) {
                        bool keep[4];
                        pp_keep4(seed, (unsigned)m, (unsigned)n, P.thr, keep);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * P.scale : 0.f;
                    }
                } else {
                    const __bf16 *ap = P.aux + (long long)m * P.ld_aux + n;
                    if (wide || n + 3 < N) {
                        const bf16x4 a4 = wide ? av[di][j] : *reinterpret_cast<const bf16x4 *>(ap);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (float)a4[e] > 0.f ? v[e] * P.scale : 0.f;
                    } else {
                        for (int e = 0; e < 4; ++e) if (n + e < N) v[e] = (float)ap[e] > 0.f ? v[e] * P.scale : 0.f;
                    }
                }
                __bf16 *op = P.out16 + (long long)m * P.ldo16 + n;
                const bf16x4 hi4 = __builtin_convertvector(v, bf16x4);
                if (wide16) {
                    pk[j] = __builtin_bit_cast(u32x2, hi4);                // stored pairwise behind the j loop
                } else if (n + 3 < N) {
                    *reinterpret_cast<bf16x4 *>(op) = hi4;
                } else {
                    for (int e = 0; e < 4; ++e) if (n + e < N) op[e] = hi4[e];
                }
                if constexpr (EPI == 1) {
                    if (P.out16_lo) {      // the second term of the value: what the bf16 rounding above dropped
                        const bf16x4 lo4 = __builtin_convertvector(v - __builtin_convertvector(hi4, f32x4), bf16x4);
                        __bf16 *lp = P.out16_lo + (long long)m * P.ldo16 + n;
                        if (wide16) pk_lo[j] = __builtin_bit_cast(u32x2, lo4);
                        else if (n + 3 < N) *reinterpret_cast<bf16x4 *>(lp) = lo4;
                        else for (int e = 0; e < 4; ++e) if (n + e < N) lp[e] = lo4[e];
                    }
                }
            }
        }
        if constexpr (EPI == 4) {
#ifdef PP_ABL
            if (P.g_out && !(PP_ABL & 8)) {
#else
            if (P.g_out) {
#endif
                // the lane that holds a gathered column hands its fp32 value over: column c sits in this lane iff
                // rel = c - (first column of the lane) is e + 16 j with e < 4, j < 4 -- one test per row block and column
                // (not one per 16-column block), the value by two levels of selects from lv (= v after the bias)
                const int c0 = en0 + wc * 64 + (lane >> 4) * 4;
                auto pick = [&](int rel) {
                    const int jj = rel >> 4, e = rel & 3;
                    const f32x4 q = jj == 0 ? lv[0] : (jj == 1 ? lv[1] : (jj == 2 ? lv[2] : lv[3]));
                    return e == 0 ? q.x : (e == 1 ? q.y : (e == 2 ? q.z : q.w));
                };
                const int rb = P.g_blank - c0, rl = g_lab - c0;
                if (rb >= 0 && !(rb & ~51) && P.g_blank < N) P.g_out[2LL * m] = pick(rb);
                if (rl >= 0 && !(rl & ~51) && g_lab < N) P.g_out[2LL * m + 1] = pick(rl);
            }
            // the next 16-row block of this lane: 16 lattice cells further
            g_u += 16;
            while (g_u >= P.g_U1) { g_u -= P.g_U1; if (++g_t == P.g_T) { g_t = 0; ++g_b; } }
        }
        if constexpr (EPI == 1 || EPI == 2 || EPI == 4) {
            if (wide16) {    // (wave-uniform; every lane of the wave takes part in the swaps: rows m >= M only skip the store)
                const int g = lane >> 4;
                unsigned short *orow = reinterpret_cast<unsigned short *>(P.out16) + (long long)m * P.ldo16 + en0 + wc * 64 + (g >> 1) * 8 + (g & 1) * 16;
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const u32x4 q = pp_pair16(pk[j], pk[j + 1]);
#ifdef PP_ABL
                    if ((PP_ABL & 16) && q[0] != 0x12345u) continue;      // timing build: no main stores
#endif
                    *reinterpret_cast<u32x4 *>(orow + j * 16) = q;
                }
                if constexpr (EPI == 1) {
                    if (P.out16_lo) {
                        unsigned short *lrow = reinterpret_cast<unsigned short *>(P.out16_lo) + (long long)m * P.ldo16 + en0 + wc * 64 + (g >> 1) * 8 + (g & 1) * 16;
#pragma unroll
                        for (int j = 0; j < 4; j += 2) {
                            const u32x4 q = pp_pair16(pk_lo[j], pk_lo[j + 1]);
                            *reinterpret_cast<u32x4 *>(lrow + j * 16) = q;
                        }
                    }
                }
            }
        }
        if constexpr (EPI == 0 || EPI == 4) {
#ifdef PP_ABL
            if (P.lse_pm && !(PP_ABL & 4)) {
#else
            if (P.lse_pm) {
#endif
                // the 64 columns this wave holds of row m sit in the 4 lanes {l, l+16, l+32, l+48}, 16 values each
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fmaxf(fmaxf(lv[j].x, lv[j].y), fmaxf(lv[j].z, lv[j].w)));
                mx = pp_max_x16(mx);
                mx = pp_max_x32(mx);
                float sm = 0.f;
                if (mx > -INFINITY) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) sm += __expf(lv[j][e] - mx);     // masked columns: exp(-inf) = 0
                }
                sm = pp_sum_x16(sm);
                sm = pp_sum_x32(sm);
                if ((lane >> 4) == 0) {
                    const long long o = (long long)m * P.lse_np + etn * 4 + wc;
                    P.lse_pm[o] = mx;
                    P.lse_ps[o] = sm;
                }
            }
        }
    }
    }
#ifdef PIKA_PP_TRACE
    if (tr_on && tr_tile < 16 && lane == 0) {
        unsigned long long *q = P.trace + ((blockIdx.x * 2 + wr) * 16 + tr_tile) * 24;
        q[6] = tr_k8;
        for (int e = 0; e < 8; ++e) q[8 + e] = tr_ph[e];
        q[0] = tr_t0; q[1] = tr_rt0; q[2] = 0; q[3] = 0; q[4] = tr_k; q[5] = __builtin_amdgcn_s_memtime();
    }
    ++tr_tile;
#endif
    cur_slot += stride;
    if (cur_slot >= ntiles) break;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wr == 1) PP_BAR();
    }
}

// ---------------------------------------------------------------------------------------------
// Same schedule for C[m,n] = sum_r A[r,m] * B[r,n] with BOTH operands reduction-major in memory
// (`trans` operands of pika_gemm_nt: weight gradients dW = dY^T X).  A K-tile is 64 reduction rows
// x 256 output columns per operand ([64][512 B] in LDS); MFMA fragments come out of it with
// ds_read_b64_tr_b16.  The 32-byte chunk index of a row is XOR-ed with
// key(r) = (r & 3) | ((r >> 3) & 1) << 2 on the global source address and on the read: the eight
// (row) x 32 B pieces a 32-lane half of a transpose read touches land on eight distinct bank
// groups.  Reduction rows past the end and column chunks past the operand width are fetched from a
// zero page.  Split-K over gridDim.z with an atomic epilogue (C pre-zeroed by the host).

struct TNOperand {
    const __bf16 *ptr;     // element (r, c): ptr + b*batch + t*row + c,  (b, t) = divmod(r, rpb)
    long long batch, row;  // element strides
    int rpb, width;        // rows per batch; number of valid view columns (multiple of 8)
    int C;                 // channels per tap (column c of the view -> tap = c / C: + tap * tap_stride)
    long long tap_stride;
};

struct TNArgs {
    TNOperand A, B;
    float *C;
    long long ldc;
    int M, N, R;           // output extents, reduction length
    int tiles_per_split;   // K-tiles (64 rows) per gridDim.z slice
    int atomic;
    float *ws;             // split-K partials [gridDim.z][M][N] (dense), or null -> atomics into C
};

// Transposed fragment read, as raw instructions: through the builtin, hipcc drains every pending LDS-DMA load
// (s_waitcnt vmcnt(0)) in front of each group of reads, because it cannot tell the rows being read from the rows
// being prefetched -- that serialises the prefetch with the very phase it is issued in.  The buffers are kept apart
// by the barrier / vmcnt protocol of the K loop instead.  The two halves land asynchronously: they may only be
// touched behind TN_WAIT_LDS().
typedef short s16x4 __attribute__((ext_vector_type(4)));
struct TrFrag { s16x4 lo, hi; };
__device__ inline void lds_tr(TrFrag &f, unsigned lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(lds_addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(f.hi) : "v"(lds_addr));
}
__device__ inline bf16x8 tr_join(const TrFrag &f) {
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
#define TN_WAIT_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// Source addresses of one operand's pieces: a scalar base (first reduction row of the K-tile being fetched) that
// advances by 64 rows per K-tile, plus 32-bit lane offsets that never change.  Piece i of a wave = reduction rows
// tn_piece_row(wave, i), + 1 of the K-tile; a row that runs past the end of its batch (rpb >= 64: at most one batch
// boundary inside a K-tile) adds `wrap`.  All loop arithmetic is scalar; a load is global_load_lds v_off, s[base].
// first of the two reduction rows of piece i of a wave: pieces 0, 1 lie in rows 0..31 of the K-tile, 2, 3 in rows 32..63
__device__ inline int tn_piece_row(int wave, int i) { return (i >> 1) * 32 + wave * 4 + (i & 1) * 2; }

struct TNStager {
    const char *base;        // row R0 of the K-tile, column block c0
    unsigned lo[2];          // lane offset for even / odd pieces: swizzled 16-byte column chunk + (lane >> 5) rows
    long long rowb;          // bytes per reduction row
    unsigned wrapb;          // extra bytes when crossing into the next batch (host: 0 <= wrap < 2^30)
    int rpb, t0;             // rows per batch; row of R0 inside its batch

    __device__ inline void init(const TNOperand &o, int c0, int r0, int wave, int lane) {
        rpb = o.rpb;
        rowb = o.row * 2;
        wrapb = (unsigned)((o.batch - (long long)o.rpb * o.row) * 2);
        const int tap = c0 / o.C;   // a 256-column block never straddles a tap (C % 256 == 0 or one tap)
        const long long coff = (long long)tap * o.tap_stride + (c0 - tap * o.C);
        const int b0 = r0 / o.rpb;
        t0 = r0 - b0 * o.rpb;
        base = reinterpret_cast<const char *>(o.ptr + (long long)b0 * o.batch + (long long)t0 * o.row + coff);
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int key = (2 * par + (lane >> 5)) | (((wave >> 1) & 1) << 2);   // key(row) = (row & 3) | ((row >> 3) & 1) << 2
            const int g16 = lane & 31, q = (g16 >> 1) ^ key;               // 32-byte chunk, swizzled
            // columns past the operand width feed output rows/columns that are never stored: re-read
            // the last valid chunk instead of running off the row
            const int col = min(q * 16 + (g16 & 1) * 8, o.width - c0 - 8);
            lo[par] = (unsigned)(col * 2) + (unsigned)((lane >> 5) * rowb);
        }
    }
    // piece i of wave `wave` (scalar); `full` = all 64 rows of the K-tile exist (else rows >= R read zeros, for BOTH
    // operands: 0 x garbage could be NaN)
    __device__ inline const char *src(int i, int wave, int lane, bool full, int r0, int R) const {
        const int row = tn_piece_row(wave, i);
        const char *sb = base + (long long)row * rowb;
        const unsigned off = lo[i & 1] + ((lane >> 5) >= rpb - t0 - row ? wrapb : 0u);
        if (full) return sb + off;
        const char *z = reinterpret_cast<const char *>(pp_zero_page) + (lane & 31) * 16;
        return r0 + row + (lane >> 5) < R ? sb + off : z;
    }
    __device__ inline void advance() {
        t0 += 64;
        base += 64 * rowb;
        if (t0 >= rpb) { t0 -= rpb; base += wrapb; }
    }
};

#define TN_MFMA(MH, KK)                                                                             \
    do {                                                                                            \
        TN_WAIT_LDS();                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                           \
                acc[(MH) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_join(fb[j]), tr_join(fa[i]), acc[(MH) * 4 + i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                              \
    } while (0)

// FULL: the reduction length is a multiple of 64 (no K-tile has rows past the end: no zero-page select per lane)
template <bool FULL>
__global__ __launch_bounds__(512) void gemm_pp_tn(TNArgs P) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int nx = gridDim.x, ntiles = nx * gridDim.y;
    int tile = blockIdx.y * nx + blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (tile / nx) * 256, n0 = (tile % nx) * 256;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wr = wave >> 2, wc = wave & 3;
    const int t_begin = blockIdx.z * P.tiles_per_split;
    const int t_total = (P.R + 63) / 64;
    const int nt = min(P.tiles_per_split, t_total - t_begin);
    if (nt <= 0) return;
    const int R = P.R;

    TNStager sa, sb;
    sa.init(P.A, m0, t_begin * 64, wave, lane);
    sb.init(P.B, n0, t_begin * 64, wave, lane);
    int r0 = t_begin * 64;   // first reduction row of the K-tile being fetched
    auto gl = [&](const char *p, unsigned char *dst) {
        __builtin_amdgcn_global_load_lds((glb_u32 *)p, (lds_u32 *)dst, 16, 0, 0);
    };
    // piece i of an operand's K-tile -> its two rows of the [64][512 B] image (`img` = buffer + 0 / PP_T)
    auto issue = [&](const TNStager &st, int i, unsigned char *img) {
        unsigned char *dst = img + tn_piece_row(wave, i) * 512;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FULL) gl(st.src(i, wave, lane, true, r0, R), dst);
        else gl(st.src(i, wave, lane, r0 + 64 <= R, r0, R), dst);
        __builtin_amdgcn_sched_barrier(0);
    };

    // transpose-read addressing: lane (g, i16 = 4j + c) supplies row kk*32 + g*8 + j (+4 for the second
    // read), 8 bytes at column 4c of the fragment's 16-column chunk; key(row) = j | (g & 1) << 2
    const int g = lane >> 4, jj = (lane & 15) >> 2, cc = lane & 3;
    const int rbase = (g * 8 + jj) * 512 + cc * 8;
    const int keyoff = (jj | ((g & 1) << 2)) << 5;
    const int achunk = wr * 8 * 32, bchunk = PP_T + (wc >> 1) * 8 * 32;
    const int keyoffb = keyoff ^ ((wc & 1) << 7);   // chunk (wc & 1) * 4 + j inside the 8-chunk group

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: K-tile 0 in full and the first two A pieces of the K-tile behind it; the stagers stay on that one
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        issue(sa, i, smem);
        issue(sb, i, smem + PP_T);
    }
    int x_kt = 1;   // the stagers' K-tile (the fetch cursor)
    sa.advance();
    sb.advance();
    r0 += 64;
    if (x_kt < nt) {
        issue(sa, 0, smem + PP_BUF);
        issue(sa, 1, smem + PP_BUF);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BAR();
    if (wr == 1) PP_BAR();

    // Phases of a K-tile: (m-half 0, k 0..31) (m-half 1, k 0..31) (m-half 0, k 32..63) (m-half 1, k 32..63); 16 MFMAs
    // each, fragment loads 8 / 4 / 8 / 4.  Every load segment also issues two 1 KB pieces of the fetch cursor's
    // K-tile into the buffer read two K-tiles earlier, as soon as both wave groups have retired their reads of
    // those rows:
    //   phase 3 of iteration t:   A rows 0..31 of K-tile t+2   (rows 0..31 are last read in phase 1)
    //   phase 0 of iteration t+1: B rows 0..31
    //   phase 1 of iteration t+1: A rows 32..63               (rows 32..63 are last read in phase 3 of t)
    //   phase 2 of iteration t+1: B rows 32..63
    // Loads are never drained: phase 3 waits for all but the last four (rows 32..63 of the next K-tile) before the
    // barrier that lets the other group read rows 0..31 of it, phase 1 for all but the last four, which retires
    // rows 32..63 of THIS K-tile one barrier before phase 2 reads them.
    TrFrag fa[4], fb[4];
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_u8 *)smem + rbase;
    for (int t = 0; t < nt; ++t) {
        const unsigned cur = lds0 + (t & 1) * PP_BUF;
        unsigned char *nxt = smem + ((t + 1) & 1) * PP_BUF, *nn = smem + (t & 1) * PP_BUF;
        // ---- phase 0
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_tr(fb[j], cur + bchunk + ((j << 5) ^ keyoffb));
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_tr(fa[i], cur + achunk + ((i << 5) ^ keyoff));
        if (x_kt < nt) {
            issue(sb, 0, nxt + PP_T);
            issue(sb, 1, nxt + PP_T);
        }
        PP_BAR();
        TN_MFMA(0, 0);
        PP_BAR();
        // ---- phase 1
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_tr(fa[i], cur + achunk + (((4 + i) << 5) ^ keyoff));
        if (x_kt < nt) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            issue(sa, 2, nxt);
            issue(sa, 3, nxt);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_BAR();
        TN_MFMA(1, 0);
        PP_BAR();
        // ---- phase 2
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_tr(fb[j], cur + bchunk + ((j << 5) ^ keyoffb) + 32 * 512);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_tr(fa[i], cur + achunk + ((i << 5) ^ keyoff) + 32 * 512);
        if (x_kt < nt) {
            issue(sb, 2, nxt + PP_T);
            issue(sb, 3, nxt + PP_T);
            sa.advance();
            sb.advance();
            r0 += 64;
            ++x_kt;
        }
        PP_BAR();
        TN_MFMA(0, 1);
        PP_BAR();
        // ---- phase 3
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_tr(fa[i], cur + achunk + (((4 + i) << 5) ^ keyoff) + 32 * 512);
        if (x_kt < nt) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            issue(sa, 0, nn);
            issue(sa, 1, nn);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_BAR();
        TN_MFMA(1, 1);
        PP_BAR();
    }
    if (wr == 0) PP_BAR();

    float *C = P.C;
    long long ldc = P.ldc;
    if (P.ws) { C = P.ws + (long long)blockIdx.z * P.M * P.N; ldc = P.N; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + (lane & 15);
        if (m >= P.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
            float *c = C + (long long)m * ldc + n;
            if (P.atomic && !P.ws) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < P.N) atomicAdd(c + e, acc[i][j][e]);
            } else if (n + 3 < P.N) {
                *reinterpret_cast<f32x4 *>(c) = acc[i][j];
            } else {
                for (int e = 0; e < 4; ++e)
                    if (n + e < P.N) c[e] = acc[i][j][e];
            }
        }
    }
}

// C[m, n] = sum_z ws[z][m][n]  (N % 4 == 0)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ ws, float *__restrict__ C,
                                                            long long ldc, int M, int N, int splits) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x, n4 = N >> 2;
    if (idx >= (long long)M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx - (long long)m * n4) * 4;
    const long long mn = (long long)M * N;
    const float *p = ws + (long long)m * N + n;
    f32x4 acc = *reinterpret_cast<const f32x4 *>(p);
    for (int z = 1; z < splits; ++z) acc += *reinterpret_cast<const f32x4 *>(p + z * mn);
    *reinterpret_cast<f32x4 *>(C + (long long)m * ldc + n) = acc;
}

// gemm_pp addresses a tile's rows by 32-bit byte offsets from the tile's first row (A) and from B
bool pp_offsets_fit(const PPArgs &P) {
    if (P.a_rpb <= 0 || P.a_batch < 0 || P.a_row < 0 || P.ldb < 0) return false;
    if ((long long)P.N * P.ldb >= (1LL << 30)) return false;
    const long long rpb = P.a_rpb;
    const long long below = P.M > rpb && (rpb - 1) * P.a_row > P.a_batch ? (rpb - 1) * P.a_row - P.a_batch : 0;
    return (255 / rpb + 1) * P.a_batch + (rpb < 256 ? rpb : 256) * P.a_row + below < (1LL << 30);
}

template <int EPI>
int launch_pp_epi(const PPArgs &P, hipStream_t s) {
    if (!pp_offsets_fit(P)) return PIKA_ETOOBIG;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp<EPI, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PP_BUF);
        if constexpr (EPI <= 1)   // padded time-delay operands only reach the plain epilogues (pika_gemm_nt, pika_gemm_bf16_ex)
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp<EPI, true>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PP_BUF);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long long nt = (long long)((P.N + 255) / 256) * ((P.M + 255) / 256);
    if (nt > 0x7fffffffLL - 65536) return PIKA_ETOOBIG;
    PPArgs Q = P;
    Q.nx = (P.N + 255) / 256; Q.ntiles = (int)nt;
    static const int gm_env = [] { const char *e = pika_knob("PIKA_GEMM_PP_GM"); return e ? atoi(e) : 0; }();
    Q.gm = gm_env > 0 ? gm_env : PP_GM;
    Q.salt = pika_internal_dropout_salt();
    // persistent: one workgroup per CU (128 KB of LDS each) walks its share of the output tiles.
    // PIKA_GEMM_PP_WGS=0 launches one workgroup per tile instead (for A/B timing).
    static const int wgs = [] {
        const char *e = pika_knob("PIKA_GEMM_PP_WGS");
        if (e) return atoi(e);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus & ~7;
    }();
    const int grid = (wgs >= 8 && nt > wgs) ? (wgs & ~7) : (int)nt;
    if (Q.f16) {
        if constexpr (EPI == 0) {
            static bool attr16 = false;
            if (!attr16) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp<0, false, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PP_BUF);
                if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp<0, true, true>),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PP_BUF);
                if (e != hipSuccess) return (int)e;
                attr16 = true;
            }
            if (Q.a_bounds) hipLaunchKernelGGL((gemm_pp<0, true, true>), dim3(grid), dim3(512), 2 * PP_BUF, s, Q);
            else hipLaunchKernelGGL((gemm_pp<0, false, true>), dim3(grid), dim3(512), 2 * PP_BUF, s, Q);
            return (int)hipGetLastError();
        } else {
            return PIKA_EINVAL;
        }
    }
    if (Q.a_bounds) {
        if constexpr (EPI <= 1) hipLaunchKernelGGL((gemm_pp<EPI, true>), dim3(grid), dim3(512), 2 * PP_BUF, s, Q);
        else return PIKA_EINVAL;
    } else {
        hipLaunchKernelGGL((gemm_pp<EPI, false>), dim3(grid), dim3(512), 2 * PP_BUF, s, Q);
    }
    return (int)hipGetLastError();
}

int launch_pp(const PPArgs &P, hipStream_t s) { return launch_pp_epi<0>(P, s); }

// out bf16 = keep(m, n) ? x * scale : 0 -- the dropout backward of an EPI 1/3 product fused with the bf16
// rounding its consumers (dX / dW products) apply anyway.  cols % 4 == 0.
__global__ __launch_bounds__(256) void pp_mask_cast_kernel(const float *__restrict__ x, long long ld, int rows,
                                                           int cols, unsigned seed, unsigned thr, float scale,
                                                           __bf16 *__restrict__ out, long long ld_out,
                                                           const unsigned *__restrict__ salt) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    if (thr && salt) seed += *salt;
    const int c4 = cols >> 2;
    const long long total = (long long)rows * c4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int m = (int)(idx / c4), n = (int)(idx - (long long)m * c4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(x + (long long)m * ld + n);
        if (thr) {
            bool keep[4];
            pp_keep4(seed, (unsigned)m, (unsigned)n, thr, keep);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * scale : 0.f;
        }
        *reinterpret_cast<bf16x4 *>(out + (long long)m * ld_out + n) = __builtin_convertvector(v, bf16x4);
    }
}

__global__ __launch_bounds__(256) void pp_keep_mask_kernel(unsigned char *mask, int rows, int cols, unsigned seed,
                                                           unsigned thr, const unsigned *__restrict__ salt) {
    if (salt) seed += *salt;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x, c4 = (cols + 3) >> 2;
    if (idx >= (long long)rows * c4) return;
    const int m = (int)(idx / c4), n = (int)(idx - (long long)m * c4) * 4;
    bool keep[4];
    pp_keep4(seed, (unsigned)m, (unsigned)n, thr, keep);
    for (int e = 0; e < 4; ++e)
        if (n + e < cols) mask[(long long)m * cols + n + e] = (!thr || keep[e]) ? 1 : 0;
}

bool tn_operand(const pika_operand_t &o, int extent, int R, TNOperand &t) {
    if (o.pad || (o.ld & 7) || (o.batch_stride & 7) || (extent & 7) || (reinterpret_cast<uintptr_t>(o.ptr) & 15)) return false;
    const bool one_tap = o.C >= extent;
    if (!one_tap && (o.C & 255)) return false;
    const int taps = one_tap ? 1 : (extent + o.C - 1) / o.C;
    if (taps > 1 && (long long)(o.rows_per_batch - 1) * o.stride + (long long)(taps - 1) * o.dil >= o.t_in) return false;
    // the kernel's stager: at most one batch boundary inside a 64-row K-tile, batches laid out one after the other
    const long long row = (long long)o.stride * o.ld;
    long long batch = o.batch_stride;
    if (R <= o.rows_per_batch) batch = (long long)o.rows_per_batch * row;   // one batch: its stride is never used
    {
        const long long wrap = batch - (long long)o.rows_per_batch * row;
        if (o.rows_per_batch < 64 || wrap < 0 || wrap >= (1LL << 29) || row >= (1LL << 28)) return false;
    }
    t.ptr = static_cast<const __bf16 *>(o.ptr);
    t.batch = batch; t.row = row; t.rpb = o.rows_per_batch;
    t.width = extent; t.C = one_tap ? (1 << 30) : o.C; t.tap_stride = (long long)o.dil * o.ld;
    return true;
}

int launch_pp_tn(const pika_operand_t *A, const pika_operand_t *B, float *C, long long ldc, int M, int N,
                 int R, const float *bias, int flags, void *ws, size_t ws_bytes, hipStream_t s) {
    if (bias || flags || (ldc & 3) || (reinterpret_cast<uintptr_t>(C) & 15)) return PIKA_NOT_APPLICABLE;
    if (M < 192 || N < 192 || R < 512) return PIKA_NOT_APPLICABLE;
    TNArgs P{};
    if (!tn_operand(*A, M, R, P.A) || !tn_operand(*B, N, R, P.B)) return PIKA_NOT_APPLICABLE;
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256), nk = (R + 63) / 64;
    int split = 1;
    long long best = -1;
    const size_t slice = (size_t)M * N * sizeof(float);
    const bool have_ws = ws && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 && ws_bytes >= 2 * slice;
    // cost model of gemm.hip; a split costs an atomic pass over C (~12 K-tile times, measured), or ~2 when
    // the partials go to the caller's workspace and one streaming kernel adds them up
    const long long per_split = have_ws ? 2 : 12;
    for (int sp = 1; sp <= 64 && sp <= nk / 4; ++sp) {
        if (have_ws && sp > 1 && (size_t)sp * slice > ws_bytes) break;
        const long long cost = (long long)((tiles * sp + 255) / 256) * ((nk + sp - 1) / sp + 40) + per_split * sp;
        if (best < 0 || cost < best) { best = cost; split = sp; }
    }
    static const int forced = [] { const char *e = pika_knob("PIKA_GEMM_TN_SPLIT"); return e ? atoi(e) : 0; }();
    if (forced > 0) {
        split = forced < nk ? forced : nk;
        if (have_ws) while (split > 1 && (size_t)split * slice > ws_bytes) --split;
    }
    P.C = C; P.ldc = ldc; P.M = M; P.N = N; P.R = R;
    P.tiles_per_split = (nk + split - 1) / split;
    split = (nk + P.tiles_per_split - 1) / P.tiles_per_split;
    P.atomic = split > 1;
    P.ws = (P.atomic && have_ws) ? static_cast<float *>(ws) : nullptr;
    if (P.atomic && !P.ws) {
        hipError_t e = hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s);
        if (e != hipSuccess) return (int)e;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp_tn<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PP_BUF);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp_tn<false>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PP_BUF);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const dim3 grid((N + 255) / 256, (M + 255) / 256, split);
    if (R & 63) hipLaunchKernelGGL(gemm_pp_tn<false>, grid, dim3(512), 2 * PP_BUF, s, P);
    else hipLaunchKernelGGL(gemm_pp_tn<true>, grid, dim3(512), 2 * PP_BUF, s, P);
    if (P.ws) {
        const long long n4 = (long long)M * (N >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, P.ws, C, ldc, M, N, split);
    }
    return (int)hipGetLastError();
}

// The A side of a PPArgs from an operand descriptor: a plain bf16 matrix or a time-delay view whose taps are whole
// K-tiles; rows whose source time leaves [0, t_in) (padded convolutions, the transposed convolution of dX) are
// redirected to a zero page inside the kernel (a_bounds).  A two-term operand (seg > 0) presents 3 * seg reduction
// columns per tap, [hi | lo | hi], over seg stored columns and a second plane lo_off elements behind the first.
bool pp_fill_a(const pika_operand_t &A, int K, PPArgs &P) {
    if (A.dtype != PIKA_BF16 || A.trans || (A.ld & 7) || (A.batch_stride & 7) || (reinterpret_cast<uintptr_t>(A.ptr) & 15))
        return false;
    const int a_C = A.C < K ? A.C : K;
    if ((a_C & 63) || K % a_C) return false;
    if (A.seg) {
        if (A.seg < 0 || (A.seg & 63) || a_C != 3 * A.seg || (A.lo_off & 7)) return false;
    } else if (A.lo_off) {
        return false;
    }
    const int taps = K / a_C;
    const bool bounds = A.pad > 0 || (long long)(A.rows_per_batch - 1) * A.stride + (long long)(taps - 1) * A.dil - A.pad >= A.t_in;
    P.A = static_cast<const __bf16 *>(A.ptr);
    P.a_rpb = A.rows_per_batch; P.a_batch = A.batch_stride; P.a_row = (long long)A.stride * A.ld;
    P.a_tap = (long long)A.dil * A.ld; P.a_C = a_C;
    P.a_tin = A.t_in; P.a_t0 = A.pad; P.a_tstep = A.stride; P.a_dtap = A.dil; P.a_bounds = bounds ? 1 : 0;
    P.a_seg = A.seg; P.a_lo = A.lo_off;
    // the row pointer of tap 0 starts `pad` source rows before the signal; only ever dereferenced in range
    P.A -= (long long)A.pad * A.ld;
    return true;
}

}  // namespace

static int g_pp_min_tiles = 160;

extern "C" int pika_gemm_set_min_tiles(int min_tiles) {
    const int old = g_pp_min_tiles;
    if (min_tiles >= 1) g_pp_min_tiles = min_tiles;
    return old;
}

// Used by pika_gemm_nt (gemm.hip): returns PIKA_NOT_APPLICABLE when the operands do not fit the
// direct-to-LDS kernels (then the register-staged kernel runs).
int pika_internal_gemm_pp(const pika_operand_t *A, const pika_operand_t *B, float *C, long long ldc,
                          int M, int N, int K, const float *bias, int flags, void *ws, size_t ws_bytes,
                          hipStream_t s) {
    if (A->dtype != PIKA_BF16 || B->dtype != PIKA_BF16 || (A->trans != 0) != (B->trans != 0)) return PIKA_NOT_APPLICABLE;
    const bool out16 = (flags & PIKA_GEMM_OUT_BF16) != 0, f16 = (flags & PIKA_GEMM_F16_OPERANDS) != 0;
    if (f16 && (A->trans || out16)) return PIKA_NOT_APPLICABLE;     // fp16 operands: the plain product with fp32 output only
    if (A->trans) return out16 ? PIKA_NOT_APPLICABLE : launch_pp_tn(A, B, C, ldc, M, N, K, bias, flags, ws, ws_bytes, s);
    if (flags & ~(PIKA_GEMM_RELU | PIKA_GEMM_OUT_BF16 | PIKA_GEMM_F16_OPERANDS | PIKA_GEMM_TERM_PRODUCT)) return PIKA_NOT_APPLICABLE;
    if ((K & 63) || (ldc & 3) || (reinterpret_cast<uintptr_t>(C) & (out16 ? 7 : 15))) return PIKA_NOT_APPLICABLE;
    // B: plain matrix
    if (B->C < K || B->pad || (B->ld & 7) || B->rows_per_batch < N || (reinterpret_cast<uintptr_t>(B->ptr) & 15))
        return PIKA_NOT_APPLICABLE;
    // fill-the-chip gate: 160 tiles for a plain product (below it the register-staged bf16 kernel's smaller tiles win), 24
    // for a K-concatenated term product (3-6 x the K-tiles per output tile amortise the fixed costs, and what it falls back
    // to is the exact kernel)
    const int min_tiles = (flags & PIKA_GEMM_TERM_PRODUCT) ? (g_pp_min_tiles < 24 ? g_pp_min_tiles : 24) : g_pp_min_tiles;
    if (M < 256 || N < 192 || (long long)((M + 255) / 256) * ((N + 255) / 256) < min_tiles) return PIKA_NOT_APPLICABLE;
    PPArgs P{};
    if (!pp_fill_a(*A, K, P)) return PIKA_NOT_APPLICABLE;
    P.B = static_cast<const __bf16 *>(B->ptr);
    P.C = C; P.bias = bias; P.ldb = B->ld; P.ldc = ldc;
    P.M = M; P.N = N; P.K = K; P.relu = (flags & PIKA_GEMM_RELU) ? 1 : 0;
    P.f16 = f16 ? 1 : 0;
    if (!pp_offsets_fit(P)) return PIKA_NOT_APPLICABLE;
    if (out16) {   // C is a bf16 matrix with pitch ldc: plain bf16 epilogue (no dropout)
        P.out16 = reinterpret_cast<__bf16 *>(C); P.ldo16 = ldc; P.thr = 0; P.scale = 1.f;
        return launch_pp_epi<1>(P, s);
    }
    return launch_pp(P, s);
}

extern "C" int pika_gemm_bf16_nt(const void *A, long long lda, const void *B, long long ldb, float *C,
                                 long long ldc, int M, int N, int K, const float *bias, void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return PIKA_EINVAL;
    if ((K % BK) || (lda & 7) || (ldb & 7) || (ldc & 3) || ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) |
                                                             reinterpret_cast<uintptr_t>(C)) & 15))
        return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N > BN) {
        PPArgs P{};
        P.A = static_cast<const __bf16 *>(A); P.B = static_cast<const __bf16 *>(B); P.C = C; P.bias = bias;
        P.ldb = ldb; P.ldc = ldc; P.a_rpb = M; P.a_batch = 0; P.a_row = lda; P.a_tap = 0; P.a_C = K;
        P.M = M; P.N = N; P.K = K; P.relu = 0;
        return launch_pp(P, s);
    }
    if ((M + BM - 1) / BM > 65535) return PIKA_ETOOBIG;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_glds),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_glds, dim3((N + BN - 1) / BN, (M + BM - 1) / BM), dim3(THREADS), 2 * BUF, s,
                       static_cast<const __bf16 *>(A), static_cast<const __bf16 *>(B), C, M, N, K, lda, ldb, ldc, bias);
    return (int)hipGetLastError();
}

extern "C" int pika_gemm_bf16_nt_lse(const void *A, long long lda, const void *B, long long ldb, float *C, long long ldc,
                                     int M, int N, int K, const float *bias, float *pmax, float *psum, int n_part,
                                     void *stream) {
    if (!A || !B || !C || !pmax || !psum || M <= 0 || N <= 256 || K <= 0 || n_part != ((N + 255) / 256) * 4) return PIKA_EINVAL;
    if ((K % BK) || (lda & 7) || (ldb & 7) || (ldc & 3) || ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) |
                                                             reinterpret_cast<uintptr_t>(C)) & 15))
        return PIKA_EINVAL;
    PPArgs P{};
    P.A = static_cast<const __bf16 *>(A); P.B = static_cast<const __bf16 *>(B); P.C = C; P.bias = bias;
    P.ldb = ldb; P.ldc = ldc; P.a_rpb = M; P.a_batch = 0; P.a_row = lda; P.a_tap = 0; P.a_C = K;
    P.M = M; P.N = N; P.K = K; P.relu = 0;
    P.lse_pm = pmax; P.lse_ps = psum; P.lse_np = n_part;
    return launch_pp(P, static_cast<hipStream_t>(stream));
}

extern "C" int pika_gemm_bf16_nt_lse_f16(const void *A, long long lda, const void *B, long long ldb, void *out16,
                                         long long ldo, int M, int N, int K, const float *bias, float *pmax, float *psum,
                                         int n_part, const int *labels, int T, int U1, int blank, float *gathered,
                                         void *stream) {
    if (!A || !B || !out16 || !pmax || !psum || M <= 0 || N <= 256 || K <= 0 || n_part != ((N + 255) / 256) * 4) return PIKA_EINVAL;
    if ((K % BK) || (lda & 7) || (ldb & 7) || (ldo & 3) || ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) ||
        (reinterpret_cast<uintptr_t>(out16) & 7))
        return PIKA_EINVAL;
    if (gathered && (T <= 0 || U1 <= 0 || blank < 0 || blank >= N || (long long)T * U1 > 0x7fffffffLL || M % (T * U1) ||
                     (U1 > 1 && !labels)))
        return PIKA_EINVAL;
    PPArgs P{};
    P.A = static_cast<const __bf16 *>(A); P.B = static_cast<const __bf16 *>(B); P.bias = bias;
    P.ldb = ldb; P.a_rpb = M; P.a_batch = 0; P.a_row = lda; P.a_tap = 0; P.a_C = K;
    P.M = M; P.N = N; P.K = K; P.relu = 0;
    P.out16 = static_cast<__bf16 *>(out16); P.ldo16 = ldo;
    P.lse_pm = pmax; P.lse_ps = psum; P.lse_np = n_part;
    P.g_labels = labels; P.g_out = gathered; P.g_T = T; P.g_U1 = U1; P.g_blank = blank;
    return launch_pp_epi<4>(P, static_cast<hipStream_t>(stream));
}

extern "C" int pika_gemm_bf16_epilogue(const void *A, long long lda, const void *B, long long ldb, void *out,
                                       long long ldo, int M, int N, int K, const float *bias, int mode,
                                       int relu, float p_drop, unsigned seed, const void *aux,
                                       long long ld_aux, float scale, void *stream) {
    if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0) return PIKA_EINVAL;
    if ((K % BK) || (lda & 7) || (ldb & 7) || (ldo & 3) ||
        ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) || (reinterpret_cast<uintptr_t>(out) & 7))
        return PIKA_EINVAL;
    if (mode != PIKA_EPI_DROPOUT_BF16 && mode != PIKA_EPI_MASK_BF16) return PIKA_EINVAL;
    if (mode == PIKA_EPI_MASK_BF16 && (!aux || (ld_aux & 3) || (reinterpret_cast<uintptr_t>(aux) & 7))) return PIKA_EINVAL;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return PIKA_EINVAL;
    PPArgs P{};
    P.A = static_cast<const __bf16 *>(A); P.B = static_cast<const __bf16 *>(B); P.bias = bias;
    P.ldb = ldb; P.a_rpb = M; P.a_batch = 0; P.a_row = lda; P.a_tap = 0; P.a_C = K;
    P.M = M; P.N = N; P.K = K; P.relu = relu ? 1 : 0;
    P.out16 = static_cast<__bf16 *>(out); P.ldo16 = ldo;
    P.aux = static_cast<const __bf16 *>(aux); P.ld_aux = ld_aux;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == PIKA_EPI_DROPOUT_BF16) {
        P.thr = (unsigned)lrintf(p_drop * 65536.f);
        P.scale = 65536.f / (float)(65536u - P.thr);
        P.seed = seed;
        return launch_pp_epi<1>(P, s);
    }
    P.scale = scale;
    return launch_pp_epi<2>(P, s);
}

extern "C" int pika_gemm_bf16_dropout_residual(const void *A, long long lda, const void *B, long long ldb,
                                               float *out, long long ldo, int M, int N, int K, const float *bias,
                                               float p_drop, unsigned seed, const float *residual,
                                               long long ld_res, void *stream) {
    if (!A || !B || !out || !residual || M <= 0 || N <= 0 || K <= 0) return PIKA_EINVAL;
    if ((K % BK) || (lda & 7) || (ldb & 7) || (ldo & 3) || (ld_res & 3) ||
        ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(out) |
          reinterpret_cast<uintptr_t>(residual)) & 15))
        return PIKA_EINVAL;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return PIKA_EINVAL;
    PPArgs P{};
    P.A = static_cast<const __bf16 *>(A); P.B = static_cast<const __bf16 *>(B); P.bias = bias; P.C = out;
    P.ldb = ldb; P.ldc = ldo; P.a_rpb = M; P.a_batch = 0; P.a_row = lda; P.a_tap = 0; P.a_C = K;
    P.M = M; P.N = N; P.K = K; P.relu = 0;
    P.res = residual; P.ld_res = ld_res;
    P.thr = (unsigned)lrintf(p_drop * 65536.f);
    P.scale = 65536.f / (float)(65536u - P.thr);
    P.seed = seed;
    return launch_pp_epi<3>(P, static_cast<hipStream_t>(stream));
}

extern "C" int pika_dropout_mask_cast_bf16(const float *x, long long ld, int rows, int cols, float p_drop,
                                           unsigned seed, void *out, long long ld_out, void *stream) {
    if (!x || !out || rows <= 0 || cols <= 0 || (cols & 3) || (ld & 3) || (ld_out & 3) || ld < cols || ld_out < cols)
        return PIKA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 7)) return PIKA_EINVAL;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return PIKA_EINVAL;
    const unsigned thr = (unsigned)lrintf(p_drop * 65536.f);
    const long long total = (long long)rows * (cols >> 2);
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pp_mask_cast_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, ld, rows, cols, seed, thr,
                       65536.f / (float)(65536u - thr), static_cast<__bf16 *>(out), ld_out, pika_internal_dropout_salt());
    return (int)hipGetLastError();
}

extern "C" int pika_dropout_keep_mask(unsigned char *mask, int rows, int cols, float p_drop, unsigned seed,
                                      void *stream) {
    if (!mask || rows <= 0 || cols <= 0 || !(p_drop >= 0.f && p_drop < 1.f)) return PIKA_EINVAL;
    const long long n4 = (long long)rows * ((cols + 3) >> 2);
    hipLaunchKernelGGL(pp_keep_mask_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), mask, rows, cols, seed, (unsigned)lrintf(p_drop * 65536.f),
                       pika_internal_dropout_salt());
    return (int)hipGetLastError();
}

// One entry for every product / epilogue combination of the direct-to-LDS kernel (include/pika_gemm.h).
extern "C" int pika_gemm_bf16_ex(const pika_gemm_ex_t *g, void *stream) {
    if (!g || !g->A.ptr || !g->B || !g->out || g->M <= 0 || g->N <= 0 || g->K <= 0) return PIKA_EINVAL;
    if ((g->K % BK) || (g->ldb & 7) || (g->ldo & 3) || (reinterpret_cast<uintptr_t>(g->B) & 15)) return PIKA_EINVAL;
    if (g->ldb < g->K) return PIKA_EINVAL;
    if (!(g->p_drop >= 0.f && g->p_drop < 1.f)) return PIKA_EINVAL;
    PPArgs P{};
    if (!pp_fill_a(g->A, g->K, P)) return PIKA_EINVAL;
    P.B = static_cast<const __bf16 *>(g->B); P.ldb = g->ldb; P.bias = g->bias;
    P.M = g->M; P.N = g->N; P.K = g->K; P.relu = g->relu ? 1 : 0;
    P.thr = (unsigned)lrintf(g->p_drop * 65536.f);
    P.scale = 65536.f / (float)(65536u - P.thr);
    P.seed = g->seed;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (g->epilogue) {
        case PIKA_EPI_F32:
            if (reinterpret_cast<uintptr_t>(g->out) & 15) return PIKA_EINVAL;
            P.C = static_cast<float *>(g->out); P.ldc = g->ldo;
            return launch_pp_epi<0>(P, s);
        case PIKA_EPI_DROPOUT_BF16:
            if ((reinterpret_cast<uintptr_t>(g->out) | reinterpret_cast<uintptr_t>(g->out_lo)) & 7) return PIKA_EINVAL;
            P.out16 = static_cast<__bf16 *>(g->out); P.out16_lo = static_cast<__bf16 *>(g->out_lo); P.ldo16 = g->ldo;
            return launch_pp_epi<1>(P, s);
        case PIKA_EPI_MASK_BF16:
            if (P.a_bounds || !g->aux || (g->ld_aux & 3) || ((reinterpret_cast<uintptr_t>(g->aux) | reinterpret_cast<uintptr_t>(g->out)) & 7))
                return PIKA_EINVAL;
            P.out16 = static_cast<__bf16 *>(g->out); P.ldo16 = g->ldo;
            P.aux = static_cast<const __bf16 *>(g->aux); P.ld_aux = g->ld_aux; P.scale = g->scale; P.relu = 0;
            return launch_pp_epi<2>(P, s);
        case PIKA_EPI_DROPOUT_RESIDUAL:
            if (P.a_bounds || !g->residual || (g->ld_res & 3) ||
                ((reinterpret_cast<uintptr_t>(g->residual) | reinterpret_cast<uintptr_t>(g->out)) & 15))
                return PIKA_EINVAL;
            P.C = static_cast<float *>(g->out); P.ldc = g->ldo; P.res = g->residual; P.ld_res = g->ld_res; P.relu = 0;
            return launch_pp_epi<3>(P, s);
        default:
            return PIKA_EINVAL;
    }
}
