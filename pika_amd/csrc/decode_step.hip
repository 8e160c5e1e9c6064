// pika_amd/csrc/decode_step.hip -- the per-step kernels of the batch beam search (include/pika_decode_step.h).
//
// The search is latency-bound (one step = ~34 GFLOP on B*beam = 1024 rows, ~290 dependent steps), so everything
// here is built for short launches on small M: weights are packed once into MFMA fragment order and streamed by
// each wave straight from L2 into registers (no LDS for the weight operand: a wave owns its output columns, so
// there is nothing to share); only the activation tile, which all four waves of a workgroup need, goes through
// LDS, where fp32 values are split into 1..3 bf16 terms on the way in.  v_mfma_f32_16x16x32_bf16 is issued as
// D^T = W_frag x A_frag: a lane ends up with 4 consecutive output columns of one row.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "pika_decode_step.h"
#include "pika_rnnt.h"  // PIKA_EINVAL
#include "pika_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// terms == 4 ("two fp16 terms"): x = hi + 2^-11 lo', hi = fp16(x), lo' = fp16((x - hi) * 2^11) -- the second term is kept
// at the magnitude of the first (never subnormal where hi is not), and the kernels accumulate hi.hi and the two cross
// products hi.lo' + lo'.hi in SEPARATE accumulators, combined as acc + 2^-11 accx at the end.  22 mantissa bits per operand:
// an fp32 product to ~2^-22 with three MFMAs (three bf16 terms: exact, six MFMAs).  Needs |x| < 65504.
constexpr int TERMS_F16X2 = 4;
constexpr float LO_SCALE = 2048.f, LO_UNSCALE = 1.f / 2048.f;
__host__ __device__ constexpr int planes_of(int terms) { return terms == TERMS_F16X2 ? 2 : terms; }


// ---- weight packing: packed[term][n_tile][k_tile][lane][8], n_tile = 16 columns, k_tile = 32 ----------------
__global__ void dpack_kernel(const float *__restrict__ W, long long ldw, int N, int K, int NT, int KT, int terms,
                             int interleave2, __bf16 *__restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)NT * KT * 64;
    if (gid >= total) return;
    const int lane = (int)(gid & 63);
    const long long tile = gid >> 6;
    const int kt = (int)(tile % KT), nt = (int)(tile / KT);
    int n = nt * 16 + (lane & 15);
    const int k0 = kt * 32 + (lane >> 4) * 8;
    float v[8];
    int src = n;
    if (interleave2 && n < N) src = (n & 1) ? (N / 2 + (n >> 1)) : (n >> 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (n < N && k0 + j < K) ? W[(long long)src * ldw + k0 + j] : 0.f;
    if (terms == TERMS_F16X2) {
        f16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h[j] = (_Float16)v[j];
            l[j] = (_Float16)((v[j] - (float)h[j]) * LO_SCALE);
        }
        *reinterpret_cast<f16x8 *>(out + (((long long)0 * NT + nt) * KT + kt) * 512 + lane * 8) = h;
        *reinterpret_cast<f16x8 *>(out + (((long long)1 * NT + nt) * KT + kt) * 512 + lane * 8) = l;
        return;
    }
    for (int t = 0; t < terms; ++t) {
        bf16x8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h[j] = (__bf16)v[j];
            v[j] -= (float)h[j];
        }
        *reinterpret_cast<bf16x8 *>(out + (((long long)t * NT + nt) * KT + kt) * 512 + lane * 8) = h;
    }
}

// ---- the shared main loop ------------------------------------------------------------------------------------
// Workgroup = 256 threads = 4 waves; tile = BM rows x (4 waves x WN x 16) columns; wave w owns column tiles
// [(ng*4 + w)*WN, +WN).  One step covers KS k-tiles of 32: A (fp32, global) -> registers -> NS bf16 terms -> LDS
// (double buffered, one barrier per step); W fragments global -> registers.  Everything of step k+1 is requested
// before the MFMAs of step k: these launches are latency-bound (one or two workgroups per CU, weights coming from
// L2 / Infinity Cache), so what counts is bytes in flight per wave -- KS*32 columns per request round.
#ifdef PIKA_CORE_TRACE
// tools/core_trace.hip: time stamps of wave 0 of workgroups 0, 8, .., 56: [wg 8][step 40][stamp 8]
__device__ unsigned long long g_core_trace[8 * 40 * 8];
#define CORE_STAMP(step, k) do { if (threadIdx.x == 0 && blockIdx.x < 64 && (blockIdx.x & 7) == 0 && (step) < 40) \
    g_core_trace[((blockIdx.x >> 3) * 40 + (step)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CORE_STAMP(step, k) do { } while (0)
#endif
#ifdef PIKA_CORE_TRACE_FINE     // stamps between the operand requests and the MFMAs (they split the block the two share)
#define CORE_STAMP_IN(step, k) CORE_STAMP(step, k)
#else
#define CORE_STAMP_IN(step, k) do { } while (0)
#endif

template <int BM, int WN, int NSM, int KS>
struct Core {
    static constexpr bool F16 = NSM == TERMS_F16X2;     // two fp16 terms (see TERMS_F16X2); else NSM bf16 terms
    static constexpr int NS = planes_of(NSM);           // 16-bit planes per operand
    static constexpr int MT = BM / 16;
    static constexpr int BK = 32 * KS;
    static constexpr int PITCH = BK + 8;          // bf16 per LDS row: the 16 rows of a fragment read start 4 banks apart
    static constexpr int APT = BM * (BK / 4) / 256;   // float4 loads of A per thread per step
    static constexpr int LDS_BYTES = 2 * NS * BM * PITCH * 2;
    static_assert((BM == 32 || BM == 64) && (BM * (BK / 4)) % 256 == 0, "tile");

    f32x4 acc[MT][WN];

    // Kvalid: columns of A that hold data (a multiple of 4): columns [Kvalid, KT*32) are NOT read -- the packed weight
    // is zero there, but garbage times zero is NaN when the garbage is not finite
    // rows: NULL, or the gather list of the launch: launch row e reads row rows[e] of A
    __device__ inline void run(const float *__restrict__ A, long long lda, int M, int m0, const __bf16 *__restrict__ W,
                               int NT, int KT, int nt0, __bf16 *lds, int Kvalid, const int *__restrict__ rows = nullptr) {
        const int tid = threadIdx.x, lane = tid & 63;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A loader: float4 index f = tid + 256*i -> row f / (BK/4), column group f % (BK/4)
        const float *abase[APT];
        bool aok[APT];
        int aoff[APT], acol[APT];
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int f = tid + 256 * i, r = f / (BK / 4), c4 = f % (BK / 4);
            aok[i] = m0 + r < M;
            abase[i] = A + (aok[i] ? (long long)(rows ? rows[m0 + r] : m0 + r) * lda : 0LL);
            aoff[i] = r * PITCH + c4 * 4;
            acol[i] = c4 * 4;
        }
        const __bf16 *wbase[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) wbase[j] = W + ((long long)(nt0 + j < NT ? nt0 + j : 0) * KT) * 512 + lane * 8;
        const long long term_stride = (long long)NT * KT * 512;
        const int Kcols = Kvalid < KT * 32 ? Kvalid : KT * 32, steps = (KT + KS - 1) / KS;
        f32x4 araw[2][APT];           // the A pieces of step s wait in araw[s & 1]: requested TWO steps ahead (see the loop)
        bf16x8 wreg[WN][KS][NS], wnext[WN][KS][NS];
        // branch-free (the loads of the coming steps sit in ONE basic block with the MFMAs of step k, so that the scheduler
        // can spread them over the MFMAs): rows beyond M read row 0, columns beyond Kcols the last valid group; both are zeroed
        auto load_a = [&](int st, f32x4 (&dst)[APT]) {
#pragma unroll
            for (int i = 0; i < APT; ++i) {
                const int k = st * BK + acol[i];
                const f32x4 v = *reinterpret_cast<const f32x4 *>(abase[i] + (k < Kcols ? k : Kcols - 4));
                dst[i] = (aok[i] && k < Kcols) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto load_w = [&](int st, bf16x8 (&dst)[WN][KS][NS]) {
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int q = 0; q < KS; ++q) {
                    const int kt = st * KS + q < KT ? st * KS + q : KT - 1;     // tail: any valid tile (its A is zero)
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        dst[j][q][s] = *reinterpret_cast<const bf16x8 *>(wbase[j] + s * term_stride + (long long)kt * 512);
                }
        };
        auto stage_a = [&](int buf, const f32x4 (&from)[APT]) {
            __bf16 *dst = lds + buf * (NS * BM * PITCH);
#pragma unroll
            for (int i = 0; i < APT; ++i) {
                f32x4 r = from[i];
                if constexpr (F16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = fminf(fmaxf(r[e], -65504.f), 65504.f);   // saturate, never inf
                    const f16x4 h = __builtin_convertvector(r, f16x4);
                    const f16x4 l = __builtin_convertvector((r - __builtin_convertvector(h, f32x4)) * LO_SCALE, f16x4);
                    *reinterpret_cast<f16x4 *>(dst + aoff[i]) = h;
                    *reinterpret_cast<f16x4 *>(dst + BM * PITCH + aoff[i]) = l;
                } else {
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const bf16x4 h = __builtin_convertvector(r, bf16x4);
                        *reinterpret_cast<bf16x4 *>(dst + s * (BM * PITCH) + aoff[i]) = h;
                        if (s + 1 < NS) r = r - __builtin_convertvector(h, f32x4);
                    }
                }
            }
        };
        [[maybe_unused]] f32x4 accx[MT][WN];     // cross products of the fp16 two-term mode
        if constexpr (F16) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) accx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        CORE_STAMP(39, 0);
        load_a(0, araw[0]);
        load_w(0, wreg);
        stage_a(0, araw[0]);
        load_a(steps > 1 ? 1 : 0, araw[1]);
        CORE_STAMP(39, 1);
        __syncthreads();
        CORE_STAMP(39, 2);
        // Step st (parity P): request the A pieces of step st + 2 and the W fragments of step st + 1, run the MFMAs of step st
        // on LDS buffer P, convert the A pieces of step st + 1 (requested during step st - 1: a whole step ago, so the
        // conversion does not wait for memory -- one step ahead, every step ended with ~0.3-0.5 us of exactly that wait) into
        // buffer P ^ 1, barrier.  The tail requests clamp to the last step (never used) instead of branching.
        auto step = [&](int st, auto parity) {
            constexpr int P = decltype(parity)::value;
            CORE_STAMP(st, 0);
            load_a(st + 2 < steps ? st + 2 : steps - 1, araw[P]);
            load_w(st + 1 < steps ? st + 1 : steps - 1, wnext);
            CORE_STAMP_IN(st, 1);
            const __bf16 *src = lds + P * (NS * BM * PITCH);
            // products of one kind across all accumulators before the next kind (consecutive MFMAs never share an
            // accumulator); smallest products first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
            constexpr int NP = NS == 3 ? 6 : (NS == 2 ? 3 : 1);
            constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PA[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                bf16x8 a[MT][NS];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        a[i][s] = *reinterpret_cast<const bf16x8 *>(src + s * (BM * PITCH) + (i * 16 + (lane & 15)) * PITCH +
                                                                    q * 32 + (lane >> 4) * 8);
                if constexpr (F16) {
                    auto f16 = [](const bf16x8 &v) { return __builtin_bit_cast(f16x8, v); };
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f16(wreg[j][q][1]), f16(a[i][0]), accx[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f16(wreg[j][q][0]), f16(a[i][1]), accx[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f16(wreg[j][q][0]), f16(a[i][0]), acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                for (int p = 6 - NP; p < 6; ++p)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[j][q][PW[p]], a[i][PA[p]], acc[i][j], 0, 0, 0);
                }
            }
#ifndef PIKA_CORE_NO_INTERLEAVE
            {
                // one operand request per few MFMAs: all four waves issuing their 12-14 loads of 1 KB at once right after the
                // barrier is 0.3-0.45 us of queueing at the CU's address unit (64 B / clk) in front of every step's products
                constexpr int NLOADS = APT + WN * KS * NS, NMFMA = MT * WN * KS * (NS == 3 ? 6 : (NS == 2 ? 3 : 1));
                constexpr int PER = NMFMA / NLOADS > 0 ? NMFMA / NLOADS : 1;
#pragma unroll
                for (int i = 0; i < NLOADS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                }
            }
#endif
            CORE_STAMP_IN(st, 2);
            stage_a(P ^ 1, araw[P ^ 1]);
            CORE_STAMP_IN(st, 3);
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int q = 0; q < KS; ++q)
#pragma unroll
                    for (int s = 0; s < NS; ++s) wreg[j][q][s] = wnext[j][q][s];
            __syncthreads();
            CORE_STAMP(st, 4);
        };
        for (int st = 0; st < steps; st += 2) {
            step(st, std::integral_constant<int, 0>());
            if (st + 1 < steps) step(st + 1, std::integral_constant<int, 1>());
        }
        if constexpr (F16) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] += accx[i][j] * LO_UNSCALE;
        }
    }
};

struct DG {   // device copy of pika_dgemm_t
    const float *A; long long lda; const __bf16 *W; const float *bias; const float *res; long long ldr;
    float *C; long long ldc; float *C2; long long ldc2; const long long *node; long long skip_node;
    const float *e_all; const long long *t_idx; int T, beam, M, N, NT, KT, flags;
    const int *m_dev; const long long *crow;
    const int *rowlist; const int *rowoff_dev;
    int Kvalid;   // K when K % 4 == 0 (columns beyond are never read), else ceil32(K) (the caller zero-pads, as the header says)
    const float *ln_g, *ln_b; float ln_eps;     // LayerNorm of the A rows on the way in (dgemm_sk_kernel only)
};

// blockIdx -> (row tile, column group) so that a column group always lands on the same XCD (block b runs on XCD
// b % 8) and its row tiles are dispatched back to back: the row tiles that share a slab of W then find it in that
// XCD's L2 instead of each pulling it from the Infinity Cache at the per-CU fetch rate (a quarter of the L2 rate),
// and an XCD works on ONE slab at a time (its 4 MiB L2 holds a slab, not all the slabs of its column groups).
// false = idle workgroup.
__device__ inline bool xcd_tile(int n_groups, int m_tiles, int &mg, int &ng) {
    const int g = blockIdx.x, xcd = g & 7, i = g >> 3;
    ng = xcd + 8 * (i / m_tiles);
    mg = i % m_tiles;
    return ng < n_groups;
}

// The other way round, for many rows (M = B*beam): an XCD owns a band of row tiles -- their A rows (a few hundred
// KB) stay in its L2 -- and walks the column groups with the band's row tiles dispatched back to back, so a slab of
// W is fetched from beyond L2 once per XCD (8x in total) instead of once per row tile.
// ... and for a product whose WEIGHTS are the large operand (the vocabulary product of a search step: 20 MB of two-term W
// against 4 MB of rows at B * beam = 1024, 128 KB at 32): an XCD owns the column groups x, x + 8, .. and takes every row tile
// of them, the row tiles of a group dispatched back to back -- a slab of W is fetched from beyond L2 by ONE XCD (1x in
// total; the rows 8x), and at one row tile the launch's workgroups spread over all eight XCDs instead of sitting on one.
// Grid: 8 * ceil(n_groups / 8) * m_tiles.
__device__ inline bool xcd_tile_cols(int n_groups, int m_tiles, int &mg, int &ng) {
    const int g = blockIdx.x, xcd = g & 7, i = g >> 3;
    ng = xcd + 8 * (i / m_tiles);
    mg = i % m_tiles;
    return ng < n_groups;
}

__device__ inline bool xcd_tile_rows(int n_groups, int m_tiles, int &mg, int &ng) {
    const int g = blockIdx.x, xcd = g & 7, i = g >> 3;
    const int band = (m_tiles + 7) >> 3;
    mg = xcd * band + i % band;
    ng = i / band;
    return mg < m_tiles && ng < n_groups;
}

// One lane's 4 consecutive result columns [c0, c0 + 4) of buffer row r: bias, ReLU, residual, the gate, the stores.
// pre: the additive terms were requested ahead of the product (pb = bias, or the gate's encoder halves in the order of v;
// pr = residual) -- only for whole aligned groups of 4 columns.
__device__ inline f32x4 dg_gate_terms(const DG &p, int r, int c0) {
    const int H = p.N >> 1, j0 = c0 >> 1;
    long long t = p.t_idx[r];
    t = t < 0 ? 0 : (t > p.T - 1 ? p.T - 1 : t);
    const float *e = p.e_all + ((long long)(r / p.beam) * p.T + t) * p.N;
    return f32x4{e[j0], e[H + j0], j0 + 1 < H ? e[j0 + 1] : 0.f, j0 + 1 < H ? e[H + j0 + 1] : 0.f};
}

__device__ inline void dg_finish(const DG &p, int r, int c0, f32x4 v, bool pre, f32x4 pb, f32x4 pr) {
    if (c0 >= p.N) return;
    if (p.flags & PIKA_DG_GATE) {
        // columns (2j, 2j+1) = (fc1, fc_gate) of joint unit j; this lane holds units c0/2 and c0/2 + 1
        const int H = p.N >> 1, j0 = c0 >> 1;
        const f32x4 e = pre ? pb : dg_gate_terms(p, r, c0);
        float o[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float z1 = v[2 * u] + e[2 * u], zg = v[2 * u + 1] + e[2 * u + 1];
            o[u] = tanhf(z1) * (1.f / (1.f + expf(-zg)));
        }
        const long long row = p.crow ? p.crow[r] : (long long)r;
        float *dst = p.C + row * p.ldc + j0;
        dst[0] = o[0];
        if (j0 + 1 < H) dst[1] = o[1];
        if (p.C2) *reinterpret_cast<f32x4 *>(p.C2 + row * p.ldc2 + c0) = v;      // (N % 4 == 0, ldc2 % 4 == 0)
        return;
    }
    const bool full = c0 + 3 < p.N;
    if (pre) {
        v += pb;
    } else if (p.bias) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c0 + u < p.N) v[u] += p.bias[c0 + u];
    }
    if (p.flags & PIKA_DG_RELU) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
    }
    if (pre) {
        v += pr;
    } else if (p.res) {
        const float *rp = p.res + (long long)r * p.ldr + c0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c0 + u < p.N) v[u] += rp[u];
    }
    const long long nd = p.node ? p.node[r] : 0;
    if ((p.flags & PIKA_DG_ROWMASK) && nd == p.skip_node) return;
    float *dst = p.C + (p.crow ? p.crow[r] : (long long)r) * p.ldc + c0;
    if (full && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
        *reinterpret_cast<f32x4 *>(dst) = v;
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c0 + u < p.N) dst[u] = v[u];
    }
    if (p.C2) {
        float *d2 = p.C2 + nd * p.ldc2 + c0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c0 + u < p.N) d2[u] = v[u];
    }
}

template <int BM, int NS>
struct DgCfg { static constexpr int KS = BM == 32 ? 4 : 2; typedef Core<BM, 1, NS, KS> core_t; };

template <int BM, int NS>
__global__ __launch_bounds__(256) void dgemm_kernel(DG p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typename DgCfg<BM, NS>::core_t core;
    const int n_groups = (p.NT + 3) / 4;
    int mg, ng;
    if (!xcd_tile_rows(n_groups, (p.M + BM - 1) / BM, mg, ng)) return;
    int M = p.M;
    if (p.m_dev) M = min(M, *p.m_dev);            // rows actually in use this step
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = mg * BM, nt0 = ng * 4 + wave;
    if (m0 >= M) return;
    // launch rows as a gather list: launch row e stands for row rows[e] of A, res, node, crow, t_idx and C
    const int *rows = p.rowlist ? p.rowlist + (p.rowoff_dev ? *p.rowoff_dev : 0) : nullptr;
    core.run(p.A, p.lda, M, m0, p.W, p.NT, p.KT, nt0, reinterpret_cast<__bf16 *>(smem), p.Kvalid, rows);
    if (nt0 >= p.NT) return;
    const int c0 = nt0 * 16 + (lane >> 4) * 4;          // first of this lane's 4 consecutive columns
    if (c0 >= p.N) return;
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) {
        const int e_ = m0 + i * 16 + (lane & 15);
        if (e_ >= M) continue;
        dg_finish(p, rows ? rows[e_] : e_, c0, core.acc[i][0], false, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f});
    }
}

// ---- wide products on some hundred rows (the LSTM gate products of a LAS scoring pass: 64..1000 rows x 4096 x 2560) --
// What bounds these launches is the rate at which ONE CU pulls operand bytes that are not in its XCD's L2 -- ~50 GB/s,
// a third of its L1 rate, with two steps of requests in flight: W (33-44 MB) streams from the Infinity Cache, and the row
// tiles that share a slab of it run in lockstep, so each of them waits for the same fills.  A tile of BM x BN needs
// 4 (BM + BN) K bytes: dgemm_kernel's 64 x 64 tiles on 2 workgroups per CU took 80 us at 470 rows (57 us at 64 rows: 40
// dependent steps); 64 x 128 tiles, one per CU: 1.5 MB per CU, 42 us at 470 rows, 37 us at 64 (tools/las_gemm_bench.py).
// Measured no better: 128-row tiles on eight waves (fewer CUs), 32-row tiles on two workgroups per CU (25 us up to 256
// rows, 80 us at 960), W fragments two or three steps ahead in registers (43 -> 34 us while one workgroup per CU
// suffices, two rounds beyond), the conversion of the next A pieces scheduled under the MFMAs (tools/core_trace.hip).
// XCD x owns the column groups x, x + 8, ..: their slabs of W are fetched into ONE L2.
template <int NS>
__global__ __launch_bounds__(256) void dgemm_wide_kernel(DG p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = 64;
    Core<BM, 2, NS, 2> core;
    const int n_groups = (p.NT + 7) / 8;
    int mg, ng;
    {   // XCD x owns the column groups x, x + 8, ..; its workgroups walk them ROW TILE BY ROW TILE: the launch is sized for
        // the largest step of a scoring pass, and the row tiles beyond this step's rows (idle workgroups: ~0.3 us of
        // dispatch each, 20 us in front of a 64-row step when they are interleaved with the live ones) come last
        const int g = blockIdx.x, xcd = g & 7, i = g >> 3, gpx = (n_groups + 7) >> 3;
        ng = xcd + 8 * (i % gpx);
        mg = i / gpx;
        if (ng >= n_groups) return;
    }
    int M = p.M;
    if (p.m_dev) M = min(M, *p.m_dev);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = mg * BM, nt0 = ng * 8 + wave * 2;
    if (m0 >= M) return;
    const int *rows = p.rowlist ? p.rowlist + (p.rowoff_dev ? *p.rowoff_dev : 0) : nullptr;
    core.run(p.A, p.lda, M, m0, p.W, p.NT, p.KT, nt0, reinterpret_cast<__bf16 *>(smem), p.Kvalid, rows);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c0 = (nt0 + j) * 16 + (lane >> 4) * 4;
        if (nt0 + j >= p.NT || c0 >= p.N) continue;
#pragma unroll
        for (int i = 0; i < BM / 16; ++i) {
            const int e_ = m0 + i * 16 + (lane & 15);
            if (e_ >= M) continue;
            dg_finish(p, rows ? rows[e_] : e_, c0, core.acc[i][j], false, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f});
        }
    }
}

// ---- products on few rows: one 16-column tile per workgroup, the reduction split over its waves --------------------
// The prediction-network products of a search step run on the ~B*beam/6 rows that emitted a label: a handful of row
// tiles.  dgemm_kernel (a wave owns 16 columns and walks the WHOLE reduction in steps of 64-128 columns) is then a chain
// of dependent memory round trips -- m_dev, first tiles, one per step, bias, residual: 13.7 us for 170 x 512 x 512 where
// the MFMAs need 0.2 us -- on 48 of 256 CUs.  Here a workgroup owns ONE 16-column tile of 16 MT rows and its KW waves
// split the reduction, at most CH k-tiles of 32 each: every operand byte of a tile (the wave's W fragments straight from
// L2 / Infinity Cache, its A pieces in MFMA fragment order straight from global memory -- 8 consecutive fp32 of one row
// per lane, no LDS staging: nothing is shared between the waves) is requested in ONE round before the first MFMA,
// together with the bias / residual / gate terms of the epilogue; the waves' partial tiles meet in LDS (summed in wave
// order: the result does not depend on timing).  Many small workgroups, each with a sliver of the weight bytes, one
// memory round trip per tile, several workgroups per CU covering each other's latency.
// Workgroups walk the tiles of their XCD's column tiles (column tile nt lives on XCD nt % 8: the row tiles sharing a
// slab of W find it in that XCD's L2).  LN: the rows of A are layer-normalised on the way in (the statistics need the
// whole row: partial sums of the waves meet in LDS, two passes like ln_fwd_kernel) -- one launch instead of two.
#ifdef PIKA_SK_TRACE      // profiling builds only (tools/sk_trace.py): time stamps of workgroup 0's first wave, summed over launches
__device__ unsigned long long g_sk_trace[8][8];
__device__ inline unsigned long long *sk_slots() { __shared__ unsigned long long s_[8]; return s_; }
#define SK_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { sk_slots()[k] = __builtin_amdgcn_s_memrealtime(); \
    if ((k) == 6) { const int v_ = (LN ? 4 : 0) + (KW == 8 ? 2 : 0) + (PIPE ? 1 : 0); \
        for (int i_ = 1; i_ < 7; ++i_) g_sk_trace[v_][i_] += sk_slots()[i_] - sk_slots()[0]; g_sk_trace[v_][0] += 1; } } } while (0)
extern "C" int pika_debug_sk_trace(unsigned long long *out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_sk_trace), sizeof(g_sk_trace));
}
#else
#define SK_STAMP(k) do { } while (0)
#endif

template <int NSM, int KW, int MT, int WN, int CH, bool LN, bool PIPE>
__global__ __launch_bounds__(64 * KW, KW == 16 ? 4 : 2) void dgemm_sk_kernel(DG p) {
    constexpr bool F16 = NSM == TERMS_F16X2;
    constexpr int NS = planes_of(NSM);
    constexpr int BMR = 16 * MT, TN = 16 * WN;              // rows / columns of a tile
    constexpr int CQ = TN / 4, ETH = BMR * CQ;              // epilogue threads: row tid / CQ, columns [4 (tid % CQ), +4)
    static_assert(ETH <= 64 * KW, "epilogue threads");
    __shared__ __attribute__((aligned(16))) float red[KW][BMR][TN];
    __shared__ float ln_part[LN ? KW : 1][BMR];
    __shared__ __attribute__((aligned(16))) float ln_gb[LN ? 2 : 1][LN ? 1024 : 4];      // gamma | beta (K <= 1024)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: the wave's k-tile range is uniform)
    SK_STAMP(0);
    int M = p.M;
    if (p.m_dev) M = min(M, *p.m_dev);
    if (M <= 0) return;
    SK_STAMP(1);
    const int m_tiles = (M + BMR - 1) / BMR;
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, jstride = gridDim.x >> 3;
    const int NG = (p.NT + WN - 1) / WN;                    // column groups of WN tiles; group g lives on XCD g % 8
    const int total = ((NG - xcd + 7) >> 3) * m_tiles;      // tiles of this XCD
    if (j0 >= total) return;
    const int *rows = p.rowlist ? p.rowlist + (p.rowoff_dev ? *p.rowoff_dev : 0) : nullptr;
    const int per = (p.KT + KW - 1) / KW;                   // k-tiles per wave, taken CH at a time
    const int kt_lo = min(wave * per, p.KT);
    const int nk = min(p.KT, kt_lo + per) - kt_lo;          // (scalar) this wave's k-tiles: [kt_lo, kt_lo + nk)
    const long long term_stride = (long long)p.NT * p.KT * 512;
    const int Kcols = p.Kvalid < p.KT * 32 ? p.Kvalid : p.KT * 32;
    const int kq = (lane >> 4) * 8;                         // this lane's 8 consecutive reduction indices inside a k-tile
    if constexpr (LN) {
        for (int i = tid * 4; i < p.KT * 32; i += 256 * KW) {
            *reinterpret_cast<f32x4 *>(&ln_gb[0][i]) = *reinterpret_cast<const f32x4 *>(p.ln_g + i);
            *reinterpret_cast<f32x4 *>(&ln_gb[1][i]) = *reinterpret_cast<const f32x4 *>(p.ln_b + i);
        }
    }

    for (int q = j0; q < total; q += jstride) {
        const int ti = q / m_tiles, mg = q - ti * m_tiles;
        const int grp = xcd + 8 * ti, m0 = mg * BMR;
        // (1) the loads other loads depend on: gather list, frame index of the gate rows -- oldest in the queue, so
        // waiting for them later does not wait for the operands
        const int e_row = m0 + tid / CQ, e_c0 = grp * TN + (tid % CQ) * 4;
        const bool e_on = tid < ETH && e_row < M && e_c0 < p.N;
        int e_r = e_on ? e_row : m0;
        int a_r[MT];
        bool aok[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int e = m0 + t * 16 + (lane & 15);
            aok[t] = e < M;
            a_r[t] = aok[t] ? e : m0;                       // rows beyond M read the tile's first row (zeroed below)
        }
        if (rows) {
            e_r = rows[e_r];
#pragma unroll
            for (int t = 0; t < MT; ++t) a_r[t] = rows[a_r[t]];
        }
        const bool gate = p.flags & PIKA_DG_GATE;
        long long e_t = 0;
        if (gate && e_on) e_t = p.t_idx[e_r];
        const __bf16 *wb[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int nt = grp * WN + j < p.NT ? grp * WN + j : p.NT - 1;       // (a group's tail tile: computed, not stored)
            wb[j] = p.W + ((long long)nt * p.KT + kt_lo) * 512 + lane * 8;
        }
        const float *ab[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) ab[t] = p.A + (long long)a_r[t] * p.lda + kt_lo * 32 + kq;
        f32x4 acc[MT][WN], accx[MT][WN];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[t][j] = accx[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bool pre = false;
        f32x4 pb = {0.f, 0.f, 0.f, 0.f}, pr = {0.f, 0.f, 0.f, 0.f};
        struct Req { f32x4 a[MT][CH][2]; bf16x8 w[WN][CH][NS]; };
        // every operand of a round, at constant offsets from one base per row tile / plane (k-tiles beyond the wave's range
        // are skipped by scalar branches; columns [Kvalid, ceil32(K)) are readable by contract, zeroed in `compute`)
        auto issue = [&](int r0, Req &r) {
            const int nr = nk - r0;
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (c < nr) {
#pragma unroll
                    for (int j = 0; j < WN; ++j)
#pragma unroll
                        for (int s_ = 0; s_ < NS; ++s_)
                            r.w[j][c][s_] = *reinterpret_cast<const bf16x8 *>(wb[j] + s_ * term_stride + (long long)(r0 + c) * 512);
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        r.a[t][c][0] = *reinterpret_cast<const f32x4 *>(ab[t] + (r0 + c) * 32);
                        r.a[t][c][1] = *reinterpret_cast<const f32x4 *>(ab[t] + (r0 + c) * 32 + 4);
                    }
                }
        };
        Req q0;
        [[maybe_unused]] Req q1;
        SK_STAMP(2);
        issue(0, q0);
        SK_STAMP(3);
        {
            // (3) the epilogue's additive terms (whole aligned groups of 4 columns; others are read in dg_finish)
            if (e_on) {
                if (gate) {
                    pre = true;
                    const int H = p.N >> 1, jj = e_c0 >> 1;
                    const long long t = e_t < 0 ? 0 : (e_t > p.T - 1 ? p.T - 1 : e_t);
                    const float *e = p.e_all + ((long long)(e_r / p.beam) * p.T + t) * p.N;
                    pb = f32x4{e[jj], e[H + jj], e[jj + 1], e[H + jj + 1]};     // (N % 4 == 0: both units exist)
                } else if (e_c0 + 3 < p.N && !(p.ldr & 3) && !(reinterpret_cast<uintptr_t>(p.res) & 15) &&
                           !(reinterpret_cast<uintptr_t>(p.bias) & 15)) {
                    pre = true;
                    if (p.bias) pb = *reinterpret_cast<const f32x4 *>(p.bias + e_c0);
                    if (p.res) pr = *reinterpret_cast<const f32x4 *>(p.res + (long long)e_r * p.ldr + e_c0);
                }
            }
        }
        [[maybe_unused]] float mu[MT], rs[MT];
        {
            if constexpr (LN) {
                // (the host entry admits LN only when K is a multiple of 32 -- no partial k-tiles -- and one round covers it)
                float s_[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) s_[t] = 0.f;
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (c < nk) {
#pragma unroll
                        for (int t = 0; t < MT; ++t)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const f32x4 x = q0.a[t][c][h];
                                s_[t] += (x[0] + x[1]) + (x[2] + x[3]);
                            }
                    }
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    s_[t] += __shfl_xor(s_[t], 16);
                    s_[t] += __shfl_xor(s_[t], 32);
                    if (lane < 16) ln_part[wave][t * 16 + lane] = s_[t];
                }
                __syncthreads();
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    float v = 0.f;
#pragma unroll
                    for (int w_ = 0; w_ < KW; ++w_) v += ln_part[w_][t * 16 + (lane & 15)];
                    mu[t] = v / (float)p.Kvalid;
                    s_[t] = 0.f;
                }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (c < nk) {
#pragma unroll
                        for (int t = 0; t < MT; ++t)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const f32x4 d = q0.a[t][c][h] - mu[t];
                                s_[t] += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                            }
                    }
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    s_[t] += __shfl_xor(s_[t], 16);
                    s_[t] += __shfl_xor(s_[t], 32);
                    if (lane < 16) ln_part[wave][t * 16 + lane] = s_[t];
                }
                __syncthreads();
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    float v = 0.f;
#pragma unroll
                    for (int w_ = 0; w_ < KW; ++w_) v += ln_part[w_][t * 16 + (lane & 15)];
                    rs[t] = rsqrtf(v / (float)p.Kvalid + p.ln_eps);
                }
            }
        }
        auto compute = [&](int r0, const Req &r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (c >= nk - r0) break;
                const int k = (kt_lo + r0 + c) * 32 + kq;
                bf16x8 afr[MT][NS];
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    f32x4 x[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        x[h] = r.a[t][c][h];
                        if constexpr (LN)
                            x[h] = (x[h] - mu[t]) * rs[t] * *reinterpret_cast<const f32x4 *>(&ln_gb[0][k + 4 * h]) +
                                   *reinterpret_cast<const f32x4 *>(&ln_gb[1][k + 4 * h]);
                        if (!(aok[t] && k + 4 * h < Kcols)) x[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    if constexpr (F16) {
                        f16x8 hi, lo;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f32x4 v = x[h];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], -65504.f), 65504.f);   // saturate, never inf
                            const f16x4 h4 = __builtin_convertvector(v, f16x4);
                            const f16x4 l4 = __builtin_convertvector((v - __builtin_convertvector(h4, f32x4)) * LO_SCALE, f16x4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { hi[4 * h + e] = h4[e]; lo[4 * h + e] = l4[e]; }
                        }
                        afr[t][0] = __builtin_bit_cast(bf16x8, hi);
                        afr[t][1] = __builtin_bit_cast(bf16x8, lo);
                    } else {
#pragma unroll
                        for (int s_ = 0; s_ < NS; ++s_) {
                            bf16x8 o;
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const bf16x4 b4 = __builtin_convertvector(x[h], bf16x4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[4 * h + e] = b4[e];
                                if (s_ + 1 < NS) x[h] = x[h] - __builtin_convertvector(b4, f32x4);
                            }
                            afr[t][s_] = o;
                        }
                    }
                }
                if constexpr (F16) {
                    auto f16 = [](const bf16x8 &v) { return __builtin_bit_cast(f16x8, v); };
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            accx[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f16(r.w[j][c][1]), f16(afr[t][0]), accx[t][j], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            accx[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f16(r.w[j][c][0]), f16(afr[t][1]), accx[t][j], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f16(r.w[j][c][0]), f16(afr[t][0]), acc[t][j], 0, 0, 0);
                } else {
                    // smallest products first, as in Core: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
                    constexpr int NP = NS == 3 ? 6 : (NS == 2 ? 3 : 1);
                    constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PA[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int pp = 6 - NP; pp < 6; ++pp)
#pragma unroll
                        for (int t = 0; t < MT; ++t)
#pragma unroll
                            for (int j = 0; j < WN; ++j)
                                acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r.w[j][c][PW[pp]], afr[t][PA[pp]], acc[t][j], 0, 0, 0);
                }
            }
        };
        if constexpr (PIPE) {
            // rounds of CH k-tiles, the next round's requests in flight while this one multiplies
            for (int r0 = 0; r0 < nk; r0 += 2 * CH) {
                if (r0 + CH < nk) issue(r0 + CH, q1);
                compute(r0, q0);
                if (r0 + CH < nk) {
                    if (r0 + 2 * CH < nk) issue(r0 + 2 * CH, q0);
                    compute(r0 + CH, q1);
                }
            }
        } else {
            compute(0, q0);             // (the host entry chose KW such that one round covers the reduction)
        }
        SK_STAMP(4);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if constexpr (F16) acc[t][j] += accx[t][j] * LO_UNSCALE;
                *reinterpret_cast<f32x4 *>(&red[wave][t * 16 + (lane & 15)][j * 16 + (lane >> 4) * 4]) = acc[t][j];
            }
        __syncthreads();
        SK_STAMP(5);
        if (e_on) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(&red[0][tid / CQ][(tid % CQ) * 4]);
#pragma unroll
            for (int w_ = 1; w_ < KW; ++w_) v += *reinterpret_cast<const f32x4 *>(&red[w_][tid / CQ][(tid % CQ) * 4]);
            dg_finish(p, e_r, e_c0, v, pre, pb, pr);
        }
        SK_STAMP(6);
        if (q + jstride < total) __syncthreads();           // `red` is rewritten by the next tile
    }
}

// ---- prediction-network bookkeeping ----------------------------------------------------------------------------
struct PrepDev {
    pika_dstep_prep_t p;
};

// The joint's prediction half follows the row's parent; rows that did not emit a label get this step's joint hidden
// (pika_dstep_joint_t).  t_new: the row's frame index after this step's increment.
__device__ inline void joint_carry(const pika_dstep_joint_t &j, int src, long long pr, int r, int beam, bool commit,
                                   long long t_new) {
    if (!j.pj[0]) return;
    const int tid = threadIdx.x, JH = j.JH;
    const float *ps = j.pj[src] + pr * 2 * JH;
    float *pd = j.pj[src ^ 1] + (long long)r * 2 * JH;
    const long long t = t_new < 0 ? 0 : (t_new > j.T - 1 ? j.T - 1 : t_new);
    const float *e = j.e_all + ((long long)(r / beam) * j.T + t) * 2 * JH;
    for (int c = tid * 4; c < JH; c += 1024) {          // units [c, c + 4): pj columns [2c, 2c + 8)
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(ps + 2 * c), p1 = *reinterpret_cast<const f32x4 *>(ps + 2 * c + 4);
        *reinterpret_cast<f32x4 *>(pd + 2 * c) = p0;
        *reinterpret_cast<f32x4 *>(pd + 2 * c + 4) = p1;
        if (commit) continue;
        const f32x4 e1 = *reinterpret_cast<const f32x4 *>(e + c), eg = *reinterpret_cast<const f32x4 *>(e + JH + c);
        const float z1[4] = {p0[0] + e1[0], p0[2] + e1[1], p1[0] + e1[2], p1[2] + e1[3]};
        const float zg[4] = {p0[1] + eg[0], p0[3] + eg[1], p1[1] + eg[2], p1[3] + eg[3]};
        f32x4 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = tanhf(z1[u]) * (1.f / (1.f + expf(-zg[u])));
        *reinterpret_cast<f32x4 *>(j.h + (long long)r * JH + c) = o;
    }
}

// (Round 6: what bounds this launch is BYTES, not its chain of dependent loads -- 17.7 MB read + 16.2 MB written per step at
// 1024 rows (profiles/r6_decode_step_pmc.json): the prediction half of the joint alone is 8 KB in and 8 KB out per row.  A
// variant with every load of a row requested up front, in three rounds instead of seven, ran 14.0 us against 14.4.)
__global__ __launch_bounds__(256) void dstep_prep_kernel(PrepDev a) {
    const pika_dstep_prep_t &p = a.p;
    if (p.stop && *p.stop) return;
    __shared__ int slot_s;
    const int r = blockIdx.x, tid = threadIdx.x;
    const long long s = p.step_t[0];
    const int src = (int)(s & 1), dst = src ^ 1;
    const int b = r / p.beam;
    const long long pr = (long long)b * p.beam + p.prev_k[r];
    const long long tok = p.y[r];
    const bool commit = tok > p.blk;
    long long pos = p.hyp_len[r];
    if (pos > p.L - 1) pos = p.L - 1;
    const long long node = 1 + s * p.rows + r;
    // state and ancestry follow the parent (transducer_decoder.py:188-202)
    const float *ss = p.state[src] + pr * p.H;
    float *sd = p.state[dst] + (long long)r * p.H;
    for (int c = tid * 4; c < p.H; c += 1024) {
        if (c + 3 < p.H) *reinterpret_cast<f32x4 *>(sd + c) = *reinterpret_cast<const f32x4 *>(ss + c);
        else for (int u = 0; c + u < p.H; ++u) sd[c + u] = ss[c + u];
    }
    joint_carry(p.joint, src, pr, r, p.beam, commit, p.t_idx[r] + (tok == p.blk ? 1 : 0));
    const long long *as = p.anc[src] + pr * p.L;
    long long *ad = p.anc[dst] + (long long)r * p.L;
    const long long ncopy = pos + 1 < p.L ? pos + 1 : p.L;      // positions > pos are never read
    for (long long j = tid; j < ncopy; j += 256) {
        long long v = as[j];
        if (commit && j == pos) v = node;
        ad[j] = v;
    }
    __syncthreads();            // (every thread has read t_idx[r] in joint_carry)
    if (tid == 0) {
        if (tok == p.blk) p.t_idx[r] += 1;                       // :129
        // rows that emitted a label get a slot in the compact row list the prediction-net launches work on
        slot_s = commit ? atomicAdd(p.count + src, 1) : -1;
    }
    __syncthreads();
    if (!commit) return;
    const int slot = slot_s;
    if (tid == 0) {
        p.rowmap[slot] = r;
        if (p.joint.rowmap32) p.joint.rowmap32[slot] = r;
        p.node[slot] = node;
        p.pos[slot] = pos;
    }
    // taps p-4 .. p-1 through the PARENT's ancestry (positions < p are unchanged by this step)
    long long tap[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long q = pos - 4 + j;
        tap[j] = q >= 0 ? as[q] : p.zero_node;
    }
    for (int l = 0; l < p.layers; ++l) {
        const int C = p.C[l];
        float *arow = p.A[l] + (long long)slot * p.lda[l];
        for (int e = tid; e < 4 * C; e += 256) {
            const int j = e / C, c = e - j * C;
            arow[e] = p.X[l][tap[j] * C + c];
        }
        if (l == 0) {
            for (int c = tid; c < C; c += 256) {
                const float x = p.emb[tok * C + c];
                arow[4 * C + c] = x;
                p.X[0][node * C + c] = x;
            }
        }
    }
}


// ---- LSTM prediction network (trainer/model/transducer.py:55-61; decoder/transducer_decoder.py:139-148) --------
// A row's state is [h_0 | c_0 | h_1 | c_1 | ...] (SP = layers * 2 * H floats), double-buffered like the transformer
// state.  Every row follows its parent; rows that emitted a label get a slot in the compact row list and the input
// rows of the layers' gate products: A[0][slot] = [emb(label) | h_0], A[l][slot] = [. | h_l] (the first block of
// A[l > 0] is written by the cell kernel of layer l-1).
__global__ __launch_bounds__(256) void dstep_prep_lstm_kernel(pika_dstep_prep_lstm_t p) {
    if (p.stop && *p.stop) return;
    __shared__ int slot_s;
    const int r = blockIdx.x, tid = threadIdx.x;
    const long long s = p.step_t[0];
    const int src = (int)(s & 1), dst = src ^ 1;
    const int b = r / p.beam;
    const long long pr = (long long)b * p.beam + p.prev_k[r];
    const long long tok = p.y[r];
    const bool commit = tok > p.blk;
    const long long SP = (long long)p.layers * 2 * p.H;
    const float *ss = p.state[src] + pr * SP;
    float *sd = p.state[dst] + (long long)r * SP;
    for (long long c = tid * 4; c < SP; c += 1024) *reinterpret_cast<f32x4 *>(sd + c) = *reinterpret_cast<const f32x4 *>(ss + c);
    joint_carry(p.joint, src, pr, r, p.beam, commit, p.t_idx[r] + (tok == p.blk ? 1 : 0));
    __syncthreads();            // (every thread has read t_idx[r])
    if (tid == 0) {
        if (tok == p.blk) p.t_idx[r] += 1;                       // :129
        slot_s = commit ? atomicAdd(p.count + src, 1) : -1;
    }
    __syncthreads();
    if (!commit) return;
    const int slot = slot_s;
    if (tid == 0) { p.rowmap[slot] = r; if (p.joint.rowmap32) p.joint.rowmap32[slot] = r; }
    float *a0 = p.A[0] + (long long)slot * p.lda[0];
    for (int c = tid; c < p.E; c += 256) a0[c] = p.emb[tok * p.E + c];
    for (int c = tid; c < p.H; c += 256) a0[p.E + c] = ss[c];
    for (int l = 1; l < p.layers; ++l) {
        float *al = p.A[l] + (long long)slot * p.lda[l] + p.H;
        const float *hl = ss + (long long)l * 2 * p.H;
        for (int c = tid; c < p.H; c += 256) al[c] = hl[c];
    }
}

// gates (slot order, [i | f | g | o], nn.LSTM order) -> c', h' of layer `layer` in the state row of the slot's beam
// row; h' also becomes the first block of the next layer's input row.  libm-grade tanh / exp: the arg-max decisions
// of the search sit downstream.
__global__ __launch_bounds__(256) void dstep_lstm_cell_kernel(const float *__restrict__ gates, long long ldg,
                                                              float *__restrict__ state, long long SP, int layer,
                                                              const long long *__restrict__ rowmap,
                                                              const int *__restrict__ m_dev, float *__restrict__ next_a,
                                                              long long ld_next, int rows, int H) {
    const int slot = blockIdx.y;
    const int m = m_dev ? min(rows, *m_dev) : rows;
    if (slot >= m) return;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= H) return;
    const long long row = rowmap[slot];
    const float *g = gates + (long long)slot * ldg + c;
    float *hp = state + row * SP + (long long)layer * 2 * H + c, *cp = hp + H;
    const float gi = g[0], gf = g[H], gg = g[2 * (long long)H], go = g[3 * (long long)H];
    const float si = 1.0f / (1.0f + expf(-gi)), sf = 1.0f / (1.0f + expf(-gf)), so = 1.0f / (1.0f + expf(-go));
    const float cn = sf * cp[0] + si * tanhf(gg);
    const float h = so * tanhf(cn);
    cp[0] = cn;
    hp[0] = h;
    if (next_a) next_a[(long long)slot * ld_next + c] = h;
}

// ---- self-attention of the new position over the cached prefix (see decode.hip incr_attn_kernel) -------------
// One workgroup per COMPACT row (slot); thread t owns dims [4t, 4t+4) of the d <= 1024 wide vectors (TPG = d/4
// threads), the G = 256 / TPG thread groups take prefix positions round-robin.  Every prefix position is a separate
// 2 KB key row + 2 KB value row somewhere in HBM, so the kernel is a chain of memory round trips: the row's ancestry
// is copied into LDS first (gathers are not chained behind index loads), keys AND values of 8 positions per group
// are requested together, and the softmax is the online form (running max / sum per head), i.e. ONE pass over the
// prefix; the groups' partial (max, sum, context) triples are merged through LDS at the end.
constexpr int ATT_UNROLL = 8;
__global__ __launch_bounds__(256) void dstep_attn_kernel(const float *__restrict__ kvq, long long ldkvq,
                                                         float *__restrict__ Kc, float *__restrict__ Vc,
                                                         const long long *__restrict__ anc, long long anc_pitch,
                                                         const long long *__restrict__ pos,
                                                         const long long *__restrict__ node,
                                                         const long long *__restrict__ rowmap,
                                                         const int *__restrict__ m_dev, int L, int d, int heads,
                                                         int tpg, float scale, float *__restrict__ out) {
    extern __shared__ float sm_f[];   // [G][d] partial contexts, [G][heads] max, [G][heads] sum, then L ints
    const int slot = blockIdx.x, t = threadIdx.x;
    if (m_dev && slot >= *m_dev) return;
    const long long r = rowmap ? rowmap[slot] : slot;
    const int G = 256 / tpg;
    const int gi = t / tpg, tl = t - gi * tpg;
    const int dh = d / heads, g = dh >> 2;           // g threads per head (power of two <= 64)
    long long p = pos[slot];
    if (p > L - 1) p = L - 1;
    const int np = (int)p;
    const long long my_node = node[slot];
    float *part = sm_f, *pm = sm_f + G * d, *ps = pm + G * heads;
    int *idx = reinterpret_cast<int *>(ps + G * heads);
    const long long *arow = anc + r * anc_pitch;
    for (int j = t; j < np; j += 256) idx[j] = (int)arow[j];
    const float *row = kvq + (long long)slot * ldkvq;
    const int col = tl * 4, hd = col / dh;
    const f32x4 kn = *reinterpret_cast<const f32x4 *>(row + col);
    const f32x4 vn = *reinterpret_cast<const f32x4 *>(row + d + col);
    const f32x4 qv = *reinterpret_cast<const f32x4 *>(row + 2 * d + col) * scale;
    if (gi == 0) {                                     // the new position joins the caches
        *reinterpret_cast<f32x4 *>(Kc + my_node * d + col) = kn;
        *reinterpret_cast<f32x4 *>(Vc + my_node * d + col) = vn;
    }
    __syncthreads();
    float m = -INFINITY, lsum = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = gi; j0 <= np; j0 += ATT_UNROLL * G) {
        f32x4 k4[ATT_UNROLL], v4[ATT_UNROLL];
#pragma unroll
        for (int u = 0; u < ATT_UNROLL; ++u) {
            const int j = j0 + u * G;
            const bool old = j < np;
            k4[u] = old ? *reinterpret_cast<const f32x4 *>(Kc + (long long)idx[j] * d + col) : kn;
            v4[u] = old ? *reinterpret_cast<const f32x4 *>(Vc + (long long)idx[j] * d + col) : vn;
        }
        float sc[ATT_UNROLL], cm = -INFINITY;
#pragma unroll
        for (int u = 0; u < ATT_UNROLL; ++u) {
            float v = qv.x * k4[u].x + qv.y * k4[u].y + qv.z * k4[u].z + qv.w * k4[u].w;
            for (int o = 1; o < g; o <<= 1) v += __shfl_xor(v, o);
            sc[u] = (j0 + u * G <= np) ? v : -INFINITY;
            cm = fmaxf(cm, sc[u]);
        }
        const float mn = fmaxf(m, cm), resc = expf(m - mn);       // m = -inf at the start: exp(-inf) = 0
        acc *= resc;
        lsum *= resc;
#pragma unroll
        for (int u = 0; u < ATT_UNROLL; ++u) {
            const float w = expf(sc[u] - mn);                       // masked positions: exp(-inf) = 0
            lsum += w;
            acc += v4[u] * w;
        }
        m = mn;
    }
    // merge the position groups: context_g * exp(m_g - M) summed, divided by the merged sum
    *reinterpret_cast<f32x4 *>(part + gi * d + col) = acc;
    if ((tl & (g - 1)) == 0) { pm[gi * heads + hd] = m; ps[gi * heads + hd] = lsum; }
    __syncthreads();
    if (gi == 0) {
        float M = -INFINITY;
        for (int k = 0; k < G; ++k) M = fmaxf(M, pm[k * heads + hd]);
        f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
        float tot = 0.f;
        for (int k = 0; k < G; ++k) {
            const float mk = pm[k * heads + hd];
            const float w = mk > -INFINITY ? expf(mk - M) : 0.f;     // a group without positions
            o4 += *reinterpret_cast<const f32x4 *>(part + k * d + col) * w;
            tot += ps[k * heads + hd] * w;
        }
        *reinterpret_cast<f32x4 *>(out + (long long)slot * d + col) = o4 * (1.f / tot);
    }
}

// ---- fc2 + log-sum-exp partials + top-K partials --------------------------------------------------------------
constexpr int FC2_WN = 3, FC2_KS = 2, FC2_COLS = 4 * FC2_WN * 16;   // 192 columns per split
static_assert(FC2_COLS == PIKA_DFC2_COLS, "pika_decode_step.h");
// floats per slab row: 4 more than the columns, so that 16 rows read at the same column sit in 16 different LDS banks
// (the row statistics of the logits mode: four lanes per row, all rows of a wave at once)
constexpr int FC2_PITCH = FC2_COLS + 4;
struct Cand { float v; int idx; };

__device__ inline bool better(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

// order-preserving map float -> unsigned (a > b <=> fkey(a) > fkey(b); -inf is the smallest non-NaN key)
__device__ inline unsigned fkey(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int NS, int BM>
__host__ __device__ constexpr size_t FC2_LDS_MAIN() {      // operand staging buffers, overlaid by the logits slab
    constexpr size_t a = Core<BM, FC2_WN, NS, FC2_KS>::LDS_BYTES, b = (size_t)BM * FC2_PITCH * 4;
    return a > b ? a : b;
}

// FC2_BM rows per workgroup: 32, or 64 when the launch has more than 512 rows -- at B * beam = 1024 rows that is 16 x 27 =
// 432 workgroups, all resident at once (two per CU), instead of 864 in two rounds, and every slab of W is read by half as
// many workgroups.  The logits slab overlays the operand staging buffers (dead after the product loop's last barrier).
// CANDS = false: the row statistics only, and the scaled logits themselves go to `logits` (row pitch ldl >= splits *
// FC2_COLS, columns >= V hold -inf): the advance then picks a row's K best with ONE thresholded pass over the row (the
// K-th largest of the row's split maxima bounds its K-th largest logit from below: a few dozen survivors of 5000) instead
// of this kernel bisecting every (row, split) pair for its own K best -- 27 x 16 candidates per row for 16 winners, and
// more than half of this kernel's time.
template <int NS, int FC2_BM>
__global__ __launch_bounds__(256, (FC2_BM == 32 && NS != 3) ? 3 : 2) void dfc2_logits_kernel(const float *__restrict__ h, long long ldh,
                                                        const __bf16 *__restrict__ W, const float *__restrict__ bias,
                                                        int rows, int V, int NT, int KT, float sm_scale,
                                                        int splits, float *__restrict__ pmax,
                                                        float *__restrict__ psum, int Kvalid,
                                                        float *__restrict__ logits, long long ldl, int by_cols) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef Core<FC2_BM, FC2_WN, NS, FC2_KS> core_t;
    core_t core;
    __bf16 *lds = reinterpret_cast<__bf16 *>(smem);
    float *slab = reinterpret_cast<float *>(smem);                       // [FC2_BM][FC2_PITCH], after the product
    int mb, sp;
    if (by_cols ? !xcd_tile_cols(splits, (rows + FC2_BM - 1) / FC2_BM, mb, sp)
                : !xcd_tile_rows(splits, (rows + FC2_BM - 1) / FC2_BM, mb, sp)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = mb * FC2_BM, nt0 = (sp * 4 + wave) * FC2_WN;
    if (m0 >= rows) return;
    // this split's bias: one value per thread, requested now, parked in ONE register across the product loop and handed
    // round through LDS afterwards (12 dependent global loads per lane after the loop were 2.5 us of every workgroup)
    float *bias_s = reinterpret_cast<float *>(smem + FC2_LDS_MAIN<NS, FC2_BM>());
    const int bc = sp * FC2_COLS + threadIdx.x;
    const float bv = (bias && threadIdx.x < FC2_COLS && bc < V) ? bias[bc] : 0.f;
    core.run(h, ldh, rows, m0, W, NT, KT, nt0, lds, Kvalid);
    CORE_STAMP(38, 0);
    if (threadIdx.x < FC2_COLS) bias_s[threadIdx.x] = bv;
    __syncthreads();
    // logits of this split -> slab (sm_scale * (acc + bias); columns >= V masked)
#pragma unroll
    for (int i = 0; i < FC2_BM / 16; ++i)
#pragma unroll
        for (int j = 0; j < FC2_WN; ++j) {
            const int lc = (wave * FC2_WN + j) * 16 + (lane >> 4) * 4;      // column inside the split
            const int c = sp * FC2_COLS + lc;
            f32x4 v = core.acc[i][j];
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias_s + lc);
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = c + u < V ? sm_scale * (v[u] + b4[u]) : -INFINITY;
            *reinterpret_cast<f32x4 *>(slab + (i * 16 + (lane & 15)) * FC2_PITCH + lc) = v;
        }
    __syncthreads();
    CORE_STAMP(38, 1);
    {
        // Row statistics with FOUR lanes per row -- a lane walks every fourth column of its row (no cross-lane step until
        // the 4-lane merge; the row pitch keeps the 16 rows of a wave in different banks) -- and the slab goes to `logits`
        // in whole 16-byte pieces.  (One row per wave pass: 12 butterfly steps per row, 16 rows per wave.)
        const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
        if (row < FC2_BM && m0 + row < rows) {
            const float *sr = slab + row * FC2_PITCH + part;
            float xv[FC2_COLS / 4];
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < FC2_COLS / 4; ++i) { xv[i] = sr[4 * i]; m = fmaxf(m, xv[i]); }
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 2));
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < FC2_COLS / 4; ++i) sum += xv[i] > -INFINITY ? expf(xv[i] - m) : 0.f;
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            if (part == 0) { pmax[(long long)(m0 + row) * splits + sp] = m; psum[(long long)(m0 + row) * splits + sp] = sum; }
        }
        for (int idx = threadIdx.x; idx < FC2_BM * (FC2_COLS / 4); idx += 256) {
            const int r = idx / (FC2_COLS / 4), c4 = idx - r * (FC2_COLS / 4);
            if (m0 + r < rows)
                *reinterpret_cast<f32x4 *>(logits + (long long)(m0 + r) * ldl + sp * FC2_COLS + 4 * c4) =
                    *reinterpret_cast<const f32x4 *>(slab + r * FC2_PITCH + 4 * c4);
        }
    }
}

int check(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

template <int BM, int NS>
void launch_dgemm(unsigned grid, hipStream_t st, const DG &p) {
    constexpr size_t lds = DgCfg<BM, NS>::core_t::LDS_BYTES;
    dgemm_kernel<BM, NS><<<dim3(grid), dim3(256), lds, st>>>(p);
}

template <int NS, int BM>
void launch_fc2_bm(unsigned grid, hipStream_t st, const float *h, long long ldh, const __bf16 *w, const float *bias, int rows,
                   int V, int NT, int KT, float sm_scale, int splits, float *pmax, float *psum, int K, float *logits,
                   long long ldl, int by_cols) {
    constexpr size_t lds = FC2_LDS_MAIN<NS, BM>() + FC2_COLS * 4;
    dfc2_logits_kernel<NS, BM><<<dim3(grid), dim3(256), lds, st>>>(h, ldh, w, bias, rows, V, NT, KT, sm_scale, splits, pmax, psum,
                                                                   (K & 3) ? KT * 32 : K, logits, ldl, by_cols);
}

int fc2_bm(int rows) {
    // 64-row tiles once the 32-row form would need a second round of workgroups (rows / 32 * 27 splits > 768 resident ones at
    // V = 5000): 74 vs 81 us at 1024 rows, but 50 vs 34 us at 32 rows (tools/dfc2_bench.py)
    static const int forced = [] { const char *e = pika_knob("PIKA_DFC2_BM"); return e ? atoi(e) : 0; }();
    if (forced == 32 || forced == 64) return forced;
    return rows > 768 ? 64 : 32;
}

template <int NS>
void launch_fc2(hipStream_t st, const float *h, long long ldh, const __bf16 *w, const float *bias, int rows,
                int V, int NT, int KT, float sm_scale, int splits, float *pmax, float *psum, int K, float *logits, long long ldl) {
    const int bm = fc2_bm(rows);
    const int m_tiles = (rows + bm - 1) / bm;
    // up to 512 rows: W slabs pinned to XCDs (xcd_tile_cols: 28.6 vs 33.0 us at 32 rows, 29.0 vs 34.3 at 128, 39.7 vs 42.3 at
    // 512; 48.7 vs 46.8 at 1024 -- tools/dfc2_bench.py); beyond: row bands per XCD.  PIKA_DFC2_MAP=rows / cols in a tuning build
    static const int forced_map = [] { const char *e = pika_knob("PIKA_DFC2_MAP"); return !e ? -1 : (e[0] == 'r' ? 0 : 1); }();
    const int by_cols = forced_map >= 0 ? forced_map : (rows <= 512 ? 1 : 0);
    const unsigned grid = by_cols ? (unsigned)(8 * ((splits + 7) / 8) * m_tiles) : (unsigned)(((m_tiles + 7) / 8) * 8 * splits);
    if (bm == 64) launch_fc2_bm<NS, 64>(grid, st, h, ldh, w, bias, rows, V, NT, KT, sm_scale, splits, pmax, psum, K, logits, ldl, by_cols);
    else launch_fc2_bm<NS, 32>(grid, st, h, ldh, w, bias, rows, V, NT, KT, sm_scale, splits, pmax, psum, K, logits, ldl, by_cols);
}

}  // namespace

extern "C" {

size_t pika_dpack_bytes(int N, int K, int terms) {
    return (size_t)planes_of(terms) * ((N + 15) / 16 * 16) * ((K + 31) / 32 * 32) * 2;
}

int pika_dpack_weight(const float *W, long long ldw, int N, int K, int terms, int interleave2, void *packed,
                      void *stream) {
    if (!W || !packed || N <= 0 || K <= 0 || terms < 1 || terms > 4 || (interleave2 && (N & 1))) return PIKA_EINVAL;
    const int NT = (N + 15) / 16, KT = (K + 31) / 32;
    const long long total = (long long)NT * KT * 64;
    hipLaunchKernelGGL(dpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K,
                       NT, KT, terms, interleave2, reinterpret_cast<__bf16 *>(packed));
    return check(hipGetLastError());
}

int pika_dgemm(const pika_dgemm_t *q, void *stream) {
    if (!q || !q->A || !q->W || !q->C || q->M <= 0 || q->N <= 0 || q->K <= 0 || q->terms < 1 || q->terms > 4)
        return PIKA_EINVAL;
    if ((q->lda & 3) || (reinterpret_cast<uintptr_t>(q->A) & 15)) return PIKA_EINVAL;
    if ((q->flags & PIKA_DG_GATE) && (!q->e_all || !q->t_idx || q->T <= 0 || q->beam <= 0 || (q->N & 3))) return PIKA_EINVAL;
    if (((q->flags & PIKA_DG_ROWMASK) || (q->C2 && !(q->flags & PIKA_DG_GATE))) && !q->node) return PIKA_EINVAL;
    if ((q->flags & PIKA_DG_GATE) && q->C2 && ((q->ldc2 & 3) || (reinterpret_cast<uintptr_t>(q->C2) & 15))) return PIKA_EINVAL;
    const bool ln = q->ln_gamma != nullptr;
    if (ln && (!q->ln_beta || (q->K & 31) || q->K > 1024 || ((reinterpret_cast<uintptr_t>(q->ln_gamma) |
                                                               reinterpret_cast<uintptr_t>(q->ln_beta)) & 15)))
        return PIKA_EINVAL;
    DG p{q->A, q->lda, reinterpret_cast<const __bf16 *>(q->W), q->bias, q->res, q->ldr, q->C, q->ldc, q->C2, q->ldc2,
         q->node, q->skip_node, q->e_all, q->t_idx, q->T, q->beam, q->M, q->N, (q->N + 15) / 16, (q->K + 31) / 32, q->flags,
         q->m_dev, q->crow, q->rowlist, q->rowoff_dev, (q->K & 3) ? ((q->K + 31) / 32) * 32 : q->K,
         q->ln_gamma, q->ln_beta, q->ln_eps};
    hipStream_t st = (hipStream_t)stream;
    // Few rows (the caller says so for a compact row list whose count lives on the device: PIKA_DG_FEW_ROWS; or M itself
    // is small): the split-reduction kernel, one 16/32-column tile of 32 rows per workgroup.  PIKA_DGEMM_SK=0 / 1 forces the
    // choice (A/B runs).
    static const int sk_env = [] { const char *e = pika_knob("PIKA_DGEMM_SK"); return e ? atoi(e) : -1; }();
    const bool sk = ln || (sk_env >= 0 ? sk_env != 0 : ((q->flags & PIKA_DG_FEW_ROWS) || q->M <= 256));
    // 4 waves per workgroup take K <= 512 in ONE request round, 8 waves K <= 1024; beyond (K up to 4096) 8 waves in
    // rounds of 4 k-tiles.  Wide products (N >= 1024) take 32-column tiles: all tiles of a ~170-row launch resident at once.
    const bool sk_fits = p.KT <= 128 && !(ln && p.KT > 32);
    if (ln && !sk_fits) return PIKA_EINVAL;
    if (sk && sk_fits) {
        const int kw = p.KT <= 16 ? 4 : 8, wn = q->N >= 1024 ? 2 : 1;
        const long long tiles = (long long)((q->M + 31) / 32) * ((p.NT + wn - 1) / wn);
        // workgroups resident at once per XCD (32 CUs): 2 per CU of 256 threads, 1 of 512
        const long long cap8 = kw == 4 ? 64 : 32;
        const unsigned grid = (unsigned)(8 * (tiles / 8 + 1 < cap8 ? tiles / 8 + 1 : cap8));
#define PIKA_SK(NS, KW, WN, LN) do { if (p.KT <= 4 * KW) dgemm_sk_kernel<NS, KW, 2, WN, 4, LN, false><<<dim3(grid), dim3(64 * KW), 0, st>>>(p); \
                                     else dgemm_sk_kernel<NS, KW, 2, WN, 2, false, true><<<dim3(grid), dim3(64 * KW), 0, st>>>(p); } while (0)
#define PIKA_SK_T(KW, WN, LN) do { if (q->terms == 1) PIKA_SK(1, KW, WN, LN); else if (q->terms == 2) PIKA_SK(2, KW, WN, LN); \
                                   else if (q->terms == 3) PIKA_SK(3, KW, WN, LN); else PIKA_SK(4, KW, WN, LN); } while (0)
#define PIKA_SK_W(KW, LN) do { if (wn == 2) PIKA_SK_T(KW, 2, LN); else PIKA_SK_T(KW, 1, LN); } while (0)
        if (ln) { if (kw == 4) PIKA_SK_W(4, true); else PIKA_SK_W(8, true); }
        else { if (kw == 4) PIKA_SK_W(4, false); else PIKA_SK_W(8, false); }
#undef PIKA_SK_W
#undef PIKA_SK_T
#undef PIKA_SK
        return check(hipGetLastError());
    }
    // wide products (N >= 2048) beyond the few-rows kernel: 64 x 128 tiles (PIKA_DGEMM_WIDE=0: the 64 x 64 tiles, A/B runs)
    static const bool wide_on = [] { const char *e = pika_knob("PIKA_DGEMM_WIDE"); return !e || atoi(e) != 0; }();
    if (q->N >= 2048 && wide_on && !(q->flags & PIKA_DG_GATE)) {
        const unsigned grid = (unsigned)(8 * ((((p.NT + 7) / 8) + 7) / 8) * ((q->M + 63) / 64));
#define PIKA_WIDE(NS) dgemm_wide_kernel<NS><<<dim3(grid), dim3(256), Core<64, 2, NS, 2>::LDS_BYTES, st>>>(p)
        if (q->terms == 1) PIKA_WIDE(1); else if (q->terms == 2) PIKA_WIDE(2); else if (q->terms == 3) PIKA_WIDE(3); else PIKA_WIDE(4);
#undef PIKA_WIDE
        return check(hipGetLastError());
    }
    const int n_groups = (p.NT + 3) / 4;
    // enough workgroups to cover the chip: 32-row tiles unless 64-row tiles already give > 256 of them
    const bool big = (long long)((q->M + 63) / 64) * n_groups >= 512;
    const int BM = big ? 64 : 32;
    const unsigned grid = (unsigned)((((q->M + BM - 1) / BM + 7) / 8) * 8 * n_groups);
    if (big) {
        if (q->terms == 1) launch_dgemm<64, 1>(grid, st, p); else if (q->terms == 2) launch_dgemm<64, 2>(grid, st, p);
        else if (q->terms == 3) launch_dgemm<64, 3>(grid, st, p); else launch_dgemm<64, 4>(grid, st, p);
    } else {
        if (q->terms == 1) launch_dgemm<32, 1>(grid, st, p); else if (q->terms == 2) launch_dgemm<32, 2>(grid, st, p);
        else if (q->terms == 3) launch_dgemm<32, 3>(grid, st, p); else launch_dgemm<32, 4>(grid, st, p);
    }
    return check(hipGetLastError());
}

int pika_dstep_prep(const pika_dstep_prep_t *q, void *stream) {
    if (!q || q->rows <= 0 || q->layers < 1 || q->layers > PIKA_DSTEP_MAX_LAYERS || q->beam <= 0 || q->H <= 0 || q->L <= 4)
        return PIKA_EINVAL;
    PrepDev a{*q};
    hipLaunchKernelGGL(dstep_prep_kernel, dim3(q->rows), dim3(256), 0, (hipStream_t)stream, a);
    return check(hipGetLastError());
}


int pika_dstep_prep_lstm(const pika_dstep_prep_lstm_t *q, void *stream) {
    if (!q || q->rows <= 0 || q->layers < 1 || q->layers > PIKA_DSTEP_MAX_LAYERS || q->beam <= 0 || q->H <= 0 || q->E <= 0 ||
        (q->H & 3))
        return PIKA_EINVAL;
    if (!q->prev_k || !q->y || !q->step_t || !q->t_idx || !q->state[0] || !q->state[1] || !q->emb || !q->rowmap || !q->count)
        return PIKA_EINVAL;
    for (int l = 0; l < q->layers; ++l)
        if (!q->A[l] || q->lda[l] < (l == 0 ? q->E + q->H : 2 * q->H)) return PIKA_EINVAL;
    hipLaunchKernelGGL(dstep_prep_lstm_kernel, dim3(q->rows), dim3(256), 0, (hipStream_t)stream, *q);
    return check(hipGetLastError());
}

int pika_dstep_lstm_cell(const float *gates, long long ldg, float *state, long long state_pitch, int layer,
                         const long long *rowmap, const int *m_dev, float *next_a, long long ld_next, int rows, int H,
                         void *stream) {
    if (!gates || !state || !rowmap || rows <= 0 || H <= 0 || layer < 0 || ldg < 4LL * H ||
        state_pitch < (long long)(layer + 1) * 2 * H)
        return PIKA_EINVAL;
    hipLaunchKernelGGL(dstep_lstm_cell_kernel, dim3((H + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream, gates, ldg,
                       state, state_pitch, layer, rowmap, m_dev, next_a, ld_next, rows, H);
    return check(hipGetLastError());
}

int pika_dstep_attention(const float *kvq, long long ldkvq, float *k_cache, float *v_cache,
                         const long long *ancestry, long long ancestry_pitch, const long long *pos,
                         const long long *node, const long long *rowmap, const int *m_dev, int rows, int L, int d,
                         int heads, float *out, void *stream) {
    if (!kvq || !k_cache || !v_cache || !ancestry || !pos || !node || !out || rows <= 0 || L <= 0 || heads <= 0)
        return PIKA_EINVAL;
    if (d % heads || d > 1024 || (d & 3) || (ldkvq & 3)) return PIKA_EINVAL;
    const int dh = d / heads, g = dh >> 2, tpg = d >> 2;
    if ((dh & 3) || g < 1 || g > 64 || (g & (g - 1)) || 256 % tpg) return PIKA_EINVAL;
    const int G = 256 / tpg;
    const size_t lds = sizeof(float) * ((size_t)G * d + 2 * (size_t)G * heads + (size_t)L);
    if (lds > 64 * 1024) return PIKA_ETOOBIG;
    hipLaunchKernelGGL(dstep_attn_kernel, dim3(rows), dim3(256), lds, (hipStream_t)stream, kvq, ldkvq, k_cache, v_cache,
                       ancestry, ancestry_pitch, pos, node, rowmap, m_dev, L, d, heads, tpg, 1.f / sqrtf((float)dh), out);
    return check(hipGetLastError());
}

int pika_dfc2_splits(int V) { return (V + FC2_COLS - 1) / FC2_COLS; }
int pika_dfc2_cols_per_split(void) { return FC2_COLS; }

int pika_dfc2_logits(const float *h, long long ldh, const void *W, const float *bias, int rows, int V, int K, int terms,
                     float sm_scale, float *pmax, float *psum, float *logits, long long ldl, void *stream) {
    if (!h || !W || !pmax || !psum || !logits || rows <= 0 || V <= 0 || K <= 0 || terms < 1 || terms > 4 || (ldh & 3) ||
        (reinterpret_cast<uintptr_t>(h) & 15) || ldl < (long long)pika_dfc2_splits(V) * FC2_COLS || (ldl & 3) ||
        (reinterpret_cast<uintptr_t>(logits) & 15))
        return PIKA_EINVAL;
    const int NT = (V + 15) / 16, KT = (K + 31) / 32, splits = pika_dfc2_splits(V);
    hipStream_t st = (hipStream_t)stream;
    const __bf16 *w = reinterpret_cast<const __bf16 *>(W);
    if (terms == 1) launch_fc2<1>(st, h, ldh, w, bias, rows, V, NT, KT, sm_scale, splits, pmax, psum, K, logits, ldl);
    else if (terms == 2) launch_fc2<2>(st, h, ldh, w, bias, rows, V, NT, KT, sm_scale, splits, pmax, psum, K, logits, ldl);
    else if (terms == 3) launch_fc2<3>(st, h, ldh, w, bias, rows, V, NT, KT, sm_scale, splits, pmax, psum, K, logits, ldl);
    else launch_fc2<4>(st, h, ldh, w, bias, rows, V, NT, KT, sm_scale, splits, pmax, psum, K, logits, ldl);
    return check(hipGetLastError());
}

}  // extern "C"
