// pika_amd/csrc/gemm.hip -- bf16-MFMA GEMM for gfx950 (see include/pika_gemm.h).
//
// 128x128x32 tile, 256 threads = 4 waves in 2x2, each wave 64x64 = 4x4 MFMA 16x16x32 tiles
// (64 fp32 accumulators per lane).  fp32 (or bf16) operands are loaded with 16-byte (8-byte)
// global loads into registers, rounded to bf16 (v_cvt_pk_bf16_f32) and staged into LDS rows of
// 32 bf16 padded to 80 bytes: ds_read_b128 of the MFMA fragments (row = lane&15, k-group =
// lane>>4) is then bank-conflict-free (16 rows x 20 dwords cover all 64 banks exactly once).
// Global loads of step k+1 are issued before the MFMAs of step k and written to the other LDS
// buffer after them: one barrier per K-step.  MFMA is issued as D^T = B_frag x A_frag so each
// lane ends up with 4 CONSECUTIVE columns of one row of C -> 16-byte epilogue stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "pika_gemm.h"
#include "pika_rnnt.h"  // PIKA_EINVAL
#include "pika_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WM_, int WN_, int BK_>
struct Cfg {
    static constexpr int WM = WM_, WN = WN_, BK = BK_;
    static constexpr int BM = WM * 64, BN = WN * 64, THREADS = WM * WN * 64;
    static constexpr int PITCH = BK + 8;  // bf16 per LDS row; (PITCH/2) dwords = 20 or 36: the 16
                                          // rows of a fragment read land on 16 distinct 4-bank groups
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct Op {  // device-side copy of pika_operand_t
    const char *ptr;
    int rows_per_batch, t_in;
    long long batch_stride, ld;
    int C, stride, dil, pad;
};

// Transposed operand: the matrix is stored with the OUTPUT index contiguous, X[k][r] (k = reduction
// = the time-like index (b,t), r = the channel-like index (tap,c)) -- dY and X in dW = dY^T X, W in
// dX = dY W.  The tile goes into LDS as it lies in memory, [BK][COLS(+8)], with coalesced 16-byte
// loads along r, and the MFMA fragments (8 consecutive k per lane) come out of the gfx950
// transpose read ds_read_b64_tr_b16: within a 16-lane group, lanes 4j..4j+3 supply row j (16
// consecutive r) and lane i receives column i of the 4 rows (measured: tools/tr_probe.hip).
template <typename T, int COLS, typename CF>
struct LoaderT {
    static constexpr int EPL = 16 / sizeof(T);
    static constexpr int TPR = COLS / EPL;              // threads per k-row
    static constexpr int RPP = CF::THREADS / TPR;       // k-rows per pass
    static constexpr int NP = CF::BK / RPP;
    static constexpr int PITCHT = COLS + 8;
    static_assert(CF::THREADS % TPR == 0 && CF::BK % RPP == 0 && NP >= 1, "tile shape");
    const T *base;
    long long ld, bstride;
    int t_in, rpb, stride_;
    int tapoff;            // tap*dil - pad of this thread's output quad (fixed)
    int c;                 // channel of this thread's output quad (fixed)
    bool out_ok;
    int kb[NP], kt[NP];    // (b,t) of this thread's reduction rows
    int k[NP], K;

    __device__ inline void init(const Op &o, long long zoff, int o0, int nout, int K_, int k0) {
        const int tid = threadIdx.x;
        base = reinterpret_cast<const T *>(o.ptr) + zoff;
        ld = o.ld; bstride = o.batch_stride; t_in = o.t_in; rpb = o.rows_per_batch; stride_ = o.stride; K = K_;
        const int oq = o0 + (tid % TPR) * EPL;
        out_ok = oq < nout;
        const int tap = oq / o.C;
        c = oq - tap * o.C;
        tapoff = tap * o.dil - o.pad;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            k[i] = k0 + tid / TPR + RPP * i;
            kb[i] = k[i] / rpb;
            kt[i] = k[i] - kb[i] * rpb;
        }
    }
    __device__ inline void advance() {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            k[i] += CF::BK;
            kt[i] += CF::BK;
            while (kt[i] >= rpb) { kt[i] -= rpb; ++kb[i]; }
        }
    }
    __device__ inline void load(f32x4 raw[NP]) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int ti = kt[i] * stride_ + tapoff;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (out_ok && k[i] < K && ti >= 0 && ti < t_in)
                x = *reinterpret_cast<const f32x4 *>(base + (long long)kb[i] * bstride + (long long)ti * ld + c);
            raw[i] = x;
        }
    }
    template <int NS>
    __device__ inline void stage(__bf16 *dst, int part, const f32x4 raw[NP]) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int off = (tid / TPR + RPP * i) * PITCHT + (tid % TPR) * EPL;
            if constexpr (sizeof(T) == 4) {
                f32x4 r = raw[i];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const bf16x4 h = __builtin_convertvector(r, bf16x4);
                    *reinterpret_cast<bf16x4 *>(dst + s * part + off) = h;
                    if (s + 1 < NS) r = r - __builtin_convertvector(h, f32x4);
                }
            } else {
                *reinterpret_cast<f32x4 *>(dst + off) = raw[i];
                if constexpr (NS > 1) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 1; s < NS; ++s) *reinterpret_cast<f32x4 *>(dst + s * part + off) = z;
                }
            }
        }
    }
};

// MFMA fragment (8 consecutive k of output row/col `r0 + (lane&15)`) from either tile layout.
template <bool TR, int PITCH, int PITCHT>
__device__ inline bf16x8 fragment(const __bf16 *tile, int r0, int kk, int lane) {
    if constexpr (!TR) {
        return *reinterpret_cast<const bf16x8 *>(tile + (r0 + (lane & 15)) * PITCH + kk + (lane >> 4) * 8);
    } else {
        const int i = lane & 15, j = i >> 2, q = i & 3, k0 = kk + (lane >> 4) * 8;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const __bf16 *p0 = tile + (k0 + j) * PITCHT + r0 + 4 * q;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p0 + 4 * PITCHT));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

// Per-thread loader of one operand tile (ROWS x BK): 16-byte loads (4 f32 or 8 bf16), TPR threads
// per row, NP passes of RPP rows.
template <typename T, int ROWS, typename CF>
struct Loader {
    static constexpr int EPL = 16 / sizeof(T);
    static constexpr int TPR = CF::BK / EPL;
    static constexpr int RPP = CF::THREADS / TPR;
    static constexpr int NP = ROWS / RPP;
    static_assert(ROWS % RPP == 0, "tile rows must be a multiple of rows per pass");
    const T *base;
    long long rowoff[NP];  // b * batch_stride
    int tbase[NP];         // t*stride - pad, or a huge negative for out-of-range rows
    long long ld;
    int t_in, C, dil;
    int tap, c;            // decomposition of this thread's current k
    int k, K;

    __device__ inline void init(const Op &o, long long zoff, int r0, int nrows, int K_, int k0) {
        const int tid = threadIdx.x;
        base = reinterpret_cast<const T *>(o.ptr) + zoff;
        ld = o.ld; t_in = o.t_in; C = o.C; dil = o.dil; K = K_;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = r0 + tid / TPR + RPP * i;
            if (r < nrows) {
                const int b = r / o.rows_per_batch, t = r - b * o.rows_per_batch;
                rowoff[i] = (long long)b * o.batch_stride;
                tbase[i] = t * o.stride - o.pad;
            } else {
                rowoff[i] = 0;
                tbase[i] = -(1 << 29);
            }
        }
        k = k0 + (tid % TPR) * EPL;
        tap = k / C;
        c = k - tap * C;
    }
    __device__ inline void advance() {
        k += CF::BK;
        c += CF::BK;
        while (c >= C) { c -= C; ++tap; }
    }
    // A 16-byte load never straddles a tap for f32 (C % 4 == 0); for bf16 it needs C % 8 == 0,
    // which the host checks (bf16 operands are our own transposed copies / plain matrices).
    __device__ inline void load(f32x4 raw[NP]) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int ti = tbase[i] + tap * dil;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (k < K && ti >= 0 && ti < t_in)
                x = *reinterpret_cast<const f32x4 *>(base + rowoff[i] + (long long)ti * ld + c);
            raw[i] = x;
        }
    }
    // Stage into LDS: NS = 1 rounds to bf16; NS = 3 writes the exact 3-way bf16 split
    // x = h + m + l (8+8+8 mantissa bits) into three consecutive tiles (stride `part`).
    template <int NS>
    __device__ inline void stage(__bf16 *dst, int part, const f32x4 raw[NP]) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int off = (tid / TPR + RPP * i) * CF::PITCH + (tid % TPR) * EPL;
            if constexpr (sizeof(T) == 4) {
                f32x4 r = raw[i];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const bf16x4 h = __builtin_convertvector(r, bf16x4);
                    *reinterpret_cast<bf16x4 *>(dst + s * part + off) = h;
                    if (s + 1 < NS) r = r - __builtin_convertvector(h, f32x4);
                }
            } else {
                *reinterpret_cast<f32x4 *>(dst + off) = raw[i];  // already bf16: parts 1,2 are zero
                if constexpr (NS > 1) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 1; s < NS; ++s) *reinterpret_cast<f32x4 *>(dst + s * part + off) = z;
                }
            }
        }
    }
};

#ifdef PIKA_NT_TRACE      // profiling builds only (tools/nt_trace.py): time stamps (100 MHz) of workgroup 0's first lane, last launch
__device__ unsigned long long g_nt_trace[64];
#define NT_STAMP(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && (k) < 64) \
    g_nt_trace[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define NT_STAMP(k) do { } while (0)
#endif

template <typename TA, typename TB, int NS, typename CF, bool TRA, bool TRB>
__global__ __launch_bounds__(CF::THREADS) void gemm_nt_kernel(
    Op A, Op B, long long a_zo, long long a_zi, long long b_zo, long long b_zi,
    float *__restrict__ Cp, long long ldc, long long c_zo, long long c_zi, int M, int N, int K,
    int z_div, int splitk, const float *__restrict__ bias, int flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = CF::BM, BN = CF::BN, BK = CF::BK, PITCH = CF::PITCH;
    using LA = typename std::conditional<TRA, LoaderT<TA, BM, CF>, Loader<TA, BM, CF>>::type;
    using LB = typename std::conditional<TRB, LoaderT<TB, BN, CF>, Loader<TB, BN, CF>>::type;
    constexpr int PTA = BM + 8, PTB = BN + 8;
    constexpr int TA_ELEMS = TRA ? BK * PTA : BM * PITCH, TB_ELEMS = TRB ? BK * PTB : BN * PITCH;
    // layout: [buf][A parts 0..NS-1 | B parts 0..NS-1]
    constexpr int BOFF = NS * TA_ELEMS;
    constexpr int PER_BUF = NS * (TA_ELEMS + TB_ELEMS);
    __bf16 *lds = reinterpret_cast<__bf16 *>(smem);

    // split-K (only with batch == 1): blockIdx.z indexes a K range and the epilogue accumulates
    // with atomics into a zeroed C
    const int split = splitk > 1 ? blockIdx.z : 0;
    const int z = splitk > 1 ? 0 : blockIdx.z, zo = z / z_div, zi = z - zo * z_div;
    // XCD-aware tile order: workgroup b runs on XCD b % 8; give every XCD a contiguous run of
    // tiles (n fastest) so the tiles that share an A row-panel meet in one L2.
    const int nx = gridDim.x, ntiles = nx * gridDim.y;
    int tile = blockIdx.y * nx + blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (tile / nx) * BM, n0 = (tile % nx) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / CF::WN, wn = wave % CF::WN;

    NT_STAMP(0);
    LA la;
    LB lb;
    const int nk_all = (K + BK - 1) / BK;
    const int nk_per = (nk_all + splitk - 1) / splitk;
    const int kb0 = split * nk_per;
    const int nk = min(nk_all, kb0 + nk_per) - kb0;
    if (nk <= 0) return;
    // a bf16 operand is read in 8-element groups: its rows are zero-padded to a multiple of 8
    // by contract, so its bound is K rounded up (the other operand supplies the zeros)
    la.init(A, zo * a_zo + zi * a_zi, m0, M, (sizeof(TA) == 2 && !TRA) ? (K + 7) & ~7 : K, kb0 * BK);
    lb.init(B, zo * b_zo + zi * b_zi, n0, N, (sizeof(TB) == 2 && !TRB) ? (K + 7) & ~7 : K, kb0 * BK);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    NT_STAMP(1);
    f32x4 ra[LA::NP], rb[LB::NP];
    la.load(ra);
    lb.load(rb);
    la.template stage<NS>(lds, TA_ELEMS, ra);
    lb.template stage<NS>(lds + BOFF, TB_ELEMS, rb);
    NT_STAMP(2);
    __syncthreads();
    NT_STAMP(3);

    for (int kb = 0; kb < nk; ++kb) {
        const __bf16 *cur = lds + (kb & 1) * PER_BUF;
        __bf16 *nxt = lds + ((kb + 1) & 1) * PER_BUF;
        const bool more = kb + 1 < nk;
        if (more) {
            la.advance();
            lb.advance();
            la.load(ra);
            lb.load(rb);
        }
        NT_STAMP(4 + 4 * kb);
        // split products kept: (0,0) [NS=1]; + (0,1),(1,0),(1,1),(0,2),(2,0) [NS=3]: every term
        // above 2^-24 of the leading one
        constexpr int NPAIR = NS == 1 ? 1 : 6;
        constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
        for (int p = NPAIR - 1; p >= 0; --p) {  // small terms first
#pragma unroll
            for (int kk = 0; kk < BK; kk += 32) {
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[i] = fragment<TRA, PITCH, PTA>(cur + PA[p] * TA_ELEMS, wm * 64 + i * 16, kk, lane);
                    fb[i] = fragment<TRB, PITCH, PTB>(cur + BOFF + PB[p] * TB_ELEMS, wn * 64 + i * 16, kk, lane);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        }
        NT_STAMP(5 + 4 * kb);
        if (more) {
            la.template stage<NS>(nxt, TA_ELEMS, ra);
            lb.template stage<NS>(nxt + BOFF, TB_ELEMS, rb);
        }
        NT_STAMP(6 + 4 * kb);
        __syncthreads();
        NT_STAMP(7 + 4 * kb);
    }
    NT_STAMP(62);

    // epilogue: lane holds C[m][n..n+3], m = tile row (lane&15), n = (lane>>4)*4
    float *Cz = Cp + zo * c_zo + zi * c_zi;
    const bool relu = flags & PIKA_GEMM_RELU, accum = flags & PIKA_GEMM_ACCUMULATE;
    const bool vec_ok = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cz) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= N) continue;
            f32x4 v = acc[i][j];
            float *dst = Cz + (long long)m * ldc + n;
            if (splitk > 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) atomicAdd(dst + e, v[e] + ((bias && split == 0) ? bias[n + e] : 0.f));
                continue;
            }
            if (n + 3 < N && vec_ok) {
                if (bias) v += *reinterpret_cast<const f32x4 *>(bias + n);  // bias + n: n%4==0
                if (accum) v += *reinterpret_cast<const f32x4 *>(dst);
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<f32x4 *>(dst) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e < N) {
                        float s = v[e] + (bias ? bias[n + e] : 0.f);
                        if (accum) s += dst[e];
                        if (relu) s = fmaxf(s, 0.f);
                        dst[e] = s;
                    }
                }
            }
        }
    }
}

Op to_op(const pika_operand_t &o) {
    Op r;
    r.ptr = static_cast<const char *>(o.ptr);
    r.rows_per_batch = o.rows_per_batch; r.t_in = o.t_in;
    r.batch_stride = o.batch_stride; r.ld = o.ld;
    r.C = o.C; r.stride = o.stride; r.dil = o.dil; r.pad = o.pad;
    return r;
}

bool operand_ok(const pika_operand_t &o, int K, int nout) {
    if (!o.ptr || o.rows_per_batch <= 0 || o.t_in <= 0 || o.C <= 0 || o.stride <= 0) return false;
    if (o.dtype != PIKA_F32 && o.dtype != PIKA_BF16) return false;
    const int g = o.dtype == PIKA_F32 ? 3 : 7;  // elements per 16-byte load - 1
    if ((o.C & g) || (o.ld & g) || (o.batch_stride & g) || (o.z_outer & g) || (o.z_inner & g))
        return false;
    if (o.trans ? (nout & g) : (K & 3)) return false;   // the 16-byte loads run along this extent
    return (reinterpret_cast<uintptr_t>(o.ptr) & 15) == 0;
}

template <typename TA, typename TB, int NS, typename CF, bool TRA, bool TRB>
int launch(const pika_operand_t *A, const pika_operand_t *B, float *C, long long ldc,
           long long c_zo, long long c_zi, int M, int N, int K, int batch, int z_div,
           const float *bias, int flags, hipStream_t s) {
    constexpr size_t ta = TRA ? CF::BK * (CF::BM + 8) : CF::BM * CF::PITCH;
    constexpr size_t tb = TRB ? CF::BK * (CF::BN + 8) : CF::BN * CF::PITCH;
    constexpr size_t smem = (size_t)2 * NS * (ta + tb) * sizeof(__bf16);
    static_assert(smem <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    static bool attr_set = false;  // idempotent; racing threads set the same value
    auto kern = gemm_nt_kernel<TA, TB, NS, CF, TRA, TRB>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    // split-K when the output grid cannot fill 256 CUs and the reduction is long (weight
    // gradients: M,N ~ 1e3, K = rows ~ 3e4).  Not combinable with ReLU / accumulate.
    const int tiles = ((N + CF::BN - 1) / CF::BN) * ((M + CF::BM - 1) / CF::BM);
    const int nk = (K + CF::BK - 1) / CF::BK;
    int splitk = 1;
    static const int min_nk = [] { const char *e = pika_knob("PIKA_GEMM_SPLIT_MIN_NK"); return e ? atoi(e) : 32; }();
    static const int per_split = [] { const char *e = pika_knob("PIKA_GEMM_SPLIT_MIN_PER"); return e ? atoi(e) : 8; }();
    // exact-mode products with the reduction contiguous (forward products: the decode path) never split: the split
    // sums through float atomics, whose order -- and with it the last bit of the encoder output, and near-ties of a
    // beam search downstream -- changes from run to run.  Weight gradients (both operands reduction-major) keep it.
    const bool exact_forward = (flags & PIKA_GEMM_FP32SPLIT) && !TRA && !TRB;
    if (batch == 1 && tiles < 256 && nk >= min_nk && !exact_forward && !(flags & (PIKA_GEMM_RELU | PIKA_GEMM_ACCUMULATE))) {
        // One resident workgroup per CU.  Cost of split s in K-tile units: rounds of 256 workgroups x
        // (K-tiles per workgroup + ~40 of prologue/epilogue) + ~6 per atomic pass over C; the minimum
        // reproduces the measured optimum on every shape of tools/dw_bench.py
        // (profiles/r1_dw_split_sweep.txt).  PIKA_GEMM_SPLIT_TARGET=n forces ceil(n / tiles).
        static const int target = [] { const char *e = pika_knob("PIKA_GEMM_SPLIT_TARGET"); return e ? atoi(e) : 0; }();
        if (target > 0) {
            splitk = (target + tiles - 1) / tiles;
        } else {
            long long best = -1;
            for (int sp = 1; sp <= 64 && sp <= nk / per_split; ++sp) {
                const long long cost = (long long)((tiles * sp + 255) / 256) * ((nk + sp - 1) / sp + 40) + 6LL * sp;
                if (best < 0 || cost < best) { best = cost; splitk = sp; }
            }
        }
        if (splitk > nk / per_split) splitk = nk / per_split;
        if (splitk > 64) splitk = 64;
        if (splitk < 1) splitk = 1;
    }
    if (splitk > 1) {
        hipError_t e = hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s);
        if (e != hipSuccess) return (int)e;
    }
    if ((M + CF::BM - 1) / CF::BM > 65535) return PIKA_ETOOBIG;
    dim3 grid((N + CF::BN - 1) / CF::BN, (M + CF::BM - 1) / CF::BM, splitk > 1 ? splitk : batch);
    hipLaunchKernelGGL(kern, grid, dim3(CF::THREADS), smem, s, to_op(*A), to_op(*B), A->z_outer,
                       A->z_inner, B->z_outer, B->z_inner, C, ldc, c_zo, c_zi, M, N, K, z_div,
                       splitk, bias, flags);
    return (int)hipGetLastError();
}

// Tile configuration: PIKA_GEMM_CFG=0..3 overrides (hardware A/B only; NT operands).
//   0: 128x128x32 / 4 waves   1: 128x128x64 / 4 waves   2: 256x128x64 / 8 waves   3: 256x128x32
int cfg_override() {
    static const int v = [] { const char *e = pika_knob("PIKA_GEMM_CFG"); return e ? atoi(e) : -1; }();
    return v;
}

template <typename TA, typename TB, bool TRA, bool TRB>
int dispatch(const pika_operand_t *A, const pika_operand_t *B, float *C, long long ldc,
             long long c_zo, long long c_zi, int M, int N, int K, int batch, int z_div,
             const float *bias, int flags, hipStream_t s) {
#define ARGS A, B, C, ldc, c_zo, c_zi, M, N, K, batch, z_div, bias, flags, s
    if (flags & PIKA_GEMM_FP32SPLIT) {
        // parity mode keeps every tensor fp32 (bf16 operands still work through the NT path)
        if constexpr (!(TRA || TRB) || (sizeof(TA) == 4 && sizeof(TB) == 4))
            return launch<TA, TB, 3, Cfg<2, 2, 32>, TRA, TRB>(ARGS);
        else
            return PIKA_EINVAL;
    }
    if constexpr (TRA || TRB) {
        static const int tcfg = [] { const char *e = pika_knob("PIKA_GEMM_CFG_T"); return e ? atoi(e) : -1; }();
        if (tcfg == 1) return launch<TA, TB, 1, Cfg<2, 2, 64>, TRA, TRB>(ARGS);
        return launch<TA, TB, 1, Cfg<4, 2, 64>, TRA, TRB>(ARGS);
    } else {
        // measured on MI355X (tools/gemm_bench.py, profiles/r1_gemm_cfg_sweep.txt): 256x128x64 / 8
        // waves wins except for f32 x bf16 operands, where 128x128x64 does
        int cfg = cfg_override();
        if (cfg < 0) cfg = (sizeof(TA) == 4 && sizeof(TB) == 2) ? 1 : 2;
        switch (cfg) {
            case 0: return launch<TA, TB, 1, Cfg<2, 2, 32>, false, false>(ARGS);
            case 2: return launch<TA, TB, 1, Cfg<4, 2, 64>, false, false>(ARGS);
            case 3: return launch<TA, TB, 1, Cfg<4, 2, 32>, false, false>(ARGS);
            default: return launch<TA, TB, 1, Cfg<2, 2, 64>, false, false>(ARGS);
        }
    }
#undef ARGS
}

template <typename TA, typename TB>
int dispatch_trans(const pika_operand_t *A, const pika_operand_t *B, float *C, long long ldc,
                   long long c_zo, long long c_zi, int M, int N, int K, int batch, int z_div,
                   const float *bias, int flags, hipStream_t s) {
#define ARGS A, B, C, ldc, c_zo, c_zi, M, N, K, batch, z_div, bias, flags, s
    switch ((A->trans ? 2 : 0) | (B->trans ? 1 : 0)) {
        case 0: return dispatch<TA, TB, false, false>(ARGS);
        case 1: return dispatch<TA, TB, false, true>(ARGS);
        case 2: return dispatch<TA, TB, true, false>(ARGS);
        default: return dispatch<TA, TB, true, true>(ARGS);
    }
#undef ARGS
}

}  // namespace

int pika_internal_gemm_pp(const pika_operand_t *A, const pika_operand_t *B, float *C, long long ldc,
                          int M, int N, int K, const float *bias, int flags, void *ws, size_t ws_bytes,
                          hipStream_t s);

#ifdef PIKA_NT_TRACE
extern "C" int pika_debug_nt_trace(unsigned long long *out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_nt_trace), sizeof(g_nt_trace));
}
#endif

extern "C" int pika_gemm_nt(const pika_operand_t *A, const pika_operand_t *B, float *C,
                            long long ldc, long long c_z_outer, long long c_z_inner, int M, int N,
                            int K, int batch, int z_div, const float *bias, int flags,
                            void *stream) {
    return pika_gemm_nt_ws(A, B, C, ldc, c_z_outer, c_z_inner, M, N, K, batch, z_div, bias, flags, nullptr, 0,
                           stream);
}

extern "C" int pika_gemm_nt_ws(const pika_operand_t *A, const pika_operand_t *B, float *C,
                               long long ldc, long long c_z_outer, long long c_z_inner, int M, int N,
                               int K, int batch, int z_div, const float *bias, int flags,
                               void *workspace, size_t workspace_bytes, void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || z_div <= 0) return PIKA_EINVAL;
    if (!operand_ok(*A, K, M) || !operand_ok(*B, K, N)) return PIKA_EINVAL;
    if (batch > 65535) return PIKA_ETOOBIG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (batch == 1 && !c_z_outer && !c_z_inner) {   // bf16 x bf16: direct-to-LDS ping-pong kernel (gemm_glds.hip)
        static const bool off = pika_knob("PIKA_GEMM_NO_PP") != nullptr;
        const int rc = off ? -100 : pika_internal_gemm_pp(A, B, C, ldc, M, N, K, bias, flags, workspace, workspace_bytes, s);
        if (rc != -100) return rc;
    }
    flags &= ~PIKA_GEMM_TERM_PRODUCT;                                                 // (a gate of the direct-to-LDS kernel only)
    if (flags & (PIKA_GEMM_OUT_BF16 | PIKA_GEMM_F16_OPERANDS)) return PIKA_EINVAL;   // only the direct-to-LDS kernel writes bf16 / reads fp16
    const int key = (A->dtype == PIKA_BF16 ? 2 : 0) | (B->dtype == PIKA_BF16 ? 1 : 0);
#define ARGS A, B, C, ldc, c_z_outer, c_z_inner, M, N, K, batch, z_div, bias, flags, s
    switch (key) {
        case 0: return dispatch_trans<float, float>(ARGS);
        case 1: return dispatch_trans<float, __bf16>(ARGS);
        case 2: return dispatch_trans<__bf16, float>(ARGS);
        default: return dispatch_trans<__bf16, __bf16>(ARGS);
    }
#undef ARGS
}
