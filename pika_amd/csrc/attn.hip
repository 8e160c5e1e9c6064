// pika_amd/csrc/attn.hip -- fused multi-head self-attention (forward, dQ, dK/dV) for gfx950.
//
// One workgroup = 4 waves = 64 query rows (forward, dQ) or 64 key rows (dK/dV) of one (batch, head);
// the other axis streams through LDS in 64-row bf16 tiles.  All products are
// v_mfma_f32_16x16x32_bf16 issued in the TRANSPOSED sense, S^T = K.Q^T, so that a lane's
// accumulator registers hold reduction-axis neighbours of ONE output column (lane & 15): the
// probabilities / score gradients can then be re-used directly as the B operand of the next MFMA
// (O^T = V^T.P^T, dQ^T = K^T.dS^T, dV^T = dO^T.P, dK^T = Q^T.dS) with no trip through LDS.  The
// 32-deep reduction of that second MFMA takes its 8 k-slots per lane from two 16-row accumulator
// tiles (rows g*4..g*4+3 of each); the matching A operand comes out of the LDS tile with two
// ds_read_b64_tr_b16 whose row addresses follow the same permutation.
// Softmax is online (running max / partial sums in the log2 domain, log2(e)/sqrt(D) folded into
// the bf16 rounding of Q); per-row statistics are lane-uniform per output column, so only the row
// maximum needs a cross-lane exchange (two xor-shuffles per 64 keys).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pika_attn.h"
#include "pika_gemm.h"
#include "pika_rnnt.h"
#include "pika_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int TILE = 64;     // rows per LDS tile and per workgroup
constexpr int THREADS = 256;

__device__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// dropout decision for probability (row = (b*H+h)*T + query, key)
__device__ inline uint32_t row_hash(uint32_t seed, uint32_t row) { return mix32(seed + row * 0x9E3779B9u); }
// one hash decides two neighbouring keys (its two 16-bit halves)
__device__ inline bool keep(uint32_t rowh, uint32_t key, uint32_t thr) {
    const uint32_t h = mix32(rowh ^ ((key >> 1) * 0x85ebca6bu));
    return ((key & 1) ? (h >> 16) : (h & 0xffffu)) >= thr;
}
// keep bits of keys kb*64 .. kb*64+63 of one (b,h,query) row
__device__ inline uint64_t keep_word(uint32_t rowh, uint32_t kb, uint32_t thr) {
    // (the word's two halves in 32-bit registers: assembling 64 bits with 64-bit shifts and ors was more work than the hashes)
    uint32_t half[2] = {0u, 0u};
#pragma unroll
    for (uint32_t hw = 0; hw < 2; ++hw) {
#pragma unroll 8
        for (uint32_t j = 0; j < 16; ++j) {
            const uint32_t h = mix32(rowh ^ ((kb * 32 + hw * 16 + j) * 0x85ebca6bu));
            half[hw] |= (uint32_t)((h & 0xffffu) >= thr) << (2 * j);
            half[hw] |= (uint32_t)((h >> 16) >= thr) << (2 * j + 1);
        }
    }
    return (uint64_t)half[0] | ((uint64_t)half[1] << 32);
}

__device__ inline float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

// x where bit `bit` of `word` is set, else +0: a sign-extended one-bit field (0 / ~0) ANDed onto the value -- two
// instructions per element instead of shift / and / compare / select
__device__ inline float keep_if(float x, uint32_t word, int bit) {      // `bit` is a constant after unrolling
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(word), "v"(bit));
    return __builtin_bit_cast(float, __builtin_bit_cast(int, x) & m);
}
template <int BIT>
__device__ inline float keep_bit(float x, uint32_t word) {
    int m;      // v_bfe_i32 spelled out: the optimizer turns the builtin back into and / compare / select
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(word), "n"(BIT));
    return __builtin_bit_cast(float, __builtin_bit_cast(int, x) & m);
}
// two fp32 -> one packed bf16 pair (v_cvt_pk_bf16_f32), as the low / high half of a 32-bit lane of an MFMA operand
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pack2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 64 x D rows of T = float | bf16 (pitch ld) -> bf16 LDS tile with pitch D+8; rows >= nvalid are zero-filled.
template <int D, typename T>
__device__ inline void load_tile(__bf16 *lds, const T *g, long long ld, int nvalid, float scale) {
    constexpr int P = D + 8;
    if constexpr (sizeof(T) == 4) {
        constexpr int F4 = D / 4, RPP = THREADS / F4;
        const int c = threadIdx.x % F4, r0 = threadIdx.x / F4;
        f32x4 v[TILE / RPP];
#pragma unroll
        for (int p = 0; p < TILE / RPP; ++p) {
            const int r = r0 + p * RPP;
            v[p] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < nvalid) v[p] = *reinterpret_cast<const f32x4 *>(g + (long long)r * ld + c * 4);
        }
#pragma unroll
        for (int p = 0; p < TILE / RPP; ++p)
            *reinterpret_cast<bf16x4 *>(lds + (r0 + p * RPP) * P + c * 4) = __builtin_convertvector(v[p] * scale, bf16x4);
    } else {
        constexpr int F8 = D / 8, RPP = THREADS / F8;
        const int c = threadIdx.x % F8, r0 = threadIdx.x / F8;
        bf16x8 v[TILE / RPP];
#pragma unroll
        for (int p = 0; p < TILE / RPP; ++p) {
            const int r = r0 + p * RPP;
            v[p] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (r < nvalid) v[p] = *reinterpret_cast<const bf16x8 *>(g + (long long)r * ld + c * 8);
        }
#pragma unroll
        for (int p = 0; p < TILE / RPP; ++p) {
            if (scale != 1.f) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[p][e] = (__bf16)((float)v[p][e] * scale);
            }
            *reinterpret_cast<bf16x8 *>(lds + (r0 + p * RPP) * P + c * 8) = v[p];
        }
    }
}

// one row of a (.,D) matrix as D/32 MFMA operand fragments (8 consecutive d per lane)
template <int D, typename T>
__device__ inline void row_frags(bf16x8 *f, const T *row, int g, float scale) {
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) {
        if constexpr (sizeof(T) == 4) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(row + ks * 32 + g * 8) * scale;
            const f32x4 b = *reinterpret_cast<const f32x4 *>(row + ks * 32 + g * 8 + 4) * scale;
            f[ks] = bf16x8{(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w,
                           (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
        } else {
            bf16x8 v = *reinterpret_cast<const bf16x8 *>(row + ks * 32 + g * 8);
            if (scale != 1.f) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] * scale);
            }
            f[ks] = v;
        }
    }
}

template <typename T>
__device__ inline void store4(T *p, f32x4 v) {
    if constexpr (sizeof(T) == 4) *reinterpret_cast<f32x4 *>(p) = v;
    else *reinterpret_cast<bf16x4 *>(p) = __builtin_convertvector(v, bf16x4);
}

// operand fragment with tile rows as the MFMA row/column index: row r0 + (lane&15), k = kk + g*8..+7
template <int P>
__device__ inline bf16x8 frag_n(const __bf16 *tile, int r0, int kk, int lane) {
    return *reinterpret_cast<const bf16x8 *>(tile + (r0 + (lane & 15)) * P + kk + (lane >> 4) * 8);
}
// operand fragment with tile COLUMNS as the MFMA row index (c0 + (lane&15)) and tile rows as the
// reduction: k-slots 0..3 = rows k0 + g*4 + {0..3}, slots 4..7 = rows k0 + 16 + g*4 + {0..3}.
template <int P>
__device__ inline bf16x8 frag_t(const __bf16 *tile, int c0, int k0, int lane) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int i = lane & 15, j = i >> 2, c = i & 3, g = lane >> 4;
    const __bf16 *p0 = tile + (k0 + g * 4 + j) * P + c0 + 4 * c;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p0 + 16 * P));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ inline f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

struct Args {
    const void *q, *k, *v, *out, *dout;   // T = float | bf16 (io dtype)
    const float *lse, *delta;
    void *o, *dq, *dk, *dv;
    float *lse_w;
    int T, H;
    long long ld, ldo;  // row pitch of q/k/v/dq/dk/dv and of out/dout
    float qscale;     // log2(e) / sqrt(D)
    float inv_keep;   // 1 / (1 - p)
    uint32_t seed, thr;
    const uint64_t *bits;   // keep bits [b*H+h][query][ceil(T/64)] (dropout only)
    int nkb;
    const uint64_t *mask;        // optional packed mask [b][query][ceil(T/64)]: bit (key & 63) of word key >> 6 set = the
                                 // score is masked to -1e18 (multi_headed_attn.py:217); pika_attention_mask_bits packs it
    long long lo_off, olo_off;   // two-term forward: element offset of the "lo" planes of q/k/v and of out
};

// masked_fill(mask, -1e18) in the kernels' log2 domain (scores carry log2(e)): a FINITE fill, as in the reference -- a
// row whose keys are all masked comes out uniform, not NaN
constexpr float MASKED = -1.4426950408889634e18f;

template <int D, typename T_>
__global__ __launch_bounds__(THREADS) void attn_fwd_kernel(Args A) {
    const T_ *Aq = static_cast<const T_ *>(A.q), *Ak = static_cast<const T_ *>(A.k), *Av = static_cast<const T_ *>(A.v);
    constexpr int P = D + 8, KS = D / 32, DT = D / 16;
    __shared__ __attribute__((aligned(16))) __bf16 Ks[TILE * P];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[TILE * P];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int T = A.T, h = blockIdx.y, b = blockIdx.z;
    const int qrow = blockIdx.x * TILE + wave * 16 + (lane & 15);
    const long long boff = (long long)b * T * A.ld + (long long)h * D;
    const uint32_t bh = (uint32_t)(b * A.H + h);
    bf16x8 qf[KS];
    row_frags<D>(qf, Aq + boff + (long long)min(qrow, T - 1) * A.ld, g, A.qscale);
    const uint64_t *bits = A.bits + ((long long)bh * T + min(qrow, T - 1)) * A.nkb;
    f32x4 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lpart = 0.f;
    for (int kb = 0; kb < T; kb += TILE) {
        __syncthreads();
        load_tile<D>(Ks, Ak + boff + (long long)kb * A.ld, A.ld, T - kb, 1.f);
        load_tile<D>(Vs, Av + boff + (long long)kb * A.ld, A.ld, T - kb, 1.f);
        __syncthreads();
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s[kt] = mfma(frag_n<P>(Ks, kt * 16, ks * 32, lane), qf[ks], s[kt]);
        }
        if (A.mask) {      // one 8-byte word per query row and key tile, as the dropout keep bits
            const uint64_t mw = A.mask[((long long)b * T + min(qrow, T - 1)) * A.nkb + (kb >> 6)] >> (g * 4);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((mw >> (kt * 16 + r)) & 1ull) s[kt][r] = MASKED;
        }
        if (kb + TILE > T) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kb + kt * 16 + g * 4 + r >= T) s[kt][r] = -INFINITY;
        }
        float mb = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) mb = fmaxf(mb, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
        mb = fmaxf(mb, __shfl_xor(mb, 16));
        mb = fmaxf(mb, __shfl_xor(mb, 32));
        const float mn = fmaxf(m, mb), alpha = ex2(m - mn);
        m = mn;
        lpart *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[dt] *= alpha;
        bf16x8 pf[2];
        const uint64_t kw = A.thr ? (bits[kb >> 6] >> (g * 4)) : ~0ull;   // bit kt*16 + r = key kt*16 + g*4 + r
        const uint32_t kw0 = (uint32_t)kw, kw1 = (uint32_t)(kw >> 32);
        {
            float p[4][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[kt][r] = ex2(s[kt][r] - mn);
                    lpart += p[kt][r];
                }
#define PIKA_KEEP(kt, r) keep_bit<((kt) & 1) * 16 + (r)>(p[kt][r], (kt) < 2 ? kw0 : kw1)
#define PIKA_PAIR(kt, r) pack2(PIKA_KEEP(kt, r), PIKA_KEEP(kt, (r) + 1))
            pf[0] = __builtin_bit_cast(bf16x8, u32x4{PIKA_PAIR(0, 0), PIKA_PAIR(0, 2), PIKA_PAIR(1, 0), PIKA_PAIR(1, 2)});
            pf[1] = __builtin_bit_cast(bf16x8, u32x4{PIKA_PAIR(2, 0), PIKA_PAIR(2, 2), PIKA_PAIR(3, 0), PIKA_PAIR(3, 2)});
#undef PIKA_PAIR
#undef PIKA_KEEP
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) oacc[dt] = mfma(frag_t<P>(Vs, dt * 16, s2 * 32, lane), pf[s2], oacc[dt]);
    }
    float l = lpart;
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (qrow < T) {
        const float sc = A.inv_keep / l;
        T_ *o = static_cast<T_ *>(A.o) + (long long)b * T * A.ldo + (long long)h * D + (long long)qrow * A.ldo + g * 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) store4(o + dt * 16, oacc[dt] * sc);
        if (g == 0) A.lse_w[(long long)bh * T + qrow] = m + __log2f(l);
    }
}


// Forward in TWO-TERM arithmetic (the parity-carrying train step: DESIGN 6.1): q, k, v arrive as bf16 "hi" planes with
// a "lo" plane A.lo_off elements behind (x = hi + lo to 16 mantissa bits: the EPI 1 two-term output of the packed
// projection GEMM), and both products keep the three leading term products on the matrix cores:
//   S^T = Kh.qh + Kl.qh + Kh.ql        O^T = Vh^T.ph + Vl^T.ph + Vh^T.pl
// with the softmax in fp32 in between (p = ph + pl split in registers).  The context goes out as two planes as well
// (A.olo_off), the two-term operand of the output projection.  The backward runs on the hi planes (bf16 kernels above).
// Same tiling as attn_fwd_kernel; the four 64-row tiles (K / V, hi / lo) take 36 / 68 KB of dynamic LDS for D = 64 / 128.
__device__ inline float hi_of(uint32_t w, int half) {     // the bf16 in the low / high half of w as a float
    return __builtin_bit_cast(float, half ? (w & 0xffff0000u) : (w << 16));
}
template <int D>
__device__ inline void row_frags2(bf16x8 *fh, bf16x8 *fl, const __bf16 *hi, const __bf16 *lo, int g, float scale) {
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(hi + ks * 32 + g * 8);
        const bf16x8 b = *reinterpret_cast<const bf16x8 *>(lo + ks * 32 + g * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = ((float)a[e] + (float)b[e]) * scale;
            const __bf16 h = (__bf16)x;
            fh[ks][e] = h;
            fl[ks][e] = (__bf16)(x - (float)h);
        }
    }
}

template <int D>
__global__ __launch_bounds__(THREADS) void attn_fwd2_kernel(Args A) {
    const __bf16 *Aq = static_cast<const __bf16 *>(A.q), *Ak = static_cast<const __bf16 *>(A.k), *Av = static_cast<const __bf16 *>(A.v);
    constexpr int P = D + 8, KS = D / 32, DT = D / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    __bf16 *Kh = reinterpret_cast<__bf16 *>(smem2), *Kl = Kh + TILE * P, *Vh = Kl + TILE * P, *Vl = Vh + TILE * P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int T = A.T, h = blockIdx.y, b = blockIdx.z;
    const int qrow = blockIdx.x * TILE + wave * 16 + (lane & 15);
    const long long boff = (long long)b * T * A.ld + (long long)h * D;
    const uint32_t bh = (uint32_t)(b * A.H + h);
    bf16x8 qh[KS], ql[KS];
    {
        const __bf16 *qp = Aq + boff + (long long)min(qrow, T - 1) * A.ld;
        row_frags2<D>(qh, ql, qp, qp + A.lo_off, g, A.qscale);
    }
    const uint64_t *bits = A.bits + ((long long)bh * T + min(qrow, T - 1)) * A.nkb;
    const uint64_t *mrow = A.mask ? A.mask + ((long long)b * T + min(qrow, T - 1)) * A.nkb : nullptr;
    f32x4 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lpart = 0.f;
    for (int kb = 0; kb < T; kb += TILE) {
        __syncthreads();
        load_tile<D>(Kh, Ak + boff + (long long)kb * A.ld, A.ld, T - kb, 1.f);
        load_tile<D>(Kl, Ak + A.lo_off + boff + (long long)kb * A.ld, A.ld, T - kb, 1.f);
        load_tile<D>(Vh, Av + boff + (long long)kb * A.ld, A.ld, T - kb, 1.f);
        load_tile<D>(Vl, Av + A.lo_off + boff + (long long)kb * A.ld, A.ld, T - kb, 1.f);
        __syncthreads();
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {     // small terms first
                const bf16x8 kh = frag_n<P>(Kh, kt * 16, ks * 32, lane);
                s[kt] = mfma(frag_n<P>(Kl, kt * 16, ks * 32, lane), qh[ks], s[kt]);
                s[kt] = mfma(kh, ql[ks], s[kt]);
                s[kt] = mfma(kh, qh[ks], s[kt]);
            }
        }
        if (mrow) {
            const uint64_t mw = mrow[kb >> 6] >> (g * 4);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((mw >> (kt * 16 + r)) & 1ull) s[kt][r] = MASKED;
        }
        if (kb + TILE > T) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kb + kt * 16 + g * 4 + r >= T) s[kt][r] = -INFINITY;
        }
        float mb = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) mb = fmaxf(mb, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
        mb = fmaxf(mb, __shfl_xor(mb, 16));
        mb = fmaxf(mb, __shfl_xor(mb, 32));
        const float mn = fmaxf(m, mb), alpha = ex2(m - mn);
        m = mn;
        lpart *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[dt] *= alpha;
        const uint64_t kw = A.thr ? (bits[kb >> 6] >> (g * 4)) : ~0ull;   // bit kt*16 + r = key kt*16 + g*4 + r
        const uint32_t kw0 = (uint32_t)kw, kw1 = (uint32_t)(kw >> 32);
        uint32_t wh[8], wl[8];      // word 2*kt + j = keys (kt*16 + g*4 + 2j, +1): the k-slots of MFMA s2 = kt / 2
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float p0 = ex2(s[kt][2 * j] - mn), p1 = ex2(s[kt][2 * j + 1] - mn);
                lpart += p0 + p1;
                const uint32_t kwd = kt < 2 ? kw0 : kw1;
                p0 = keep_if(p0, kwd, (kt & 1) * 16 + 2 * j);
                p1 = keep_if(p1, kwd, (kt & 1) * 16 + 2 * j + 1);
                const uint32_t w = pack2(p0, p1);
                wh[2 * kt + j] = w;
                wl[2 * kt + j] = pack2(p0 - hi_of(w, 0), p1 - hi_of(w, 1));
            }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 ph = __builtin_bit_cast(bf16x8, u32x4{wh[4 * s2], wh[4 * s2 + 1], wh[4 * s2 + 2], wh[4 * s2 + 3]});
            const bf16x8 pl = __builtin_bit_cast(bf16x8, u32x4{wl[4 * s2], wl[4 * s2 + 1], wl[4 * s2 + 2], wl[4 * s2 + 3]});
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 vh = frag_t<P>(Vh, dt * 16, s2 * 32, lane);
                oacc[dt] = mfma(frag_t<P>(Vl, dt * 16, s2 * 32, lane), ph, oacc[dt]);
                oacc[dt] = mfma(vh, pl, oacc[dt]);
                oacc[dt] = mfma(vh, ph, oacc[dt]);
            }
        }
    }
    float l = lpart;
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (qrow < T) {
        const float sc = A.inv_keep / l;
        __bf16 *o = static_cast<__bf16 *>(A.o) + (long long)b * T * A.ldo + (long long)h * D + (long long)qrow * A.ldo + g * 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const f32x4 v = oacc[dt] * sc;
            const bf16x4 hi4 = __builtin_convertvector(v, bf16x4);
            *reinterpret_cast<bf16x4 *>(o + dt * 16) = hi4;
            *reinterpret_cast<bf16x4 *>(o + A.olo_off + dt * 16) = __builtin_convertvector(v - __builtin_convertvector(hi4, f32x4), bf16x4);
        }
        if (g == 0) A.lse_w[(long long)bh * T + qrow] = m + __log2f(l);
    }
}

// INFERENCE forward on two FP16 terms per operand (the decoder's encoder pass: fp32-grade products, ~2^-22):
// x = hi + 2^-11 lo' with hi = fp16(x), lo' = fp16((x - hi) 2^11) -- the second term is kept at the magnitude of the first
// (never subnormal where hi is not), the leading products hi.hi accumulate in one accumulator and the cross products
// lo'.hi + hi.lo' in a second one that joins scaled by 2^-11:
//   S^T = Kh.qh + 2^-11 (Kl'.qh + Kh.ql')        O^T = Vh^T.ph + 2^-11 (Vl'^T.ph + Vh^T.pl')
// Softmax in fp32; no dropout; the context leaves as fp32.  Same tiling / LDS budget as attn_fwd2_kernel.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float F16_LO = 2048.f, F16_LO_INV = 1.f / 2048.f;
__device__ inline f32x4 mfma_h(bf16x8 a, bf16x8 b, f32x4 c) {      // the operands carry fp16 bit patterns
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ inline uint32_t pack2h(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2));
}
__device__ inline float half_of(uint32_t w, int half) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(half ? (w >> 16) : (w & 0xffffu)));
}

template <int D>
__global__ __launch_bounds__(THREADS) void attn_infer_f16x2_kernel(Args A) {
    const _Float16 *Aq = static_cast<const _Float16 *>(A.q);
    const __bf16 *Ak = static_cast<const __bf16 *>(A.k), *Av = static_cast<const __bf16 *>(A.v);   // 16-bit patterns
    constexpr int P = D + 8, KS = D / 32, DT = D / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    __bf16 *Kh = reinterpret_cast<__bf16 *>(smem2), *Kl = Kh + TILE * P, *Vh = Kl + TILE * P, *Vl = Vh + TILE * P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int T = A.T, h = blockIdx.y, b = blockIdx.z;
    const int qrow = blockIdx.x * TILE + wave * 16 + (lane & 15);
    const long long boff = (long long)b * T * A.ld + (long long)h * D;
    bf16x8 qh[KS], ql[KS];
    {
        const _Float16 *qp = Aq + boff + (long long)min(qrow, T - 1) * A.ld;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 a = *reinterpret_cast<const f16x8 *>(qp + ks * 32 + g * 8);
            const f16x8 c = *reinterpret_cast<const f16x8 *>(qp + A.lo_off + ks * 32 + g * 8);
            f16x8 fh, fl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = ((float)a[e] + (float)c[e] * F16_LO_INV) * A.qscale;
                const _Float16 hh = (_Float16)x;
                fh[e] = hh;
                fl[e] = (_Float16)((x - (float)hh) * F16_LO);
            }
            qh[ks] = __builtin_bit_cast(bf16x8, fh);
            ql[ks] = __builtin_bit_cast(bf16x8, fl);
        }
    }
    const uint64_t *mrow = A.mask ? A.mask + ((long long)b * T + min(qrow, T - 1)) * A.nkb : nullptr;
    f32x4 oacc[DT], oaccx[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = oaccx[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lpart = 0.f;
    for (int kb = 0; kb < T; kb += TILE) {
        __syncthreads();
        load_tile<D>(Kh, Ak + boff + (long long)kb * A.ld, A.ld, T - kb, 1.f);
        load_tile<D>(Kl, Ak + A.lo_off + boff + (long long)kb * A.ld, A.ld, T - kb, 1.f);
        load_tile<D>(Vh, Av + boff + (long long)kb * A.ld, A.ld, T - kb, 1.f);
        load_tile<D>(Vl, Av + A.lo_off + boff + (long long)kb * A.ld, A.ld, T - kb, 1.f);
        __syncthreads();
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 sx = f32x4{0.f, 0.f, 0.f, 0.f};
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kh = frag_n<P>(Kh, kt * 16, ks * 32, lane);
                sx = mfma_h(frag_n<P>(Kl, kt * 16, ks * 32, lane), qh[ks], sx);
                sx = mfma_h(kh, ql[ks], sx);
                s[kt] = mfma_h(kh, qh[ks], s[kt]);
            }
            s[kt] += sx * F16_LO_INV;
        }
        if (mrow) {
            const uint64_t mw = mrow[kb >> 6] >> (g * 4);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((mw >> (kt * 16 + r)) & 1ull) s[kt][r] = MASKED;
        }
        if (kb + TILE > T) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kb + kt * 16 + g * 4 + r >= T) s[kt][r] = -INFINITY;
        }
        float mb = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) mb = fmaxf(mb, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
        mb = fmaxf(mb, __shfl_xor(mb, 16));
        mb = fmaxf(mb, __shfl_xor(mb, 32));
        const float mn = fmaxf(m, mb), alpha = ex2(m - mn);
        m = mn;
        lpart *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            oacc[dt] *= alpha;
            oaccx[dt] *= alpha;
        }
        uint32_t wh[8], wl[8];      // word 2*kt + j = keys (kt*16 + g*4 + 2j, +1): the k-slots of MFMA s2 = kt / 2
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float p0 = ex2(s[kt][2 * j] - mn), p1 = ex2(s[kt][2 * j + 1] - mn);
                lpart += p0 + p1;
                const uint32_t w = pack2h(p0, p1);
                wh[2 * kt + j] = w;
                wl[2 * kt + j] = pack2h((p0 - half_of(w, 0)) * F16_LO, (p1 - half_of(w, 1)) * F16_LO);
            }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 ph = __builtin_bit_cast(bf16x8, u32x4{wh[4 * s2], wh[4 * s2 + 1], wh[4 * s2 + 2], wh[4 * s2 + 3]});
            const bf16x8 pl = __builtin_bit_cast(bf16x8, u32x4{wl[4 * s2], wl[4 * s2 + 1], wl[4 * s2 + 2], wl[4 * s2 + 3]});
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 vh = frag_t<P>(Vh, dt * 16, s2 * 32, lane);
                oaccx[dt] = mfma_h(frag_t<P>(Vl, dt * 16, s2 * 32, lane), ph, oaccx[dt]);
                oaccx[dt] = mfma_h(vh, pl, oaccx[dt]);
                oacc[dt] = mfma_h(vh, ph, oacc[dt]);
            }
        }
    }
    float l = lpart;
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (qrow < T) {
        const float sc = 1.f / l;
        float *o = static_cast<float *>(A.o) + (long long)b * T * A.ldo + (long long)h * D + (long long)qrow * A.ldo + g * 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4 *>(o + dt * 16) = (oacc[dt] + oaccx[dt] * F16_LO_INV) * sc;
    }
}

// delta[bh*T + q] = sum_d out[b,q,h,d] * dout[b,q,h,d]; one workgroup per (b,q) row.
template <typename T_>
__device__ inline f32x4 load4(const T_ *p) {
    if constexpr (sizeof(T_) == 4) {
        return *reinterpret_cast<const f32x4 *>(p);
    } else {
        const bf16x4 v = *reinterpret_cast<const bf16x4 *>(p);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
}

template <int D, typename T_>
__global__ __launch_bounds__(THREADS) void attn_delta_kernel(Args A, float *delta) {
    const int T = A.T, HD = A.H * D;
    const long long row = blockIdx.x;  // b*T + q
    const int b = (int)(row / T), qi = (int)(row - (long long)b * T);
    for (int c = threadIdx.x * 4; c < HD; c += THREADS * 4) {
        const f32x4 o = load4(static_cast<const T_ *>(A.out) + row * A.ldo + c);
        const f32x4 d = load4(static_cast<const T_ *>(A.dout) + row * A.ldo + c);
        float s = o.x * d.x + o.y * d.y + o.z * d.z + o.w * d.w;
#pragma unroll
        for (int w = 1; w < D / 4; w <<= 1) s += __shfl_xor(s, w);
        if ((threadIdx.x & (D / 4 - 1)) == 0) delta[((long long)b * A.H + c / D) * T + qi] = s;
    }
}

template <int D, typename T_, bool MASK>
__global__ __launch_bounds__(THREADS) void attn_bwd_q_kernel(Args A) {
    const T_ *Aq = static_cast<const T_ *>(A.q), *Ak = static_cast<const T_ *>(A.k), *Av = static_cast<const T_ *>(A.v);
    const T_ *Ado = static_cast<const T_ *>(A.dout);
    constexpr int P = D + 8, KS = D / 32, DT = D / 16;
    __shared__ __attribute__((aligned(16))) __bf16 Ks[TILE * P];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[TILE * P];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int T = A.T, h = blockIdx.y, b = blockIdx.z;
    const int qrow = blockIdx.x * TILE + wave * 16 + (lane & 15);
    const long long boff = (long long)b * T * A.ld + (long long)h * D;
    const uint32_t bh = (uint32_t)(b * A.H + h);
    bf16x8 qf[KS], dof[KS];
    row_frags<D>(qf, Aq + boff + (long long)min(qrow, T - 1) * A.ld, g, A.qscale);
    row_frags<D>(dof, Ado + (long long)b * T * A.ldo + (long long)h * D + (long long)min(qrow, T - 1) * A.ldo, g, 1.f);
    const float lse = qrow < T ? A.lse[(long long)bh * T + qrow] : INFINITY;
    const float delta = qrow < T ? A.delta[(long long)bh * T + qrow] : 0.f;
    const uint64_t *bits = A.bits + ((long long)bh * T + min(qrow, T - 1)) * A.nkb;
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < T; kb += TILE) {
        __syncthreads();
        load_tile<D>(Ks, Ak + boff + (long long)kb * A.ld, A.ld, T - kb, 1.f);
        load_tile<D>(Vs, Av + boff + (long long)kb * A.ld, A.ld, T - kb, 1.f);
        __syncthreads();
        const uint64_t kw = A.thr ? (bits[kb >> 6] >> (g * 4)) : ~0ull;
        const uint32_t kw0 = (uint32_t)kw, kw1 = (uint32_t)(kw >> 32);
        [[maybe_unused]] uint64_t mw = 0;
        if constexpr (MASK) mw = A.mask[((long long)b * T + min(qrow, T - 1)) * A.nkb + (kb >> 6)] >> (g * 4);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint32_t dsw[4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kt = 2 * s2 + half;
                f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    sa = mfma(frag_n<P>(Ks, kt * 16, ks * 32, lane), qf[ks], sa);
                    dp = mfma(frag_n<P>(Vs, kt * 16, ks * 32, lane), dof[ks], dp);
                }
                float ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kb + kt * 16 + g * 4 + r;
                    if constexpr (MASK) {      // a separate instantiation: these kernels are bound by VALU issue
                        if ((mw >> (kt * 16 + r)) & 1ull) sa[r] = MASKED;
                    }
                    const float p = key < T ? ex2(sa[r] - lse) : 0.f;
                    const float d = keep_if(dp[r] * A.inv_keep, s2 == 0 ? kw0 : kw1, half * 16 + r);
                    ds[r] = p * (d - delta);
                }
                dsw[half * 2] = pack2(ds[0], ds[1]);
                dsw[half * 2 + 1] = pack2(ds[2], ds[3]);
            }
            const bf16x8 dsf = __builtin_bit_cast(bf16x8, u32x4{dsw[0], dsw[1], dsw[2], dsw[3]});
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] = mfma(frag_t<P>(Ks, dt * 16, s2 * 32, lane), dsf, acc[dt]);
        }
    }
    if (qrow < T) {
        const float sc = rsqrtf((float)D);
        T_ *o = static_cast<T_ *>(A.dq) + boff + (long long)qrow * A.ld + g * 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) store4(o + dt * 16, acc[dt] * sc);
    }
}

template <int D, typename T_, bool MASK>
__global__ __launch_bounds__(THREADS) void attn_bwd_kv_kernel(Args A) {
    const T_ *Aq = static_cast<const T_ *>(A.q), *Ak = static_cast<const T_ *>(A.k), *Av = static_cast<const T_ *>(A.v);
    const T_ *Ado = static_cast<const T_ *>(A.dout);
    constexpr int P = D + 8, KS = D / 32, DT = D / 16;
    __shared__ __attribute__((aligned(16))) __bf16 Qs[TILE * P];
    __shared__ __attribute__((aligned(16))) __bf16 Os[TILE * P];
    __shared__ float lse_s[TILE], delta_s[TILE];
    __shared__ uint32_t wlo_s[TILE], whi_s[TILE];     // the two halves of every query's keep word for this key block
    __shared__ uint32_t mlo_s[TILE], mhi_s[TILE];     // ... and of its mask word (MASK)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int T = A.T, h = blockIdx.y, b = blockIdx.z;
    const int krow = blockIdx.x * TILE + wave * 16 + (lane & 15);
    const bool kbit_hi = (krow & 32) != 0;
    const int kbit = krow & 31;
    const long long boff = (long long)b * T * A.ld + (long long)h * D;
    const uint32_t bh = (uint32_t)(b * A.H + h);
    bf16x8 kf[KS], vf[KS];
    row_frags<D>(kf, Ak + boff + (long long)min(krow, T - 1) * A.ld, g, 1.f);
    row_frags<D>(vf, Av + boff + (long long)min(krow, T - 1) * A.ld, g, 1.f);
    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int qb = 0; qb < T; qb += TILE) {
        __syncthreads();
        load_tile<D>(Qs, Aq + boff + (long long)qb * A.ld, A.ld, T - qb, A.qscale);
        load_tile<D>(Os, Ado + (long long)b * T * A.ldo + (long long)h * D + (long long)qb * A.ldo, A.ldo, T - qb, 1.f);
        if (threadIdx.x < TILE) {
            const int qi = qb + threadIdx.x;
            lse_s[threadIdx.x] = qi < T ? A.lse[(long long)bh * T + qi] : INFINITY;
            delta_s[threadIdx.x] = qi < T ? A.delta[(long long)bh * T + qi] : 0.f;
            const uint64_t w64 = (A.thr && qi < T) ? A.bits[((long long)bh * T + qi) * A.nkb + blockIdx.x] : ~0ull;
            wlo_s[threadIdx.x] = (uint32_t)w64;
            whi_s[threadIdx.x] = (uint32_t)(w64 >> 32);
            if constexpr (MASK) {
                const uint64_t m64 = qi < T ? A.mask[((long long)b * T + qi) * A.nkb + blockIdx.x] : 0ull;
                mlo_s[threadIdx.x] = (uint32_t)m64;
                mhi_s[threadIdx.x] = (uint32_t)(m64 >> 32);
            }
        }
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint32_t pdw[4], dsw[4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qt = 2 * s2 + half;
                f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    sa = mfma(frag_n<P>(Qs, qt * 16, ks * 32, lane), kf[ks], sa);
                    dp = mfma(frag_n<P>(Os, qt * 16, ks * 32, lane), vf[ks], dp);
                }
                float pdv[4], dsv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qi = qt * 16 + g * 4 + r;
                    if constexpr (MASK) {
                        if (((kbit_hi ? mhi_s[qi] : mlo_s[qi]) >> kbit) & 1u) sa[r] = MASKED;
                    }
                    const float p = ex2(sa[r] - lse_s[qi]);
                    // this lane's key is bit (krow & 63) of the query's keep word: the half that holds it, then one
                    // sign-extended bit field as an AND mask for both products
                    const uint32_t wd = kbit_hi ? whi_s[qi] : wlo_s[qi];
                    const int m = __builtin_amdgcn_sbfe((int)wd, (unsigned)kbit, 1u);
                    pdv[r] = __builtin_bit_cast(float, __builtin_bit_cast(int, p * A.inv_keep) & m);
                    const float d = __builtin_bit_cast(float, __builtin_bit_cast(int, dp[r] * A.inv_keep) & m);
                    dsv[r] = p * (d - delta_s[qi]);
                }
                pdw[half * 2] = pack2(pdv[0], pdv[1]); pdw[half * 2 + 1] = pack2(pdv[2], pdv[3]);
                dsw[half * 2] = pack2(dsv[0], dsv[1]); dsw[half * 2 + 1] = pack2(dsv[2], dsv[3]);
            }
            const bf16x8 pdf = __builtin_bit_cast(bf16x8, u32x4{pdw[0], pdw[1], pdw[2], pdw[3]});
            const bf16x8 dsf = __builtin_bit_cast(bf16x8, u32x4{dsw[0], dsw[1], dsw[2], dsw[3]});
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dv[dt] = mfma(frag_t<P>(Os, dt * 16, s2 * 32, lane), pdf, dv[dt]);
                dk[dt] = mfma(frag_t<P>(Qs, dt * 16, s2 * 32, lane), dsf, dk[dt]);
            }
        }
    }
    if (krow < T) {
        const float ln2 = 0.6931471805599453f;  // Qs carries log2(e)/sqrt(D)
        T_ *ok = static_cast<T_ *>(A.dk) + boff + (long long)krow * A.ld + g * 4;
        T_ *ov = static_cast<T_ *>(A.dv) + boff + (long long)krow * A.ld + g * 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            store4(ok + dt * 16, dk[dt] * ln2);
            store4(ov + dt * 16, dv[dt]);
        }
    }
}

// bits[row][kb] for row = (b*H+h)*T + query: one thread per 64-key word
__global__ __launch_bounds__(THREADS) void keep_bits_kernel(uint64_t *bits, long long rows, int nkb,
                                                            uint32_t seed, uint32_t thr, const unsigned *__restrict__ salt) {
    if (salt) seed += *salt;       // pika_set_dropout_salt: new masks per replay of a captured launch sequence
    const long long idx = (long long)blockIdx.x * THREADS + threadIdx.x;
    if (idx >= rows * nkb) return;
    const long long row = idx / nkb;
    bits[idx] = keep_word(row_hash(seed, (uint32_t)row), (uint32_t)(idx - row * nkb), thr);
}

// packed attention mask: bits[(b*T + q)*nkb + w] bit j = mask[b][q][w*64 + j] != 0 (keys beyond T: clear)
__global__ __launch_bounds__(THREADS) void mask_bits_kernel(const unsigned char *__restrict__ mask, uint64_t *__restrict__ bits,
                                                            long long rows, int T, int nkb) {
    const long long idx = (long long)blockIdx.x * THREADS + threadIdx.x;
    if (idx >= rows * nkb) return;
    const long long row = idx / nkb;
    const int w = (int)(idx - row * nkb);
    const unsigned char *m = mask + row * T + (long long)w * 64;
    uint64_t v = 0;
    const int n = min(64, T - w * 64);
    for (int j = 0; j < n; ++j) v |= (uint64_t)(m[j] != 0) << j;
    bits[idx] = v;
}

__global__ __launch_bounds__(THREADS) void keep_mask_kernel(unsigned char *mask, long long rows, int T,
                                                            uint32_t seed, uint32_t thr, const unsigned *__restrict__ salt) {
    if (salt) seed += *salt;
    const long long row = blockIdx.x;  // bh*T + q
    const uint32_t rowh = row_hash(seed, (uint32_t)row);
    for (int key = threadIdx.x; key < T; key += THREADS)
        mask[row * T + key] = (!thr || keep(rowh, (uint32_t)key, thr)) ? 1 : 0;
}

inline bool args_ok(const void *a, const void *b, const void *c, const void *d, int B, int T, int H, int D,
                    long long ld, float p, int g) {
    if (!a || !b || !c || !d || B <= 0 || T <= 0 || H <= 0) return false;
    if ((D != 64 && D != 128) || ld < (long long)H * D || (ld & (g - 1))) return false;
    if (!(p >= 0.f && p < 1.f)) return false;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
         reinterpret_cast<uintptr_t>(d)) & 15)
        return false;
    return true;
}

inline void dropout_consts(Args &A, float p, unsigned seed) {
    A.thr = (uint32_t)lrintf(p * 65536.f);
    A.inv_keep = 65536.f / (float)(65536u - A.thr);
    A.seed = seed;
}

template <int D, typename T_>
void launch_fwd(const Args &A, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((attn_fwd_kernel<D, T_>), grid, dim3(THREADS), 0, s, A);
}

template <int D, typename T_>
void launch_bwd(const Args &A, dim3 grid, int rows, float *delta, hipStream_t s) {
    hipLaunchKernelGGL((attn_delta_kernel<D, T_>), dim3((unsigned)rows), dim3(THREADS), 0, s, A, delta);
    if (A.mask) {
        hipLaunchKernelGGL((attn_bwd_kv_kernel<D, T_, true>), grid, dim3(THREADS), 0, s, A);
        hipLaunchKernelGGL((attn_bwd_q_kernel<D, T_, true>), grid, dim3(THREADS), 0, s, A);
    } else {
        hipLaunchKernelGGL((attn_bwd_kv_kernel<D, T_, false>), grid, dim3(THREADS), 0, s, A);
        hipLaunchKernelGGL((attn_bwd_q_kernel<D, T_, false>), grid, dim3(THREADS), 0, s, A);
    }
}

}  // namespace

extern "C" {

static int attention_fwd_impl(const void *q, const void *k, const void *v, void *out, int io_dtype, float *lse,
                              void *keep_bits, const uint64_t *mask, long long lo_off, long long olo_off, int B, int T,
                              int H, int D, long long ld, long long ldo, float p_drop, unsigned seed, void *stream) {
    if (io_dtype != PIKA_F32 && io_dtype != PIKA_BF16 && io_dtype != -2) return PIKA_EINVAL;
    const bool two_term = io_dtype == -2;
    if (two_term) io_dtype = PIKA_BF16;
    const int g = io_dtype == PIKA_F32 ? 4 : 8;
    if (!args_ok(q, k, v, out, B, T, H, D, ld, p_drop, g) || !lse || ldo < (long long)H * D || (ldo & (g - 1))) return PIKA_EINVAL;
    if (B > 65535 || H > 65535 || (long long)B * H * T > 0x7fffffffLL) return PIKA_ETOOBIG;
    Args A{};
    A.q = q; A.k = k; A.v = v; A.o = out; A.lse_w = lse; A.T = T; A.H = H; A.ld = ld; A.ldo = ldo;
    A.qscale = 1.4426950408889634f / sqrtf((float)D);
    A.mask = mask; A.lo_off = lo_off; A.olo_off = olo_off;
    dropout_consts(A, p_drop, seed);
    const dim3 grid((T + TILE - 1) / TILE, H, B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    A.nkb = (T + 63) / 64;
    if (A.thr) {
        if (!keep_bits || (reinterpret_cast<uintptr_t>(keep_bits) & 7)) return PIKA_EINVAL;
        const long long words = (long long)B * H * T * A.nkb;
        hipLaunchKernelGGL(keep_bits_kernel, dim3((unsigned)((words + THREADS - 1) / THREADS)), dim3(THREADS), 0, s,
                           static_cast<uint64_t *>(keep_bits), (long long)B * H * T, A.nkb, A.seed, A.thr,
                           pika_internal_dropout_salt());
        A.bits = static_cast<const uint64_t *>(keep_bits);
    }
    if (two_term) {
        if ((lo_off & 7) || (olo_off & 7)) return PIKA_EINVAL;
        const size_t lds = (size_t)4 * TILE * (D + 8) * sizeof(__bf16);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fwd2_kernel<128>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE * 136 * 2);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        if (D == 64) hipLaunchKernelGGL(attn_fwd2_kernel<64>, grid, dim3(THREADS), lds, s, A);
        else hipLaunchKernelGGL(attn_fwd2_kernel<128>, grid, dim3(THREADS), lds, s, A);
    } else if (io_dtype == PIKA_F32) {
        if (D == 64) launch_fwd<64, float>(A, grid, s); else launch_fwd<128, float>(A, grid, s);
    } else {
        if (D == 64) launch_fwd<64, __bf16>(A, grid, s); else launch_fwd<128, __bf16>(A, grid, s);
    }
    return (int)hipGetLastError();
}

int pika_attention_infer_f16x2(const void *q, const void *k, const void *v, long long lo_off, float *out, const void *mask,
                               int B, int T, int H, int D, long long ld, long long ldo, void *stream) {
    if (!q || !k || !v || !out || B <= 0 || T <= 0 || H <= 0 || (D != 64 && D != 128)) return PIKA_EINVAL;
    if ((ld & 7) || (ldo & 3) || (lo_off & 7) || ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
                                                   reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(out)) & 15) ||
        (mask && (reinterpret_cast<uintptr_t>(mask) & 7)))
        return PIKA_EINVAL;
    if (B > 65535 || H > 65535 || (long long)B * H * T > 0x7fffffffLL) return PIKA_ETOOBIG;
    Args A{};
    A.q = q; A.k = k; A.v = v; A.o = out; A.T = T; A.H = H; A.ld = ld; A.ldo = ldo;
    A.qscale = 1.4426950408889634f / sqrtf((float)D);
    A.mask = static_cast<const uint64_t *>(mask); A.lo_off = lo_off; A.nkb = (T + 63) / 64;
    const dim3 grid((T + TILE - 1) / TILE, H, B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)4 * TILE * (D + 8) * sizeof(__bf16);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_infer_f16x2_kernel<128>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE * 136 * 2);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (D == 64) hipLaunchKernelGGL(attn_infer_f16x2_kernel<64>, grid, dim3(THREADS), lds, s, A);
    else hipLaunchKernelGGL(attn_infer_f16x2_kernel<128>, grid, dim3(THREADS), lds, s, A);
    return (int)hipGetLastError();
}

int pika_attention_fwd(const void *q, const void *k, const void *v, void *out, int io_dtype, float *lse,
                       void *keep_bits, const void *mask, int B, int T, int H, int D, long long ld, long long ldo,
                       float p_drop, unsigned seed, void *stream) {
    if (io_dtype != PIKA_F32 && io_dtype != PIKA_BF16) return PIKA_EINVAL;
    return attention_fwd_impl(q, k, v, out, io_dtype, lse, keep_bits, static_cast<const uint64_t *>(mask), 0, 0, B, T, H,
                              D, ld, ldo, p_drop, seed, stream);
}

int pika_attention_fwd_two_term(const void *q, const void *k, const void *v, long long lo_off, void *out,
                                long long out_lo_off, float *lse, void *keep_bits, const void *mask, int B, int T,
                                int H, int D, long long ld, long long ldo, float p_drop, unsigned seed, void *stream) {
    return attention_fwd_impl(q, k, v, out, -2, lse, keep_bits, static_cast<const uint64_t *>(mask), lo_off, out_lo_off,
                              B, T, H, D, ld, ldo, p_drop, seed, stream);
}

int pika_attention_bwd(const void *q, const void *k, const void *v, const void *out, const void *dout,
                       int io_dtype, const float *lse, const void *keep_bits, const void *mask, float *delta, void *dq,
                       void *dk, void *dv, int B, int T, int H, int D, long long ld, long long ldo,
                       float p_drop, unsigned seed, void *stream) {
    if (io_dtype != PIKA_F32 && io_dtype != PIKA_BF16) return PIKA_EINVAL;
    const int g = io_dtype == PIKA_F32 ? 4 : 8;
    if (!args_ok(q, k, v, out, B, T, H, D, ld, p_drop, g) || !args_ok(dout, dq, dk, dv, B, T, H, D, ld, p_drop, g) ||
        !lse || !delta || ldo < (long long)H * D || (ldo & (g - 1)))
        return PIKA_EINVAL;
    if (B > 65535 || H > 65535 || (long long)B * H * T > 0x7fffffffLL) return PIKA_ETOOBIG;
    Args A{};
    A.q = q; A.k = k; A.v = v; A.out = out; A.dout = dout; A.lse = lse; A.delta = delta;
    A.dq = dq; A.dk = dk; A.dv = dv; A.T = T; A.H = H; A.ld = ld; A.ldo = ldo;
    A.qscale = 1.4426950408889634f / sqrtf((float)D);
    A.mask = static_cast<const uint64_t *>(mask);
    dropout_consts(A, p_drop, seed);
    A.nkb = (T + 63) / 64;
    if (A.thr) {
        if (!keep_bits || (reinterpret_cast<uintptr_t>(keep_bits) & 7)) return PIKA_EINVAL;
        A.bits = static_cast<const uint64_t *>(keep_bits);
    }
    const dim3 grid((T + TILE - 1) / TILE, H, B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (io_dtype == PIKA_F32) {
        if (D == 64) launch_bwd<64, float>(A, grid, B * T, delta, s); else launch_bwd<128, float>(A, grid, B * T, delta, s);
    } else {
        if (D == 64) launch_bwd<64, __bf16>(A, grid, B * T, delta, s); else launch_bwd<128, __bf16>(A, grid, B * T, delta, s);
    }
    return (int)hipGetLastError();
}

int pika_attention_mask_bits(const unsigned char *mask, int B, int T, void *bits, void *stream) {
    if (!mask || !bits || B <= 0 || T <= 0 || (reinterpret_cast<uintptr_t>(bits) & 7)) return PIKA_EINVAL;
    const int nkb = (T + 63) / 64;
    const long long words = (long long)B * T * nkb;
    hipLaunchKernelGGL(mask_bits_kernel, dim3((unsigned)((words + THREADS - 1) / THREADS)), dim3(THREADS), 0,
                       static_cast<hipStream_t>(stream), mask, static_cast<uint64_t *>(bits), (long long)B * T, T, nkb);
    return (int)hipGetLastError();
}

int pika_attention_keep_mask(unsigned char *mask, int BH, int T, float p_drop, unsigned seed,
                             void *stream) {
    if (!mask || BH <= 0 || T <= 0 || !(p_drop >= 0.f && p_drop < 1.f)) return PIKA_EINVAL;
    if ((long long)BH * T > 0x7fffffffLL) return PIKA_ETOOBIG;
    Args A{};
    dropout_consts(A, p_drop, seed);
    hipLaunchKernelGGL(keep_mask_kernel, dim3((unsigned)(BH * T)), dim3(THREADS), 0,
                       static_cast<hipStream_t>(stream), mask, (long long)BH * T, T, A.seed, A.thr,
                       pika_internal_dropout_salt());
    return (int)hipGetLastError();
}

}  // extern "C"
