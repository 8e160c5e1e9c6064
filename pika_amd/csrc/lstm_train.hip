// pika_amd/csrc/lstm_train.hip -- the recurrence of a unidirectional LSTM layer in TRAINING (forward with everything the
// backward needs, and the backward through time) as ONE persistent launch each for gfx950 (include/pika_lstm.h;
// reference trainer/model/transducer.py:55-61,93-96: the prediction network of every shipped recipe, nn.LSTM over the
// padded (B, U+1) label matrix, zero initial state).
//
// A library LSTM runs the chain step by step: at B = 32, H = 1024, U + 1 = 51 and two layers that is 102 steps of
// {recurrent product, cell kernel} forward (24.7 + 4.6 us) and as many backward (8.3 + 4.8 us): 4.3 ms of a 47 ms
// training step (profiles/r6_lstm_*).  Here the recurrent weights never move, as in blstm.hip: a workgroup (8 waves) owns
// 16 hidden units (all four gates: 64 rows of W_hh) for 16 batch rows and keeps its 64 x H slice in registers as two
// bf16 terms in MFMA fragment order (128 VGPRs per lane at H = 1024).
//
//   forward, step t:  gates[rows, my 64] = gx_t + h_{t-1}[rows, :] . W_hh[my 64, :]^T   (reduction over H: every wave an
//     eighth of it, partial tiles summed through LDS) -> cell -> h_t[rows, my 16 units] published as ONE word per value
//     (both bf16 terms), read by the H/16 workgroups of the row block.
//   backward, step t: dh = dy_t + sum over the H/16 producers of their partial products (below) -> gate gradients of my
//     64 gate rows (local: the cell state gradient stays in a register) -> stored for the weight-gradient products ->
//     partial[rows, ALL H units] = dgates[rows, my 64] . W_hh[my 64, :]  (reduction over MY gate rows only: the same
//     weight slice as the forward, no transposed copy; 64 x fp32 partials per value are summed by the consumer in a
//     fixed order) published as fp32 words.
//
// Exchange protocol of blstm.hip: agent-scope stores / loads (the workgroups sit on different XCDs), a word is EMPTY
// or final, every step has its own slot (EMPTY-filled by a memset in front of the launch), no flags, no fences.  All
// workgroups must be resident: the host entry refuses grids larger than the CU count; a waiting wave gives up after ~2 s,
// raises the error word and poisons its outputs with NaN.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pika_lstm.h"
#include "pika_rnnt.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int RB = 16;      // batch rows per workgroup (one MFMA row tile)
constexpr int UB = 16;      // hidden units per workgroup (x 4 gates = 64 rows of W_hh)
constexpr int NW = 8;       // waves per workgroup
constexpr unsigned EMPTY = 0xffffffffu;

__device__ inline float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline float tanh_fast(float x) { return 1.0f - 2.0f * rcp(__expf(2.0f * x) + 1.0f); }
__device__ inline float sigmoid_fast(float x) { return rcp(1.0f + __expf(-x)); }

__device__ inline unsigned pack_terms(float h) {
    const __bf16 hi = (__bf16)h;
    const __bf16 lo = (__bf16)(h - (float)hi);
    const unsigned w = ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16) | __builtin_bit_cast(unsigned short, lo);
    return w == EMPTY ? 0xfffffffeu : w;
}
__device__ inline unsigned word_of(float v) {
    const unsigned w = __builtin_bit_cast(unsigned, v);
    return w == EMPTY ? 0xfffffffeu : w;        // (a NaN either way)
}

// Both packings of W_hh (4H, H), two bf16 planes each (plane 0 = bf16(w), plane 1 = bf16(w - plane 0)):
//  fwd[g][wave][gate][kt][plane][lane] x 8: B fragment, column = unit g*16 + (lane & 15) of gate `gate`, reduction index
//      (wave*KTW + kt)*32 + (lane >> 4)*8 + e over the hidden units of h_{t-1}                      (KTW = H / 256)
//  bwd[g][wave][ct][ks][plane][lane] x 8: B fragment, column = hidden unit (wave*CTW + ct)*16 + (lane & 15), reduction
//      index ks*32 + (lane >> 4)*8 + e over MY gate rows, row k = gate*16 + u -> W_hh row gate*H + g*16 + u  (CTW = H / 128)
__global__ __launch_bounds__(256) void lstm_pack_kernel(const float *__restrict__ w, int H, bf16x8 *__restrict__ packed) {
    const int KTW = H >> 8, CTW = H >> 7, ng = H / UB;
    const long long per = (long long)ng * NW * 4 * KTW;           // fragments of the forward packing (= of the backward's)
    const long long frag = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frag >= 2 * per) return;
    const int lane = threadIdx.x & 63;
    const float *src;
    long long stride;       // between consecutive reduction indices
    if (frag < per) {
        long long f = frag;
        const int kt = (int)(f % KTW); f /= KTW;
        const int gate = (int)(f % 4); f /= 4;
        const int wave = (int)(f % NW);
        const int g = (int)(f / NW);
        src = w + ((long long)gate * H + g * UB + (lane & 15)) * H + (wave * KTW + kt) * 32 + (lane >> 4) * 8;
        stride = 1;
    } else {
        long long f = frag - per;
        const int ks = (int)(f % 2); f /= 2;
        const int ct = (int)(f % CTW); f /= CTW;
        const int wave = (int)(f % NW);
        const int g = (int)(f / NW);
        const int k0 = ks * 32 + (lane >> 4) * 8;                 // 8 consecutive k: one gate, units (k0 & 15) .. + 7
        src = w + ((long long)(k0 >> 4) * H + g * UB + (k0 & 15)) * H + (wave * CTW + ct) * 16 + (lane & 15);
        stride = H;
    }
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = src[e * stride];
        hi[e] = (__bf16)v;
        lo[e] = (__bf16)(v - (float)hi[e]);
    }
    packed[(frag * 2 + 0) * 64 + lane] = hi;
    packed[(frag * 2 + 1) * 64 + lane] = lo;
}

struct LT {
    const float *gx;        // fwd: (B, S, 4H) input projections + both biases        bwd: dy (B, S, H)
    const bf16x8 *w;        // the packing this kernel reads
    float *out;             // fwd: h (B, S, H)                                        bwd: dgates (B, S, 4H)
    float *gates;           // (B, S, 4H) activated gates [i | f | g | o]   (fwd writes, bwd reads)
    float *cells;           // (B, S, H) cell states                        (fwd writes, bwd reads)
    unsigned *xbuf;
    int *err;
    int S, B, H, nq, ng;
};

// 16-byte loads at agent scope (`sc1`: past this XCD's L2 -- what __hip_atomic_load compiles to, which exists for 4 and
// 8 bytes only).  The compiler does not count these loads: wait_loads names the registers they fill.
__device__ inline void load16_agent(u32x4 &d, const unsigned *p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(d) : "v"(p) : "memory");
}
template <int N>
__device__ inline void wait_loads(u32x4 (&v)[N]) {
    static_assert(N == 2 || N == 4 || N == 6 || N == 8, "registers named one by one");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1])::"memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
    if constexpr (N == 6)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])::"memory");
    if constexpr (N == 8)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
}
// Gives up after ~2 s of polling: a peer is not running.  Returns true when the wait is over for good.
__device__ inline bool timed_out(unsigned spin, unsigned long long &t0, int *err) {
    if ((spin & 63) != 63) return false;
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
    const unsigned long long now = wall_clock64();      // 100 MHz
    if (!t0) t0 = now;
    else if (now - t0 > 200000000ull) {
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    return false;
}

// Polls `n` words per lane (addresses base + i * stride) until none is EMPTY.  Returns false after the timeout.
template <int N, typename F>
__device__ inline bool poll_words(unsigned (&v)[N], int *err, F addr) {
    unsigned long long t0 = 0;
    for (unsigned spin = 0;; ++spin) {
        bool empty = false;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            v[i] = __hip_atomic_load(addr(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            empty |= v[i] == EMPTY;
        }
        if (!__any(empty)) return true;
        if ((spin & 63) == 63) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            const unsigned long long now = wall_clock64();      // 100 MHz
            if (!t0) t0 = now;
            else if (now - t0 > 200000000ull) {                 // 2 s: a peer is not running -- give up
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

template <int KTW>
__global__ __launch_bounds__(64 * NW) void lstm_fwd_kernel(LT p) {
    __shared__ float part[NW][4][RB * UB];          // [wave (reduction eighth)][gate][row][unit]   (32 KB)
    __shared__ int dead;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x % p.nq, g = blockIdx.x / p.nq;       // the row blocks of a unit group on neighbouring XCDs
    const int H = p.H, B = p.B, S = p.S;
    unsigned *xb = p.xbuf + (long long)q * S * RB * H;            // [q][step][H / 8][row][8]
    if (tid == 0) dead = 0;

    bf16x8 wr[4][KTW][2];
    {
        const bf16x8 *wp = p.w + (((long long)g * NW + wave) * 4 * KTW * 2) * 64 + lane;
#pragma unroll
        for (int gate = 0; gate < 4; ++gate)
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) wr[gate][kt][pl] = wp[((gate * KTW + kt) * 2 + pl) * 64];
    }
    const bool cell = tid < RB * UB;                  // thread -> (row r of the block, unit u of the block)
    const int r = (tid >> 4) & 15, u = tid & 15;
    const int row = q * RB + r, unit = g * UB + u;
    const bool live = cell && row < B;
    float c = 0.f, h = 0.f;
    __syncthreads();

    for (int s = 0; s < S; ++s) {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (live) {         // independent of the recurrence: requested first
            const float *gp = p.gx + ((long long)row * S + s) * 4 * H + unit;
#pragma unroll
            for (int gate = 0; gate < 4; ++gate) z[gate] = gp[(long long)gate * H];
        }
        f32x4 acc[4];
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) acc[gate] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            // the 16 x (KTW*32) words of step s-1 this wave reduces over, in FRAGMENT order: [group of 8 units][row][8]
            // -- lane (row lane & 15, quarter lane >> 4 of a k-tile) finds its 8 words of a k-tile in 32 contiguous bytes and
            // the wave reads 2 KB in one piece per k-tile
            const unsigned *xs = xb + (long long)(s - 1) * RB * H + (((wave * KTW) * 4 + (lane >> 4)) * RB + (lane & 15)) * 8;
            u32x4 wq[KTW * 2];
            unsigned long long t0 = 0;
            for (unsigned spin = 0;; ++spin) {
#pragma unroll
                for (int kt = 0; kt < KTW; ++kt) {
                    load16_agent(wq[2 * kt], xs + kt * 4 * RB * 8);
                    load16_agent(wq[2 * kt + 1], xs + kt * 4 * RB * 8 + 4);
                }
                wait_loads(wq);
                bool empty = false;
#pragma unroll
                for (int i = 0; i < KTW * 2; ++i)
                    empty |= wq[i][0] == EMPTY || wq[i][1] == EMPTY || wq[i][2] == EMPTY || wq[i][3] == EMPTY;
                if (!__any(empty)) break;
                if (timed_out(spin, t0, p.err)) { dead = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            unsigned wd[KTW * 8];
#pragma unroll
            for (int i = 0; i < KTW * 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) wd[i * 4 + e] = wq[i][e];
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) {
                u32x4 hb, lb;       // bf16 pairs: element 2i in the low half
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned w0 = wd[kt * 8 + 2 * j], w1 = wd[kt * 8 + 2 * j + 1];
                    hb[j] = (w0 >> 16) | (w1 & 0xffff0000u);
                    lb[j] = (w0 & 0xffffu) | (w1 << 16);
                }
                const bf16x8 ah = __builtin_bit_cast(bf16x8, hb), al = __builtin_bit_cast(bf16x8, lb);
#pragma unroll
                for (int gate = 0; gate < 4; ++gate) {
                    acc[gate] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wr[gate][kt][1], acc[gate], 0, 0, 0);
                    acc[gate] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wr[gate][kt][0], acc[gate], 0, 0, 0);
                    acc[gate] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wr[gate][kt][0], acc[gate], 0, 0, 0);
                }
            }
        }
        // lane holds rows (lane >> 4)*4 + e of column lane & 15
#pragma unroll
        for (int gate = 0; gate < 4; ++gate)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[wave][gate][((lane >> 4) * 4 + e) * UB + (lane & 15)] = acc[gate][e];
        __syncthreads();
        if (dead) break;
        if (cell) {
#pragma unroll
            for (int gate = 0; gate < 4; ++gate)
#pragma unroll
                for (int w = 0; w < NW; ++w) z[gate] += part[w][gate][tid];
            const float gi = sigmoid_fast(z[0]), gf = sigmoid_fast(z[1]), gg = tanh_fast(z[2]), go = sigmoid_fast(z[3]);
            c = gf * c + gi * gg;
            h = go * tanh_fast(c);
            if (s + 1 < S)
                __hip_atomic_store(xb + (long long)s * RB * H + ((unit >> 3) * RB + r) * 8 + (unit & 7), pack_terms(h),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (live) {
                const long long o = (long long)row * S + s;
                p.out[o * H + unit] = h;
                p.cells[o * H + unit] = c;
                float *gs = p.gates + o * 4 * H + unit;
                gs[0] = gi; gs[(long long)H] = gf; gs[2LL * H] = gg; gs[3LL * H] = go;
            }
        }
        __syncthreads();        // `part` is rewritten by the next step
    }
    if (dead && live)       // a peer never arrived: nothing of this launch may look like a result
        for (int s = 0; s < S; ++s) p.out[((long long)row * S + s) * H + unit] = __builtin_nanf("");
}

template <int KTW>
__global__ __launch_bounds__(64 * NW) void lstm_bwd_kernel(LT p) {
    constexpr int CTW = 2 * KTW;            // column tiles (16 hidden units) of a wave: H / 16 / 8
    constexpr int NGH = 8 * KTW;            // producers per half: H / 16 / 2
    constexpr int LDA = 64 + 8;             // bf16 elements per staged row (16-byte aligned, no 2-way conflict on the reads)
    __shared__ __attribute__((aligned(16))) __bf16 stage[2][RB][LDA];      // gate gradients of my 64 gate rows, two planes
    __shared__ float half_sum[RB * UB];
    __shared__ int dead;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x % p.nq, g = blockIdx.x / p.nq;
    const int H = p.H, B = p.B, S = p.S, ng = p.ng;
    if (tid == 0) dead = 0;

    bf16x8 wr[CTW][2][2];
    {
        const bf16x8 *wp = p.w + (((long long)g * NW + wave) * CTW * 2 * 2) * 64 + lane;
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) wr[ct][ks][pl] = wp[((ct * 2 + ks) * 2 + pl) * 64];
    }
    // cell ownership: thread -> (unit u, row r) so that word u*16 + r of a producer's block is this thread's
    const int half = tid >> 8, ct_ = tid & 255;
    const int u = ct_ >> 4, r = ct_ & 15;
    const int row = q * RB + r, unit = g * UB + u;
    const bool live = row < B;
    float dc_next = 0.f;
    // slot t holds what the step-t gradients send to step t-1:  [t][q][consumer][producer][u][r] fp32 words
    const long long slot_words = (long long)p.nq * ng * ng * (RB * UB);
    __syncthreads();

    for (int t = S - 1; t >= 0; --t) {
        float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, ct = 0.f, cp = 0.f, dy = 0.f;
        if (half == 0 && live) {        // independent of the recurrence: requested first
            const long long o = (long long)row * S + t;
            const float *gs = p.gates + o * 4 * H + unit;
            gi = gs[0]; gf = gs[(long long)H]; gg = gs[2LL * H]; go = gs[3LL * H];
            ct = p.cells[o * H + unit];
            if (t > 0) cp = p.cells[(o - 1) * H + unit];
            dy = p.gx[o * H + unit];
        }
        float ps = 0.f;
        if (t + 1 < S) {
            const unsigned *src = p.xbuf + (long long)(t + 1) * slot_words + (((long long)q * ng + g) * ng + half * NGH) * (RB * UB) + ct_;
            unsigned v[NGH];
            if (!poll_words(v, p.err, [&](int i) { return src + i * (RB * UB); })) dead = 1;
#pragma unroll
            for (int i = 0; i < NGH; ++i) ps += __builtin_bit_cast(float, v[i]);
            // consumed: every word has this one reader -- left EMPTY for the next launch on this scratch (no memset of the
            // S x 8 MB of slots in front of every launch: include/pika_lstm.h `armed`)
#pragma unroll
            for (int i = 0; i < NGH; ++i)
                __hip_atomic_store(const_cast<unsigned *>(src) + i * (RB * UB), EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (half == 1) half_sum[ct_] = ps;
        __syncthreads();
        if (dead) break;
        if (half == 0) {
            const float dh = dy + (ps + half_sum[ct_]);
            const float tc = tanh_fast(ct);
            const float dct = dh * go * (1.f - tc * tc) + dc_next;
            const float dz[4] = {dct * gg * gi * (1.f - gi), dct * cp * gf * (1.f - gf), dct * gi * (1.f - gg * gg),
                                 dh * tc * go * (1.f - go)};
            dc_next = dct * gf;
            if (live) {
                float *d = p.out + ((long long)row * S + t) * 4 * H + unit;
#pragma unroll
                for (int gate = 0; gate < 4; ++gate) d[(long long)gate * H] = dz[gate];
            }
#pragma unroll
            for (int gate = 0; gate < 4; ++gate) {
                const __bf16 hi = (__bf16)dz[gate];
                stage[0][r][gate * 16 + u] = hi;
                stage[1][r][gate * 16 + u] = (__bf16)(dz[gate] - (float)hi);
            }
        }
        __syncthreads();
        if (t > 0) {
            f32x4 acc[CTW];
#pragma unroll
            for (int c_ = 0; c_ < CTW; ++c_) acc[c_] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(&stage[0][lane & 15][ks * 32 + (lane >> 4) * 8]);
                const bf16x8 al = *reinterpret_cast<const bf16x8 *>(&stage[1][lane & 15][ks * 32 + (lane >> 4) * 8]);
#pragma unroll
                for (int c_ = 0; c_ < CTW; ++c_) {
                    acc[c_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wr[c_][ks][1], acc[c_], 0, 0, 0);
                    acc[c_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wr[c_][ks][0], acc[c_], 0, 0, 0);
                    acc[c_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wr[c_][ks][0], acc[c_], 0, 0, 0);
                }
            }
            // lane holds rows (lane >> 4)*4 + e of hidden unit (wave*CTW + c_)*16 + (lane & 15): the consumer of that unit
            // group finds them as words [u = lane & 15][r = (lane >> 4)*4 + e] of MY block in its slot
            unsigned *dst = p.xbuf + (long long)t * slot_words + (((long long)q * ng + wave * CTW) * ng + g) * (RB * UB)
                            + (lane & 15) * RB + (lane >> 4) * 4;
#pragma unroll
            for (int c_ = 0; c_ < CTW; ++c_) {
                // (two 8-byte stores the compiler emits itself: a hand-written global_store_dwordx4 sc1 here produced wrong
                // partial sums at H >= 512 -- the hazard recogniser does not look into inline assembly)
                unsigned long long *d64 = reinterpret_cast<unsigned long long *>(dst + (long long)c_ * ng * (RB * UB));
                const unsigned long long lo = (unsigned long long)word_of(acc[c_][0]) | ((unsigned long long)word_of(acc[c_][1]) << 32);
                const unsigned long long hi = (unsigned long long)word_of(acc[c_][2]) | ((unsigned long long)word_of(acc[c_][3]) << 32);
                __hip_atomic_store(d64, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(d64 + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (dead && half == 0 && live)
        for (int t = 0; t < S; ++t) p.out[((long long)row * S + t) * 4 * H + unit] = __builtin_nanf("");
}

inline int geometry(int B, int H, int *nq, int *ng) {
    if (B <= 0 || H <= 0) return PIKA_EINVAL;
    if ((H & 255) || H > 1024) return PIKA_ETOOBIG;          // 128 registers of weights per lane at H = 1024
    *nq = (B + RB - 1) / RB;
    *ng = H / UB;
    return PIKA_OK;
}

inline long long fwd_words(int S, int nq, int H) { return (long long)nq * S * RB * H; }
inline long long bwd_words(int S, int nq, int ng) { return (long long)S * nq * ng * ng * (RB * UB); }     // (slot 0 stays unused)

int resident(int grid) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return PIKA_EINVAL;
    return grid > cus ? PIKA_ETOOBIG : PIKA_OK;
}

}  // namespace

extern "C" {

long long pika_lstm_train_packed_bytes(int H) {
    int nq, ng;
    if (geometry(1, H, &nq, &ng) != PIKA_OK) return -1;
    return 2LL * 4 * H * H * 2 * 2;          // two packings x two bf16 planes
}

long long pika_lstm_train_fwd_work_bytes(int S, int B, int H) {
    int nq, ng;
    if (geometry(B, H, &nq, &ng) != PIKA_OK || S <= 0) return -1;
    return 256 + 4 * fwd_words(S, nq, H);
}

long long pika_lstm_train_bwd_work_bytes(int S, int B, int H) {
    int nq, ng;
    if (geometry(B, H, &nq, &ng) != PIKA_OK || S <= 0) return -1;
    return 256 + 4 * bwd_words(S, nq, ng);
}

int pika_lstm_train_pack(const float *w_hh, int H, void *packed, void *stream) {
    int nq, ng;
    const int rc = geometry(1, H, &nq, &ng);
    if (rc != PIKA_OK) return rc;
    if (!w_hh || !packed || (reinterpret_cast<uintptr_t>(packed) & 15)) return PIKA_EINVAL;
    const long long nfrag = 2LL * ng * NW * 4 * (H >> 8);
    hipLaunchKernelGGL(lstm_pack_kernel, dim3((unsigned)((nfrag + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       w_hh, H, static_cast<bf16x8 *>(packed));
    return (int)hipGetLastError();
}

static int run(bool bwd, const float *in, const void *packed, float *out, float *gates, float *cells, void *work,
               long long work_bytes, int armed, int S, int B, int H, void *stream) {
    int nq, ng;
    int rc = geometry(B, H, &nq, &ng);
    if (rc != PIKA_OK) return rc;
    if (!in || !packed || !out || !gates || !cells || !work || S <= 0) return PIKA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(work)) & 15) return PIKA_EINVAL;
    if (work_bytes < (bwd ? pika_lstm_train_bwd_work_bytes(S, B, H) : pika_lstm_train_fwd_work_bytes(S, B, H))) return PIKA_EINVAL;
    const int grid = nq * ng;
    if ((rc = resident(grid)) != PIKA_OK) return rc;          // every workgroup waits for others inside the kernel
    hipStream_t st = static_cast<hipStream_t>(stream);
    LT p;
    p.gx = in; p.out = out; p.gates = gates; p.cells = cells;
    const long long half = 4LL * H * H * 2 * 2 / 16;          // bf16x8 elements of one packing
    p.w = static_cast<const bf16x8 *>(packed) + (bwd ? half : 0);
    p.err = static_cast<int *>(work);
    p.xbuf = reinterpret_cast<unsigned *>(static_cast<char *>(work) + 256);
    p.S = S; p.B = B; p.H = H; p.nq = nq; p.ng = ng;
    hipError_t e = hipMemsetAsync(work, 0, 256, st);
    if (e == hipSuccess && S > 1 && !armed)
        e = hipMemsetAsync(p.xbuf, 0xff, (size_t)(bwd ? bwd_words(S, nq, ng) : fwd_words(S, nq, H)) * 4, st);
    if (e != hipSuccess) return (int)e;
#define PIKA_LSTM_LAUNCH(K)                                                                          \
    switch (H >> 8) {                                                                                \
    case 1: hipLaunchKernelGGL(K<1>, dim3(grid), dim3(64 * NW), 0, st, p); break;                    \
    case 2: hipLaunchKernelGGL(K<2>, dim3(grid), dim3(64 * NW), 0, st, p); break;                    \
    case 3: hipLaunchKernelGGL(K<3>, dim3(grid), dim3(64 * NW), 0, st, p); break;                    \
    default: hipLaunchKernelGGL(K<4>, dim3(grid), dim3(64 * NW), 0, st, p); break;                   \
    }
    if (bwd) { PIKA_LSTM_LAUNCH(lstm_bwd_kernel) } else { PIKA_LSTM_LAUNCH(lstm_fwd_kernel) }
#undef PIKA_LSTM_LAUNCH
    return (int)hipGetLastError();
}

int pika_lstm_train_fwd(const float *gx, const void *packed, float *out, float *gates, float *cells, void *work,
                        long long work_bytes, int S, int B, int H, void *stream) {
    return run(false, gx, packed, out, gates, cells, work, work_bytes, 0, S, B, H, stream);
}

int pika_lstm_train_bwd(const float *dy, const void *packed, const float *gates, const float *cells, float *dgates,
                        void *work, long long work_bytes, int armed, int S, int B, int H, void *stream) {
    return run(true, dy, packed, dgates, const_cast<float *>(gates), const_cast<float *>(cells), work, work_bytes, armed, S,
               B, H, stream);
}

int pika_lstm_train_status(const void *work, int *host_out, void *stream) {
    if (!work || !host_out) return PIKA_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(host_out, work, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return (int)e;
}

}  // extern "C"
