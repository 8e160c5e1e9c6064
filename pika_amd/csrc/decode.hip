// pika_amd/csrc/decode.hip -- fused beam-search step for gfx950 (include/pika_decode.h).
//
// One 256-thread workgroup per utterance.  Each of the 4 waves stages beam rows (V scaled logits)
// in a wave-private 20 KiB LDS slab: row max / log-sum-exp by wave reductions, then the row's K
// best entries by K rounds of "lane-local arg-max over the slab + wave arg-max" (ties -> lowest
// index).  The K*K row winners are merged by a rank sort; the bookkeeping of the reference's
// `advance` (finish rule, hypotheses, finished list, history) runs on K lanes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pika_decode.h"
#include "pika_decode_step.h"
#include "pika_rnnt.h"

namespace {

constexpr float DEAD = -1e20f;
constexpr long long EOS = -1;
constexpr int MAXV = 8192, MAXK = 64, WAVES = 4;      // (the shipped recipes' vocabulary is 6268)

struct Cand { float v; int idx; };

// order-preserving map float -> unsigned (a > b <=> fkey_h(a) > fkey_h(b))
__device__ inline unsigned fkey_h(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ inline bool better(float va, int ia, float vb, int ib) {
    return va > vb || (va == vb && ia < ib);
}

// Workgroup barrier for hand-offs through LDS only: __syncthreads() also waits for the wave's outstanding global STORES
// (vmcnt counts them on gfx9), a ~2 us write round trip in kernels that publish results as they go.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- kernel A: one wavefront per beam row -> the row's K best candidates -----------------------
__global__ __launch_bounds__(WAVES * 64) void beam_row_topk_kernel(
    const float *__restrict__ logits, float sm_scale, int first, const float *__restrict__ scores,
    const float *__restrict__ lm_scores, float lm_scale, const long long *__restrict__ y,
    const long long *__restrict__ hyp, const long long *__restrict__ hyp_len, int L, int B, int K,
    int V, int beam_prune, Cand *__restrict__ cand) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rowid = blockIdx.x * WAVES + wave;
    if (rowid >= B * K) return;
    float *x = reinterpret_cast<float *>(smem) + wave * ((V + 3) & ~3);      // (a row of logits per wave)
    const int b = rowid / K, k = rowid - b * K;
    const long long bk = (long long)b * K;
    Cand *out = cand + (long long)rowid * K;

    // disabled row: eos, or a duplicate of an earlier live slot's partial hypothesis (:100-114)
    bool d;
    if (first) {
        d = k != 0;
    } else {
        d = y[bk + k] == EOS;
        const long long len = hyp_len[bk + k];
        if (!d && beam_prune && len > 0) {
            for (int j = 0; j < k && !d; ++j) {
                if (y[bk + j] == EOS || hyp_len[bk + j] != len) continue;
                bool same = true;
                for (long long p = lane; p < len; p += 64)
                    same &= hyp[(bk + j) * L + p] == hyp[(bk + k) * L + p];
                d = __all(same);
            }
        }
    }
    if (d) {  // the whole row is -1e20: its K lowest indices stand in (ties -> index order)
        if (lane < K) out[lane] = Cand{first ? -3.0e38f : DEAD, k * V + lane};
        return;
    }
    const float *row = logits + (bk + k) * (long long)V;
    float m = -INFINITY;
    if ((V & 3) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
        // 16-byte loads, 8 in flight per lane: a 5000-float row is 3 rounds instead of 10
        typedef float tk_f4 __attribute__((ext_vector_type(4)));
        const int V4 = V >> 2;
        for (int v0 = lane; v0 < V4; v0 += 64 * 8) {
            tk_f4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (v0 + 64 * j < V4) t[j] = reinterpret_cast<const tk_f4 *>(row)[v0 + 64 * j];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (v0 + 64 * j < V4) {
                    const tk_f4 u = t[j] * sm_scale;
                    reinterpret_cast<tk_f4 *>(x)[v0 + 64 * j] = u;
                    m = fmaxf(m, fmaxf(fmaxf(u.x, u.y), fmaxf(u.z, u.w)));
                }
        }
    } else {
        for (int v0 = lane; v0 < V; v0 += 64 * 8) {   // 8 independent loads in flight per lane
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = v0 + 64 * j < V ? row[v0 + 64 * j] : -INFINITY;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (v0 + 64 * j < V) { x[v0 + 64 * j] = sm_scale * t[j]; m = fmaxf(m, sm_scale * t[j]); }
        }
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(x[v] - m);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float logsum = logf(s);
    const float add_s = scores[bk + k], add_l = lm_scale * lm_scores[bk + k];
    // Lane-local two best entries cached in registers; a lane re-scans its slice only when both
    // were consumed, so a round costs one wave arg-max instead of a V/64-element scan.
    float a1 = -INFINITY, a2 = -INFINITY;
    int i1 = 0x7fffffff, i2 = 0x7fffffff;
    auto rescan = [&]() {
        a1 = a2 = -INFINITY; i1 = i2 = 0x7fffffff;
        for (int v = lane; v < V; v += 64) {
            const float t = x[v];
            if (t > a1) { a2 = a1; i2 = i1; a1 = t; i1 = v; }
            else if (t > a2) { a2 = t; i2 = v; }
        }
    };
    rescan();
    for (int r = 0; r < K; ++r) {
        if (__any(i1 == 0x7fffffff)) {
            if (i1 == 0x7fffffff) rescan();   // consumed entries are -inf in the slab
        }
        float bv = a1;
        int bi = i1;
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (bi == i1 && bi != 0x7fffffff) {   // the owning lane pops its cache
            x[bi] = -INFINITY;
            a1 = a2; i1 = i2;
            a2 = -INFINITY; i2 = 0x7fffffff;
        }
        if (lane == 0) {
            float val = (bv - m) - logsum;                 // log_softmax, torch's operation order
            if (!first) val = (val + add_s) + add_l;       // (:94-97)
            out[r] = Cand{val, k * V + (bi == 0x7fffffff ? 0 : bi)};
        }
    }
}

#ifdef PIKA_ADV_TRACE     // profiling builds only (tools/adv_trace.py): time stamps of utterance 0's wave PIKA_ADV_TRACE, summed over launches
__device__ unsigned long long g_adv_trace[16];
__device__ inline unsigned long long *adv_slots() { __shared__ unsigned long long s_[16]; return s_; }
// stamps go to LDS and reach memory once, at the end (a global read-modify-write per stamp is a ~2 us round trip of its own)
#define ADV_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 64 * (PIKA_ADV_TRACE)) { adv_slots()[k] = __builtin_amdgcn_s_memrealtime(); \
    if ((k) == 9) { for (int i_ = 1; i_ < 16; ++i_) g_adv_trace[i_] += adv_slots()[i_] - adv_slots()[0]; g_adv_trace[0] += 1; } } } while (0)
extern "C" int pika_debug_adv_trace(unsigned long long *out16) {
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_adv_trace), sizeof(g_adv_trace));
}
#else
#define ADV_STAMP(k) do { } while (0)
#endif

// ---- kernel B: one workgroup per utterance: merge + the bookkeeping of `advance` -------------
struct BeamState {
    float *scores; const float *lm_scores; float lm_scale; long long *y; long long *t_idx;
    const long long *num_frames; const long long *max_len; long long *hyp; long long *hyp_len; int L;
    long long *ks_hist; long long *ys_hist; const long long *step_t; unsigned char *eos_top; float *fin_score;
    long long *fin_step; long long *fin_k; long long *fin_n; int fin_cap; long long *prev_k_out; long long *y_raw;
    int B, K, V, blk;
};

// Shared scratch of one utterance (static part; hyp_l [K][L] ints and cand [K][K] live in dynamic LDS)
struct BeamShared {
    float lm_old[MAXK], best_v[MAXK], new_score[MAXK];
    long long t_old[MAXK], len_old[MAXK], y_old[MAXK], new_len[MAXK];
    int best_i[MAXK], fin_flag[MAXK];
    int rank[MAXK * MAXK];
    // per-utterance scalars, read ONCE with the slot state (every later use would be a dependent ~2 us round trip of its own)
    long long nf, ml, fin_n, s_now;
    int eos;
};

// Everything in this kernel is a handful of dependent memory round trips (~2 us each at this occupancy), so loops
// over global memory are issued as batches of 8 independent loads per thread, never one load per iteration.
__device__ inline void beam_load_state(const BeamState &a, BeamShared &sh, int *hyp_l, int b) {
    const int tid = threadIdx.x;
    const long long bk = (long long)b * a.K;
    if (tid < a.K) {
        sh.lm_old[tid] = a.lm_scores[bk + tid];
        sh.t_old[tid] = a.t_idx[bk + tid];
        sh.len_old[tid] = a.hyp_len[bk + tid];
        sh.y_old[tid] = a.y[bk + tid];
    } else if (tid == 64) {
        sh.nf = a.num_frames[b];
        sh.ml = a.max_len[b];
        sh.fin_n = a.fin_n[b];
        sh.s_now = a.step_t[0];
        sh.eos = a.eos_top[b];
    }
    __syncthreads();
    long long ml = 0;
    for (int i = 0; i < a.K; ++i) ml = max(ml, sh.len_old[i]);
    const int used = (int)min((long long)a.L, ml + 1);      // labels beyond a slot's length are never read
    const int n = a.K * used;
    const int nth = blockDim.x;
    for (int base = tid; base < n; base += nth * 8) {
        long long v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = base + u * nth;
            if (e < n) v[u] = a.hyp[(bk + e / used) * a.L + e % used];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = base + u * nth;
            if (e < n) hyp_l[(e / used) * a.L + e % used] = (int)v[u];
        }
    }
}

// cand [K][K] in LDS holds the K best candidates of every row; everything of `advance` after that
// (beam_transducer.py:119-187) for utterance b.  Ends with every thread past its last global write of the step.
__device__ inline void beam_merge_and_book(const BeamState &a, BeamShared &sh, const int *hyp_l, const Cand *cand,
                                           int b) {
    const int tid = threadIdx.x, K = a.K, L = a.L, V = a.V, B = a.B;
    const long long bk = (long long)b * K;
    const int nc = K * K;
    // rank sort of the K*K row winners, value desc / flat index asc (:119-121).  The comparisons of a candidate are
    // split over blockDim / nc threads (partial ranks meet in LDS): one thread walking all K*K candidates was a chain of
    // 256 dependent LDS reads, the longest single item of this kernel.
    // Every candidate against every other: 64 x 64 blocks, one per wave and turn -- the wave keeps its 64 "others" in
    // registers and broadcasts them lane by lane through the scalar unit (v_readlane); partial ranks meet in LDS.
    // (Reading the others from LDS, 16 waves at once, was 4 us of LDS traffic for K = 16.)
    {
        const int lane = tid & 63, wave = tid >> 6, nwaves = (int)blockDim.x >> 6;
        const int chunks = (nc + 63) >> 6;
        for (int c = tid; c < nc; c += blockDim.x) sh.rank[c] = 0;
        lds_barrier();
        for (int task = wave; task < chunks * chunks; task += nwaves) {
            const int cm = task % chunks, co = task / chunks;
            const int ci = cm * 64 + lane, oi = co * 64 + lane;
            const Cand me = cand[ci < nc ? ci : nc - 1];
            const Cand ot = oi < nc ? cand[oi] : Cand{-INFINITY, 0x7fffffff};     // (padding: better than nothing)
            const int no = min(64, nc - co * 64);
            int rank = 0;
            for (int j = 0; j < no; ++j) {
                const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ot.v), j));
                const int ox = __builtin_amdgcn_readlane(ot.idx, j);
                rank += better(ov, ox, me.v, me.idx) ? 1 : 0;
            }
            if (rank && ci < nc) atomicAdd(&sh.rank[ci], rank);
        }
        lds_barrier();
        for (int c2 = tid; c2 < nc; c2 += blockDim.x)
            if (sh.rank[c2] < K) { sh.best_v[sh.rank[c2]] = cand[c2].v; sh.best_i[sh.rank[c2]] = cand[c2].idx; }
    }
    lds_barrier();
    ADV_STAMP(10);

    // ---- bookkeeping on K lanes (:125-187) ---------------------------------------------------
    const long long s = sh.s_now;                   // steps taken before this one
    const long long n_ys = s + 2;                   // len(next_ys) after the append
    if (tid < K) {
        const int id = sh.best_i[tid];
        const int pk = id / V, ysym = id - pk * V;
        const float ns = sh.best_v[tid] - a.lm_scale * sh.lm_old[pk];
        const bool fin = (ysym == a.blk && sh.t_old[pk] == sh.nf - 1) || (n_ys > sh.ml);
        sh.fin_flag[tid] = fin ? 1 : 0;
        sh.new_score[tid] = ns;
        sh.new_len[tid] = fin ? sh.len_old[tid] : sh.len_old[pk] + ((ysym != a.blk) ? 1 : 0);
        a.scores[bk + tid] = ns;
        a.prev_k_out[bk + tid] = pk;
        a.ks_hist[(s * B + b) * K + tid] = pk;
        const long long yn = fin ? EOS : (long long)ysym;
        if (a.y_raw) a.y_raw[bk + tid] = ysym;      // the symbol before the eos substitution (FST fusion)
        a.y[bk + tid] = yn;
        a.ys_hist[((s + 1) * B + b) * K + tid] = yn;
        a.t_idx[bk + tid] = sh.t_old[pk];           // transducer_decoder.py:201-202
        if (tid == 0 && yn == EOS) { a.eos_top[b] = 1; sh.eos = 1; }
        if (!fin) a.hyp_len[bk + tid] = sh.new_len[tid];
    }
    lds_barrier();
    ADV_STAMP(11);
    if (tid == 0) {                                  // finished list, slot order (:165-181)
        long long n = sh.fin_n;
        for (int i = 0; i < K; ++i) {
            if (!sh.fin_flag[i]) continue;
            const long long pos = n < a.fin_cap - 2 ? n : a.fin_cap - 2;
            a.fin_score[(long long)b * a.fin_cap + pos] = sh.new_score[i];
            a.fin_step[(long long)b * a.fin_cap + pos] = n_ys - 1;
            a.fin_k[(long long)b * a.fin_cap + pos] = i;
            ++n;
        }
        a.fin_n[b] = n;
        sh.fin_n = n;
    }
    ADV_STAMP(12);
    // partial hypotheses: slot i <- parent's labels (+ y), finished slots keep their own (:217-226)
    // (a wave per slot: the slots one after the other were K dependent LDS-read chains for the whole workgroup)
    const int lane = tid & 63, nwaves = (int)blockDim.x >> 6;
    for (int i = tid >> 6; i < K; i += nwaves) {
        if (sh.fin_flag[i]) continue;
        const int p = sh.best_i[i] / V, ys_ = sh.best_i[i] - p * V;
        const int plen = (int)sh.len_old[p];
        long long *dst = a.hyp + (bk + i) * L;
        const int ncopy = min(L, plen + 1);          // the parent's labels + the new one; the rest is never read
        for (int q = lane; q < ncopy; q += 64) {
            long long v = hyp_l[p * L + q];
            if (q == plen && ys_ != a.blk) v = ys_;
            dst[q] = v;
        }
    }
}

__global__ __launch_bounds__(256) void beam_merge_kernel(const Cand *__restrict__ cand_g, BeamState a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *hyp_l = reinterpret_cast<int *>(smem);                          // [K][L]
    Cand *cand = reinterpret_cast<Cand *>(hyp_l + a.K * a.L);            // [K][K]
    __shared__ BeamShared sh;
    const int b = blockIdx.x, tid = threadIdx.x;
    beam_load_state(a, sh, hyp_l, b);
    const int nc = a.K * a.K;
    for (int c = tid; c < nc; c += blockDim.x) cand[c] = cand_g[(long long)b * a.K * a.K + c];
    __syncthreads();
    beam_merge_and_book(a, sh, hyp_l, cand, b);
}

// The K best of a row's pool of n partial candidates (LDS) -> out[0..K), value = log-softmax (+ running scores):
// bisection on the order-preserving integer image of the values with ballots + popcounts only (no shuffle chains; see
// dfc2_topk_kernel), keys held in registers (PLM per lane, n <= 64*PLM).  Candidate indices are unique within a row,
// so ties at the threshold value are resolved by a second bisection on the index (lowest indices win).  Output order
// is arbitrary: the merge ranks the K*K row winners anyway.
template <int PLM>
__device__ inline void select_row(const Cand *pool, int n, int K, int lane, float m, float logsum, bool first,
                                  float add_s, float add_l, int kV, Cand *out) {
    unsigned key[PLM];
    int cidx[PLM];
#pragma unroll
    for (int q = 0; q < PLM; ++q) {
        const int e = lane + 64 * q;
        const Cand c = e < n ? pool[e] : Cand{-INFINITY, 0x7fffffff};
        key[q] = e < n ? fkey_h(c.v) : 0u;      // 0 is below every real key (fkey_h(-inf) = 0x007fffff)
        cidx[q] = c.idx;
    }
    unsigned T = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned mid = T | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < PLM; ++q) cnt += __popcll(__ballot(key[q] >= mid));
        if (cnt >= K) T = mid;
    }
    int n_gt = 0, n_eq = 0;
#pragma unroll
    for (int q = 0; q < PLM; ++q) {
        n_gt += __popcll(__ballot(key[q] > T));
        n_eq += __popcll(__ballot(key[q] == T));
    }
    const int need = K - n_gt;
    unsigned I = 0xffffffffu;
    if (n_eq > need) {                          // keep the `need` lowest indices among the ties
        unsigned lo = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned mid = lo | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < PLM; ++q) cnt += __popcll(__ballot(key[q] == T && (unsigned)cidx[q] < mid));
            if (cnt <= need) lo = mid;
        }
        I = lo;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    int base = 0;
#pragma unroll
    for (int q = 0; q < PLM; ++q) {
        const bool sel = key[q] > T || (key[q] == T && T != 0u && (unsigned)cidx[q] < I);
        const unsigned long long mk = __ballot(sel);
        const int rk = base + __popcll(mk & lt);
        if (sel && rk < K) {
            const float cv = pool[lane + 64 * q].v;
            float val = (cv - m) - logsum;                 // log_softmax, torch's operation order
            if (!first) val = (val + add_s) + add_l;       // (:94-97)
            out[rk] = Cand{val, kV + (cidx[q] == 0x7fffffff ? 0 : cidx[q])};
        }
        base += __popcll(mk);
    }
}

// ---- kernel B': the same, with the row log-softmax / top-K built from the row statistics + scaled logits of pika_dfc2_logits ----
// (include/pika_decode_step.h).  Phase 1, one wave per beam row: disabled rows as in kernel A; otherwise
// log-sum-exp = log sum_s psum_s exp(pmax_s - max), and the K best of the S sorted partial lists by K rounds of
// "wave arg-max over the S list heads".  Phase 2 = kernel B.  Phase 3: done flags, the all-done stop flag, the
// longest partial hypothesis and the step counter (last workgroup to arrive).
// Launched with min(16, K) waves per utterance: every beam row gets a wave of its own for phase 1 (the four rows a wave
// took in turn at 256 threads were four dependent chains of row-sized round trips: 70 us of a 600 us step).

// The partials are the row statistics only; the candidates come from the scaled logits themselves -- see row_survivors.
constexpr int POOL_CAP = 256;     // candidates a row's wave holds in LDS for the selection (select_row<4>)

// The row's candidates for its K best logits WITHOUT looking at most of them twice: the K-th largest of the row's split
// maxima (one per lane, bisected like select_row's keys) is a lower bound on the row's K-th largest logit, so ONE pass
// over the row keeps the few dozen values at or above it (ballot prefix -> dense LDS pool, scan order = index order).
// More than POOL_CAP survivors (rows of near-equal logits, tiny S): the bound is raised by bisection with counting passes
// until they fit; values tied at a bound that cannot be raised any more are taken lowest index first after everything
// above them -- what the selection would have done with them.  Returns the pool size.
__device__ inline int row_survivors(const float *__restrict__ row, int V, int K, int S, float split_max, int lane,
                                    Cand *pool) {
    constexpr int COLS = PIKA_DFC2_COLS, PL = COLS / 64, SB = 16;    // SB ranges per batch of requests
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned km = lane < S ? fkey_h(split_max) : 0u;
    unsigned lo = 0;                                            // keys >= lo: at least K of them
    if (S >= K) {
        // (any bound with at least K maxima at or above it will do: the first that has exactly K ends the bisection)
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned mid = lo | (1u << bit);
            const int cnt = __popcll(__ballot(km >= mid));
            if (cnt >= K) lo = mid;
            if (cnt == K) break;
        }
    }
    // Only the ranges whose maximum reaches the bound can hold a survivor (K of 27 as a rule): their columns in ascending
    // order, SB ranges' requests in flight together.
    // mode 0: collect key >= lo_; 1: count key >= lo_; 2: collect key > lo_; 3: append key == lo_ (from pool[c0])
    auto pass = [&](unsigned lo_, int mode, int c0) {
        int c = c0;
        unsigned long long todo = __ballot(lane < S && km >= lo_);
        while (todo) {
            float t[SB][PL];
            int sp[SB];
            unsigned long long td = todo;
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                sp[u] = td ? (int)__builtin_ctzll(td) : -1;
                td &= td - 1;
#pragma unroll
                for (int q = 0; q < PL; ++q) {
                    const int v = sp[u] * COLS + 64 * q + lane;
                    t[u][q] = (sp[u] >= 0 && v < V) ? row[v] : -INFINITY;
                }
            }
            todo = td;
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                if (sp[u] < 0) break;
#pragma unroll
                for (int q = 0; q < PL; ++q) {
                    const int v = sp[u] * COLS + 64 * q + lane;
                    const unsigned key = fkey_h(t[u][q]);
                    const bool sel = v < V && (mode == 2 ? key > lo_ : (mode == 3 ? key == lo_ : key >= lo_));
                    const unsigned long long mk = __ballot(sel);
                    const int pos = c + __popcll(mk & lt);
                    if (mode != 1 && sel && pos < POOL_CAP) pool[pos] = Cand{t[u][q], v};
                    c += __popcll(mk);
                }
            }
        }
        return c;
    };
    int c = pass(lo, 0, 0);
    if (c <= POOL_CAP) return c;
    unsigned hi = 0xffffffffu;                                  // keys >= hi: fewer than K (no key is all ones: NaN-free rows)
    while (hi - lo > 1u) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        const int cm = pass(mid, 1, 0);
        if (cm >= K) { lo = mid; c = cm; if (c <= POOL_CAP) break; } else hi = mid;
    }
    if (c <= POOL_CAP) return pass(lo, 0, 0);
    // more than POOL_CAP values tie at key lo and fewer than K lie above it
    const int above = pass(lo, 2, 0);
    const int all = pass(lo, 3, above);
    return all < POOL_CAP ? all : POOL_CAP;
}

__global__ __launch_bounds__(1024) void beam_partials_kernel(const float *__restrict__ pmax,
                                                            const float *__restrict__ psum,
                                                            int S, BeamState a,
                                                            int beam_prune, int n_best, int *__restrict__ stop,
                                                            long long *__restrict__ max_hyp, int *__restrict__ sync,
                                                            long long *__restrict__ step_rw,
                                                            const float *__restrict__ logits, long long ldl) {
    ADV_STAMP(0);
    if (*stop) {                                      // a replay after the search has ended: nothing happens, and the
        if (blockIdx.x == 0 && threadIdx.x == 0) {    // FST advance of this (skipped) step must not run either
            sync[4] = 1;
            // both compact-row counters: the prediction-network launches of the remaining replays then run on ZERO rows
            // (the LSTM cell kernel updates its state in place -- a stale count would advance the final states again)
            sync[5] = 0;
            sync[6] = 0;
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *hyp_l = reinterpret_cast<int *>(smem);                          // [K][L]
    Cand *cand = reinterpret_cast<Cand *>(hyp_l + a.K * a.L);            // [K][K]
    Cand *pool_all = cand + a.K * a.K;                                   // [waves][S*K]
    __shared__ BeamShared sh;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, L = a.L, V = a.V;
    const long long bk = (long long)b * K;
    ADV_STAMP(1);
    // the row statistics of this wave's (first) row: requested before the slot state, they do not depend on it
    float pm_first = -INFINITY, ps_first = 0.f;
    {
        const int k0 = threadIdx.x >> 6, l0 = threadIdx.x & 63;
        if (k0 < a.K && l0 < S) {
            pm_first = pmax[((long long)blockIdx.x * a.K + k0) * S + l0];
            ps_first = psum[((long long)blockIdx.x * a.K + k0) * S + l0];
        }
    }
    beam_load_state(a, sh, hyp_l, b);
    __syncthreads();
    ADV_STAMP(2);
    const long long s_now = sh.s_now;
    const int first = s_now == 0;
    const int nwaves = blockDim.x >> 6;
    for (int k = wave; k < K; k += nwaves) {
        bool d;
        if (first) {
            d = k != 0;
        } else {
            d = sh.y_old[k] == EOS;           // (from LDS: a global load per candidate slot here was a chain of
            const long long len = sh.len_old[k];   //  up to K dependent ~2 us round trips per row)
            if (!d && beam_prune && len > 0) {
                // lane j < k compares slot j with slot k, last label first (the hypotheses of a beam share their beginnings):
                // 15 slots one after the other, each a wave-wide compare + vote, were 6 us of the last row's wave
                bool same = lane < k && sh.y_old[lane] != EOS && sh.len_old[lane] == len;
                for (long long p = len - 1; p >= 0 && __any(same); --p)
                    if (same) same = hyp_l[lane * L + p] == hyp_l[k * L + p];
                d = __any(same);
            }
        }
        Cand *out = cand + k * K;
        if (d) {
            if (lane < K) out[lane] = Cand{first ? -3.0e38f : DEAD, k * V + lane};
            continue;
        }
        ADV_STAMP(3);
        const long long pi = (bk + k) * S;
        float m = k == wave ? pm_first : (lane < S ? pmax[pi + lane] : -INFINITY);
        const float mine = m;
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float s = lane < S ? (k == wave ? ps_first : psum[pi + lane]) * expf(mine - m) : 0.f;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float logsum = logf(s);
        const float add_s = a.scores[bk + k], add_l = a.lm_scale * sh.lm_old[k];
        // the row's few dozen candidates for its K best into this wave's LDS pool (row_survivors), then the selection
        ADV_STAMP(4);
        Cand *pool = pool_all + wave * POOL_CAP;
        const int n = row_survivors(logits + (bk + k) * ldl, V, K, S, mine, lane, pool);
        __builtin_amdgcn_wave_barrier();
        ADV_STAMP(5);
        if (n <= 64) select_row<1>(pool, n, K, lane, m, logsum, first, add_s, add_l, k * V, out);
        else if (n <= 128) select_row<2>(pool, n, K, lane, m, logsum, first, add_s, add_l, k * V, out);
        else select_row<4>(pool, n, K, lane, m, logsum, first, add_s, add_l, k * V, out);
        ADV_STAMP(6);
    }
    lds_barrier();
    ADV_STAMP(7);
    beam_merge_and_book(a, sh, hyp_l, cand, b);
    lds_barrier();
    ADV_STAMP(8);
    if (tid == 0) {
        long long mh = 0;
        for (int i = 0; i < K; ++i) mh = max(mh, sh.new_len[i]);
        atomicMax(reinterpret_cast<unsigned long long *>(max_hyp), (unsigned long long)mh);
        const int done = (sh.eos && sh.fin_n >= n_best) ? 1 : 0;   // beam_transducer.py:189-194
        const int par = (int)(s_now & 1);
        // ONE device-scope atomic carries both the arrival and the done count (high half): no fence, no second
        // atomic whose order would matter; the last arriver sees everybody's contribution in the value it gets back
        const unsigned old = atomicAdd(reinterpret_cast<unsigned *>(&sync[par * 2]), (unsigned)(done << 16) + 1u);
        if ((old & 0xffffu) == (unsigned)(a.B - 1)) {   // everybody has read step_t and finished its utterance
            const int ndone = (int)(old >> 16) + done;
            if (ndone == a.B) {
                atomicExch(stop, 1);
                // the launches of the remaining (skipped) steps address the counters by the parity they were captured
                // with: this step's counter must read zero too, or an in-place LSTM cell update would run again on the
                // rows of this step (final_state() would then return advanced states)
                sync[5 + par] = 0;
            }
            sync[(par ^ 1) * 2] = 0;
            sync[(par ^ 1) * 2 + 1] = 0;
            sync[5 + (par ^ 1)] = 0;                  // the compact-row counter the NEXT step's prep will fill
            step_rw[0] = s_now + 1;
        }
    }
    ADV_STAMP(9);
}

// ---------------------------------------------------------------------------------------------
// One new position of the conv-transformer prediction net's self-attention for every beam row
// (decoder/prednet_cache.py): the keys / values of a row's prefix live in flat caches addressed through its
// ancestry list, so attention is a gather-dot-softmax-gather chain over <= L cache rows of d floats.
// One workgroup per row, thread t owns the 4 dims [4t, 4t+4) (+1024 per extra chunk); the dh/4 threads of a
// head reduce their partial dots with xor-shuffles; scores sit in LDS (H x L floats).  Reads 2 x (p+1) x d x 4
// bytes per row -- the unfused chain (two index_select, two permute copies, two batched M=1 GEMMs, mask,
// softmax) moved ~12x that.
typedef float at_f4 __attribute__((ext_vector_type(4)));

// Thread layout: the d/4 float4 columns of a row are covered by TPG = d/4 threads (d <= 1024) and the 256-thread
// workgroup holds G = 256 / TPG such groups that take the prefix positions round-robin (d = 512: two positions
// in flight per step, twice the loads in flight); for d > 1024 every thread owns NQ column chunks and G = 1.
template <int NQ>
__global__ __launch_bounds__(256) void incr_attn_kernel(const float *__restrict__ q, const float *__restrict__ Kc,
                                                        const float *__restrict__ Vc,
                                                        const long long *__restrict__ anc, long long anc_pitch,
                                                        const long long *__restrict__ pos,
                                                        const long long *__restrict__ node, int L, int d, int heads,
                                                        int tpg, float scale, float *__restrict__ out) {
    extern __shared__ float sc[];   // [heads][L] scores, then [G][d] partial contexts
    const int r = blockIdx.x, t = threadIdx.x;
    const int G = NQ == 1 ? 256 / tpg : 1;           // position groups
    const int gi = NQ == 1 ? t / tpg : 0, tl = NQ == 1 ? t - gi * tpg : t;
    const int dh = d / heads, g = dh >> 2;           // g threads per head (power of two <= 64)
    const long long p = min(pos[r], (long long)(L - 1));
    const long long my_node = node[r];
    const long long *arow = anc + (long long)r * anc_pitch;
    at_f4 qv[NQ];
    bool act[NQ];
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
        const int col = (tl + c * 256) * 4;
        act[c] = col < d && gi < G;
        qv[c] = act[c] ? *reinterpret_cast<const at_f4 *>(q + (long long)r * d + col) * scale : at_f4{0.f, 0.f, 0.f, 0.f};
    }
    // pass 1: scores; group gi takes positions gi, gi + G, ...
    for (long long j0 = gi; j0 <= p; j0 += 4 * G) {
        float part[4][NQ];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long j = j0 + (long long)u * G;
            const long long idx = j >= p ? my_node : arow[j];
#pragma unroll
            for (int c = 0; c < NQ; ++c) {
                part[u][c] = 0.f;
                if (act[c] && j <= p) {
                    const at_f4 k4 = *reinterpret_cast<const at_f4 *>(Kc + idx * d + (tl + c * 256) * 4);
                    part[u][c] = qv[c].x * k4.x + qv[c].y * k4.y + qv[c].z * k4.z + qv[c].w * k4.w;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < NQ; ++c) {
                float v = part[u][c];
                for (int o = 1; o < g; o <<= 1) v += __shfl_xor(v, o);
                const int col = (tl + c * 256) * 4;
                const long long j = j0 + (long long)u * G;
                if (act[c] && j <= p && (tl & (g - 1)) == 0) sc[(col / dh) * L + (int)j] = v;
            }
    }
    __syncthreads();
    // pass 2: softmax per head (the g threads of a head in group 0 cooperate), normalised weights back into LDS
    if (gi == 0) {
#pragma unroll
        for (int c = 0; c < NQ; ++c) {
            if (!act[c]) continue;
            const int h = ((tl + c * 256) * 4) / dh, lane = tl & (g - 1);
            float m = -INFINITY;
            for (int j = lane; j <= p; j += g) m = fmaxf(m, sc[h * L + j]);
            for (int o = 1; o < g; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
            float sum = 0.f;
            for (int j = lane; j <= p; j += g) sum += __expf(sc[h * L + j] - m);
            for (int o = 1; o < g; o <<= 1) sum += __shfl_xor(sum, o);
            const float inv = 1.f / sum;
            for (int j = lane; j <= p; j += g) sc[h * L + j] = __expf(sc[h * L + j] - m) * inv;
        }
    }
    __syncthreads();
    // pass 3: context, partial per position group
    at_f4 acc[NQ];
#pragma unroll
    for (int c = 0; c < NQ; ++c) acc[c] = at_f4{0.f, 0.f, 0.f, 0.f};
    for (long long j0 = gi; j0 <= p; j0 += 4 * G) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long j = j0 + (long long)u * G;
            if (j > p) break;
            const long long idx = j == p ? my_node : arow[j];
#pragma unroll
            for (int c = 0; c < NQ; ++c)
                if (act[c]) {
                    const int col = (tl + c * 256) * 4;
                    const float a = sc[(col / dh) * L + (int)j];
                    acc[c] += *reinterpret_cast<const at_f4 *>(Vc + idx * d + col) * a;
                }
        }
    }
    if (G > 1) {     // sum the groups' partial contexts through LDS (after everyone is done with the weights)
        __syncthreads();
        float *part = sc;                              // [G][d]
        if (act[0]) *reinterpret_cast<at_f4 *>(part + gi * d + tl * 4) = acc[0];
        __syncthreads();
        if (gi == 0 && act[0]) {
            at_f4 s4 = acc[0];
            for (int k = 1; k < G; ++k) s4 += *reinterpret_cast<const at_f4 *>(part + k * d + tl * 4);
            *reinterpret_cast<at_f4 *>(out + (long long)r * d + tl * 4) = s4;
        }
    } else {
#pragma unroll
        for (int c = 0; c < NQ; ++c)
            if (act[c]) *reinterpret_cast<at_f4 *>(out + (long long)r * d + (tl + c * 256) * 4) = acc[c];
    }
}

// ---------------------------------------------------------------------------------------------
// On-the-fly n-gram FST scoring of the surviving candidates (reference beam_transducer.py:135-181 +
// decoder/sorted_matcher.py:24-111), device resident: one thread per beam slot, the LM state set of a slot is a
// small ordered array (insertion order = the reference's dict order, which its "first smaller wins" update
// depends on), costs in fp64 like the Python floats.  One workgroup per utterance so that every slot reads its
// PARENT's set before any slot overwrites its own.
constexpr int FST_SM = 8;        // states per slot (an n-gram LM needs 1-3); overflow raises a flag
constexpr int FST_MAXDIS = 4;

struct FstDev {
    const long long *off;
    const int *il;
    const float *wt;
    const int *ns;
    const float *fin;
    int max_num_arcs, max_id, backoff_id, ndis;
    int dis[FST_MAXDIS];
};

// sorted_matcher.py:30-48: lower bound over a window of max_num_arcs slots whose tail reads as max_id
__device__ inline long long fst_search(const FstDev &F, int state, int label) {
    const long long lo = F.off[state], hi = F.off[state + 1];
    const long long n = hi - lo;
    const long long w = n < F.max_num_arcs ? n : F.max_num_arcs;
    long long a = 0, b = w;
    while (a < b) {
        const long long m = (a + b) >> 1;
        if (F.il[lo + m] < label) a = m + 1; else b = m;
    }
    long long idx = a;
    if (idx >= w) idx = (w < F.max_num_arcs && F.max_id >= label) ? w : F.max_num_arcs - 1;
    if (idx >= n || F.il[lo + idx] != label) return -1;
    return lo + idx;
}

__global__ __launch_bounds__(64) void fst_advance_kernel(FstDev F, const long long *__restrict__ prev_k,
                                                         const long long *__restrict__ y_raw,
                                                         const long long *__restrict__ y_now, int blk,
                                                         double nonblk_reward, float lm_scale,
                                                         int *__restrict__ set_n, int *__restrict__ set_st,
                                                         double *__restrict__ set_cs,
                                                         float *__restrict__ lm_scores, float *__restrict__ scores,
                                                         float *__restrict__ fin_score,
                                                         const long long *__restrict__ fin_n, int fin_cap, int K,
                                                         int *__restrict__ err, const int *__restrict__ skip) {
    if (skip && *skip) return;
    __shared__ int fin_flag[64];
    const int b = blockIdx.x, i = threadIdx.x;
    const long long bk = (long long)b * K;
    int pn = 0, pst[FST_SM];
    double pcs[FST_SM];
    long long yr = blk;
    if (i < K) {
        const long long pk = prev_k[bk + i];
        yr = y_raw[bk + i];
        pn = set_n[bk + pk];
        for (int s = 0; s < FST_SM; ++s) { pst[s] = set_st[(bk + pk) * FST_SM + s]; pcs[s] = set_cs[(bk + pk) * FST_SM + s]; }
    }
    __syncthreads();                                   // every parent set is in registers before any write
    int nn = 0, nst[FST_SM];
    double ncs[FST_SM];
    bool overflow = false;
    if (i < K) {
        if (yr == blk) {                               // :151-152
            nn = pn;
            for (int s = 0; s < FST_SM; ++s) { nst[s] = pst[s]; ncs[s] = pcs[s]; }
        } else {
            const int ilabel = (int)yr + 1;            // :139
            for (int s = 0; s < pn; ++s) {
                // get_scores: the state itself, then the destinations of its disambiguation arcs
                for (int d = -1; d < F.ndis; ++d) {
                    int cur = pst[s];
                    double bf = 0.0;
                    if (d >= 0) {
                        const long long a = fst_search(F, pst[s], F.dis[d]);
                        if (a < 0) continue;
                        bf = (double)F.wt[a];
                        cur = F.ns[a];
                    }
                    for (;;) {                         // get_scores_wodisambig: walk the back-off chain
                        const long long a = fst_search(F, cur, ilabel);
                        if (a >= 0) {
                            const double next_cost = pcs[s] + (bf + (double)F.wt[a]);
                            const int nxt = F.ns[a];
                            int f = -1;
                            for (int q = 0; q < nn; ++q) if (nst[q] == nxt) { f = q; break; }
                            if (f < 0) {
                                if (nn < FST_SM) { nst[nn] = nxt; ncs[nn] = next_cost - nonblk_reward; ++nn; }
                                else overflow = true;
                            } else if (next_cost < ncs[f]) {   // :147-149 (sic: compared without the reward)
                                ncs[f] = next_cost - nonblk_reward;
                            }
                        }
                        const long long bo = fst_search(F, cur, F.backoff_id);
                        if (bo < 0) break;
                        bf += (double)F.wt[bo];
                        cur = F.ns[bo];
                    }
                }
            }
        }
        double mn = INFINITY;
        for (int q = 0; q < nn; ++q) mn = fmin(mn, ncs[q]);
        lm_scores[bk + i] = nn ? (float)(-mn) : -1e20f;      // :153-157
        set_n[bk + i] = nn;
        for (int s = 0; s < FST_SM; ++s) { set_st[(bk + i) * FST_SM + s] = nst[s]; set_cs[(bk + i) * FST_SM + s] = ncs[s]; }
    }
    fin_flag[i] = (i < K && y_now[bk + i] == EOS) ? 1 : 0;
    __syncthreads();
    if (i < K && fin_flag[i]) {
        // final cost (:165-181, sorted_matcher.final_score): min over active states and their disambiguation
        // destinations of cost + (back-off chain to a final state)
        double best = INFINITY;
        for (int s = 0; s < nn; ++s) {
            for (int d = -1; d < F.ndis; ++d) {
                int cur = nst[s];
                double sc = 0.0;
                if (d >= 0) {
                    const long long a = fst_search(F, nst[s], F.dis[d]);
                    if (a < 0) continue;
                    sc = (double)F.wt[a];
                    cur = F.ns[a];
                }
                bool ok = true;
                for (;;) {
                    const float fw = F.fin[cur];
                    if (isinf(fw)) {
                        const long long bo = fst_search(F, cur, F.backoff_id);
                        if (bo < 0) { ok = false; break; }
                        sc += (double)F.wt[bo];
                        cur = F.ns[bo];
                    } else { sc += (double)fw; break; }
                }
                if (ok) best = fmin(best, ncs[s] + sc);
            }
        }
        if (isinf(best)) overflow = true;              // the reference would raise on an empty candidate set
        int total = 0, rank = 0;
        for (int q = 0; q < K; ++q) { total += fin_flag[q]; rank += (q < i) ? fin_flag[q] : 0; }
        const float v = scores[bk + i] + lm_scale * (float)(-best);
        scores[bk + i] = v;                            // `s = self.scores[i]` is a view: the final cost lands here too
        long long pos = fin_n[b] - total + rank;
        if (pos > fin_cap - 2) pos = fin_cap - 2;
        fin_score[(long long)b * fin_cap + pos] = v;
    }
    if (overflow) atomicOr(err, 1);
}

}  // namespace

extern "C" int pika_beam_advance(const float *logits, float sm_scale, int first, float *scores,
                                 const float *lm_scores, float lm_scale, long long *y,
                                 long long *t_idx, const long long *num_frames,
                                 const long long *max_len, long long *hyp, long long *hyp_len, int L,
                                 long long *ks_hist, long long *ys_hist, const long long *step_t,
                                 unsigned char *eos_top, float *fin_score, long long *fin_step,
                                 long long *fin_k, long long *fin_n, int fin_cap,
                                 long long *prev_k_out, long long *y_raw, void *cand_ws, int B, int K,
                                 int V, int blk, int beam_prune, void *stream) {
    if (!logits || !scores || !lm_scores || !y || !t_idx || !num_frames || !max_len || !hyp ||
        !hyp_len || !ks_hist || !ys_hist || !step_t || !eos_top || !fin_score || !fin_step ||
        !fin_k || !fin_n || !prev_k_out || B <= 0 || K <= 0 || V <= 0 || L <= 0 || fin_cap < 3)
        return PIKA_EINVAL;
    if (K > MAXK || V > MAXV || V < K || (size_t)K * L * 4 > 64 * 1024) return PIKA_ETOOBIG;
    if (!cand_ws) return PIKA_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(beam_row_topk_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, WAVES * MAXV * 4);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(beam_merge_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    Cand *cand = static_cast<Cand *>(cand_ws);
    hipLaunchKernelGGL(beam_row_topk_kernel, dim3((B * K + WAVES - 1) / WAVES), dim3(WAVES * 64),
                       (size_t)WAVES * ((V + 3) & ~3) * 4, st, logits, sm_scale, first, scores, lm_scores, lm_scale,
                       y, hyp, hyp_len, L, B, K, V, beam_prune, cand);
    BeamState a{scores, lm_scores, lm_scale, y, t_idx, num_frames, max_len, hyp, hyp_len, L, ks_hist, ys_hist, step_t,
                eos_top, fin_score, fin_step, fin_k, fin_n, fin_cap, prev_k_out, y_raw, B, K, V, blk};
    hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(256),
                       (size_t)K * L * 4 + (size_t)K * K * sizeof(Cand), st, cand, a);
    return (int)hipGetLastError();
}

// Also exported: the LDS the launch needs and whether it fits, so that the host-side gate (decoder/fused_step.py::supported)
// asks the library instead of restating its arithmetic.  Returns the bytes, or 0 when the shape is not taken.
extern "C" size_t pika_beam_advance_logits_lds(int K, int L, int splits) {
    if (K <= 0 || L <= 0 || splits < 1 || K > MAXK || splits > 64) return 0;
    int waves = K < 16 ? K : 16;
    auto lds_for = [&](int w) { return (size_t)K * L * 4 + (size_t)K * K * sizeof(Cand) + (size_t)w * POOL_CAP * sizeof(Cand); };
    while (waves > 4 && lds_for(waves) > 96 * 1024) waves >>= 1;
    if (waves < 4) waves = 4;
    return lds_for(waves) > 96 * 1024 ? 0 : lds_for(waves);
}

extern "C" int pika_beam_advance_logits(const float *pmax, const float *psum, const float *logits, long long ldl,
                                        int splits,
                                        float *scores, const float *lm_scores, float lm_scale, long long *y,
                                        long long *t_idx, const long long *num_frames, const long long *max_len,
                                        long long *hyp, long long *hyp_len, int L, long long *ks_hist,
                                        long long *ys_hist, long long *step_t, unsigned char *eos_top,
                                        float *fin_score, long long *fin_step, long long *fin_k, long long *fin_n,
                                        int fin_cap, long long *prev_k_out, long long *y_raw, int B, int K, int V,
                                        int blk, int beam_prune, int n_best, int *stop, long long *max_hyp,
                                        int *sync, void *stream) {
    if (!logits || ldl < V) return PIKA_EINVAL;
    if (!pmax || !psum || !scores || !lm_scores || !y || !t_idx || !num_frames || !max_len || !hyp ||
        !hyp_len || !ks_hist || !ys_hist || !step_t || !eos_top || !fin_score || !fin_step || !fin_k || !fin_n ||
        !prev_k_out || !stop || !max_hyp || !sync || B <= 0 || K <= 0 || V <= 0 || L <= 0 || fin_cap < 3 || splits < 1)
        return PIKA_EINVAL;
    // a wave per beam row (at most 16), fewer when their candidate pools would not fit the LDS budget
    int waves = K < 16 ? K : 16;
    auto lds_for = [&](int w) { return (size_t)K * L * 4 + (size_t)K * K * sizeof(Cand) + (size_t)w * POOL_CAP * sizeof(Cand); };
    while (waves > 4 && lds_for(waves) > 96 * 1024) waves >>= 1;
    if (waves < 4) waves = 4;
    const size_t lds_bytes = lds_for(waves);
    if (K > MAXK || splits > 64 || lds_bytes > 96 * 1024) return PIKA_ETOOBIG;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(beam_partials_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    BeamState a{scores, lm_scores, lm_scale, y, t_idx, num_frames, max_len, hyp, hyp_len, L, ks_hist, ys_hist, step_t,
                eos_top, fin_score, fin_step, fin_k, fin_n, fin_cap, prev_k_out, y_raw, B, K, V, blk};
    hipLaunchKernelGGL(beam_partials_kernel, dim3(B), dim3(64 * waves), lds_bytes, static_cast<hipStream_t>(stream), pmax, psum,
                       splits, a, beam_prune, n_best, stop, max_hyp, sync, step_t, logits, ldl);
    return (int)hipGetLastError();
}

extern "C" int pika_incremental_attention(const float *q, const float *k_cache, const float *v_cache,
                                          const long long *ancestry, long long ancestry_pitch,
                                          const long long *pos, const long long *node, int rows, int L, int d,
                                          int heads, float *out, void *stream) {
    if (!q || !k_cache || !v_cache || !ancestry || !pos || !node || !out || rows <= 0 || L <= 0 || d <= 0 || heads <= 0)
        return PIKA_EINVAL;
    if (d % heads || (d & 3)) return PIKA_EINVAL;
    const int dh = d / heads, g = dh >> 2;
    if ((dh & 3) || g < 1 || g > 64 || (g & (g - 1))) return PIKA_EINVAL;     // dh/4 threads per head, power of two
    if (d > 2048 || (size_t)heads * L * sizeof(float) > 64 * 1024) return PIKA_ETOOBIG;
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k_cache) | reinterpret_cast<uintptr_t>(v_cache) |
         reinterpret_cast<uintptr_t>(out)) & 15)
        return PIKA_EINVAL;
    const float scale = 1.0f / sqrtf((float)dh);
    // threads per position group: the smallest power of two >= d/4 that divides 256 (d <= 1024)
    int tpg = 256;
    if (d <= 1024) { tpg = 1; while (tpg < d / 4) tpg <<= 1; }
    const int G = d <= 1024 ? 256 / tpg : 1;
    size_t smem = (size_t)heads * L * sizeof(float);
    if ((size_t)G * d * sizeof(float) > smem) smem = (size_t)G * d * sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (d <= 1024)
        hipLaunchKernelGGL(incr_attn_kernel<1>, dim3(rows), dim3(256), smem, s, q, k_cache, v_cache, ancestry,
                           ancestry_pitch, pos, node, L, d, heads, tpg, scale, out);
    else
        hipLaunchKernelGGL(incr_attn_kernel<2>, dim3(rows), dim3(256), smem, s, q, k_cache, v_cache, ancestry,
                           ancestry_pitch, pos, node, L, d, heads, 256, scale, out);
    return (int)hipGetLastError();
}

extern "C" int pika_fst_advance(const long long *fst_offsets, const int *fst_ilabel, const float *fst_weight,
                                const int *fst_nextstate, const float *fst_final, int max_num_arcs, int max_id,
                                int backoff_id, const int *disambig_ids, int n_disambig,
                                const long long *prev_k, const long long *y_raw, const long long *y, int blk,
                                double nonblk_reward, float lm_scale, int *set_n, int *set_state,
                                double *set_cost, float *lm_scores, float *scores, float *fin_score,
                                const long long *fin_n, int fin_cap, int B, int K, int *err, const int *skip,
                                void *stream) {
    if (!fst_offsets || !fst_ilabel || !fst_weight || !fst_nextstate || !fst_final || !prev_k || !y_raw || !y ||
        !set_n || !set_state || !set_cost || !lm_scores || !scores || !fin_score || !fin_n || !err || B <= 0 || K <= 0)
        return PIKA_EINVAL;
    if (K > 64 || n_disambig < 0 || n_disambig > FST_MAXDIS || (n_disambig && !disambig_ids)) return PIKA_ETOOBIG;
    FstDev F{fst_offsets, fst_ilabel, fst_weight, fst_nextstate, fst_final, max_num_arcs, max_id, backoff_id, n_disambig, {0, 0, 0, 0}};
    for (int i = 0; i < n_disambig; ++i) F.dis[i] = disambig_ids[i];     // host array
    hipLaunchKernelGGL(fst_advance_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), F, prev_k, y_raw,
                       y, blk, nonblk_reward, lm_scale, set_n, set_state, set_cost, lm_scores, scores, fin_score,
                       fin_n, fin_cap, K, err, skip);
    return (int)hipGetLastError();
}

extern "C" int pika_fst_states_per_slot(void) { return FST_SM; }

// ---------------------------------------------------------------------------------------------------------------------
// get_hyp for every n-best entry at once (include/pika_decode.h: pika_beam_backtrack): one thread walks one entry's
// back-pointers from its finishing step down to step 0 -- two dependent loads per step from histories that sit in L2.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void beam_backtrack_kernel(const long long *__restrict__ ys_hist,
                                                            const long long *__restrict__ ks_hist,
                                                            const int *__restrict__ sel_step, const int *__restrict__ sel_k,
                                                            int n, int per_utt, int B, int K, int smax, int blk,
                                                            int *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const long long plane = (long long)B * K;
    const long long base = (long long)(e / per_utt) * K;
    int steps = sel_step[e];
    steps = steps < 0 ? 0 : (steps > smax ? smax : steps);
    int *o = out + (long long)e * smax;
    for (int j = smax - 1; j >= steps; --j) o[j] = blk;
    long long idx = sel_k[e];
    for (int j = steps - 1; j >= 0; --j) {
        idx = idx < 0 ? 0 : (idx >= K ? K - 1 : idx);
        o[j] = (int)ys_hist[(long long)(j + 1) * plane + base + idx];
        idx = ks_hist[(long long)j * plane + base + idx];
    }
}
}  // namespace

extern "C" int pika_beam_backtrack(const long long *ys_hist, const long long *ks_hist, const int *sel_step, const int *sel_k,
                                   int n, int per_utt, int B, int K, int smax, int blk, int *out, void *stream) {
    if (!ys_hist || !ks_hist || !sel_step || !sel_k || !out || n <= 0 || per_utt <= 0 || B <= 0 || K <= 0 || smax <= 0 ||
        (long long)B * per_utt < n)
        return PIKA_EINVAL;
    hipLaunchKernelGGL(beam_backtrack_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream),
                       ys_hist, ks_hist, sel_step, sel_k, n, per_utt, B, K, smax, blk, out);
    return (int)hipGetLastError();
}
