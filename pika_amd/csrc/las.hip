// pika_amd/csrc/las.hip -- per-token kernels of LAS n-best rescoring for gfx950 (include/pika_las.h;
// reference trainer/model/las.py:649-668, modules/global_attention.py:162-248, modules/stacked_rnn.py:20-34).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "pika_las.h"
#include "pika_rnnt.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// 1 - 2/(e^{2x}+1): exact limits at +-inf (exp -> inf: rcp -> 0; exp -> 0: 1 - 2), abs error ~2e-7
__device__ inline float tanh_fast(float x) { return 1.0f - 2.0f * rcp(__expf(2.0f * x) + 1.0f); }
__device__ inline float sigmoid_fast(float x) { return rcp(1.0f + __expf(-x)); }

// one thread per 4 channels of one row
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float *__restrict__ gates, long long ldg,
                                                        const float *__restrict__ c_prev, float *__restrict__ c_out,
                                                        float *__restrict__ h_out, long long ldh,
                                                        float *__restrict__ h_out2, long long ldh2, int N, int H,
                                                        const int *__restrict__ n_dev, const int *__restrict__ rowlist,
                                                        const int *__restrict__ rowoff_dev) {
    const int H4 = H >> 2;
    if (n_dev) N = min(N, *n_dev);       // rows in use this step (a launch captured once, replayed per token)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * H4) return;
    const int e_ = (int)(idx / H4), c = (int)(idx - (long long)e_ * H4) << 2;
    const int n = rowlist ? rowlist[(rowoff_dev ? *rowoff_dev : 0) + e_] : e_;     // launch rows as a gather list
    const float *g = gates + (long long)n * ldg + c;
    const f32x4 gi = *reinterpret_cast<const f32x4 *>(g), gf = *reinterpret_cast<const f32x4 *>(g + H);
    const f32x4 gg = *reinterpret_cast<const f32x4 *>(g + 2 * H), go = *reinterpret_cast<const f32x4 *>(g + 3 * H);
    const f32x4 cp = *reinterpret_cast<const f32x4 *>(c_prev + (long long)n * H + c);
    f32x4 cn, h;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        cn[e] = sigmoid_fast(gf[e]) * cp[e] + sigmoid_fast(gi[e]) * tanh_fast(gg[e]);
        h[e] = sigmoid_fast(go[e]) * tanh_fast(cn[e]);
    }
    *reinterpret_cast<f32x4 *>(c_out + (long long)n * H + c) = cn;
    *reinterpret_cast<f32x4 *>(h_out + (long long)n * ldh + c) = h;
    if (h_out2) *reinterpret_cast<f32x4 *>(h_out2 + (long long)n * ldh2 + c) = h;
}

constexpr int KQ = 4;         // float4 slots per lane: D <= 64 * 4 * KQ = 1024

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ inline float block_reduce(float v, bool is_max, float *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, w) : v + w;
    }
    __syncthreads();                       // `red` may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    v = red[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) v = is_max ? fmaxf(v, red[w]) : v + red[w];
    return v;
}

// ---- the "mlp" attention, utterance by utterance --------------------------------------------------------------
// One query per workgroup (rounds 3-4) reads an utterance's U_a h_s and h_s rows once per QUERY: 1.9 MB per workgroup at S = 240,
// D = 1024, and a CU pulls bytes from beyond its L1 at ~50 GB/s -- 111 us for the ~470 queries of an average step of a
// rescoring pass, three times what its exp2 / rcp need.  Here a workgroup takes ALL the queries of one utterance (the
// query list is ordered by utterance) and a chunk of ACS positions: every row is read once per workgroup; a second launch
// merges the chunks' partial (max, sum, context sums).  Measured (tools/las_gemm_bench.py): 130 -> 73 us at 470 queries,
// 158 -> 99 at 960, 36 -> 39 at 64 (one query per utterance: the workgroup's chain of dependent round trips -- row count,
// ranges, query rows, position rows, two barriers, the chunk's h_s rows -- is ~20 us, the merge launch the rest).
// Batching the loads of phases 1 and 3 further cost registers (> 128: one workgroup per CU) and measured worse.
//   phase 1  wave w walks positions w, w + 8, .. of the chunk: the position's U_a h_s row in registers, the queries from
//            LDS (staged scaled by 2 log2 e), one score per (query, position) into LDS;
//   phase 2  thread (query, position): the chunk's max, the weights exp(score - max) (into LDS, [position][query]), their sum;
//   phase 3  wave w owns channels [128 w, 128 w + 128): sum_s weight[s][g] h_s[channel] for every query of the block.
constexpr int ACS = 32;       // positions per chunk
constexpr int AQB = 16;       // queries per block of an utterance's queries (AQB * ACS = the 512 threads of phase 2)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void las_att_chunk_kernel(const float *__restrict__ wq, long long ldq,
                                                            const float *__restrict__ proj,
                                                            const float *__restrict__ context,
                                                            const int *__restrict__ lens, const int *__restrict__ qidx,
                                                            const int *__restrict__ uoff, const float *__restrict__ v,
                                                            float *__restrict__ work, int NC, int N, int B, int S, int D,
                                                            const int *__restrict__ n_dev,
                                                            const int *__restrict__ qoff_dev,
                                                            const int *__restrict__ step_dev) {
    extern __shared__ __attribute__((aligned(16))) float att_lds[];
    const int b = blockIdx.x / NC, c = blockIdx.x - b * NC;
    if (n_dev) N = min(N, *n_dev);
    if (N <= 0) return;                  // (a step beyond the pass: its table of ranges ends with the pass)
    if (qoff_dev) qidx += *qoff_dev;
    if (step_dev) uoff += (long long)*step_dev * (B + 1);
    const int u0 = min(uoff[b], N), u1 = min(uoff[b + 1], N);
    const int len = min(lens[b], S), s0 = c * ACS;
    if (u0 >= u1 || s0 >= len) return;                         // block-uniform
    const int ns = min(ACS, len - s0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, D4 = D >> 2;
    float *qs = att_lds;                                       // [AQB][D]: 2 log2(e) (W_q h_t) of the block's queries
    float *sc = att_lds + (size_t)AQB * D;                     // [AQB][ACS] scores, then [ACS][AQB] weights
    constexpr float C2 = 2.8853900817779268f;                  // 2 log2(e): tanh(x) = 1 - 2 / (exp2(C2 x) + 1)
    f32x4 vv[KQ];
    float vsum = 0.f;
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int j = lane + 64 * k;
        const f32x4 v4 = j < D4 ? reinterpret_cast<const f32x4 *>(v)[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        vsum += (v4.x + v4.y) + (v4.z + v4.w);
        vv[k] = v4 * -2.0f;
    }
    vsum = wave_sum(vsum);
    for (int qb = u0; qb < u1; qb += AQB) {
        const int nqb = min(AQB, u1 - qb);
        auto load_p = [&](int i, f32x4 (&dst)[KQ]) {
            const f32x4 *prow = reinterpret_cast<const f32x4 *>(proj + ((long long)b * S + s0 + (i < ns ? i : ns - 1)) * D);
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
                const int j = lane + 64 * k;
                dst[k] = j < D4 ? prow[j] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        {   // wave w stages queries w and w + 8 of the block: their rows requested together
            int qn[AQB / 8];
#pragma unroll
            for (int u = 0; u < AQB / 8; ++u) qn[u] = wave + 8 * u < nqb ? qidx[qb + wave + 8 * u] : -1;
            f32x4 t[AQB / 8][KQ];
#pragma unroll
            for (int u = 0; u < AQB / 8; ++u)
#pragma unroll
                for (int k = 0; k < KQ; ++k) {
                    const int j = lane + 64 * k;
                    t[u][k] = (qn[u] >= 0 && j < D4) ? reinterpret_cast<const f32x4 *>(wq + (long long)qn[u] * ldq)[j]
                                                     : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            __syncthreads();                                   // the previous block's phase 3 has read sc (and phase 1 qs)
#pragma unroll
            for (int u = 0; u < AQB / 8; ++u)
#pragma unroll
                for (int k = 0; k < KQ; ++k) {
                    const int j = lane + 64 * k;
                    if (qn[u] >= 0 && j < D4) reinterpret_cast<f32x4 *>(qs + (size_t)(wave + 8 * u) * D)[j] = t[u][k] * C2;
                }
        }
        __syncthreads();
        for (int i = wave; i < ns; i += 8) {
            f32x4 p[KQ];
            load_p(i, p);
#pragma unroll
            for (int k = 0; k < KQ; ++k) p[k] *= C2;
            for (int g = 0; g < nqb; ++g) {
                const f32x4 *qrow = reinterpret_cast<const f32x4 *>(qs + (size_t)g * D);
                float part = 0.f;
#pragma unroll
                for (int k = 0; k < KQ; ++k) {
                    const int j = lane + 64 * k;
                    const f32x4 q4 = qrow[j < D4 ? j : 0];     // (lanes past D: vv = 0)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ex = __builtin_amdgcn_exp2f(p[k][e] + q4[e]);       // e^{2z}
                        part = __builtin_fmaf(vv[k][e], rcp(ex + 1.0f), part);         // inf -> 0, 0 -> -2 v
                    }
                }
                const float sv = vsum + wave_sum(part);
                if (lane == 0) sc[g * ACS + i] = sv;
            }
        }
        __syncthreads();
        {   // phase 2: thread (g, i)
            const int g = tid >> 5, i = tid & 31;
            const bool on = g < nqb && i < ns;
            const float sv = on ? sc[g * ACS + i] : -INFINITY;
            float m = sv;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            const float w = on ? __expf(sv - m) : 0.f;
            float l = w;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) l += __shfl_xor(l, o);
            __syncthreads();                                   // every score has been read
            sc[i * AQB + g] = w;                               // [position][query]
            if (i == 0 && g < nqb) {
                float *pq = work + ((size_t)(qb + g) * NC + c) * (D + 2);
                pq[0] = m;
                pq[1] = l;
            }
        }
        __syncthreads();
        {   // phase 3: channels (2 j2, 2 j2 + 1) of every query of the block
            const int j2 = wave * 64 + lane;
            if (2 * j2 < D) {
                float ax[AQB], ay[AQB];
#pragma unroll
                for (int g = 0; g < AQB; ++g) { ax[g] = 0.f; ay[g] = 0.f; }
                const float *xcol = context + ((long long)b * S + s0) * D + 2 * j2;
                for (int i = 0; i < ns; i += 4) {              // 4 positions' values requested together (rows >= ns: zero weights)
                    float2 x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const float2 *>(xcol + (long long)min(i + u, ns - 1) * D);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const f32x4 *wrow = reinterpret_cast<const f32x4 *>(sc + (i + u) * AQB);
#pragma unroll
                        for (int g4 = 0; g4 < AQB / 4; ++g4) {
                            const f32x4 w4 = wrow[g4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                ax[g4 * 4 + e] = __builtin_fmaf(w4[e], x[u].x, ax[g4 * 4 + e]);
                                ay[g4 * 4 + e] = __builtin_fmaf(w4[e], x[u].y, ay[g4 * 4 + e]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);     // (one position's weights in registers at a time)
                    }
                }
                float *dst = work + ((size_t)qb * NC + c) * (D + 2) + 2 + 2 * j2;
                const size_t pitch = (size_t)NC * (D + 2);
#pragma unroll
                for (int g = 0; g < AQB; ++g) {
                    if (g < nqb) *reinterpret_cast<float2 *>(dst) = float2{ax[g], ay[g]};
                    dst += pitch;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
}

// list entry e: merge its chunks' partials {max, sum, context sums[D]} (pitch D + 2) into row qidx[e] of ctx_out
__global__ __launch_bounds__(256) void las_att_merge_kernel(const float *__restrict__ work, const int *__restrict__ owner,
                                                            const int *__restrict__ lens, const int *__restrict__ qidx,
                                                            float *__restrict__ ctx_out, long long ldo, int NC, int N, int S,
                                                            int D, const int *__restrict__ n_dev,
                                                            const int *__restrict__ qoff_dev) {
    __shared__ float f_s[64];                                 // exp(max_c - M) / L per chunk (NC <= 2048 / ACS = 64)
    if (n_dev) N = min(N, *n_dev);
    const int e = blockIdx.x;
    if (e >= N) return;
    if (qoff_dev) qidx += *qoff_dev;
    const int q = qidx[e], len = min(lens[owner[q]], S), nc = min(NC, (len + ACS - 1) / ACS);
    const float *pw = work + (size_t)e * NC * (D + 2);
    if (threadIdx.x < 64) {                                    // one wave: chunk c on lane c, all chunks requested at once
        const int c = threadIdx.x;
        const float2 ml = c < nc ? *reinterpret_cast<const float2 *>(pw + (size_t)c * (D + 2)) : float2{-INFINITY, 0.f};
        float M = ml.x;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o));
        const float f = c < nc ? __expf(ml.x - M) : 0.f;
        const float L = wave_sum(ml.y * f);
        f_s[c] = f / L;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < (D >> 1); j += 256) {
        float2 t = float2{0.f, 0.f};
        for (int c0 = 0; c0 < nc; c0 += 8) {                   // 8 chunks' sums requested together
            float2 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                a[u] = c0 + u < nc ? *reinterpret_cast<const float2 *>(pw + (size_t)(c0 + u) * (D + 2) + 2 + 2 * j) : float2{0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float f = f_s[min(c0 + u, 63)];           // (0 beyond nc)
                t.x = __builtin_fmaf(f, a[u].x, t.x);
                t.y = __builtin_fmaf(f, a[u].y, t.y);
            }
        }
        *reinterpret_cast<float2 *>(ctx_out + (long long)q * ldo + 2 * j) = t;
    }
}

// ---- the per-token bookkeeping of a rescoring pass captured as ONE launch sequence (include/pika_las.h) ----------
// step = {t, n, qoff, -}: advanced by one thread in front of every token's launches
__global__ void las_step_advance_kernel(int *__restrict__ step, const int *__restrict__ n_active,
                                        const int *__restrict__ qoff, int L) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int t = step[0] + 1;
        step[0] = t;
        step[1] = t < L ? n_active[t] : 0;
        step[2] = t < L ? qoff[t] : 0;
    }
}

// rows r < n of the decoder's layer-0 input get the embedding of token (t, r); crow[r] = t * N + r addresses the row of
// the (L, N, H) result this token's output projection writes
__global__ __launch_bounds__(256) void las_embed_rows_kernel(const int *__restrict__ step,
                                                             const long long *__restrict__ tokens,
                                                             const float *__restrict__ emb, float *__restrict__ x0,
                                                             long long ldx, long long *__restrict__ crow, int N, int E,
                                                             const int *__restrict__ rowlist) {
    const int t = step[0], n = min(N, step[1]);
    const int E4 = E >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * E4) return;
    const int e_ = (int)(idx / E4), c = (int)(idx - (long long)e_ * E4) << 2;
    const int r = rowlist ? rowlist[step[2] + e_] : e_;       // the step's active rows (the attention's query list)
    const long long tok = tokens[(long long)t * N + r];
    *reinterpret_cast<f32x4 *>(x0 + (long long)r * ldx + c) = *reinterpret_cast<const f32x4 *>(emb + tok * E + c);
    if (c == 0) crow[r] = (long long)t * N + r;
}

// hypotheses that share a prefix share the decoder rows of that prefix; at the step a hypothesis leaves the prefix it
// inherits the recurrent state of the row that computed it: segment s of row dst <- the same columns of row src
struct ForkSegs { float *base[PIKA_LAS_FORK_SEGS]; long long ld[PIKA_LAS_FORK_SEGS]; int col0[PIKA_LAS_FORK_SEGS], ncols[PIKA_LAS_FORK_SEGS]; };
__global__ __launch_bounds__(256) void las_fork_rows_kernel(const int *__restrict__ step, const int *__restrict__ fork_off,
                                                            const int *__restrict__ fork_dst,
                                                            const int *__restrict__ fork_src, ForkSegs g) {
    const int t = step[0];
    if (step[1] <= 0) return;           // no active rows: a step beyond the pass (a longer pass shares the replayed graph)
    const int k0 = fork_off[t], k1 = fork_off[t + 1];
    const int sgm = blockIdx.y;
    float *base = g.base[sgm];
    const long long ld = g.ld[sgm];
    const int c0 = g.col0[sgm], n4 = g.ncols[sgm] >> 2;
    for (int k = k0 + blockIdx.x; k < k1; k += gridDim.x) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(base + (long long)fork_src[k] * ld + c0);
        f32x4 *dst = reinterpret_cast<f32x4 *>(base + (long long)fork_dst[k] * ld + c0);
        for (int i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
    }
}

}  // namespace

extern "C" {

int pika_las_fork_rows(const int *step, const int *fork_off, const int *fork_dst, const int *fork_src, int max_forks,
                       int nseg, float *const *base, const long long *ld, const int *col0, const int *ncols, void *stream) {
    if (!step || !fork_off || !fork_dst || !fork_src || !base || !ld || !col0 || !ncols) return PIKA_EINVAL;
    if (nseg < 1 || nseg > PIKA_LAS_FORK_SEGS || max_forks < 0) return PIKA_EINVAL;
    if (max_forks == 0) return PIKA_OK;
    ForkSegs g{};
    for (int i = 0; i < nseg; ++i) {
        if (!base[i] || (ld[i] & 3) || (col0[i] & 3) || (ncols[i] & 3) || ncols[i] <= 0 ||
            (reinterpret_cast<uintptr_t>(base[i]) & 15))
            return PIKA_EINVAL;
        g.base[i] = base[i]; g.ld[i] = ld[i]; g.col0[i] = col0[i]; g.ncols[i] = ncols[i];
    }
    hipLaunchKernelGGL(las_fork_rows_kernel, dim3((unsigned)(max_forks < 256 ? max_forks : 256), (unsigned)nseg), dim3(256), 0,
                       static_cast<hipStream_t>(stream), step, fork_off, fork_dst, fork_src, g);
    return (int)hipGetLastError();
}

int pika_las_step_advance(int *step, const int *n_active, const int *qoff, int L, void *stream) {
    if (!step || !n_active || !qoff || L <= 0) return PIKA_EINVAL;
    hipLaunchKernelGGL(las_step_advance_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), step, n_active, qoff, L);
    return (int)hipGetLastError();
}

int pika_las_embed_rows(const int *step, const long long *tokens, const float *emb, float *x0, long long ldx,
                        long long *crow, int N, int E, const int *rowlist, void *stream) {
    if (!step || !tokens || !emb || !x0 || !crow || N <= 0 || E <= 0 || (E & 3) || (ldx & 3) || ldx < E) return PIKA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(emb) | reinterpret_cast<uintptr_t>(x0)) & 15) return PIKA_EINVAL;
    const long long total = (long long)N * (E >> 2);
    hipLaunchKernelGGL(las_embed_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), step, tokens, emb, x0, ldx, crow, N, E, rowlist);
    return (int)hipGetLastError();
}

int pika_lstm_cell(const float *gates, long long ldg, const float *c_prev, float *c_out, float *h_out, long long ldh,
                   float *h_out2, long long ldh2, int N, int H, const int *n_dev, const int *rowlist,
                   const int *rowoff_dev, void *stream) {
    if (!gates || !c_prev || !c_out || !h_out || N <= 0 || H <= 0) return PIKA_EINVAL;
    if ((H & 3) || (ldg & 3) || (ldh & 3) || (h_out2 && (ldh2 & 3)) || ldg < 4LL * H || ldh < H) return PIKA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(gates) | reinterpret_cast<uintptr_t>(c_prev) | reinterpret_cast<uintptr_t>(c_out) |
         reinterpret_cast<uintptr_t>(h_out) | reinterpret_cast<uintptr_t>(h_out2)) & 15)
        return PIKA_EINVAL;
    const long long total = (long long)N * (H >> 2);
    if ((total + 255) / 256 > 0x7fffffffLL) return PIKA_ETOOBIG;
    hipLaunchKernelGGL(lstm_cell_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), gates, ldg, c_prev, c_out, h_out, ldh, h_out2, ldh2, N, H, n_dev,
                       rowlist, rowoff_dev);
    return (int)hipGetLastError();
}

size_t pika_las_attention_work_floats(int N, int S, int D) {
    if (N <= 0 || S <= 0 || D <= 0) return 0;
    return (size_t)N * ((S + ACS - 1) / ACS) * ((size_t)D + 2);
}

int pika_las_mlp_attention_by_utterance(const float *wq, long long ldq, const float *proj, const float *context,
                                        const int *owner, const int *lens, const int *qidx, const int *uoff, const float *v,
                                        float *ctx_out, long long ldo, float *work, int N, int B, int S, int D,
                                        const int *n_dev, const int *qoff_dev, const int *step_dev, void *stream) {
    if (!wq || !proj || !context || !owner || !lens || !qidx || !uoff || !v || !ctx_out || !work || N <= 0 || B <= 0 ||
        S <= 0 || D <= 0)
        return PIKA_EINVAL;
    if ((D & 3) || (ldq & 3) || (ldo & 3) || ldq < D || ldo < D) return PIKA_EINVAL;
    if (D > 64 * 4 * KQ || S > 2048) return PIKA_ETOOBIG;
    if ((reinterpret_cast<uintptr_t>(wq) | reinterpret_cast<uintptr_t>(proj) | reinterpret_cast<uintptr_t>(context) |
         reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(ctx_out) | reinterpret_cast<uintptr_t>(work)) & 15)
        return PIKA_EINVAL;
    const int NC = (S + ACS - 1) / ACS;
    const size_t lds = ((size_t)AQB * D + (size_t)AQB * ACS) * sizeof(float);
    if (lds > 64 * 1024) {
        static const int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void *>(las_att_chunk_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (rc) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(las_att_chunk_kernel, dim3((unsigned)(B * NC)), dim3(512), lds, st, wq, ldq, proj, context, lens, qidx,
                       uoff, v, work, NC, N, B, S, D, n_dev, qoff_dev, step_dev);
    hipLaunchKernelGGL(las_att_merge_kernel, dim3((unsigned)N), dim3(256), 0, st, work, owner, lens, qidx, ctx_out, ldo, NC,
                       N, S, D, n_dev, qoff_dev);
    return (int)hipGetLastError();
}

}  // extern "C"
