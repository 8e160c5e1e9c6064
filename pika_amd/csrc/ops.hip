// pika_amd/csrc/ops.hip -- transposes, column sums and col2im around the MFMA GEMM (gfx950).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pika_ops.h"
#include "pika_internal.h"
#include "pika_rnnt.h"

namespace {

struct Op {
    const char *ptr;
    int dtype, rows_per_batch, t_in;
    long long batch_stride, ld;
    int C, stride, dil, pad;
};

__device__ inline float load_elem(const Op &o, int r, int k) {
    const int b = r / o.rows_per_batch, t = r - b * o.rows_per_batch;
    const int tap = k / o.C, c = k - tap * o.C;
    const int ti = t * o.stride + tap * o.dil - o.pad;
    if (ti < 0 || ti >= o.t_in) return 0.f;
    const long long off = (long long)b * o.batch_stride + (long long)ti * o.ld + c;
    if (o.dtype == PIKA_F32) return reinterpret_cast<const float *>(o.ptr)[off];
    return (float)reinterpret_cast<const __bf16 *>(o.ptr)[off];
}

// 64x64 tile through LDS (pitch 65: conflict-free column reads); reads coalesced along k,
// writes coalesced along r.
template <typename TOUT>
__global__ __launch_bounds__(256) void transpose_cast_kernel(Op X, int rows, int K,
                                                             TOUT *__restrict__ out,
                                                             long long ld_out) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, k = k0 + tx;
        tile[i][tx] = (r < rows && k < K) ? load_elem(X, r, k) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int k = k0 + i, r = r0 + tx;
        if (k < K && r < ld_out) out[(long long)k * ld_out + r] = (TOUT)tile[tx][i];
    }
}

// grid (ceil(cols/64), chunks): each block sums a slab of rows for 64 columns, 4 row-groups,
// then one atomicAdd per column per block (out pre-zeroed by the host wrapper).  Fallback for
// unaligned matrices.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T *__restrict__ x, long long ld,
                                                     int rows, int cols, int rows_per_block,
                                                     float *__restrict__ out) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int rbeg = blockIdx.y * rows_per_block, rend = min(rows, rbeg + rows_per_block);
    float s = 0.f;
    if (c < cols)
        for (int r = rbeg + g; r < rend; r += 4) s += (float)x[(long long)r * ld + c];
    part[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < cols) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] +
                                                   part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// Vector variant: every lane owns one 16-byte granule (VEC = 4 f32 / 8 bf16 columns), a wave covers
// 64*VEC contiguous columns of a row, the 4 waves of a block take rows r, r+1, r+2, r+3 and four
// row-steps are kept in flight.  Requires ld % VEC == 0, cols % VEC == 0 and a 16-byte aligned base.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T *__restrict__ x, long long ld,
                                                         int rows, int cols, int rows_per_block,
                                                         float *__restrict__ out) {
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ float part[4][64 * VEC];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * VEC;
    const int rbeg = blockIdx.y * rows_per_block, rend = min(rows, rbeg + rows_per_block);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    if (c < cols) {
        const T *col = x + c;
        int r = rbeg + g;
        for (; r + 12 < rend; r += 16) {
            vec_t v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                v[q] = *reinterpret_cast<const vec_t *>(col + (long long)(r + 4 * q) * ld);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] += (float)v[q][i];
        }
        for (; r < rend; r += 4) {
            const vec_t v = *reinterpret_cast<const vec_t *>(col + (long long)r * ld);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] += (float)v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) part[g][i * 64 + lane] = acc[i];
    __syncthreads();
    if (g == 0 && c < cols) {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
            atomicAdd(out + c + i, part[0][i * 64 + lane] + part[1][i * 64 + lane] +
                                       part[2][i * 64 + lane] + part[3][i * 64 + lane]);
    }
}

// out[c][(taps - 1 - t) * N + n] = bf16(W[n][t * C + c]): the weight operand of the transposed convolution / of a plain
// d(input) product (taps = 1: W^T) in ONE launch -- torch's flip + permuted copy + cast were three (35 us per TDNN layer).
// 32 x 32 tiles through LDS: reads coalesced along c, writes along n.  grid (ceil(C/32), ceil(N/32), taps), 256 threads.
__global__ __launch_bounds__(256) void weight_taps_transposed_kernel(const float *__restrict__ W, int N, int taps, int C,
                                                                     __bf16 *__restrict__ out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, t = blockIdx.z;
    const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const long long ldw = (long long)taps * C, ldo = (long long)taps * N;
#pragma unroll
    for (int j = ty; j < 32; j += 8)
        tile[j][tx] = (n0 + j < N && c0 + tx < C) ? W[(long long)(n0 + j) * ldw + (long long)t * C + c0 + tx] : 0.f;
    __syncthreads();
#pragma unroll
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && n0 + tx < N)
            out[(long long)(c0 + j) * ldo + (long long)(taps - 1 - t) * N + n0 + tx] = (__bf16)tile[tx][j];
}

// The same walk with EIGHT row-steps in flight and many more, shorter row chunks (64 rows: the launch of a 31808 x 512 bf16
// matrix was 124 workgroups with 16 KB in flight each -- 34 us for 32 MB), the chunk's sums written side by side
// (partials[chunk][cols]) for colsum_fold_kernel instead of added onto `out` with float atomics.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void colsum_part_kernel(const T *__restrict__ x, long long ld, int rows, int cols,
                                                          int rows_per_block, float *__restrict__ partials) {
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ float part[4][64 * VEC];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * VEC;
    const int rbeg = blockIdx.y * rows_per_block, rend = min(rows, rbeg + rows_per_block);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    if (c < cols) {
        const T *col = x + c;
        int r = rbeg + g;
        for (; r + 28 < rend; r += 32) {
            vec_t v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const vec_t *>(col + (long long)(r + 4 * q) * ld);
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] += (float)v[q][i];
        }
        for (; r < rend; r += 4) {
            const vec_t v = *reinterpret_cast<const vec_t *>(col + (long long)r * ld);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] += (float)v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) part[g][i * 64 + lane] = acc[i];
    __syncthreads();
    if (g == 0 && c < cols) {
        float *dst = partials + (long long)blockIdx.y * cols + c;
#pragma unroll
        for (int i = 0; i < VEC; i += 4) {
            f4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                t[e] = (part[0][(i + e) * 64 + lane] + part[1][(i + e) * 64 + lane]) +
                       (part[2][(i + e) * 64 + lane] + part[3][(i + e) * 64 + lane]);
            *reinterpret_cast<f4 *>(dst + i) = t;
        }
    }
}

// out[c] = sum over the nb chunks of partials[chunk][c], 4 columns per lane: 16 groups of 64 lanes take the chunks round-robin
// (4 loads in flight each), their sums meet in LDS in group order -- the same bits whatever the timing.  cols % 4 == 0.
__global__ __launch_bounds__(1024) void colsum_fold_kernel(const float *__restrict__ partials, int nb, int cols,
                                                           float *__restrict__ out) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ f4 red[16][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6, i = blockIdx.x * 64 + lane, c4 = cols >> 2;
    f4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = f4{0.f, 0.f, 0.f, 0.f};
    if (i < c4) {
        int b = rg;
        for (; b + 48 < nb; b += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += reinterpret_cast<const f4 *>(partials + (long long)(b + 16 * u) * cols)[i];
        }
        for (; b < nb; b += 16) a[0] += reinterpret_cast<const f4 *>(partials + (long long)b * cols)[i];
    }
    red[rg][lane] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (rg == 0 && i < c4) {
        f4 t = red[0][lane];
#pragma unroll
        for (int g = 1; g < 16; ++g) t += red[g][lane];
        reinterpret_cast<f4 *>(out)[i] = t;
    }
}

inline int colsum_part_rows(int rows) {         // rows per chunk of the partials form: 64, more beyond 1024 chunks
    int rpb = 64;
    if ((rows + rpb - 1) / rpb > 1024) rpb = (((rows + 1023) / 1024) + 3) & ~3;
    return rpb;
}

template <typename T>
bool colsum_part_ok(const T *x, long long ld, int cols, const float *out, const float *partials) {
    constexpr int VEC = 16 / (int)sizeof(T);
    return partials && ld % VEC == 0 && cols % VEC == 0 &&
           ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(partials)) & 15) == 0;
}

template <typename T>
int colsum_impl(const T *x, long long ld, int rows, int cols, float *out, float *partials, void *stream) {
    if (!x || !out || rows <= 0 || cols <= 0) return PIKA_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (colsum_part_ok(x, ld, cols, out, partials)) {
        constexpr int VEC = 16 / (int)sizeof(T);
        const int rpb = colsum_part_rows(rows), chunks = (rows + rpb - 1) / rpb;
        hipLaunchKernelGGL((colsum_part_kernel<T, VEC>), dim3((cols + 64 * VEC - 1) / (64 * VEC), chunks), dim3(256), 0, s, x, ld,
                           rows, cols, rpb, partials);
        hipLaunchKernelGGL(colsum_fold_kernel, dim3((cols / 4 + 63) / 64), dim3(1024), 0, s, partials, chunks, cols, out);
        return (int)hipGetLastError();
    }
    hipError_t e = hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    constexpr int VEC = 16 / (int)sizeof(T);
    const int chunks = max(1, min(256, rows / 256));
    const int rpb = (rows + chunks - 1) / chunks;
    if (ld % VEC == 0 && cols % VEC == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        hipLaunchKernelGGL((colsum_vec_kernel<T, VEC>), dim3((cols + 64 * VEC - 1) / (64 * VEC), chunks),
                           dim3(256), 0, s, x, ld, rows, cols, rpb, out);
    else
        hipLaunchKernelGGL(colsum_kernel<T>, dim3((cols + 63) / 64, chunks), dim3(256), 0, s, x, ld,
                           rows, cols, rpb, out);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void col2im_kernel(const float *__restrict__ dcol,
                                                     float *__restrict__ dx, int B, int t_out,
                                                     int t_in, int C, int taps, int stride, int dil,
                                                     int pad) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * t_in * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int ti = (int)((idx / C) % t_in);
    const int b = (int)(idx / ((long long)C * t_in));
    float s = 0.f;
    for (int tap = 0; tap < taps; ++tap) {
        const int num = ti + pad - tap * dil;
        if (num < 0 || num % stride) continue;
        const int t = num / stride;
        if (t >= t_out) continue;
        s += dcol[((long long)b * t_out + t) * ((long long)taps * C) + (long long)tap * C + c];
    }
    dx[idx] = s;
}


// bf16 term split of an f32 operand for the K-concatenated products (include/pika_ops.h): every thread takes 8
// consecutive source columns (two 16-byte loads), computes the terms t0 = bf16(x), t1 = bf16(x - t0),
// t2 = bf16(x - t0 - t1) (exact: 8 + 8 + 8 mantissa bits) and writes the NSEG 16-byte segments of its granule, segment s
// holding term (pattern >> 2s) & 3.  seg_stride = elements between consecutive segments of one source element; pad
// columns [C, Cp) of the concat layout are zero-filled by the threads that own them.
struct SplitJob {
    const float *x; int t_in, C, Cp; long long batch_stride, ld; unsigned pattern;
    long long seg_stride, dst_batch, dst_ld, n_gran; __bf16 *dst; int f16_role;
};

template <int NSEG>
__device__ inline void split_terms_body(const float *__restrict__ x, int t_in, int C, int Cp,
                                        long long batch_stride, long long ld, unsigned pattern,
                                        long long seg_stride, long long dst_batch,
                                        long long dst_ld, long long n_gran,
                                        __bf16 *__restrict__ dst, int f16_role, long long g) {
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    if (g >= n_gran) return;
    const int gpr = Cp >> 3;                       // granules per destination row
    const long long row = g / gpr;
    const int c = (int)(g - row * gpr) << 3;
    const long long b = row / t_in;
    const int t = (int)(row - b * t_in);
    bf8 term[3];
    if (c < C) {                                   // C % 8 == 0: a granule is all source or all padding
        const float4 *src = reinterpret_cast<const float4 *>(x + b * batch_stride + (long long)t * ld + c);
        const float4 v0 = src[0], v1 = src[1];
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (f16_role) {
            // two fp16 terms x = hi + lo (22 mantissa bits) for ONE accumulator: the three segments of the two sides are
            //   A: [hi | lo * 32 | hi / 64]      B: [hi | hi / 32 | lo * 64]     -> hi.hi + lo_a.hi_b + hi_a.lo_b
            // the power-of-two factors keep the small terms out of fp16's subnormal range (lo ~ 2^-12 |x|) and cancel in
            // every product; inputs saturate at +-65504
            h8 t0, t1, t2;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float x_ = fminf(fmaxf(v[i], -65504.f), 65504.f);
                const _Float16 h = (_Float16)x_;
                const float lo = x_ - (float)h;        // exact in fp32
                t0[i] = h;
                if (f16_role == 3) {        // the pair of pika_attention_infer_f16x2: x = hi + 2^-11 lo'
                    t1[i] = (_Float16)(lo * 2048.f);
                    t2[i] = (_Float16)0.f;
                    continue;
                }
                t1[i] = f16_role == 1 ? (_Float16)(lo * 32.f) : (_Float16)((float)h * (1.f / 32.f));
                t2[i] = f16_role == 1 ? (_Float16)((float)h * (1.f / 64.f)) : (_Float16)(lo * 64.f);
            }
            term[0] = __builtin_bit_cast(bf8, t0);
            term[1] = __builtin_bit_cast(bf8, t1);
            term[2] = __builtin_bit_cast(bf8, t2);
        } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const __bf16 h = (__bf16)v[i];
            const float r1 = v[i] - (float)h;      // exact in fp32
            const __bf16 m = (__bf16)r1;
            term[0][i] = h;
            term[1][i] = m;
            term[2][i] = (__bf16)(r1 - (float)m);  // exact: what is left fits 8 bits
        }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) term[k][i] = (__bf16)0.f;
    }
    __bf16 *o = dst + b * dst_batch + (long long)t * dst_ld + c;
#pragma unroll
    for (int s2 = 0; s2 < NSEG; ++s2) {
        const unsigned k = (pattern >> (2 * s2)) & 3u;
        *reinterpret_cast<bf8 *>(o + s2 * seg_stride) = k == 0 ? term[0] : (k == 1 ? term[1] : term[2]);
    }
}

template <int NSEG>
__global__ __launch_bounds__(256) void split_terms_kernel(const float *__restrict__ x, int t_in, int C, int Cp,
                                                          long long batch_stride, long long ld, unsigned pattern,
                                                          long long seg_stride, long long dst_batch,
                                                          long long dst_ld, long long n_gran,
                                                          __bf16 *__restrict__ dst, int f16_role) {
    split_terms_body<NSEG>(x, t_in, C, Cp, batch_stride, ld, pattern, seg_stride, dst_batch, dst_ld, n_gran, dst, f16_role,
                           (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

// BOTH operands of a K-concatenated product in one launch (a weight's split is a 5-us launch of its own otherwise, 56 of
// them in a train step): the first blocks_a workgroups take job a, the rest job b.
template <int NSEG>
__global__ __launch_bounds__(256) void split_terms2_kernel(SplitJob a, SplitJob b, unsigned blocks_a) {
    if (blockIdx.x < blocks_a)
        split_terms_body<NSEG>(a.x, a.t_in, a.C, a.Cp, a.batch_stride, a.ld, a.pattern, a.seg_stride, a.dst_batch, a.dst_ld,
                               a.n_gran, a.dst, a.f16_role, (long long)blockIdx.x * blockDim.x + threadIdx.x);
    else
        split_terms_body<NSEG>(b.x, b.t_in, b.C, b.Cp, b.batch_stride, b.ld, b.pattern, b.seg_stride, b.dst_batch, b.dst_ld,
                               b.n_gran, b.dst, b.f16_role, (long long)(blockIdx.x - blocks_a) * blockDim.x + threadIdx.x);
}

// argument checks + derived quantities of one split (pika_split_bf16_terms); nseg out
int make_split_job(const float *x, int n_batch, int t_in, int C, long long batch_stride, long long ld, int role, int n_terms,
                   int layout, int Cp, void *dst, SplitJob &j, int &nseg) {
    if (!x || !dst || n_batch <= 0 || t_in <= 0 || C <= 0 || (role != 0 && role != 1)) return PIKA_EINVAL;
    if (n_terms != 2 && n_terms != 3 && n_terms != 4) return PIKA_EINVAL;
    if (n_terms == 4 && layout == PIKA_SPLIT_STACK) return PIKA_EINVAL;
    if (layout != PIKA_SPLIT_CONCAT && layout != PIKA_SPLIT_STACK && layout != PIKA_SPLIT_PAIR) return PIKA_EINVAL;
    if (layout == PIKA_SPLIT_PAIR && n_terms != 2 && n_terms != 4) return PIKA_EINVAL;
    if ((C & 7) || (ld & 3) || (batch_stride & 3) || Cp < C || (Cp & 7) || (layout == PIKA_SPLIT_STACK && Cp != C))
        return PIKA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dst)) & 15) return PIKA_EINVAL;
    const long long rows = (long long)n_batch * t_in;
    const long long n_gran = rows * (Cp >> 3);
    if ((n_gran + 255) / 256 > 0x3fffffffLL) return PIKA_ETOOBIG;
    nseg = layout == PIKA_SPLIT_PAIR ? 2 : (n_terms == 3 ? 6 : 3);
    const int f16_role = n_terms == 4 ? (layout == PIKA_SPLIT_PAIR ? 3 : 1 + role) : 0;
    long long seg_stride, dst_batch, dst_ld;
    if (layout == PIKA_SPLIT_PAIR) { dst_ld = Cp; dst_batch = (long long)t_in * dst_ld; seg_stride = rows * dst_ld; }
    else if (layout == PIKA_SPLIT_CONCAT) { dst_ld = (long long)nseg * Cp; dst_batch = (long long)t_in * dst_ld; seg_stride = Cp; }
    else { dst_ld = Cp; dst_batch = (long long)t_in * dst_ld; seg_stride = rows * dst_ld; }
    // term index per segment, two bits each (segment 0 in the low bits); the pairs (A side, B side) of one segment
    // are the products kept: two terms  h.h + l.h + h.l;  three terms  h.h + h.m + m.h + h.l + l.h + m.m
    static const unsigned pat[2][2] = {{0u | 1u << 2 | 0u << 4, 0u | 0u << 2 | 1u << 4},
                                       {0u | 0u << 2 | 1u << 4 | 0u << 6 | 2u << 8 | 1u << 10,
                                        0u | 1u << 2 | 0u << 4 | 2u << 6 | 0u << 8 | 1u << 10}};
    const unsigned pattern = layout == PIKA_SPLIT_PAIR ? (0u | 1u << 2)
                             : (n_terms == 4 ? (0u | 1u << 2 | 2u << 4) : pat[n_terms - 2][role]);
    j = SplitJob{x, t_in, C, Cp, batch_stride, ld, pattern, seg_stride, dst_batch, dst_ld, n_gran, static_cast<__bf16 *>(dst),
                 f16_role};
    return 0;
}

}  // namespace

static const unsigned *g_dropout_salt = nullptr;
const unsigned *pika_internal_dropout_salt() { return g_dropout_salt; }

extern "C" {

int pika_set_dropout_salt(const unsigned *device_word) {
    g_dropout_salt = device_word;
    return 0;
}

int pika_transpose_cast(const pika_operand_t *X, int rows, int K, void *out, long long ld_out,
                        int out_dtype, void *stream) {
    if (!X || !X->ptr || !out || rows <= 0 || K <= 0 || ld_out < rows) return PIKA_EINVAL;
    if (X->rows_per_batch <= 0 || X->C <= 0 || X->stride <= 0) return PIKA_EINVAL;
    Op o{static_cast<const char *>(X->ptr), X->dtype, X->rows_per_batch, X->t_in, X->batch_stride,
         X->ld, X->C, X->stride, X->dil, X->pad};
    dim3 grid((K + 63) / 64, (unsigned)((ld_out + 63) / 64));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_dtype == PIKA_F32)
        hipLaunchKernelGGL(transpose_cast_kernel<float>, grid, dim3(256), 0, s, o, rows, K,
                           static_cast<float *>(out), ld_out);
    else if (out_dtype == PIKA_BF16)
        hipLaunchKernelGGL(transpose_cast_kernel<__bf16>, grid, dim3(256), 0, s, o, rows, K,
                           static_cast<__bf16 *>(out), ld_out);
    else
        return PIKA_EINVAL;
    return (int)hipGetLastError();
}

int pika_weight_taps_transposed_bf16(const float *W, int N, int taps, int C, void *out, void *stream) {
    if (!W || !out || N <= 0 || taps <= 0 || C <= 0 || taps > 65535) return PIKA_EINVAL;
    hipLaunchKernelGGL(weight_taps_transposed_kernel, dim3((C + 31) / 32, (N + 31) / 32, taps), dim3(256), 0,
                       static_cast<hipStream_t>(stream), W, N, taps, C, static_cast<__bf16 *>(out));
    return (int)hipGetLastError();
}

int pika_colsum(const float *x, long long ld, int rows, int cols, float *out, float *partials, void *stream) {
    return colsum_impl<float>(x, ld, rows, cols, out, partials, stream);
}

int pika_colsum_bf16(const void *x, long long ld, int rows, int cols, float *out, float *partials, void *stream) {
    return colsum_impl<__bf16>(static_cast<const __bf16 *>(x), ld, rows, cols, out, partials, stream);
}

long long pika_colsum_partial_floats(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    const int rpb = colsum_part_rows(rows);
    return (long long)((rows + rpb - 1) / rpb) * cols;
}

int pika_col2im(const float *dcol, float *dx, int B, int t_out, int t_in, int C, int taps,
                int stride, int dil, int pad, void *stream) {
    if (!dcol || !dx || B <= 0 || t_out <= 0 || t_in <= 0 || C <= 0 || taps <= 0 || stride <= 0)
        return PIKA_EINVAL;
    const long long total = (long long)B * t_in * C;
    hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dcol, dx, B, t_out, t_in, C, taps, stride,
                       dil, pad);
    return (int)hipGetLastError();
}

int pika_split_bf16_terms(const float *x, int n_batch, int t_in, int C, long long batch_stride, long long ld,
                          int role, int n_terms, int layout, int Cp, void *dst, void *stream) {
    SplitJob j;
    int nseg = 0;
    const int rc = make_split_job(x, n_batch, t_in, C, batch_stride, ld, role, n_terms, layout, Cp, dst, j, nseg);
    if (rc) return rc;
    const dim3 grid((unsigned)((j.n_gran + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (nseg == 2)
        hipLaunchKernelGGL(split_terms_kernel<2>, grid, dim3(256), 0, s, j.x, j.t_in, j.C, j.Cp, j.batch_stride, j.ld, j.pattern,
                           j.seg_stride, j.dst_batch, j.dst_ld, j.n_gran, j.dst, j.f16_role);
    else if (nseg == 3)
        hipLaunchKernelGGL(split_terms_kernel<3>, grid, dim3(256), 0, s, j.x, j.t_in, j.C, j.Cp, j.batch_stride, j.ld, j.pattern,
                           j.seg_stride, j.dst_batch, j.dst_ld, j.n_gran, j.dst, j.f16_role);
    else
        hipLaunchKernelGGL(split_terms_kernel<6>, grid, dim3(256), 0, s, j.x, j.t_in, j.C, j.Cp, j.batch_stride, j.ld, j.pattern,
                           j.seg_stride, j.dst_batch, j.dst_ld, j.n_gran, j.dst, j.f16_role);
    return (int)hipGetLastError();
}

int pika_split_bf16_terms2(const pika_split_job_t *a, const pika_split_job_t *b, void *stream) {
    if (!a || !b || a->n_terms != b->n_terms || a->layout != b->layout) return PIKA_EINVAL;
    SplitJob ja, jb;
    int na = 0, nb = 0;
    int rc = make_split_job(a->x, a->n_batch, a->t_in, a->C, a->batch_stride, a->ld, a->role, a->n_terms, a->layout, a->Cp,
                            a->dst, ja, na);
    if (!rc) rc = make_split_job(b->x, b->n_batch, b->t_in, b->C, b->batch_stride, b->ld, b->role, b->n_terms, b->layout, b->Cp,
                                 b->dst, jb, nb);
    if (rc) return rc;
    if (na != nb) return PIKA_EINVAL;
    const unsigned ga = (unsigned)((ja.n_gran + 255) / 256), gb = (unsigned)((jb.n_gran + 255) / 256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (na == 2) hipLaunchKernelGGL(split_terms2_kernel<2>, dim3(ga + gb), dim3(256), 0, s, ja, jb, ga);
    else if (na == 3) hipLaunchKernelGGL(split_terms2_kernel<3>, dim3(ga + gb), dim3(256), 0, s, ja, jb, ga);
    else hipLaunchKernelGGL(split_terms2_kernel<6>, dim3(ga + gb), dim3(256), 0, s, ja, jb, ga);
    return (int)hipGetLastError();
}

}  // extern "C"
