/*
 * include/pika_ops.h -- C ABI of the data-movement / reduction helpers around the GEMM.
 * Conventions as in pika_rnnt.h.  These serve the backward passes of the model layers
 * (autograd of the reference model files under /root/reference/trainer/model, run on the GPU).
 */
#ifndef PIKA_OPS_H
#define PIKA_OPS_H

#include "pika_gemm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* out[k][r] = X(r,k) for r < rows, k < K, where X is a (possibly virtual time-delay) operand as
 * in pika_gemm.h; columns r in [rows, ld_out) are zero-filled so that `ld_out` (a multiple of 4)
 * can serve as the reduction length of the weight-gradient GEMM dW = dY^T X.
 * out_dtype: PIKA_F32 or PIKA_BF16. */
int pika_transpose_cast(const pika_operand_t *X, int rows, int K, void *out, long long ld_out,
                        int out_dtype, void *stream);

/* out (C, taps * N) bf16, out[c][(taps - 1 - t) * N + n] = W[n][t * C + c] for W (N, taps * C) f32 contiguous: the weight
 * operand of a layer's d(input) product -- the time-delay layers' transposed convolution (taps reversed;
 * /root/reference/trainer/model/encoder/tdnn.py via nn.Conv1d's backward) and, at taps = 1, plain W^T -- in one launch. */
int pika_weight_taps_transposed_bf16(const float *W, int N, int taps, int C, void *out, void *stream);

/* out[c] = sum_r x[r][c]  (bias gradients).  x (rows, cols) f32 with pitch ld.
 * partials: device scratch of pika_colsum_partial_floats(rows, cols) floats (16-byte aligned) or NULL.  With it (and 16-byte
 * aligned x / out, ld and cols multiples of 4 f32 / 8 bf16) chunks of 64 rows are summed side by side and folded by a
 * second launch in a fixed order: no atomics, no memset, the same bits every time (31808 x 512 bf16: 34 -> ~12 us);
 * otherwise float atomics onto `out`, which the call zeroes first. */
int pika_colsum(const float *x, long long ld, int rows, int cols, float *out, float *partials, void *stream);
/* the same over a bf16 matrix (fp32 accumulation) */
int pika_colsum_bf16(const void *x, long long ld, int rows, int cols, float *out, float *partials, void *stream);
long long pika_colsum_partial_floats(int rows, int cols);

/* Adjoint of the virtual time-delay operand: dx[b,ti,c] = sum over (t,tap) with
 * t*stride + tap*dil - pad == ti of dcol[(b,t)][tap*C + c].  dx (B,t_in,C) contiguous f32 is
 * fully overwritten (deterministic gather, no atomics). */
int pika_col2im(const float *dcol, float *dx, int B, int t_out, int t_in, int C, int taps,
                int stride, int dil, int pad, void *stream);

/* bf16 term split of an f32 operand for K-concatenated products on the bf16 kernels:
 *   t0 = bf16(x), t1 = bf16(x - t0), t2 = bf16(x - t0 - t1)      (x = t0 + t1 + t2 exactly: 8 + 8 + 8 mantissa bits)
 * A product A B^T of two split operands is taken by ONE bf16 GEMM over a longer reduction whose segments pair the
 * terms to keep.  The A side (role 0) and the B side (role 1) lay their segments out as
 *   n_terms = 2 ("bf16x3"):   A [t0 | t1 | t0]                 B [t0 | t0 | t1]
 *       -> t0.t0 + t1.t0 + t0.t1; the dropped t1.t1 is below 2^-16 of the leading product: ~1e-5 relative, inside
 *       the 1e-3 the parity statement of the training path asks for, at a third of the bf16 rate;
 *   n_terms = 3 (fp32-exact):  A [t0 | t0 | t1 | t0 | t2 | t1]   B [t0 | t1 | t0 | t2 | t0 | t1]
 *       -> the six products PIKA_GEMM_FP32SPLIT issues (everything above 2^-24 of the leading one), at a sixth of
 *       the bf16 rate of the direct-to-LDS kernels instead of a sixth of the register-staged kernel's.
 *   n_terms = 4 (two FP16 terms, PIKA_SPLIT_CONCAT only; the product must be issued with PIKA_GEMM_F16_OPERANDS):
 *       x = hi + lo, hi = fp16(x), 22 mantissa bits;   A [hi | 32 lo | hi / 64]   B [hi | hi / 32 | 64 lo]
 *       -> hi.hi + lo.hi + hi.lo in ONE accumulator: the power-of-two factors cancel in every product and keep the
 *       small terms out of fp16's subnormal range.  An fp32 product to ~2^-22 at a third of the 16-bit rate (the
 *       three-term bf16 split: exact, a sixth).  Inputs saturate at +-65504.
 * Source: n_batch blocks (batch_stride apart) of t_in rows (pitch ld) x C columns, f32, C % 8 == 0.  S = 3 or 6
 * segments.
 *   PIKA_SPLIT_CONCAT: dst (n_batch, t_in, S*Cp) bf16, segment s of a row at columns [s*Cp, s*Cp + C), the pad
 *       columns up to Cp zero (Cp >= C, Cp % 8 == 0; Cp % 64 == 0 keeps the direct-to-LDS kernels applicable):
 *       for operands whose reduction index is the contiguous one; a time-delay view over dst has S*Cp channels.
 *   PIKA_SPLIT_STACK: dst (S, n_batch, t_in, C) bf16, segment s = block s: for `trans` operands, whose reduction
 *       runs over the rows (Cp must equal C).
 *   PIKA_SPLIT_PAIR (n_terms = 2, role ignored): dst (2, n_batch, t_in, Cp) bf16, plane 0 = t0 ("hi"), plane 1 = t1
 *       ("lo"), pad columns zero: the two-term A operand of pika_gemm_bf16_ex (pika_operand_t.seg = Cp,
 *       lo_off = n_batch * t_in * Cp).  With n_terms = 4: fp16 planes hi = fp16(x), lo' = fp16((x - hi) 2^11) -- the
 *       operands of pika_attention_infer_f16x2 (pika_attn.h) straight from a packed fp32 [q | k | v] projection. */
#define PIKA_SPLIT_CONCAT 0
#define PIKA_SPLIT_STACK 1
#define PIKA_SPLIT_PAIR 2
int pika_split_bf16_terms(const float *x, int n_batch, int t_in, int C, long long batch_stride, long long ld,
                          int role, int n_terms, int layout, int Cp, void *dst, void *stream);

/* The splits of BOTH operands of one product in a single launch (same n_terms and layout; the fields are the arguments of
 * pika_split_bf16_terms): a weight's split is otherwise a launch of ~5 us of its own, 56 of them in a train step. */
typedef struct {
    const float *x;
    int n_batch, t_in, C;
    long long batch_stride, ld;
    int role, n_terms, layout, Cp;
    void *dst;
} pika_split_job_t;
int pika_split_bf16_terms2(const pika_split_job_t *a, const pika_split_job_t *b, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_OPS_H */
