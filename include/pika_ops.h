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

/* out[c] = sum_r x[r][c]  (bias gradients).  x (rows, cols) f32 with pitch ld. */
int pika_colsum(const float *x, long long ld, int rows, int cols, float *out, void *stream);
/* the same over a bf16 matrix (fp32 accumulation) */
int pika_colsum_bf16(const void *x, long long ld, int rows, int cols, float *out, void *stream);

/* Adjoint of the virtual time-delay operand: dx[b,ti,c] = sum over (t,tap) with
 * t*stride + tap*dil - pad == ti of dcol[(b,t)][tap*C + c].  dx (B,t_in,C) contiguous f32 is
 * fully overwritten (deterministic gather, no atomics). */
int pika_col2im(const float *dcol, float *dx, int B, int t_out, int t_in, int C, int taps,
                int stride, int dil, int pad, void *stream);

/* Two-term bf16 split of an f32 operand for "bf16x3" products: x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi),
 * |r| <= 2^-17 |x|.  A product A B^T of two such operands is taken as hi.hi + lo.hi + hi.lo (the dropped lo.lo term
 * is below 2^-16 of the leading one) by ONE bf16 GEMM over a three times longer reduction: the A side lays its
 * segments out as [hi | lo | hi], the B side as [hi | hi | lo]  (role 0 / 1).  This gives products accurate to
 * ~1e-5 relative -- inside the 1e-3 the parity statement of the training path asks for -- on the direct-to-LDS
 * bf16 kernels at a third of their rate, instead of the exact 6-MFMA PIKA_GEMM_FP32SPLIT path at a sixth of the
 * register-staged kernel's.
 * Source: n_batch blocks (batch_stride apart) of t_in rows (pitch ld) x C columns, f32, C % 8 == 0.
 *   PIKA_SPLIT_CONCAT: dst (n_batch, t_in, 3*Cp) bf16, segment s of a row at columns [s*Cp, s*Cp + C), the pad
 *       columns up to Cp zero (Cp >= C, Cp % 8 == 0; Cp % 64 == 0 keeps the direct-to-LDS kernels applicable):
 *       for operands whose reduction index is the contiguous one; a time-delay view over dst has 3*Cp channels.
 *   PIKA_SPLIT_STACK: dst (3, n_batch, t_in, C) bf16, segment s = block s: for `trans` operands, whose reduction
 *       runs over the rows (Cp must equal C). */
#define PIKA_SPLIT_CONCAT 0
#define PIKA_SPLIT_STACK 1
int pika_split_bf16x3(const float *x, int n_batch, int t_in, int C, long long batch_stride, long long ld,
                      int role, int layout, int Cp, void *dst, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_OPS_H */
