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

#ifdef __cplusplus
}
#endif
#endif /* PIKA_OPS_H */
