/*
 * include/pika_norm.h -- C ABI of the BatchNorm kernels of the TDNN encoder.
 *
 * Replaces nn.BatchNorm1d over the (B*T, C) activation matrix in training mode
 * (/root/reference/trainer/model/rnnt_tdnn_transformer.py:76-78,80-82,85: `bn(relu(conv(x)))`,
 * batch statistics over ALL B*T rows, padded frames included).  Conventions as in pika_rnnt.h.
 */
#ifndef PIKA_NORM_H
#define PIKA_NORM_H

#ifdef __cplusplus
extern "C" {
#endif

/* Rows that count (optional, NULL = all): the matrix is B blocks of rows_per_batch rows (one utterance each, row = frame)
 * of which the first (*t_valid - sub) / div hold data and the rest is padding up to a fixed shape -- statistics, the row
 * count of the mean / variance, and the backward then run over the data rows only, padding rows come out as zeros and
 * receive no gradient.  *t_valid is read ON THE DEVICE when the kernels run (a launch sequence captured into a hipGraph for
 * a padded shape serves every batch length that fits it: pika_amd/train_graph.py); sub / div map the valid length of the
 * network input to this layer's (time-delay layers shorten it, the last one strides by 4). */
typedef struct {
    const int *t_valid;      /* device word: valid frames of the network input */
    int rows_per_batch;      /* rows per utterance in THIS matrix (padded); rows % rows_per_batch == 0 */
    int sub, div;            /* valid rows per utterance here = max(0, (*t_valid - sub)) / div, capped at rows_per_batch */
} pika_bn_valid_t;

/* Per-channel sums: stats[0..C) = sum_r x[r][c], stats[C..2C) = sum_r x[r][c]^2 (fp64, zeroed by
 * the call).  x (rows, C) f32 contiguous. */
int pika_bn_stats(const float *x, long long rows, int C, double *stats, const pika_bn_valid_t *valid, void *stream);

/* y = (x - mean) * rstd * gamma + beta with mean/var from `stats` (biased variance, eps inside the
 * sqrt); writes save_mean / save_rstd (C each, for the backward) and, when running_mean != NULL,
 * running = (1-momentum)*running + momentum*{mean, unbiased var}.  y_dtype PIKA_F32 | PIKA_BF16 (pika_gemm.h):
 * bf16 when y only feeds an MFMA product (the next time-delay layer).  y_lo (NULL, or a second bf16 plane of the same
 * shape; needs y_dtype PIKA_BF16): receives bf16(value - y), making (y, y_lo) the two-term operand of a forward product
 * that carries activations to 16 mantissa bits (pika_gemm.h: pika_operand_t.seg). */
int pika_bn_apply(const float *x, long long rows, int C, const double *stats, const float *gamma,
                  const float *beta, float eps, float momentum, float *running_mean,
                  float *running_var, float *save_mean, float *save_rstd, void *y, int y_dtype,
                  void *y_lo, const pika_bn_valid_t *valid, void *stream);

/* Backward, two launches: sums[0..C) = sum dy, sums[C..2C) = sum dy*xhat (fp64); then
 * dx = gamma*rstd*(dy - sum_dy/rows - xhat*sum_dy_xhat/rows), dgamma = sum dy*xhat, dbeta = sum dy.
 * relu_mask != 0: x is the output of a ReLU (the reference computes bn(relu(conv(.)))) and dx is
 * additionally multiplied by (x > 0), i.e. the ReLU backward is folded into this pass.  dy may arrive as bf16
 * (y was produced as bf16) and dx may be written as bf16 (it only feeds the dX / dW products of the layer that
 * produced x). */
int pika_bn_backward(const void *dy, int dy_dtype, const float *x, long long rows, int C, const float *gamma,
                     const float *save_mean, const float *save_rstd, double *sums, void *dx, int dx_dtype,
                     float *dgamma, float *dbeta, int relu_mask, const pika_bn_valid_t *valid, void *stream);

/* nn.LayerNorm over the last dimension of x (rows, C) f32 contiguous (pre-LN transformer layers,
 * /root/reference/trainer/model/transformer.py:85-100, position_ffn.py:36): C % 4 == 0, C <= 2048.
 * y may be written as bf16 (y_dtype PIKA_BF16, pika_gemm.h) when it only feeds MFMA products; mean /
 * rstd (rows each) are kept for the backward, whose incoming gradient may be bf16 as well; y_lo as in pika_bn_apply.
 * dx = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma; dgamma = sum dy*xhat; dbeta = sum dy. */
int pika_layer_norm_fwd(const float *x, long long rows, int C, const float *gamma, const float *beta,
                        float eps, void *y, int y_dtype, void *y_lo, float *mean, float *rstd, void *stream);
int pika_layer_norm_bwd(const void *dy, int dy_dtype, const float *x, long long rows, int C, const float *gamma,
                        const float *mean, const float *rstd, float *dx, float *dgamma, float *dbeta,
                        float *partials, const float *dx_add, void *stream);
/* dx_add: NULL, or (rows, C) f32 added to dx on the way out -- the gradient of the skip connection around the LN
 * (transformer.py:95-99: out = drop(attn(LN(x))) + x), which autograd would otherwise add in a launch of its own.
 * partials: device scratch of pika_layer_norm_bwd_partial_floats(rows, C) floats (16-byte aligned), or NULL.  With it the
 * workgroups of the backward write their column sums side by side and a second launch adds them up in a fixed order:
 * dgamma / dbeta do not depend on timing and cost no atomics (31808 x 512: 95 -> ~60 us); NULL = float atomics onto
 * dgamma / dbeta (which this call zeroes first). */
long long pika_layer_norm_bwd_partial_floats(long long rows, int C);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_NORM_H */
