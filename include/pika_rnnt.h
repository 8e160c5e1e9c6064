/*
 * include/pika_rnnt.h -- C ABI of the MI355X-native RNN-T loss (libpika_amd.so).
 *
 * Replaces, for PIKA's RNN-T training path, the third-party `warp_rnnt` CUDA
 * extension that the reference binds at
 *   /root/reference/trainer/train_transducer_bmuf_otfaug.py:25,58,97-99
 *   /root/reference/trainer/train_transducer_mbr_bmuf_otfaug.py:23,64,157-159
 * (`RNNTLoss(blank=0, reduction='sum').apply(log_probs, labels, frames_lengths,
 *   labels_lengths)` -> per-utterance costs, differentiable w.r.t. log_probs).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory
 *     owned by the caller; the library never allocates or frees device memory;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and is stream-ordered: nothing synchronises the host;
 *   - thread-safe per stream (no global mutable state);
 *   - return value: 0 on success, a negative PIKA_E* code for bad arguments,
 *     a positive hipError_t value if a launch failed.
 *
 * Tensor contract:
 *   log_probs      f32 (B,T,U1,V) contiguous, already log-softmaxed
 *   labels         i32 (B,U1-1)   entries >= labels_lengths[n] are padding and never read
 *   frames_lengths i32 (B,)       T_n, clamped on device to [1,T]
 *   labels_lengths i32 (B,)       U_n, clamped on device to [0,U1-1]
 *   costs          f32 (B,)       -log P(y_n | x_n)
 *   grads          f32 (B,T,U1,V) d(sum_n grad_costs[n]*cost_n)/d log_probs, DENSE:
 *                                 every element is written (zeros included), so the
 *                                 caller may hand over uninitialised memory.
 */
#ifndef PIKA_RNNT_H
#define PIKA_RNNT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIKA_OK 0
#define PIKA_EINVAL (-1)   /* null pointer / non-positive dimension / blank out of range */
#define PIKA_ETOOBIG (-2)  /* U1 > 1024 (one workgroup spans the label axis) */

/* Library/ABI version, bumped on any signature change. */
int pika_amd_abi_version(void);

/* Bytes of device scratch `workspace` needed by the calls below for a (B,T,U1)
 * lattice batch.  Layout (DESIGN.md "RNNT loss / HBM layout"): four skewed
 * f32 planes [B][T+U1-1][W] (W = workgroup width covering U1, a multiple of 64) (blank log-prob, emit log-prob, alpha,
 * beta) + per-utterance log-likelihoods. */
size_t pika_rnnt_workspace_bytes(int B, int T, int U1);

/* Forward: gathers the two log-probs each lattice cell needs, runs the alpha and
 * beta anti-diagonal recurrences (one wavefront per direction per utterance),
 * writes costs[B] and leaves the lattice in `workspace` for the backward call. */
int pika_rnnt_loss_forward(const float *log_probs, const int *labels,
                           const int *frames_lengths, const int *labels_lengths,
                           int B, int T, int U1, int V, int blank,
                           float *costs, void *workspace, void *stream);

/* Backward: one streaming pass that writes the dense gradient tensor.
 * `grad_costs` (B,) f32 scales utterance n's gradient (autograd's grad_output);
 * NULL means all ones.  `workspace` must be the buffer filled by the matching
 * forward call (same B,T,U1 and lengths).
 * grads == NULL: only the per-row metadata (the at most two non-zeros of every V-row, already scaled by
 * grad_costs) is left in `workspace`; pika_rnnt_loss_dense_grads writes the dense tensor from it later, and
 * pika_rnnt_dlogits_compact_bf16 needs nothing else. */
int pika_rnnt_loss_backward(const int *labels, const int *frames_lengths,
                            const int *labels_lengths,
                            int B, int T, int U1, int V, int blank,
                            const float *grad_costs, const void *workspace,
                            float *grads, void *stream);

/* The streaming pass of pika_rnnt_loss_backward on its own: dense (B,T,U1,V) gradient from the row metadata a
 * pika_rnnt_loss_backward call (with or without `grads`) left in `workspace`. */
int pika_rnnt_loss_dense_grads(const void *workspace, int B, int T, int U1, int V, int blank,
                               float *grads, void *stream);

/* warp_rnnt-shaped one-shot: forward + backward with unit grad_costs. */
int pika_rnnt_loss_fwd_bwd(const float *log_probs, const int *labels,
                           const int *frames_lengths, const int *labels_lengths,
                           int B, int T, int U1, int V, int blank,
                           float *costs, float *grads, void *workspace, void *stream);

/* Test/diagnostic hook: un-skews the lattice held in `workspace` into dense
 * (B,T,U1) f32 planes (invalid cells = -1e30).  Either output may be NULL. */
int pika_rnnt_export_lattice(const void *workspace, const int *frames_lengths,
                             const int *labels_lengths, int B, int T, int U1,
                             float *alphas, float *betas, void *stream);

/* For a producer that computed log_probs = log_softmax(scale * logits) itself (the joint network,
 * trainer/model/transducer.py:108-111): d(loss)/d(logits) as a bf16 matrix (rows = B*T*U1, pitch ld_out,
 * columns [V, ld_out) zero) straight from the two non-zeros per row that the matching
 * pika_rnnt_loss_backward call left in `workspace` -- the dense gradient (B,T,U1,V) need not be read back,
 * nor even written (grads == NULL above):
 *   out[r, v] = scale * (grad[r, v] - exp(log_probs[r, v]) * sum_v' grad[r, v']).
 * colsum (V floats, may be NULL) receives sum_r out[r, :] before the bf16 rounding: the bias gradient of the
 * layer that produced the logits, without another pass over `out`.
 * lse (rows floats) non-NULL: `log_probs` holds the RAW logits and lse their per-row log-sum-exp (as written by
 * pika_rnnt_fused_forward) -- the log-probabilities never have to exist (scale must then be 1).
 * V % 4 == 0, V <= 8192, ld_out % 4 == 0 (a wave covers a row in 64 x 4 x 20 columns up to 5120, x 32 beyond). */
int pika_rnnt_dlogits_compact_bf16(const float *log_probs, const float *lse, const void *workspace, int B, int T,
                                   int U1, int V, int blank, void *out, long long ld_out, float scale,
                                   float *colsum, void *stream);

/* pika_rnnt_fused_forward with the row log-sum-exp taken from partial statistics instead of a pass over the logits:
 * pmax / psum (rows, n_part) as written by pika_gemm_bf16_nt_lse (pika_gemm.h) for the SAME logits.  Reads
 * 8*n_part bytes + two 64-byte sectors per lattice cell instead of 4*V bytes. */
int pika_rnnt_fused_forward_partials(const float *logits, const float *pmax, const float *psum, int n_part,
                                     const int *labels, const int *frames_lengths, const int *labels_lengths, int B,
                                     int T, int U1, int V, int blank, float *costs, float *lse, void *workspace,
                                     void *stream);

/* pika_rnnt_fused_forward_partials for the 16-bit logits of pika_gemm_bf16_nt_lse_f16 (pika_gemm.h): logits16 is the
 * fp16 matrix (row pitch ld16 halves), gathered / g_labels / g_blank what that product's epilogue left in fp32 -- per
 * lattice row the logits of column g_blank and of the row's label g_labels[b][u].  A lattice cell whose label (blank) is
 * the gathered one takes the fp32 value, so costs and log-sum-exp are those of the fp32 logits; any other label reads the
 * fp16 logit.  g_labels may be NULL (nothing gathered for labels). */
int pika_rnnt_fused_forward_gathered(const void *logits16, long long ld16, const float *gathered, const int *g_labels,
                                     int g_blank, const float *pmax, const float *psum, int n_part, const int *labels,
                                     const int *frames_lengths, const int *labels_lengths, int B, int T, int U1, int V,
                                     int blank, float *costs, float *lse, void *workspace, void *stream);

/* pika_rnnt_dlogits_compact_bf16 reading the RAW logits as fp16 (row pitch ld_in halves; lse = their row log-sum-exp from
 * the forward call above, required): d(logits) = scale * (g - softmax * sum(g)) as one bf16 matrix + column sums.
 * gathered (optional; with g_labels / g_blank as in pika_rnnt_fused_forward_gathered: the (rows, 2) f32 blank and label
 * logits of every row): the softmax of those two columns -- the only entries that also carry the loss' own gradient terms -- is then taken from
 * the fp32 values instead of the fp16 copy. */
int pika_rnnt_dlogits_compact_bf16_f16in(const void *logits16, long long ld_in, const float *lse, const void *workspace,
                                         int B, int T, int U1, int V, int blank, void *out, long long ld_out, float scale,
                                         float *colsum, const float *gathered, const int *g_labels, int g_blank,
                                         void *stream);

/* Fused boundary logits -> (costs, d loss / d logits)  (SURVEY.md 8d M1'): replaces
 * F.log_softmax (trainer/model/transducer.py:111) + the loss + the log-softmax backward for a caller that owns
 * the joint output.  `logits` (B,T,U1,V) f32 are the RAW fc2 outputs, V % 4 == 0, V <= 8192; lse (B*T*U1) f32
 * receives the per-row log-sum-exp and must be handed to the backward together with the same logits and
 * workspace.  grad_logits: f32 (out_dtype 0, ld_out >= V) or bf16 (out_dtype 1; columns [V, ld_out) zeroed).
 * Traffic 3 x B*T*U1*V*4 bytes (one read for lse + gather, one read + one write for the gradient) against
 * 6 x for the three separate passes. */
int pika_rnnt_fused_forward(const float *logits, const int *labels, const int *frames_lengths,
                            const int *labels_lengths, int B, int T, int U1, int V, int blank, float *costs,
                            float *lse, void *workspace, void *stream);
int pika_rnnt_fused_backward(const float *logits, const float *lse, const int *labels,
                             const int *frames_lengths, const int *labels_lengths, int B, int T, int U1, int V,
                             int blank, const float *grad_costs, const void *workspace, void *grad_logits,
                             int out_dtype, long long ld_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_RNNT_H */
