/*
 * include/pika_lstm.h -- C ABI of the LSTM recurrence in TRAINING (SURVEY 8a row 8': the prediction network of every
 * shipped recipe, `dec_type=rnn`).
 *
 * Replaces the recurrence inside nn.LSTM as the reference model calls it
 * (/root/reference/trainer/model/transducer.py:55-61 builds it -- unidirectional, batch_first, `dec_layers` layers with
 * dropout between them -- and :93-96 runs it over the padded (B, U+1) label matrix from a zero state), one layer per call:
 *
 *   gx  = x W_ih^T + b_ih + b_hh                     for all steps at once: the caller's product (pika_gemm.h)
 *   [i | f | g | o]_t = gx_t + h_{t-1} W_hh^T         (nn.LSTM's gate order)
 *   c_t = sigmoid(f) c_{t-1} + sigmoid(i) tanh(g),    h_t = sigmoid(o) tanh(c_t),       h_{-1} = c_{-1} = 0
 *
 * and its gradient through time.  The library the reference reaches (cuDNN there, MIOpen on this hardware) runs two
 * launches per step and direction; here one persistent launch per layer and direction of time keeps its slice of W_hh in
 * registers for the whole sequence (csrc/lstm_train.hip).  W_hh enters the products as two bf16 terms (hi.hi + lo.hi +
 * hi.lo, fp32 accumulation: ~2^-17 per product), h and the gate gradients likewise.
 * Conventions as in pika_rnnt.h (caller-owned device memory, stream-ordered, no allocation, return code).
 * H % 256 == 0, H <= 1024; ceil(B / 16) * (H / 16) workgroups must be resident at once (<= the CU count), else PIKA_ETOOBIG.
 */
#ifndef PIKA_LSTM_H
#define PIKA_LSTM_H

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes of the packed recurrent weights (both packings: forward and backward), or -1. */
long long pika_lstm_train_packed_bytes(int H);
/* Bytes of exchange scratch of a (B, S) batch for either launch, or -1.  16-byte aligned.  The backward's is
 * S * ceil(B / 16) * (H / 16)^2 KB (428 MB at B = 32, S = 51, H = 1024): a partial product per (step, consumer, producer). */
long long pika_lstm_train_fwd_work_bytes(int S, int B, int H);
long long pika_lstm_train_bwd_work_bytes(int S, int B, int H);

/* w_hh (4H, H) f32 contiguous (nn.LSTM.weight_hh_l*) -> packed (16-byte aligned).  Once per weight update. */
int pika_lstm_train_pack(const float *w_hh, int H, void *packed, void *stream);

/* gx (B, S, 4H) f32 contiguous.  out (B, S, H) = h_t; gates (B, S, 4H) = the ACTIVATED gates and cells (B, S, H) = c_t
 * are what pika_lstm_train_bwd reads.  The first word of `work` is an error flag (pika_lstm_train_status): non-zero when a
 * workgroup waited ~2 s for a peer that never ran; its outputs are then NaN. */
int pika_lstm_train_fwd(const float *gx, const void *packed, float *out, float *gates, float *cells, void *work,
                        long long work_bytes, int S, int B, int H, void *stream);

/* dy (B, S, H) f32 = d(loss)/d(out).  dgates (B, S, 4H) = d(loss)/d(gx): the gradient of the PRE-activation gates, from
 * which the caller's products take dW_ih = dgates^T x, dW_hh = dgates_t^T h_{t-1} summed over t >= 1, the bias
 * gradients (column sums) and dx = dgates W_ih.
 * armed != 0: the caller vouches that every byte of `work` behind its first 256 is 0xff -- as a memset leaves it, and as
 * every COMPLETED pika_lstm_train_bwd launch leaves it again (a consumer resets the words it has read) -- so the launch
 * skips its own memset of the scratch (0.5 ms at the size above).  armed == 0: the launch fills the scratch itself. */
int pika_lstm_train_bwd(const float *dy, const void *packed, const float *gates, const float *cells, float *dgates,
                        void *work, long long work_bytes, int armed, int S, int B, int H, void *stream);

/* Copies the error flag of the last launch on `work` to host_out (synchronises the stream). */
int pika_lstm_train_status(const void *work, int *host_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
