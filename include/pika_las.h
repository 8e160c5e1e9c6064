/*
 * include/pika_las.h -- C ABI of the per-token kernels of LAS n-best rescoring.
 *
 * Replaces, inside the input-feed decoder loop of the reference rescorer
 * (/root/reference/trainer/model/las.py:649-668, one Python iteration per token and per hypothesis):
 *   - nn.LSTMCell's gate arithmetic (modules/stacked_rnn.py:20-34 stacks the cells), and
 *   - the "mlp" global attention (modules/global_attention.py:162-248: score(h_t, h_s) = v^T tanh(W_q h_t + U_a h_s),
 *     softmax over the source positions, context = sum_s a_s h_s),
 * for ALL n-best hypotheses of a decode batch at once.  The reference materialises tanh(W_q h_t + U_a h_s) as an
 * (N, S, D) tensor per token (1 GB at 1024 hypotheses x 240 positions x 1024) and re-reads the expanded context; here
 * the scores, the softmax and the context sum of a hypothesis never leave the workgroup, and the hypotheses of one
 * utterance share the utterance's (S, D) projections through L2.  Conventions as in pika_rnnt.h (caller-owned device
 * memory, stream-ordered, no allocation, return code).
 */
#ifndef PIKA_LAS_H
#define PIKA_LAS_H

#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* gates (N, 4H) f32 = W_ih x + b_ih + W_hh h + b_hh in nn.LSTMCell order [i | f | g | o] (pitch ldg);
 * c_prev (N, H) contiguous.  c_out (N, H) contiguous (may alias c_prev) = sigmoid(f) c_prev + sigmoid(i) tanh(g);
 * h = sigmoid(o) tanh(c_out) is written to h_out (pitch ldh) and, when h_out2 != NULL, also to h_out2 (pitch ldh2) --
 * the decoder keeps h both as the next layer's input block and as the recurrent block of its own next step.
 * H % 4 == 0, 16-byte aligned rows.  n_dev: NULL, or a device int: only rows < min(N, *n_dev) are processed.
 * rowlist / rowoff_dev: NULL, or a gather list (device int32) and its device offset: launch row e stands for row
 * rowlist[*rowoff_dev + e] of every operand (the rows of a launch are then any subset of the buffers' rows). */
int pika_lstm_cell(const float *gates, long long ldg, const float *c_prev, float *c_out, float *h_out, long long ldh,
                   float *h_out2, long long ldh2, int N, int H, const int *n_dev, const int *rowlist,
                   const int *rowoff_dev, void *stream);

/* "mlp" attention of the queries of a list over the source positions of their utterances; query n = qidx[i]:
 *   align[n, s] = sum_d v[d] tanh(wq[n, d] + proj[owner[n], s, d])          s < lens[owner[n]]  (else excluded)
 *   a[n, :]     = softmax_s align[n, :]
 *   ctx_out[n, :] = sum_s a[n, s] context[owner[n], s, :]
 * wq (., D) f32 pitch ldq = W_q h_t + b_q (the caller's GEMM); proj, context (B, S, D) f32 contiguous = U_a h_s and
 * h_s; owner (.,) int32 utterance of each query; lens (B,) int32 valid positions (>= 1); ctx_out (., D) f32 pitch ldo.
 * D % 4 == 0, D <= 1024, S <= 2048.
 * The query list is ORDERED BY UTTERANCE and the attention runs utterance by utterance (two launches): a workgroup takes
 * all the queries of one utterance and a chunk of 32 of its positions -- the utterance's U_a h_s and h_s rows are read
 * once per workgroup instead of once per query (130 -> 73 us at 470 queries of 64 utterances, S = 240, D = 1024) -- and a
 * second launch merges the chunks' partial (max, sum, context sums) into ctx_out.
 * uoff int32 (B + 1): the queries of utterance b are entries [uoff[b], uoff[b + 1]) of the list (qidx + *qoff_dev, of
 * min(N, *n_dev) entries; one launch captured into a hipGraph serves every token of a pass, N then sizes the grid for the
 * largest step); with step_dev the table of the current step, uoff + *step_dev * (B + 1) (step[0] of
 * pika_las_step_advance).  work: f32 scratch of pika_las_attention_work_floats(N, S, D) floats, 16-byte aligned.
 * (Rounds 3-4 launched one query per workgroup -- pika_las_mlp_attention, removed in ABI 20.) */
size_t pika_las_attention_work_floats(int N, int S, int D);
int pika_las_mlp_attention_by_utterance(const float *wq, long long ldq, const float *proj, const float *context,
                                        const int *owner, const int *lens, const int *qidx, const int *uoff, const float *v,
                                        float *ctx_out, long long ldo, float *work, int N, int B, int S, int D,
                                        const int *n_dev, const int *qoff_dev, const int *step_dev, void *stream);

/* The token loop of a rescoring pass as ONE captured launch sequence replayed once per token: every per-token quantity
 * lives on the device.  step int32[4] = {t, n, qoff, -} (the caller starts it at {-1, 0, 0, 0});
 * pika_las_step_advance: t += 1, n = n_active[t], qoff = qoffs[t] (0 beyond L) -- n is what the m_dev / n_dev
 * parameters of pika_dgemm, pika_lstm_cell and pika_las_mlp_attention_by_utterance point at;
 * pika_las_embed_rows: for e < n, r = rowlist ? rowlist[qoff + e] : e: x0[r, 0:E] = emb[tokens[t, r], :] (the decoder's
 * layer-0 input rows, pitch ldx) and crow[r] = t * N + r (the row of the (L, N, H) result the token's output projection
 * writes through pika_dgemm's crow).  tokens (L, N) int64, E % 4 == 0.  rowlist: the per-step lists of ACTIVE rows
 * (the same lists the attention kernel takes as its query lists), concatenated; step[2] is the offset of step t's. */
int pika_las_step_advance(int *step, const int *n_active, const int *qoffs, int L, void *stream);
/* Prefix sharing: the n-best entries of an utterance that share a token prefix share the decoder rows of that prefix (the
 * reference scores every entry from scratch, decoder/transducer_decoder.py:219-253: same values).  At the step t a
 * hypothesis leaves the shared prefix its row inherits the recurrent state of the row that computed the prefix: for
 * k in [fork_off[t], fork_off[t+1]) and every segment i < nseg (<= PIKA_LAS_FORK_SEGS):
 *   base[i][fork_dst[k], col0[i] : col0[i] + ncols[i]] = base[i][fork_src[k], same columns]   (row pitch ld[i], floats)
 * t = step[0] (after pika_las_step_advance); fork_off int32[L + 1]; max_forks = the largest number of forks of a step
 * (sizes the grid).  Source and destination rows of a step are disjoint.  Columns / pitches multiples of 4.   A step without active rows (step[1] == 0: beyond the pass, when a longer pass shares the
 * replayed launch sequence) copies nothing. */
#define PIKA_LAS_FORK_SEGS 8
int pika_las_fork_rows(const int *step, const int *fork_off, const int *fork_dst, const int *fork_src, int max_forks,
                       int nseg, float *const *base, const long long *ld, const int *col0, const int *ncols, void *stream);
int pika_las_embed_rows(const int *step, const long long *tokens, const float *emb, float *x0, long long ldx,
                        long long *crow, int N, int E, const int *rowlist, void *stream);

/* The rescorer's encoder: one (bi)directional nn.LSTM layer over a padded batch of packed sequences
 * (/root/reference/trainer/model/las.py:44-75: pack_padded_sequence -> nn.LSTM -> pad_packed_sequence), the whole
 * recurrence as ONE persistent launch (pika_amd/csrc/blstm.hip).  D directions (1 or 2: [forward, reverse]), hidden size
 * H per direction (H % 128 == 0, H <= 512), B utterances of lens[b] <= S valid positions.
 *   pika_blstm_pack: w_hh (D, 4H, H) f32 = weight_hh_l{k}[_reverse], gate order [i | f | g | o] -> `packed`
 *     (pika_blstm_packed_bytes(D, H) bytes, 16-byte aligned): two bf16 terms in MFMA fragment order.
 *   pika_blstm_layer: gx (S, B, D*4H) f32 = W_ih x_t + b_ih + b_hh of every position and direction (the caller's GEMM);
 *     out (S, B, D*H) f32 = the layer's output, zeros at positions >= lens[b] (fully written); h_n, c_n (D, B, H) f32 =
 *     the state after each sequence's last position (forward) / first position (reverse).  Recurrent products:
 *     hi.hi + lo.hi + hi.lo over bf16 terms of h and W_hh with fp32 accumulation (an fp32 product to ~2^-17).
 *     work: pika_blstm_work_bytes(S, B, D, H) bytes of device scratch (16-byte aligned; one 16 x H word slot per step and
 *     (direction, 16-row block): 65 MB at S = 250, B = 64, H = 512), no initialisation needed.
 *     The D * ceil(B/16) * H/16 workgroups of the launch exchange hidden states inside the kernel, so all of them must
 *     be resident: PIKA_ETOOBIG when the device has fewer CUs (the caller falls back to its library LSTM).
 *   pika_blstm_status: synchronises the stream and returns the launch's error word in *host_out (non-zero: a workgroup
 *     gave up waiting for its peers after ~2 s, the outputs are invalid). */
long long pika_blstm_packed_bytes(int D, int H);
long long pika_blstm_work_bytes(int S, int B, int D, int H);
int pika_blstm_pack(const float *w_hh, int D, int H, void *packed, void *stream);
int pika_blstm_layer(const float *gx, const void *w_packed, const int *lens, float *out, float *h_n, float *c_n,
                     void *work, long long work_bytes, int S, int B, int D, int H, void *stream);
int pika_blstm_status(const void *work, int *host_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_LAS_H */
