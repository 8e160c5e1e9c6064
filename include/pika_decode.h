/*
 * include/pika_decode.h -- C ABI of the fused beam-search step.
 *
 * One launch performs, for every utterance of the batch, what the reference does per utterance in
 * Python with per-element host reads (/root/reference/decoder/beam_transducer.py:82-187
 * `BeamMergeTransducer.advance`, plus the frame-index part of `_beam_update`,
 * /root/reference/decoder/transducer_decoder.py:188-202):
 *   log-softmax(sm_scale * logits) per beam row, + running scores (+ lm_scale * lm_scores), rows
 *   whose last symbol is eos and duplicate partial hypotheses disabled (-1e20), sorted top-K over
 *   the K*V candidates, parent / symbol split, score update, finish rule
 *   ((y == blank and t_idx[parent] == frames-1) or len > max_len), partial-hypothesis update,
 *   `finished` list append, back-pointer / symbol history, eos_top, frame-index re-ordering.
 * All state is int64 / f32 device memory laid out as pika_amd/decoder/beam_search.py keeps it.
 * Two launches: one wavefront per beam row (row top-K), then one workgroup per utterance (merge +
 * bookkeeping).  `cand_ws`: B*K*K*8 bytes of device scratch.
 * Requirements: K <= 64, V <= 8192, K*L*4 bytes <= 64 KiB; otherwise PIKA_ETOOBIG.
 * `first` != 0 selects the reference's first-step branch (only row 0 competes, no score added).
 */
#ifndef PIKA_DECODE_H
#define PIKA_DECODE_H

#ifdef __cplusplus
extern "C" {
#endif

int pika_beam_advance(const float *logits, float sm_scale, int first, float *scores,
                      const float *lm_scores, float lm_scale, long long *y, long long *t_idx,
                      const long long *num_frames, const long long *max_len, long long *hyp,
                      long long *hyp_len, int L, long long *ks_hist, long long *ys_hist,
                      const long long *step_t, unsigned char *eos_top, float *fin_score,
                      long long *fin_step, long long *fin_k, long long *fin_n, int fin_cap,
                      long long *prev_k_out, long long *y_raw, void *cand_ws, int B, int K, int V,
                      int blk, int beam_prune, void *stream);
/* y_raw (B,K) i64, may be NULL: the selected symbols BEFORE finished slots are overwritten with eos -- what the
 * FST update below needs. */

/* N-gram FST shallow fusion on the device, called right after pika_beam_advance of the same step (reference
 * beam_transducer.py:135-181 with decoder/sorted_matcher.py:24-111).  The FST is an ilabel-sorted CSR table
 * (offsets i64 (S+1), ilabel i32, weight f32, nextstate i32, final f32 with +inf = not final; DEVICE memory);
 * disambig_ids is a HOST array of n_disambig <= 4 labels.  Per beam slot a set of at most
 * pika_fst_states_per_slot() LM states is kept in set_n (B*K) i32, set_state (B*K*S) i32, set_cost (B*K*S) f64
 * (initialise slot sets to {state 0: 0.0}).  The call re-orders the sets by prev_k, advances them by y_raw + 1
 * (non-blank symbols; same back-off / disambiguation walk and the same "first smaller wins" update as the
 * reference), writes lm_scores (B,K), and for the slots that finished in this step (y == eos) adds
 * lm_scale * (- best final cost) to scores and to the finished-list entries the advance call just appended.
 * *err is OR-ed with 1 if a set overflowed or a finishing slot had no final state (the reference raises). */
int pika_fst_advance(const long long *fst_offsets, const int *fst_ilabel, const float *fst_weight,
                     const int *fst_nextstate, const float *fst_final, int max_num_arcs, int max_id,
                     int backoff_id, const int *disambig_ids, int n_disambig, const long long *prev_k,
                     const long long *y_raw, const long long *y, int blk, double nonblk_reward, float lm_scale,
                     int *set_n, int *set_state, double *set_cost, float *lm_scores, float *scores,
                     float *fin_score, const long long *fin_n, int fin_cap, int B, int K, int *err,
                     const int *skip, void *stream);
/* skip (device int32, may be NULL): the call does nothing when *skip != 0 (graph replays after the search ended,
 * include/pika_decode_step.h). */
int pika_fst_states_per_slot(void);

/* get_hyp (/root/reference/decoder/beam_transducer.py:234-243) for all n = B * per_utt n-best entries at once, after the
 * search: entry e (of utterance e / per_utt) finished at step sel_step[e] in beam slot sel_k[e]; its symbols
 *   out[e, j] = ys_hist[j + 1, b, k_j],   k_{sel_step - 1} = sel_k[e],   k_{j-1} = ks_hist[j, b, k_j]     (j < sel_step[e])
 * and blk for j >= sel_step[e].  ys_hist (S+1, B, K), ks_hist (S, B, K) int64 as pika_beam_advance keeps them;
 * out (n, smax) int32, smax >= every sel_step.  Replaces a host loop over the steps on 5 MB of copied histories. */
int pika_beam_backtrack(const long long *ys_hist, const long long *ks_hist, const int *sel_step, const int *sel_k,
                        int n, int per_utt, int B, int K, int smax, int blk, int *out, void *stream);

/* Self-attention of ONE new position per beam row over that row's cached prefix, for the incremental
 * conv-transformer prediction network (pika_amd/decoder/prednet_cache.py; reference arithmetic
 * trainer/model/multi_headed_attn.py:199-231 at a single query position with the causal mask, i.e. keys at
 * positions <= pos).  k_cache / v_cache are flat (nodes, d) f32 tables; row r's prefix is
 * ancestry[r * ancestry_pitch + j] for j < pos[r] and `node[r]` (the freshly stored position) for j == pos[r];
 * pos is clamped to L-1.  q, out (rows, d) f32; d % heads == 0, head width a multiple of 4 whose quarter is a
 * power of two (<= 256), d <= 2048, heads * L * 4 bytes <= 64 KiB. */
int pika_incremental_attention(const float *q, const float *k_cache, const float *v_cache,
                               const long long *ancestry, long long ancestry_pitch, const long long *pos,
                               const long long *node, int rows, int L, int d, int heads, float *out,
                               void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_DECODE_H */
