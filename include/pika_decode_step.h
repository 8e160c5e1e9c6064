/*
 * include/pika_decode_step.h -- C ABI of the per-step kernels of the batch beam search (gfx950).
 *
 * One search step of /root/reference/decoder/transducer_decoder.py:123-186 (`decode_batch` loop body:
 * prediction-network step for the rows that emitted a label :139-171, joint :173-175, log-softmax :177,
 * BeamMergeTransducer.advance per utterance :182 = decoder/beam_transducer.py:82-187, `_beam_update` :188-202)
 * as a fixed chain of launches with NO host involvement, so that the chain can be captured once in a hipGraph
 * and replayed several times per host read:
 *
 *   pika_dstep_prep           re-order the prediction-net state / ancestry by the previous step's parents, advance
 *                             the frame indices of blank rows, embed the new labels, gather the causal-conv taps
 *   pika_dgemm (x12)          every Linear / causal Conv1d of the conv-transformer prediction net at ONE new
 *                             position per row, and the prediction half of fc1/fc_gate with the gate
 *                             tanh(.)*sigmoid(.) (+ gathered encoder half) in its epilogue
 *   pika_dstep_attention (x2) self-attention of the new position over the row's cached prefix (keys / values of
 *                             the new position are stored into the caches by the same launch)
 *   pika_dfc2_logits          fc2 (+bias, x sm_scale): the scaled logits (B*K, V) and per-row, per-range log-sum-exp
 *                             statistics (max, sum exp) from its epilogue
 *   pika_beam_advance_logits  a row's K best by ONE thresholded pass over its logits, then the whole of `advance` (as
 *                             pika_beam_advance, pika_decode.h), the all-done test and the step counter
 * (Rounds 2-4 selected K candidates per (row, range) inside the product's epilogue -- pika_dfc2_topk /
 *  pika_beam_advance_partials -- which cost more than the product; removed in ABI 20.)
 *
 * Weights are constant while decoding: they are packed ONCE (pika_dpack_weight) into MFMA fragment order, as 1, 2
 * or 3 bf16 terms (w = hi [+ mid [+ lo]]: 3 terms reproduce fp32 products exactly, 6 MFMAs per product pair), or --
 * terms = 4 -- as two fp16 terms w = hi + 2^-11 lo' (lo' = fp16((w - hi) 2^11): 22 mantissa bits per operand; the kernels
 * keep hi.hi and the cross products hi.lo' + lo'.hi in separate accumulators: an fp32 product to ~2^-22 with 3 MFMAs;
 * needs |values| < 65504).  Activations stay fp32 in memory and are split the same way while they are staged into LDS.
 *
 * Conventions as in pika_rnnt.h: caller-owned device memory, no allocation, no host sync, stream = hipStream_t,
 * returns 0 / PIKA_EINVAL (<0) / hipError_t (>0).  `stop` (int32, device, may be NULL): when *stop != 0 the
 * state-mutating launches return without touching anything (replays after the search has finished).
 */
#ifndef PIKA_DECODE_STEP_H
#define PIKA_DECODE_STEP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- packed weights ---------------------------------------------------------------------------------------- */
/* bytes of the packed form of a (N,K) matrix: planes * ceil16(N) * ceil32(K) * 2, planes = terms (2 for terms = 4) */
size_t pika_dpack_bytes(int N, int K, int terms);
/* W (N,K) f32, row stride ldw -> packed.  interleave2 != 0: W has 2*N2 rows = [first half | second half]; packed
 * row 2j = W[j], row 2j+1 = W[N2 + j] (the gate epilogue wants fc1 / fc_gate outputs of one unit side by side). */
int pika_dpack_weight(const float *W, long long ldw, int N, int K, int terms, int interleave2, void *packed,
                      void *stream);

/* ---- C = epilogue(A . W^T) --------------------------------------------------------------------------------- */
#define PIKA_DG_RELU 1      /* max(.,0) after bias                                                   */
#define PIKA_DG_GATE 2      /* packed W is interleave2: C[r,j] = tanh(acc[2j] + e[g,j]) * sigmoid(acc[2j+1] + e[g,N/2+j]),
                               g = (r / beam) * T + clamp(t_idx[r], 0, T-1); N counts the 2H interleaved columns;
                               C2 (optional, no node needed): the raw acc, (M, N) at the row C is written to       */
#define PIKA_DG_ROWMASK 4   /* rows r with node[r] == skip_node are not stored                        */
#define PIKA_DG_FEW_ROWS 8  /* hint: *m_dev is expected to be a small fraction of M (the rows of a search step that emitted
                               a label, ~M/6): many small workgroups with ONE request round each instead of M-sized tiling */
typedef struct {
    const float *A;          /* (M, Kp) f32, Kp = ceil32(K) columns readable (caller zero-pads)      */
    long long lda;
    const void *W;           /* packed, `terms` terms                                                */
    const float *bias;       /* N or NULL                                                            */
    const float *res;        /* residual (M, N) added after bias / relu, or NULL                     */
    long long ldr;
    float *C;                /* (M, N) [GATE: (M, N/2)]                                              */
    long long ldc;
    float *C2;               /* optional second destination, row r -> C2 + node[r] * ldc2, or NULL   */
    long long ldc2;
    const long long *node;   /* (M) i64: C2 scatter / ROWMASK                                        */
    long long skip_node;
    const float *e_all;      /* GATE: (B*T, N) f32 = [fc1 encoder half + bias | fc_gate half + bias] */
    const long long *t_idx;  /* GATE: (M) i64                                                        */
    int T, beam;
    int M, N, K, terms, flags;
    const int *m_dev;        /* device int32 or NULL: only rows < min(M, *m_dev) are computed (compact row lists) */
    const long long *crow;   /* (M) i64 or NULL: row r of the result is stored at row crow[r] of C     */
    const int *rowlist;      /* NULL, or a gather list (device int32): launch row e < M stands for row
                                r = rowlist[*rowoff_dev + e] of A, res, node, crow, t_idx and C (the rows of a launch are
                                then any subset of the buffers' rows; buffers stay where they are)          */
    const int *rowoff_dev;   /* device int32 offset into rowlist, or NULL (0)                               */
    const float *ln_gamma;   /* NULL, or: the rows of A are layer-normalised over their K columns on the way in
                                (reference onmt LayerNorm in front of every projection of the prediction net,
                                trainer/model/rnnt_conv_transformer_lm.py:59-80): A' = (A - mean) * rsqrt(var + ln_eps) *
                                ln_gamma + ln_beta, biased variance, two passes.  Needs K % 32 == 0, K <= 1024,
                                16-byte aligned ln_gamma / ln_beta (K floats each); PIKA_EINVAL otherwise           */
    const float *ln_beta;
    float ln_eps;
} pika_dgemm_t;
int pika_dgemm(const pika_dgemm_t *p, void *stream);

/* ---- prediction-network bookkeeping of one step ---------------------------------------------------------------
 * rows = B*beam, row r = b*beam + k.  state / anc are double-buffered: step s reads buffer s&1 and writes (s+1)&1.
 * For every row: parent pr = b*beam + prev_k[r]; state_dst[r] = state_src[pr]; anc_dst[r,:] = anc_src[pr,:];
 * tok = y[r]; t_idx[r] += (tok == blk); p = min(hyp_len[r], L-1).  Rows with tok > blk (a label was emitted: the
 * only rows whose prediction-net state changes, transducer_decoder.py:139-171) get a SLOT in a compact row list
 * (slot = count[s&1]++, order arbitrary): rowmap[slot] = r; node[slot] = 1 + s*rows + r; pos[slot] = p;
 * anc_dst[r,p] = node;
 * layer 0: x = emb[tok]; X[0][node] = x; A[0][slot] = [X[0][anc[p-4]] .. X[0][anc[p-1]] | x]  (zero left of 0)
 * layer l>0: A[l][slot, :4*C[l]] = [X[l][anc[p-4]] .. X[l][anc[p-1]]]  (the fifth block is written by the layer below)
 * s = *step_t (steps taken so far).  The prediction-net launches of the step then run on count[s&1] rows. */
#define PIKA_DSTEP_MAX_LAYERS 4
/* The joint's prediction half travels with the rows (optional; pj[0] == NULL: off).  A row's prediction-net state only
 * changes when it emits a label, so its product with the prediction halves of fc1 / fc_gate (transducer.py:107-109) is
 * computed once, when the state is, and kept: pj[(s+1)&1][r] = pj[s&1][parent] for every row (the same double buffering as
 * the state); rows that did NOT emit a label get their joint hidden for this step right here,
 *   h[r, j] = tanh(pj[2j] + e[j]) * sigmoid(pj[2j+1] + e[JH + j]),  e = e_all[(r / beam) * T + clamp(t_idx[r], 0, T-1)]
 * (t_idx after this step's increment); the rows of the compact list get pj and h from the PIKA_DG_GATE product of their
 * new state (pika_dgemm with rowlist = rowmap32, C = h, C2 = pj[(s+1)&1]) -- count rows instead of all of them. */
typedef struct {
    float *pj[2];            /* (rows, 2*JH) f32: columns (2j, 2j+1) = (fc1, fc_gate) prediction halves of unit j  */
    float *h;                /* (rows, JH) f32 out                                                                 */
    const float *e_all;      /* (B*T, 2*JH) f32 = [fc1 encoder half + bias | fc_gate encoder half + bias]          */
    int *rowmap32;           /* (rows) int32 out: slot -> row, the gather list of the compact rows' joint product  */
    int T, JH;
} pika_dstep_joint_t;
typedef struct {
    const long long *prev_k, *y, *hyp_len, *step_t;
    long long *t_idx;
    float *state[2];         /* (rows, H)                                                            */
    long long *anc[2];       /* (rows, L)                                                            */
    const float *emb;        /* (vocab+1, C[0])                                                      */
    float *X[PIKA_DSTEP_MAX_LAYERS];   /* (nodes, C[l]) layer-input caches                            */
    float *A[PIKA_DSTEP_MAX_LAYERS];   /* (rows, lda[l]) causal-conv input matrices, lda[l] >= 5*C[l] */
    int C[PIKA_DSTEP_MAX_LAYERS];
    long long lda[PIKA_DSTEP_MAX_LAYERS];
    long long *node, *pos;   /* (rows) outputs, indexed by SLOT                                      */
    long long *rowmap;       /* (rows) output: slot -> row                                           */
    int *count;              /* int32[2]: count[s & 1] += rows that emitted a label (the caller zeroes it; the
                                advance call re-zeroes the other parity's counter)                   */
    long long dump_node, zero_node;
    int layers, rows, beam, H, L, blk;
    const int *stop;
    pika_dstep_joint_t joint;
} pika_dstep_prep_t;
int pika_dstep_prep(const pika_dstep_prep_t *p, void *stream);

/* Self-attention of the new position of every row over its cached prefix (arithmetic of
 * pika_incremental_attention, pika_decode.h).  kvq (rows, 3d) f32 = [k | v | q] of the new position; k and v are
 * stored into k_cache / v_cache at node[r] by this launch and read back from kvq for position pos[r]. */
int pika_dstep_attention(const float *kvq, long long ldkvq, float *k_cache, float *v_cache,
                         const long long *ancestry, long long ancestry_pitch, const long long *pos,
                         const long long *node, const long long *rowmap, const int *m_dev, int rows, int L, int d,
                         int heads, float *out, void *stream);
/* kvq / pos / node / out are indexed by slot (< min(rows, *m_dev)); the ancestry row of slot i is rowmap[i]
 * (identity when rowmap is NULL). */

/* ---- LSTM prediction network (the shipped recipes' `dec_type=rnn`; reference trainer/model/transducer.py:55-61,
 * stepped by decoder/transducer_decoder.py:139-148: nn.LSTM on the embedded label, state kept for rows whose last
 * symbol is not a label) --------------------------------------------------------------------------------------
 * A row's state is [h_0 | c_0 | h_1 | c_1 | ...] (layers * 2 * H floats), double-buffered like the transformer
 * state: step s reads buffer s&1 and writes (s+1)&1.  For every row: parent pr = b*beam + prev_k[r];
 * state_dst[r] = state_src[pr]; tok = y[r]; t_idx[r] += (tok == blk).  Rows with tok > blk get a SLOT in the compact
 * row list (slot = count[s&1]++): rowmap[slot] = r; A[0][slot] = [emb[tok] (E) | h_0 of the parent (H)];
 * A[l][slot, H:2H] = h_l of the parent for l > 0 (columns [0, H) of A[l > 0] are written by the cell kernel of layer
 * l-1; pad columns up to lda[l] are never written: the caller zeroes them once).  The gate products of the step are
 * then pika_dgemm(A[l], [W_ih | W_hh] packed, bias = b_ih + b_hh) on count rows. */
typedef struct {
    const long long *prev_k, *y, *step_t;
    long long *t_idx;
    float *state[2];         /* (rows, layers*2*H)                                                   */
    const float *emb;        /* (vocab+1, E)                                                         */
    float *A[PIKA_DSTEP_MAX_LAYERS];   /* (rows, lda[l]), lda[0] >= E+H, lda[l>0] >= 2H               */
    long long lda[PIKA_DSTEP_MAX_LAYERS];
    long long *rowmap;       /* (rows) output: slot -> row                                           */
    int *count;              /* int32[2], as in pika_dstep_prep_t                                    */
    int layers, rows, beam, H, E, blk;
    const int *stop;
    pika_dstep_joint_t joint;        /* (JH = H: the joint reads h of the last layer)                   */
} pika_dstep_prep_lstm_t;
int pika_dstep_prep_lstm(const pika_dstep_prep_lstm_t *p, void *stream);

/* gates (slot order, pitch ldg) = [i | f | g | o] pre-activations of layer `layer` (nn.LSTM order) for the first
 * min(rows, *m_dev) slots -> c' = sigmoid(f) c + sigmoid(i) tanh(g), h' = sigmoid(o) tanh(c') written into the state
 * row rowmap[slot] (pitch state_pitch, blocks 2*layer and 2*layer+1) and, when next_a != NULL, h' also into
 * next_a[slot, 0:H) (pitch ld_next): the next layer's input row. */
int pika_dstep_lstm_cell(const float *gates, long long ldg, float *state, long long state_pitch, int layer,
                         const long long *rowmap, const int *m_dev, float *next_a, long long ld_next, int rows, int H,
                         void *stream);

/* ---- fc2: scaled logits + log-sum-exp statistics per column range ------------------------------------------------
 * h (rows, K) f32, W packed (V, K).  Column range s of `splits` (= pika_dfc2_splits(V)) covers
 * [s*cols, (s+1)*cols), cols = pika_dfc2_cols_per_split().  For every row and range:
 *   pmax[r*splits+s] = max_c x,  psum[...] = sum_c exp(x - pmax),  x = sm_scale * (h.W^T + bias)[r, c]
 * and the scaled values x themselves go to logits (rows, ldl) f32, ldl >= splits * cols (columns [V, splits*cols) are
 * written as -inf). */
#define PIKA_DFC2_COLS 192        /* columns per range (= pika_dfc2_cols_per_split()) */
int pika_dfc2_splits(int V);
int pika_dfc2_cols_per_split(void);
int pika_dfc2_logits(const float *h, long long ldh, const void *W, const float *bias, int rows, int V, int K,
                     int terms, float sm_scale, float *pmax, float *psum, float *logits, long long ldl, void *stream);

/* ---- advance from the scaled logits ----------------------------------------------------------------------------
 * As pika_beam_advance (pika_decode.h) with the row log-softmax taken from the statistics above and the row's K best from
 * its scaled logits: the K-th largest of the row's `splits` range maxima bounds the row's K-th largest value from below;
 * one pass over the row keeps the values at or above the bound (a few dozen of V), and the K best of those are the row's K
 * best (ties: lowest column first).  Rows whose survivors outnumber the 256-entry pool (many near-equal values; splits < K)
 * have the bound raised by bisection with counting passes.  Results identical to pika_beam_advance.  Plus:
 * `first` is read from the device (*step_t == 0); the step counter is incremented by the call;
 * done[b] (u8) = eos_top[b] && fin_n[b] >= n_best; *stop = all utterances done; *max_hyp = max hyp_len.
 * sync (int32[8], zeroed once by the caller): [0..3] scratch for the cross-workgroup arrival counts of even / odd
 * steps; [4] is set once a call was skipped because *stop was already set (the gate of the FST advance that follows);
 * [5], [6] are the compact-row counters of pika_dstep_prep (count = sync + 5): the call zeroes the one the next step
 * will fill.  LDS: pika_beam_advance_logits_lds(K, L, splits) bytes (0: the shape is not taken -> PIKA_ETOOBIG). */
size_t pika_beam_advance_logits_lds(int K, int L, int splits);
int pika_beam_advance_logits(const float *pmax, const float *psum, const float *logits, long long ldl, int splits,
                             float *scores, const float *lm_scores, float lm_scale, long long *y,
                             long long *t_idx, const long long *num_frames, const long long *max_len,
                             long long *hyp, long long *hyp_len, int L, long long *ks_hist,
                             long long *ys_hist, long long *step_t, unsigned char *eos_top,
                             float *fin_score, long long *fin_step, long long *fin_k, long long *fin_n,
                             int fin_cap, long long *prev_k_out, long long *y_raw, int B, int K, int V,
                             int blk, int beam_prune, int n_best, int *stop, long long *max_hyp, int *sync,
                             void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_DECODE_STEP_H */
