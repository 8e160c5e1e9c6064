/*
 * include/pika_joint.h -- C ABI of the joint-network kernels around the fc2 GEMM.
 *
 * Reference: /root/reference/trainer/model/transducer.py:98-111
 *   out = fc2(tanh(fc1(z)) * sigmoid(fc_gate(z))),  z = cat(enc[b,t], pred[b,u]);  log_softmax
 * and its decode-time twin /root/reference/decoder/transducer_decoder.py:173-177.
 * With fc1/fc_gate split into encoder/prediction halves:
 *   h[b,t,u,:] = tanh(e1[b,t,:] + p1[b,u,:]) * sigmoid(eg[b,t,:] + pg[b,u,:])
 * Conventions as in pika_rnnt.h.
 */
#ifndef PIKA_JOINT_H
#define PIKA_JOINT_H

#ifdef __cplusplus
extern "C" {
#endif

/* h (B,T,U,H) = tanh(e1+p1)*sigmoid(eg+pg); e* (B,T,H), p* (B,U,H) f32 contiguous; H % 4 == 0.
 * out_dtype PIKA_F32 | PIKA_BF16 (pika_gemm.h). */
int pika_joint_gate_fwd(const float *e1, const float *p1, const float *eg, const float *pg,
                        void *h, int out_dtype, int B, int T, int U, int H, void *stream);

/* Backward of the gate given dh (B,T,U,H), dh_dtype PIKA_F32 | PIKA_BF16: de1/deg (B,T,H) = sum over u, dp1/dpg (B,U,H) =
 * sum over t of  dz1 = dh*sig(zg)*(1-tanh(z1)^2),  dzg = dh*tanh(z1)*sig(zg)*(1-sig(zg)).
 * tanh/sigmoid are recomputed from e*,p* (nothing of size B*T*U*H is kept from the forward). */
int pika_joint_gate_bwd(const void *dh, int dh_dtype, const float *e1, const float *p1, const float *eg,
                        const float *pg, float *de1, float *dp1, float *deg, float *dpg,
                        int B, int T, int U, int H, void *stream);

/* In place on x (rows, cols) f32 with pitch ld: x = log_softmax(scale * x) per row. */
int pika_log_softmax_rows(float *x, long long rows, int cols, long long ld, float scale,
                          void *stream);

/* In place on g: g = scale * (g - exp(lp) * rowsum(g))  (log-softmax backward; lp = the
 * forward output, same shape/pitch). */
int pika_log_softmax_bwd_rows(const float *lp, float *g, long long rows, int cols, long long ld,
                              float scale, void *stream);

/* Same, written as bf16 into `out` (rows, ld_out) instead of in place; columns [cols, ld_out) are
 * zero-filled so `out` can feed pika_gemm_bf16_nt with K = ld_out (a multiple of 64). */
int pika_log_softmax_bwd_rows_bf16(const float *lp, const float *g, void *out, long long rows,
                                   int cols, long long ld, long long ld_out, float scale,
                                   void *stream);

/* MBR risk gradient (reference: trainer/train_transducer_mbr_bmuf_otfaug.py:225-235, where a
 * dense (rows, V) tensor holding ONE non-zero per row is pushed through log_softmax backward).
 * In place on lp (rows, V) = log_softmax(scale * logits):
 *   lp[r, v] <- scale * val[r] * ((v == sym[r]) - exp(lp[r, v]))      (rows with val == 0 -> zeros)
 * i.e. d/dlogits of sum_r val[r] * lp[r, sym[r]].  sym i32, val f32. */
int pika_mbr_risk_grad_rows(float *lp, const int *sym, const float *val, long long rows, int cols,
                            long long ld, float scale, void *stream);

/* Levenshtein distances of the N-best against their references (reference: editdistance.eval(hyp, ref) once per
 * hypothesis, trainer/train_transducer_mbr_bmuf_otfaug.py:186-190 -- editdistance==0.5.2, requirements.txt:1).
 * HOST function (no device work, no stream): pair i compares the int32 sequences seqs[a_off[i] .. a_off[i] + a_len[i])
 * and seqs[b_off[i] .. b_off[i] + b_len[i]); out[i] = minimum number of insertions, deletions and substitutions. */
int pika_edit_distances(const int *seqs, const long long *a_off, const int *a_len, const long long *b_off,
                        const int *b_len, int n_pairs, int *out);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_JOINT_H */
