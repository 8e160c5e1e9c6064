/*
 * include/pika_attn.h -- C ABI of the fused multi-head self-attention core (libpika_amd.so).
 *
 * Replaces, for the encoder's transformer layers (no mask, no layer cache, no relative positions:
 * the only branch the RNN-T training path reaches), the chain
 *   /root/reference/trainer/model/multi_headed_attn.py:199-231
 *     query = query / sqrt(D); scores = query @ key^T; attn = softmax(scores);
 *     drop_attn = dropout(attn); context = drop_attn @ value
 * whose (B,H,T,T) score/probability tensors (2 GB per layer at config 2) are never written here.
 *
 * io_dtype PIKA_F32 | PIKA_BF16 (pika_gemm.h) is the element type of q, k, v, out, dout, dq, dk, dv: bf16
 * when the projections come from / go to MFMA products only (the kernels round to bf16 anyway; pitches
 * then % 8 == 0).
 * Tensors: q, k, v, dq, dk, dv are (B,T,H*D) with row pitch `ld` elements (so they may be the three
 * column blocks of one packed (B,T,3*H*D) projection), out and dout with row pitch `ldo`; head h owns
 * columns [h*D, (h+1)*D); batches are T rows apart; D is 64 or 128; pitches % 4 == 0 and 16-byte
 * aligned bases.
 * lse (B*H*T,) f32 = log2-domain log-sum-exp of the scaled scores, written by the forward and read
 * by the backward.  Arithmetic: operands rounded to bf16 for the MFMAs, fp32 softmax/accumulation.
 *
 * Dropout (p_drop in [0,1), 0 = none): the keep decision of (b, h, query, key) is a counter-based hash
 * of (seed, b, h, query, key); the forward packs the decisions once into `keep_bits`
 * (u64 [B*H][T][ceil(T/64)], 8 bytes per query row and 64 keys, caller-owned, may be NULL when
 * p_drop == 0) and all three passes read bits instead of re-hashing.  Kept probabilities are scaled by
 * 1/(1-p), p = round(p_drop*65536)/65536.  pika_attention_keep_mask materialises the mask as bytes
 * (tests only).  Conventions as in pika_rnnt.h.
 */
#ifndef PIKA_ATTN_H
#define PIKA_ATTN_H

#ifdef __cplusplus
extern "C" {
#endif

/* mask: NULL, or the PACKED attention mask of pika_attention_mask_bits, u64 [B][T][ceil(T/64)], shared by the heads of a
 * batch element: bit (key & 63) of word key >> 6 of query row q set = the score of (q, key) is replaced by -1e18 before
 * the softmax (multi_headed_attn.py:215-217; the prediction network's causal + padding mask,
 * rnnt_conv_transformer_lm.py:65-69).  One 8-byte word per query row and 64 keys, read exactly as the dropout keep
 * bits are.  The backward must be given the same mask. */
int pika_attention_fwd(const void *q, const void *k, const void *v, void *out, int io_dtype, float *lse,
                       void *keep_bits, const void *mask, int B, int T, int H, int D, long long ld, long long ldo,
                       float p_drop, unsigned seed, void *stream);

/* The forward in TWO-TERM arithmetic: q, k, v are the bf16 "hi" planes of two-term tensors x = hi + lo whose "lo"
 * planes lie lo_off elements behind (the packed projection written by pika_gemm_bf16_ex with out_lo); both products
 * keep hi.hi + lo.hi + hi.lo on the matrix cores, the softmax runs in fp32, and the context is written as two bf16
 * planes (out, out + out_lo_off) as well: scores and context to ~1e-5 instead of the 4e-3 of one bf16 term -- the
 * forward of the train step that carries the 1e-3 parity statement.  lse / keep_bits / mask as above;
 * the backward is pika_attention_bwd on the hi planes (io_dtype PIKA_BF16). */
int pika_attention_fwd_two_term(const void *q, const void *k, const void *v, long long lo_off, void *out,
                                long long out_lo_off, float *lse, void *keep_bits, const void *mask, int B, int T,
                                int H, int D, long long ld, long long ldo, float p_drop, unsigned seed, void *stream);

/* INFERENCE forward with fp32-grade products on two FP16 terms per operand (the decoder's encoder pass, reference
 * multi_headed_attn.py:199-231 at T ~ 1000): q, k, v are the fp16 "hi" planes of x = hi + 2^-11 lo' (hi = fp16(x),
 * lo' = fp16((x - hi) 2^11): the second term at the first one's magnitude), the "lo'" planes lo_off elements behind;
 * both products keep hi.hi in one accumulator and lo'.hi + hi.lo' in a second one (joined scaled by 2^-11): ~2^-22 per
 * product at three MFMAs; softmax in fp32, no dropout, context out as fp32 (pitch ldo).  mask as in pika_attention_fwd. */
int pika_attention_infer_f16x2(const void *q, const void *k, const void *v, long long lo_off, float *out, const void *mask,
                               int B, int T, int H, int D, long long ld, long long ldo, void *stream);

/* delta (B*H*T,) f32 is scratch (sum_d out*dout per query row). */
int pika_attention_bwd(const void *q, const void *k, const void *v, const void *out, const void *dout,
                       int io_dtype, const float *lse, const void *keep_bits, const void *mask, float *delta, void *dq,
                       void *dk, void *dv, int B, int T, int H, int D, long long ld, long long ldo,
                       float p_drop, unsigned seed, void *stream);

/* bits u64 [B][T][ceil(T/64)] from a byte mask (B, T, T) (non-zero = masked). */
int pika_attention_mask_bits(const unsigned char *mask, int B, int T, void *bits, void *stream);

/* mask (B*H, T, T) u8: 1 where the probability of (query row, key column) is kept. */
int pika_attention_keep_mask(unsigned char *mask, int BH, int T, float p_drop, unsigned seed,
                             void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_ATTN_H */
