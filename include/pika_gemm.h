/*
 * include/pika_gemm.h -- C ABI of the MFMA GEMM used by the RNN-T model kernels.
 *
 * C[z][m][n] (+)= act( sum_k A[z](m,k) * B[z](n,k) + bias[n] )        ("NT": both operands are
 * indexed (output-row, reduction) with the reduction index contiguous)
 *
 * Replaces the cuBLAS/cuDNN calls the reference reaches through nn.Linear, the TDNN
 * Conv2d(1,C,(3,C),dilation) (rnnt_tdnn_transformer.py:44-57), the causal Conv1d(k=5)
 * (rnnt_conv_transformer_lm.py:36-45), torch.matmul in attention (multi_headed_attn.py:207,223)
 * and the joint's fc2 (transducer.py:108).  An operand may be a VIRTUAL time-delay matrix:
 * row r = (b,t), reduction k = (tap,c) reads x[b, t*stride + tap*dil - pad, c] (zero outside
 * [0,t_in)), so convolutions run as ONE GEMM without materialising im2col.
 *
 * Arithmetic: operands are rounded to bf16 (RNE) on their way into LDS, products accumulate in
 * fp32 on v_mfma_f32_16x16x32_bf16.  PIKA_GEMM_FP32SPLIT splits each fp32 operand exactly into
 * three bf16 parts (8+8+8 mantissa bits) and issues the 6 MFMAs whose products exceed 2^-24 of
 * the leading one: fp32-class accuracy (~1e-7 relative) at 1/6 of the bf16 rate, used for the
 * parity runs against the reference's fp32 arithmetic.
 */
#ifndef PIKA_GEMM_H
#define PIKA_GEMM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIKA_F32 0
#define PIKA_BF16 1

typedef struct {
    const void *ptr;
    int dtype;           /* PIKA_F32 | PIKA_BF16 */
    int rows_per_batch;  /* row r -> b = r / rows_per_batch, t = r % rows_per_batch */
    int t_in;            /* source time extent: rows with ti outside [0,t_in) read as zero */
    long long batch_stride; /* elements between consecutive b */
    long long ld;        /* elements between consecutive source time rows */
    int C;               /* channels per tap: k -> tap = k / C, c = k % C */
    int stride, dil, pad;/* ti = t*stride + tap*dil - pad */
    long long z_outer, z_inner; /* element offsets per batched-GEMM index z: (z / z_div, z % z_div) */
    int trans;           /* 0: (row, k) = (time-like (b,t), channel-like (tap,c)): reduction contiguous.
                          * 1: roles swapped -- the REDUCTION runs over the time-like index and the output
                          *    row over the channel-like one (matrix stored [k][row], output index
                          *    contiguous): dY and X in dW = dY^T X, W in dX = dY W.  No transposed copy
                          *    is made: the kernel uses the gfx950 LDS transpose read. */
    int seg;             /* pika_gemm_bf16_ex only, bf16, else 0: a TWO-TERM operand x = hi + lo stored as two bf16
                          * planes of `seg` columns per source row (seg % 64 == 0), the "lo" plane lo_off elements
                          * behind the "hi" plane `ptr` addresses.  It presents C = 3 * seg reduction columns per tap,
                          * the segments [hi | lo | hi]; against a B whose rows hold [hi | hi | lo] per tap
                          * (pika_split_bf16_terms, role 1) the product is hi.hi + lo.hi + hi.lo: both operands to
                          * 16 mantissa bits, ~1e-5 relative, at a third of the bf16 rate. */
    long long lo_off;
} pika_operand_t;

#define PIKA_GEMM_RELU 1
#define PIKA_GEMM_ACCUMULATE 2
#define PIKA_GEMM_FP32SPLIT 4
#define PIKA_GEMM_OUT_BF16 8   /* C points to a bf16 matrix (pitch ldc elements); only for bf16 x bf16 products the
                                * direct-to-LDS kernel takes (K % 64 == 0, >= 160 output tiles), else PIKA_EINVAL */
#define PIKA_GEMM_F16_OPERANDS 16 /* the 16-bit operands (dtype PIKA_BF16 in the descriptors) hold FP16 bit patterns: the
                                * segments pika_split_bf16_terms(n_terms = 4) writes.  Only the plain product with fp32
                                * output on the direct-to-LDS kernel (v_mfma_f32_16x16x32_f16), else PIKA_EINVAL */

#define PIKA_GEMM_TERM_PRODUCT 32 /* the operands are term-segment copies (a K-concatenated product of pika_split_bf16_terms:
                                * 3 or 6 times the reduction of the plain product): the direct-to-LDS kernel takes the
                                * product from 24 output tiles on instead of 160 -- its alternative is not the bf16
                                * register-staged kernel but the EXACT one at six MFMA products per tile (round 6: the
                                * decoder's encoder pass at B = 8, 128 tiles, spent 6.9 of its 9 ms there).  Changes which
                                * kernel runs, never the values of a product. */

/* Requirements (16-byte operand loads): K % 4 == 0 unless both operands are `trans` (then the output
 * extents M / N must be multiples of g instead); with g = 4 for f32 / 8 for bf16 operands, C,
 * ld and the batch strides must be multiples of g and the base pointer 16-byte aligned; the rows
 * of a bf16 operand must be readable and ZERO up to the next multiple of 8 past K.  Returns PIKA_EINVAL otherwise (the caller decides; nothing falls back silently). */
int pika_gemm_nt(const pika_operand_t *A, const pika_operand_t *B, float *C, long long ldc,
                 long long c_z_outer, long long c_z_inner, int M, int N, int K, int batch,
                 int z_div, const float *bias, int flags, void *stream);

/* Same, with caller-owned device scratch (16-byte aligned; may be NULL / 0): split-K products whose
 * partial tiles fit (splits * M * N * 4 bytes) are written there and summed by one streaming kernel
 * instead of float atomics into C (deterministic, and ~5x cheaper per split on small outputs). */
int pika_gemm_nt_ws(const pika_operand_t *A, const pika_operand_t *B, float *C, long long ldc,
                    long long c_z_outer, long long c_z_inner, int M, int N, int K, int batch,
                    int z_div, const float *bias, int flags, void *workspace,
                    size_t workspace_bytes, void *stream);

/* C[M,N] f32 = A[M,K] bf16 * B[N,K]^T bf16 + bias[n] with direct global->LDS operand loads
 * (gemm_glds.hip).  Requirements: K % 64 == 0, lda/ldb % 8 == 0, ldc % 4 == 0, 16-byte aligned
 * bases.  Rows beyond M / N are never stored. */
int pika_gemm_bf16_nt(const void *A, long long lda, const void *B, long long ldb, float *C,
                      long long ldc, int M, int N, int K, const float *bias, void *stream);

/* pika_gemm_bf16_nt (N > 256) whose epilogue also emits, per output row m and 64-column block b < n_part =
 * ceil(N/256)*4, the partial log-sum-exp statistics of C's row: pmax[m*n_part + b] = max of the block's columns
 * (-inf for a block beyond N), psum[...] = sum exp(x - pmax).  The RNN-T loss merges them instead of re-reading the
 * (B,T,U,V) logits (pika_rnnt_fused_forward_partials, pika_rnnt.h). */
int pika_gemm_bf16_nt_lse(const void *A, long long lda, const void *B, long long ldb, float *C, long long ldc,
                          int M, int N, int K, const float *bias, float *pmax, float *psum, int n_part,
                          void *stream);

/* The joint's logits over the RNN-T lattice in 16 bits (reference trainer/model/transducer.py:107-111, fc2 over (B,T,U1)):
 * out16 (M = B*T*U1 rows, pitch ldo halves) receives A B^T + bias as IEEE fp16 (saturated at +-65504) -- half the bytes
 * of the largest tensor of a training step, written once and read once (pika_rnnt_dlogits_compact_bf16_f16in) -- while
 * everything the LOSS needs leaves in fp32 from the accumulators: the partial log-sum-exp statistics of
 * pika_gemm_bf16_nt_lse, and gathered[2 m] = C[m][blank], gathered[2 m + 1] = C[m][labels[b (U1-1) + u]] for row
 * m = (b T + t) U1 + u (u < U1 - 1; unwritten otherwise).  gathered may be NULL (then labels / T / U1 / blank are
 * ignored). */
int pika_gemm_bf16_nt_lse_f16(const void *A, long long lda, const void *B, long long ldb, void *out16, long long ldo,
                              int M, int N, int K, const float *bias, float *pmax, float *psum, int n_part,
                              const int *labels, int T, int U1, int blank, float *gathered, void *stream);

/* The same product with a fused bf16 epilogue, for chains whose wide intermediate only ever feeds
 * another MFMA product (the transformer feed-forward block, reference trainer/model/position_ffn.py:27-39:
 * w_2(dropout(relu(w_1(x)))) -- the (rows, d_ff) hidden exists only in bf16, ReLU/dropout never run as passes):
 *   PIKA_EPI_DROPOUT_BF16: out bf16 (pitch ldo) = dropout_p(relu?(A B^T + bias)); the keep decision of element
 *       (m, n) is a counter-based hash of (seed, m, n), kept values are scaled by 1/(1-p),
 *       p = round(p_drop * 65536) / 65536; pika_dropout_keep_mask materialises the mask (tests);
 *   PIKA_EPI_MASK_BF16:    out bf16 = scale * (A B^T) where aux[m,n] > 0 (bf16, pitch ld_aux), else 0 --
 *       with aux = the forward's dropped hidden this is the ReLU and dropout backward in one.
 * Requirements as pika_gemm_bf16_nt; ldo, ld_aux % 4 == 0. */
#define PIKA_EPI_F32 0
#define PIKA_EPI_DROPOUT_BF16 1
#define PIKA_EPI_MASK_BF16 2
#define PIKA_EPI_DROPOUT_RESIDUAL 3
int pika_gemm_bf16_epilogue(const void *A, long long lda, const void *B, long long ldb, void *out,
                            long long ldo, int M, int N, int K, const float *bias, int mode, int relu,
                            float p_drop, unsigned seed, const void *aux, long long ld_aux, float scale,
                            void *stream);
int pika_dropout_keep_mask(unsigned char *mask, int rows, int cols, float p_drop, unsigned seed,
                           void *stream);

/* The direct-to-LDS kernels take a (non-transposed) product only when it fills the chip: at least `min_tiles` 256 x 256
 * output tiles (default 160); smaller products run on the register-staged kernel.  The gate is a performance choice, not
 * an arithmetic one -- but the K-concatenated term products (fp16x2 / six-segment exact) exist on the direct-to-LDS kernels
 * only, so a parity test at a small batch lowers it to put that arithmetic on its path.  Returns the previous value;
 * min_tiles < 1 only reads.  Process-wide. */
int pika_gemm_set_min_tiles(int min_tiles);

/* Registers (NULL: clears) a device word that EVERY kernel of this library that takes a dropout seed adds to it when it
 * runs (GEMM epilogues, pika_dropout_mask_cast_bf16, the attention keep bits, the two keep-mask helpers).  A training
 * step captured once into a hipGraph replays the seeds it was captured with; with the caller changing this word between
 * replays every replay draws new masks, forward and backward of one replay agree.  Process-wide; the word must stay
 * allocated while registered. */
int pika_set_dropout_salt(const unsigned *device_word);

/* out f32 (pitch ldo) = dropout_p(A B^T + bias) + residual (f32, pitch ld_res): a projection with its residual
 * dropout and residual add (reference trainer/model/transformer.py:98-99 `self.dropout(context) + inputs`,
 * position_ffn.py:38-39 `output + x`) in the product's epilogue; same keep hash as PIKA_EPI_DROPOUT_BF16. */
int pika_gemm_bf16_dropout_residual(const void *A, long long lda, const void *B, long long ldb, float *out,
                                    long long ldo, int M, int N, int K, const float *bias, float p_drop,
                                    unsigned seed, const float *residual, long long ld_res, void *stream);
/* out bf16 (pitch ld_out) = keep(m,n) ? x * 1/(1-p) : 0 for x f32 (rows, cols), cols % 4 == 0: the backward of
 * that dropout fused with the bf16 rounding of the gradient's consumers. */
int pika_dropout_mask_cast_bf16(const float *x, long long ld, int rows, int cols, float p_drop, unsigned seed,
                                void *out, long long ld_out, void *stream);

/* Every product / epilogue combination of the direct-to-LDS kernel behind ONE entry: A is an operand descriptor (bf16;
 * plain or time-delay view with C % 64 == 0; optionally two-term, see pika_operand_t.seg), B a plain bf16 (N, K) matrix.
 *   PIKA_EPI_F32:              out f32 = relu?(A B^T + bias)
 *   PIKA_EPI_DROPOUT_BF16:     out bf16 = dropout_p(relu?(A B^T + bias)); with out_lo != NULL the value is written as TWO
 *                              bf16 terms, out = bf16(v) and out_lo = bf16(v - out) (same pitch): the two-term operand of
 *                              the next product of a forward pass that carries activations to 16 mantissa bits
 *   PIKA_EPI_MASK_BF16:        out bf16 = scale * (A B^T) where aux > 0
 *   PIKA_EPI_DROPOUT_RESIDUAL: out f32 = dropout_p(A B^T + bias) + residual
 * (semantics of the dedicated entries above).  No size gate: small products run correctly, on 256 x 256 tiles.
 * Returns PIKA_EINVAL for operands the kernel does not take (K % 64, alignments, padded views with the last two). */
typedef struct {
    pika_operand_t A;
    const void *B;
    long long ldb;
    int M, N, K;
    const float *bias;
    int relu;
    int epilogue;
    void *out;
    void *out_lo;
    long long ldo;
    float p_drop;
    unsigned seed;
    const void *aux;
    long long ld_aux;
    float scale;
    const float *residual;
    long long ld_res;
} pika_gemm_ex_t;
int pika_gemm_bf16_ex(const pika_gemm_ex_t *g, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_GEMM_H */
