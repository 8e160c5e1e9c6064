/*
 * oracle/rnnt_loss_ref.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the RNN-T loss that the reference obtains from the
 * third-party `warp_rnnt` CUDA extension (github.com/1ytic/warp-rnnt, version
 * UNPINNED by the reference: only linked from README.md:36, absent from
 * requirements.txt:1-7, sources not vendored under /root/reference).
 *
 * PARITY UNPINNED: the reference ships no test, golden vector or known-answer
 * for this boundary.  The restatement is therefore pinned by
 * first principles instead (tests/test_oracle_rnnt.py):
 *   (1) brute-force enumeration of every alignment path on tiny lattices,
 *   (2) central finite differences of the fp64 cost w.r.t. log_probs,
 *   (3) alpha-side and beta-side log-likelihoods agree.
 *
 * Contract restated (reference call sites
 *   trainer/train_transducer_bmuf_otfaug.py:58,97-99 and
 *   trainer/train_transducer_mbr_bmuf_otfaug.py:64,157-159):
 *   log_probs (B,T,U1,V) f32, already log-softmaxed (trainer/model/transducer.py:110-111)
 *   labels    (B,U1-1)   i32, entries beyond labels_lengths[n] hold the padding
 *                              value V (egs/train_transducer_bmuf_otfaug.sh:38,178-179)
 *                              and must never be used as an index
 *   frames_lengths (B,) i32 = T_n, labels_lengths (B,) i32 = U_n
 *   alpha[0,0] = 0
 *   alpha[t,u] = logaddexp(alpha[t-1,u] + lp[t-1,u,blank],
 *                          alpha[t,u-1] + lp[t,u-1,y_u])
 *   cost_n     = -(alpha[T_n-1,U_n] + lp[T_n-1,U_n,blank])
 *   beta symmetric; gradients w.r.t. log_probs (Graves 2012, eq. 20, in the
 *   log_probs parameterisation used by warp_rnnt):
 *   g[t,u,blank]   = -exp(alpha[t,u] + beta[t+1,u] + lp[t,u,blank] - ll)   (t < T_n-1)
 *   g[T_n-1,U_n,blank] = -exp(alpha + lp - ll)
 *   g[t,u,y_{u+1}] = -exp(alpha[t,u] + beta[t,u+1] + lp[t,u,y_{u+1}] - ll) (u < U_n)
 *   zero elsewhere, and zero outside the (T_n, U_n+1) sub-lattice.
 *
 * Two instantiations: REAL=double (truth) and REAL=float (the fp32 tolerance
 * budget, also the `cpu_baseline` "port" leg of bench.py).  Utterances are
 * independent, so the outer loop is an OpenMP parallel-for.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NEG_INF_D (-INFINITY)

static inline double lae_d(double a, double b) {
    if (a == NEG_INF_D) return b;
    if (b == NEG_INF_D) return a;
    double m = a > b ? a : b;
    return m + log1p(exp(-fabs(a - b)));
}
static inline float lae_f(float a, float b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    float m = a > b ? a : b;
    return m + log1pf(expf(-fabsf(a - b)));
}

#define DEFINE_RNNT(NAME, REAL, LAE, EXPF)                                            \
int NAME(const float *log_probs, const int *labels, const int *frames_lengths,         \
         const int *labels_lengths, int B, int T, int U1, int V, int blank,            \
         REAL *costs, REAL *grads /* (B,T,U1,V) or NULL */,                            \
         REAL *alphas /* (B,T,U1) or NULL */, REAL *betas /* (B,T,U1) or NULL */)      \
{                                                                                      \
    int status = 0;                                                                    \
    _Pragma("omp parallel for schedule(dynamic, 1)")                                   \
    for (int n = 0; n < B; ++n) {                                                      \
        const int Tn = frames_lengths[n], Un = labels_lengths[n];                      \
        const float *lp = log_probs + (size_t)n * T * U1 * V;                          \
        const int *y = labels + (size_t)n * (U1 - 1);                                  \
        if (Tn < 1 || Tn > T || Un < 0 || Un > U1 - 1) { status = 1; continue; }       \
        REAL *a = (REAL *)malloc(sizeof(REAL) * (size_t)T * U1);                       \
        REAL *b = (REAL *)malloc(sizeof(REAL) * (size_t)T * U1);                       \
        for (size_t i = 0; i < (size_t)T * U1; ++i) { a[i] = -INFINITY; b[i] = -INFINITY; } \
        /* forward variable */                                                         \
        for (int t = 0; t < Tn; ++t) {                                                 \
            for (int u = 0; u <= Un; ++u) {                                            \
                if (t == 0 && u == 0) { a[0] = 0; continue; }                          \
                REAL no_emit = -INFINITY, emit = -INFINITY;                            \
                if (t > 0)                                                             \
                    no_emit = a[(size_t)(t - 1) * U1 + u] +                            \
                              (REAL)lp[((size_t)(t - 1) * U1 + u) * V + blank];        \
                if (u > 0)                                                             \
                    emit = a[(size_t)t * U1 + u - 1] +                                 \
                           (REAL)lp[((size_t)t * U1 + u - 1) * V + y[u - 1]];          \
                a[(size_t)t * U1 + u] = LAE(no_emit, emit);                            \
            }                                                                          \
        }                                                                              \
        /* backward variable */                                                        \
        for (int t = Tn - 1; t >= 0; --t) {                                            \
            for (int u = Un; u >= 0; --u) {                                            \
                REAL lb = (REAL)lp[((size_t)t * U1 + u) * V + blank];                  \
                if (t == Tn - 1 && u == Un) { b[(size_t)t * U1 + u] = lb; continue; }  \
                REAL no_emit = -INFINITY, emit = -INFINITY;                            \
                if (t < Tn - 1) no_emit = b[(size_t)(t + 1) * U1 + u] + lb;            \
                if (u < Un)                                                            \
                    emit = b[(size_t)t * U1 + u + 1] +                                 \
                           (REAL)lp[((size_t)t * U1 + u) * V + y[u]];                  \
                b[(size_t)t * U1 + u] = LAE(no_emit, emit);                            \
            }                                                                          \
        }                                                                              \
        const REAL ll = b[0];                                                          \
        costs[n] = -ll;                                                                \
        if (grads) {                                                                   \
            REAL *g = grads + (size_t)n * T * U1 * V;                                  \
            memset(g, 0, sizeof(REAL) * (size_t)T * U1 * V);                           \
            for (int t = 0; t < Tn; ++t) {                                             \
                for (int u = 0; u <= Un; ++u) {                                        \
                    const size_t c = (size_t)t * U1 + u;                               \
                    const REAL al = a[c];                                              \
                    if (t < Tn - 1)                                                    \
                        g[c * V + blank] = -EXPF(al + b[c + U1] +                      \
                                                 (REAL)lp[c * V + blank] - ll);        \
                    else if (u == Un)                                                  \
                        g[c * V + blank] = -EXPF(al + (REAL)lp[c * V + blank] - ll);   \
                    if (u < Un)                                                        \
                        g[c * V + y[u]] = -EXPF(al + b[c + 1] +                        \
                                                (REAL)lp[c * V + y[u]] - ll);          \
                }                                                                      \
            }                                                                          \
        }                                                                              \
        if (alphas) memcpy(alphas + (size_t)n * T * U1, a, sizeof(REAL) * (size_t)T * U1); \
        if (betas) memcpy(betas + (size_t)n * T * U1, b, sizeof(REAL) * (size_t)T * U1);   \
        free(a); free(b);                                                              \
    }                                                                                  \
    return status;                                                                     \
}

DEFINE_RNNT(oracle_rnnt_loss_f64, double, lae_d, exp)
DEFINE_RNNT(oracle_rnnt_loss_f32, float, lae_f, expf)

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
