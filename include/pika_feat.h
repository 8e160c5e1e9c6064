/*
 * include/pika_feat.h -- C ABI of the on-device feature-side kernels of the RNN-T path.
 *
 * Replaces, on the hot path of /root/reference/trainer/train_transducer_bmuf_otfaug.py:
 *   :86-91  per-batch CMN + global CMVN (three in-place PyTorch passes)
 *   :92-93  SpecAugment.apply -> /root/reference/utils/spec_augment.py:10-20
 * Conventions as in pika_rnnt.h (device pointers, stream-ordered, no allocation).
 */
#ifndef PIKA_FEAT_H
#define PIKA_FEAT_H

#ifdef __cplusplus
extern "C" {
#endif

/* In place on x f32 (B,T,F) contiguous:
 *   if cmn:   x[b,t,f] -= mean_t x[b,:,f]   (mean over ALL T rows, padding included,
 *                                            train_transducer_bmuf_otfaug.py:88-89)
 *   x = (x + offset[f]) * scale[f]          (:90-91; offset/scale may be NULL = skip) */
int pika_cmvn_apply(float *x, int B, int T, int F, const float *offset, const float *scale,
                    int cmn, void *stream);

/* In place on x f32 (B,T,F): zero x[:, :, f0:f0+fs] and x[:, t0:t0+ts, :]
 * (spec_augment.py:16-20; one band pair for the whole batch).  fs/ts may be 0. */
int pika_specaug_apply(float *x, int B, int T, int F, int f0, int fs, int t0, int ts, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_FEAT_H */
