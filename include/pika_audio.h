/*
 * include/pika_audio.h -- C ABI of the on-the-fly feature front end (GPU side of the loader).
 *
 * Replaces the per-utterance CPU work of /root/reference/loader/otf_utt_loader.py:213-270:
 *   :218-230  AudioSegment.change_speed / normalize / int16 conversion  (loader/audio.py:217-262,578-603)
 *   :231-234  Kaldi Fbank.compute_features (third-party; options egs/fbank.conf:1-6)
 *   :249-270  splice (+-ctx, :28-46), stride, padding with the last frame
 * Utterances of one batch are concatenated; `*_off` arrays hold B+1 element offsets (device
 * memory, int64).  Conventions as in pika_rnnt.h.
 */
#ifndef PIKA_AUDIO_H
#define PIKA_AUDIO_H

#ifdef __cplusplus
extern "C" {
#endif

/* Speed + volume perturbation.  pcm int16 [in_off[B]] -> out f32 [out_off[B]] holding
 * int16-valued samples (what Kaldi receives).  Utterance b: m = out_off[b+1]-out_off[b] output
 * samples; if m != n: linear interpolation at x_i = i*n/(m-1) (np.interp on linspace(0,n,m),
 * clamped to the last sample) in fp64; gain = 10^(min(300, target_db[b] - rms_db)/20) with
 * rms_db = 10 log10(max(1e-20, mean(x^2))); result trunc(clip(x*gain*32768, -32768, 32767)).
 * sumsq: B doubles of scratch (zeroed by the call). */
int pika_audio_perturb(const short *pcm, const long long *in_off, const long long *out_off,
                       const double *target_db, int B, long long max_out, float *out,
                       double *sumsq, void *stream);

/* Kaldi-compatible log-mel filterbank.  wave f32 [wave_off[B]] -> feats f32 [frame_off[B]][num_bins],
 * frames per utterance = 1 + (n - frame_len)/frame_shift (snip-edges; frame_off from the host).
 * Per frame: (+ dither*N(0,1)) - mean; pre-emphasis; Hamming; zero-pad to nfft (power of two
 * <= 1024); |FFT|^2 over bins [0,nfft/2); mel filters given as CSR (mel_lo[b], mel_cnt[b], weights
 * mel_w packed); log(max(e, FLT_EPSILON)).  dither_seed: any value; used only if dither != 0. */
int pika_fbank(const float *wave, const long long *wave_off, const long long *frame_off, int B,
               long long total_frames, int frame_len, int frame_shift, int nfft, float preemph,
               float dither, unsigned long long dither_seed, int num_bins, const int *mel_lo,
               const int *mel_cnt, const int *mel_ptr, const float *mel_w, float *feats,
               void *stream);

/* Splice + subsample + pad into the batch tensor out (B, t_max, dim*(lctx+1+rctx)):
 * out[b,t,:] = concat_{j=-lctx..rctx} feats_b[clamp(t*stride + j, 0, n_b-1)] for t < len_b =
 * ceil(n_b/stride); rows t >= len_b repeat row len_b-1 (otf_utt_loader.py:262-268). */
int pika_splice_pad(const float *feats, const long long *frame_off, int B, int dim, int lctx,
                    int rctx, int stride, int t_max, float *out, void *stream);

/* Noise / reverberation augmentation on float samples (reference loader/audio.py: add_noise :467-513,
 * convolve / convolve_and_normalize :426-465; the hooks in otf_utt_loader.py:224-228 are commented out in the
 * reference).  Building blocks, composed by pika_amd/loader/augment.py:
 *   pika_audio_sumsq         *out (f64, zeroed by the call) = sum x[i]^2  -> rms_db = 10 log10(max(1e-20, sum/n))
 *   pika_audio_axpby         y = b*y + a*x   (x may be NULL: plain gain, AudioSegment.gain_db / superimpose)
 *   pika_audio_convolve_same out[i] = sum_k h[k] x[i + (m-1)/2 - k], i < n  (fftconvolve(x, h, "same"), m <= n) */
int pika_audio_sumsq(const float *x, long long n, double *out, void *stream);
int pika_audio_axpby(float *y, const float *x, long long n, float a, float b, void *stream);
int pika_audio_convolve_same(const float *x, long long n, const float *h, int m, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_AUDIO_H */
