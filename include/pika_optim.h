/*
 * include/pika_optim.h -- C ABI of the per-step optimizer passes of the training scripts.
 *
 * Replaces, for fp32 parameters on the device,
 *   torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip, norm_type=inf)
 *   optimizer.step()   with optim.SGD(..., momentum, nesterov=True)
 * as the reference calls them once per batch (/root/reference/trainer/train_transducer_bmuf_otfaug.py:105-110;
 * optimizer construction :53-55,121-123 -- it is rebuilt after every BMUF sync, so momentum buffers restart).
 * torch runs these as ~10 multi-tensor launches over the ~300 parameter tensors (1.1 ms per step at 90 M
 * parameters); here: ONE launch for the infinity norm, ONE for the (rarely needed) in-place scaling, ONE for the
 * Nesterov update, all tensors addressed through a device table of chunks so that no tensor is copied or flattened.
 *
 * A chunk table describes the tensors: chunk c covers elements [chunk_off[c], chunk_off[c] + chunk_len[c]) of tensor
 * chunk_tensor[c]; tensor t lives at ptr[t].  All tables are device memory (built once per parameter set).
 * Conventions as in pika_rnnt.h (caller-owned device memory, stream-ordered, no allocation, 0 / <0 / hipError_t).
 */
#ifndef PIKA_OPTIM_H
#define PIKA_OPTIM_H

#ifdef __cplusplus
extern "C" {
#endif

/* *out (device f32, zeroed by the call) = max_t max_i |grad_t[i]|; NaN anywhere -> NaN (as torch's max of norms). */
int pika_multi_absmax(const float *const *grad_ptrs, const int *chunk_tensor, const long long *chunk_off,
                      const int *chunk_len, int n_chunks, float *out, void *stream);

/* clip_grad_norm_'s second half: coef = max_norm / (*total_norm + 1e-6); if coef < 1 every gradient is multiplied by
 * coef in place (torch multiplies by clamp(coef, max=1) always; x * 1 == x, so skipping it is the same result). */
int pika_multi_scale_by_clip(float *const *grad_ptrs, const int *chunk_tensor, const long long *chunk_off,
                             const int *chunk_len, int n_chunks, const float *total_norm, float max_norm,
                             void *stream);

/* torch.optim.SGD.step with momentum > 0, nesterov, dampening 0, weight_decay 0, maximize False:
 *   buf = first ? g : momentum * buf + g;   p -= lr * (g + momentum * buf)
 * (`first`: the optimizer has no momentum buffers yet -- torch clones the gradient). */
int pika_multi_sgd_nesterov(float *const *param_ptrs, const float *const *grad_ptrs, float *const *buf_ptrs,
                            const int *chunk_tensor, const long long *chunk_off, const int *chunk_len, int n_chunks,
                            float lr, float momentum, int first, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_OPTIM_H */
