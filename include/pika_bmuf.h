/*
 * include/pika_bmuf.h -- C ABI of the fused BMUF (block-momentum model averaging) kernels.
 *
 * Replaces the chain of PyTorch ops in the reference's
 *   /root/reference/trainer/bmuf.py:76-100  (BmufTrainer.update_and_sync)
 * and the per-parameter copy loop /root/reference/trainer/bmuf.py:14-35 (_copy_vec_to_param).
 *
 * MI355X formulation (DESIGN.md "BMUF"): parameters live as views of ONE flat fp32 vector
 * `local` (no flatten/unflatten copies); every rank keeps `global` and `delta_prev`; the
 * exchange is ONE all-reduce(SUM) of `delta` over RCCL/xGMI (host side, torch.distributed),
 * bracketed by the two single-pass kernels below.
 *
 * Same conventions as pika_rnnt.h: device pointers owned by the caller, stream-ordered on
 * `stream` (hipStream_t as void*), no allocation, return 0 / negative PIKA_E* / hipError_t.
 */
#ifndef PIKA_BMUF_H
#define PIKA_BMUF_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* delta[i] = global[i] - local[i]                                   (bmuf.py:83-84) */
int pika_bmuf_delta(const float *global, const float *local, float *delta, size_t n, void *stream);

/* *flag (device int, caller zeroes it) is set to 1 if any delta[i] is NaN   (bmuf.py:89) */
int pika_bmuf_nan_flag(const float *delta, size_t n, int *flag, void *stream);

/* One pass over four vectors (bmuf.py:93-98, executed identically on every rank):
 *   d          = delta[i] / world          (world = float(world_size); a true division as bmuf.py:93, not a multiplication by
 *                                            the reciprocal: the two round differently when world is not a power of two)
 *   delta_prev = block_momentum * delta_prev + block_lr * (1 - block_momentum) * d
 *   global    -= (1 + block_momentum) * delta_prev
 *   local      = global                      (replaces broadcast + _copy_vec_to_param)
 * skip_flag: NULL, or the device int pika_bmuf_nan_flag wrote: when it is set the call changes nothing -- the STOP branch
 * of bmuf.py:89-90 decided on the device, so the host can read the flag later without a blocking read in between. */
int pika_bmuf_update(const float *delta, float *delta_prev, float *global, float *local, size_t n,
                     float world, float block_momentum, float block_lr, const int *skip_flag, void *stream);

/* BMUF-Adam (/root/reference/trainer/bmuf.py:191-333: the block update of Adam's first / second moments, :291-313).
 * sum_* (n) IN: the moments summed over the ranks (the all-reduced optimizer state itself); OUT: the block moments, which
 * the optimizer continues from.  blk_* (n): the block moments kept from block to block.  Per element, every product rounded:
 *   blk = (c1 * blk + c2 * (sum / world)) / c3
 * with c1 = beta^tau (beta^(rho bm) - 1), c2 = 1 - beta^tau beta^(rho bm), c3 = 1 - beta^tau formed by the caller in
 * float64 (tau = sync period, rho as in the reference).  skip_flag as in pika_bmuf_update. */
int pika_bmuf_adam_moments(float *sum_avg, float *blk_avg, float *sum_sq, float *blk_sq, size_t n, float world,
                           float c1_avg, float c2_avg, float c3_avg, float c1_sq, float c2_sq, float c3_sq,
                           const int *skip_flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKA_BMUF_H */
