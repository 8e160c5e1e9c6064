"""Writes tests/golden/rnnt_kat.npz: the PUBLISHED known-answer test of the RNN-T loss that the reference's
third-party loss (`warp_rnnt`, github.com/1ytic/warp-rnnt, README.md:36 of the reference) and the library it
re-implements (warp-transducer, github.com/HawkAaron/warp-transducer) both ship in their own test suites
(warp-transducer `tests/test_cpu.cpp::small_test` and `pytorch_binding/tests`, warp_rnnt `pytorch_binding/warp_rnnt/test.py`):
activations (1,2,3,5), labels [1,2], blank 0 -> cost 4.495666, and the gradient with respect to the ACTIVATIONS
(the loss applied to log_softmax(acts)).

The numbers below are the literals of those upstream tests -- NOT outputs of this repository's oracle or kernels.
tests/test_oracle_rnnt.py and tests/test_rnnt_loss_gpu.py check the oracle / the HIP loss against them."""
import os

import numpy as np

ACTS = [[[[0.1, 0.6, 0.1, 0.1, 0.1],
          [0.1, 0.1, 0.6, 0.1, 0.1],
          [0.1, 0.1, 0.2, 0.8, 0.1]],
         [[0.1, 0.6, 0.1, 0.1, 0.1],
          [0.1, 0.1, 0.2, 0.1, 0.1],
          [0.7, 0.1, 0.2, 0.1, 0.1]]]]
LABELS = [[1, 2]]
EXPECTED_COST = [4.495666]
EXPECTED_GRADS_WRT_ACTS = [[[[-0.13116688, -0.3999269, 0.17703125, 0.17703125, 0.17703125],
                             [-0.18572757, 0.12247056, -0.18168412, 0.12247056, 0.12247056],
                             [-0.32091254, 0.06269141, 0.06928472, 0.12624499, 0.06269141]],
                            [[0.05456069, -0.21824276, 0.05456069, 0.05456069, 0.05456069],
                             [0.12073959, 0.12073959, -0.48295835, 0.12073959, 0.12073959],
                             [-0.6925882, 0.16871116, 0.18645467, 0.16871116, 0.16871116]]]]

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rnnt_kat.npz")
    np.savez(out, acts=np.array(ACTS, np.float32), labels=np.array(LABELS, np.int32),
             frames_lengths=np.array([2], np.int32), labels_lengths=np.array([2], np.int32),
             cost=np.array(EXPECTED_COST, np.float64), grads_wrt_acts=np.array(EXPECTED_GRADS_WRT_ACTS, np.float64))
    print("wrote", out)
