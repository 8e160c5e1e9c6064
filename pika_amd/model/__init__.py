"""RNN-T model of the PIKA hot path: TDNN-Transformer encoder,
conv/transformer (or LSTM) prediction network, gated joint.

The module tree, parameter names and registration order are the reference's (they are the
checkpoint / BMUF-vector / decoder contract: `model.encoder`, `.decoder`, `.embed`, `.fc1`,
`.fc_gate`, `.fc2`, ...), so state_dicts and whole-module pickles move both ways.  The forward
passes are ours: they route through `pika_amd.model.ops`, which dispatches to the HIP kernels
of libpika_amd.so for GPU tensors.
"""
