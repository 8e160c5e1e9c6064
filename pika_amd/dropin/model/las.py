"""Drop-in for `importlib.import_module("model." + args.nnet_proto)` with nnet_proto = las
(/root/reference/trainer/train_las_bmuf_otfaug.py:489-495: `nnet_module.Net(args, input_dim, output_dim, padding_idx)`)."""
from pika_amd.model.las import *  # noqa: F401,F403
from pika_amd.model.las import Net  # noqa: F401
