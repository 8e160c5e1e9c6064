"""Generic transducer (reference: trainer/model/transducer.py:27-112): encoder + embedding +
prediction network + gated joint + optional log-softmax."""
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from . import ops
from .encoder import Net as TdnnTransformerEncoder
from .prednet import Net as ConvTransformerPredNet


def _lstm_forward(rnn, x):
    """nn.LSTM(x)[0] (trainer/model/transducer.py:93-96).  Training on a HIP device with dropout between the layers: the
    layers are run one at a time with torch's own dropout in between -- same function, same parameters -- because the
    dropout INSIDE the library's multi-layer call keeps the mask it drew when a hipGraph was captured: every replay of
    the graphed training step (pika_amd/train_graph.py) would drop the same units for the rest of the run
    (tests/test_train_step_gpu.py::test_graphed_lstm_prediction_net_draws_new_dropout_masks_per_replay).  torch's dropout
    takes its Philox offset from the graph-registered generator state and draws a new mask per replay.

    First choice on a HIP device: the recurrence of every layer as one persistent launch per direction of time
    (pika_amd/model/lstm.py, include/pika_lstm.h) -- the library's step-by-step chain is 4.3 ms of a 47 ms training step."""
    from . import lstm
    if lstm.applies(rnn, x):
        return lstm.forward(rnn, x)[0]
    if not (rnn.training and rnn.dropout > 0.0 and rnn.num_layers > 1 and x.is_cuda and rnn.batch_first
            and not rnn.bidirectional and getattr(rnn, "proj_size", 0) == 0 and rnn.bias):
        return rnn(x)[0]
    out = x
    zeros = x.new_zeros(1, x.shape[0], rnn.hidden_size)
    for l in range(rnn.num_layers):
        w = [getattr(rnn, n % l) for n in ("weight_ih_l%d", "weight_hh_l%d", "bias_ih_l%d", "bias_hh_l%d")]
        # (the dropout argument of a ONE-layer call drops nothing -- there is no layer below the output -- but it selects the
        # library's step-wise recurrence; without it the call takes a hipBLASLt path that cannot be captured at all:
        # "operation would make the legacy stream depend on a capturing blocking stream", tools/lstm_capture_probe.py)
        out = torch._VF.lstm(out, (zeros, zeros), w, True, 1, rnn.dropout, True, False, True)[0]
        if l + 1 < rnn.num_layers:
            out = torch.nn.functional.dropout(out, rnn.dropout, True)
    return out


class Net(nn.Module):
    """`opt` supplies rnn_size, local_rank, decoder_type, brnn, encoder_type, dropout,
    enc_layers, dec_layers, embd_dim, padding_idx (transducer.py:27-68)."""

    def __init__(self, opt, input_dim, output_dim):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.hid_dim = opt.rnn_size
        self.local_rank = opt.local_rank
        self.pack_seq = True
        self.decoder_type = opt.decoder_type
        if opt.encoder_type == 'rnn':
            dirs = 2 if opt.brnn else 1
            self.encoder = nn.LSTM(input_size=input_dim, hidden_size=self.hid_dim // dirs,
                                   dropout=opt.dropout, num_layers=opt.enc_layers,
                                   bidirectional=opt.brnn, batch_first=True)
        else:
            # widths are hard-coded in the reference (transducer.py:46-50), not flag-controlled
            self.encoder = TdnnTransformerEncoder(input_dim=input_dim, input_ctx=0,
                                                  output_dim=self.hid_dim, tdnn_nhid=1024,
                                                  tdnn_layers=9)
            self.pack_seq = False
        self.embed = nn.Embedding(output_dim + 1, opt.embd_dim, padding_idx=opt.padding_idx)
        if opt.decoder_type == 'rnn':
            self.decoder = nn.LSTM(input_size=opt.embd_dim, hidden_size=self.hid_dim,
                                   dropout=opt.dropout, num_layers=opt.dec_layers,
                                   bidirectional=False, batch_first=True)
        else:
            self.decoder = ConvTransformerPredNet(embeddings=self.embed, output_dim=self.hid_dim,
                                                  d_model=512, num_layers=opt.dec_layers, heads=8,
                                                  d_ff=2048, dropout=opt.dropout)
        self.fc1 = nn.Linear(2 * self.hid_dim, self.hid_dim)
        self.fc_gate = nn.Linear(2 * self.hid_dim, self.hid_dim)
        self.fc2 = nn.Linear(self.hid_dim, output_dim)

    def encode(self, x, x_len=None, valid_frames=None):
        if self.pack_seq and x_len is not None:
            packed = pack_padded_sequence(x, x_len, batch_first=True, enforce_sorted=True)
            packed, _ = self.encoder(packed)
            return pad_packed_sequence(packed, batch_first=True)[0]
        if valid_frames is not None:        # the time axis is padded beyond its data (pika_amd/train_graph.py)
            return self.encoder(x, valid_frames=valid_frames)
        return self.encoder(x)

    def predict(self, y):
        """Prediction network on label sequences that already start with SOS (= blank = 0)."""
        if self.decoder_type == 'rnn':
            return _lstm_forward(self.decoder, self.embed(y))
        return self.decoder(y)

    def forward(self, x, y, x_len=None, softmax=True):
        """transducer.py:73-112.  A TRAINING call on a HIP device may be served by a hipGraph replay of this very
        forward (pika_amd/train_graph.py: enabled by pika_amd.launch for the training scripts and by bench.py); same
        values, and `_forward_eager` is what gets captured."""
        from .. import train_graph
        if train_graph.wanted(self, x, softmax):
            return train_graph.forward(self, x, y, x_len, softmax)
        return self._forward_eager(x, y, x_len, softmax)

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_step_graphs", None)         # captured graphs are not part of a checkpoint (torch.save(model))
        return state

    def _forward_eager(self, x, y, x_len=None, softmax=True, valid_frames=None):
        enc = self.encode(x, x_len, valid_frames)
        sos = torch.zeros(y.shape[0], 1, dtype=torch.long, device=y.device)  # SOS = blank = 0
        pred = self.predict(torch.cat((sos, y), dim=1))
        return ops.joint(enc, pred, self.fc1, self.fc_gate, self.fc2, log_softmax=softmax, labels=y)

    def clean_hidden(self):
        """interface kept for the training scripts"""

    def reset_hidden(self, h, reset_idx):
        """interface kept for the training scripts"""
