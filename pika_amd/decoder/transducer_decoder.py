"""`TransducerDecoder` with the reference's constructor and `decode_batch` contract
(decoder/transducer_decoder.py:28-217), driving the vectorised beam state of beam_search.py.

Differences in mechanics, not in results:
* the encoder halves of fc1/fc_gate are applied ONCE to the whole encoder output; each step only
  gathers the rows of the current frame indices (the reference re-multiplies the 2H-wide
  concatenation every step, transducer_decoder.py:173-175);
* row layout is (B,K) utterance-major instead of the reference's beam-major k = beam*B + b;
* one host read per step (the loop condition), instead of thousands of `.item()` calls.
"""
import torch
import torch.nn.functional as F

from ..model import ops
from .beam_search import BeamState, EOS


class GlobalScorer(object):
    """decoder/beam_transducer.py:246-258: identity re-scorer."""

    def score(self, beam, logprobs):
        return logprobs


class TransducerDecoder(object):
    def __init__(self, model, batch_size, beam_size, n_best=1, blk=0, global_scorer=None,
                 sm_scale=1.0, lm=None, lm_scale=1.0, lm_scorer=None, lm_scorer_scale=1.0,
                 cuda=False, beam_prune=True, args=None):
        self.model = model
        self.batch_size = batch_size
        self.beam_size = beam_size
        self.n_best = n_best
        self.blk = blk
        self.global_scorer = global_scorer
        self.sm_scale = sm_scale
        self.cuda = cuda
        self.beam_prune = beam_prune
        self.lm, self.lm_scale = lm, lm_scale
        self.lm_scorer, self.lm_scorer_scale = lm_scorer, lm_scorer_scale
        self.las_rescorer = getattr(args, "las_rescorer", None)
        self.las_rescorer_bw = getattr(args, "las_rescorer_bw", None)
        if getattr(args, "bilas_rescorer", None) is not None:
            self.bilas_rescorer = args.bilas_rescorer
        self.args = args
        self.t_idx = None
        self.dec_states = None

    # ---- prediction network stepping --------------------------------------------------------
    def _pred_init(self, n, device):
        sos = torch.full((n, 1), self.blk, dtype=torch.long, device=device)
        if self.model.decoder_type == 'rnn':
            _, st = self.model.decoder(self.model.embed(sos))                 # :116-117
            return st[0].contiguous(), st[1].contiguous()
        return self.model.decoder(sos)[:, -1, :]                              # :121

    def _pred_step(self, state, inp, beam):
        """Advance the rows whose last symbol is a label (> blank); transducer_decoder.py:139-171."""
        nonblk = inp.gt(self.blk)
        if not bool(nonblk.any()):
            return state
        if self.model.decoder_type == 'rnn':
            h, c = state
            idx = nonblk.nonzero(as_tuple=False).squeeze(1)
            dec_in = self.model.embed(inp[idx].unsqueeze(1))
            _, (h2, c2) = self.model.decoder(dec_in, (h[:, idx].contiguous(), c[:, idx].contiguous()))
            h = h.index_copy(1, idx, h2)
            c = c.index_copy(1, idx, c2)
            return h, c
        # transformer prediction net: re-run on [blank] + partial hypothesis (:153-171)
        idx = nonblk.nonzero(as_tuple=False).squeeze(1)
        hyp = beam.hyp.view(-1, beam.hyp.shape[2])[idx]
        ln = beam.hyp_len.view(-1)[idx]
        L = int(ln.max()) + 1
        pad = self.model.embed.padding_idx
        seq = torch.full((idx.numel(), L), pad, dtype=torch.long, device=inp.device)
        seq[:, 0] = self.blk
        pos = torch.arange(1, L, device=inp.device).unsqueeze(0)
        body = torch.where(pos <= ln.unsqueeze(1), hyp[:, :L - 1], torch.full_like(hyp[:, :L - 1], pad))
        seq[:, 1:] = body
        out = self.model.decoder(seq)
        last = out[torch.arange(idx.numel(), device=inp.device), ln]          # position len(cur_hyp)-1
        return state.index_copy(0, idx, last)

    # ---- the search ---------------------------------------------------------------------------
    @torch.no_grad()
    def decode_batch(self, x, x_len, max_len=None):
        model, K = self.model, self.beam_size
        if model.pack_seq and x_len is not None:
            enc_out = model.encode(x, x_len)
        else:
            enc_out = model.encoder(x)                                        # :92-100
        B, T, H = enc_out.shape
        dev = enc_out.device
        V = model.output_dim
        if max_len is None:
            max_len = [None] * B
        mlen = [int(m) if (m is not None and int(m)) else 10000 for m in max_len]   # :83
        beam = BeamState(B, K, self.blk, self.n_best, mlen, V, dev, beam_prune=self.beam_prune)
        num_frames = torch.as_tensor(x_len, device=dev).long()

        # encoder halves of the joint, once
        w1, wg = model.fc1, model.fc_gate
        e1_all = ops.linear(enc_out, w1.weight[:, :H].contiguous(), w1.bias)
        eg_all = ops.linear(enc_out, wg.weight[:, :H].contiguous(), wg.bias)
        w1p, wgp = w1.weight[:, H:].contiguous(), wg.weight[:, H:].contiguous()

        t_idx = torch.full((B, K), -1, dtype=torch.long, device=dev)          # :107
        state = self._pred_init(B * K, dev)
        bidx = torch.arange(B, device=dev).unsqueeze(1).expand(B, K)

        while not bool(beam.done().all()):                                    # :123
            inp = beam.y                                                      # (B,K)  :127
            t_idx = t_idx + inp.eq(self.blk).long()                           # :129
            tg = t_idx.clamp(0, T - 1)
            state = self._pred_step(state, inp.reshape(-1), beam)
            dec_hid = state[0][-1] if model.decoder_type == 'rnn' else state  # (B*K,H)
            z1 = e1_all[bidx, tg] + ops.linear(dec_hid, w1p).view(B, K, H)
            zg = eg_all[bidx, tg] + ops.linear(dec_hid, wgp).view(B, K, H)
            h = torch.tanh(z1) * torch.sigmoid(zg)
            logits = ops.linear(h, model.fc2.weight, model.fc2.bias)
            logp = F.log_softmax(self.sm_scale * logits, dim=-1)              # :177
            prev_k = beam.advance(logp, t_idx, num_frames, self.lm_scorer_scale)   # :182
            # _beam_update :188-202: re-order prediction-net state and frame indices by parent
            flat = (bidx * K + prev_k).reshape(-1)
            if model.decoder_type == 'rnn':
                state = (state[0].index_select(1, flat), state[1].index_select(1, flat))
            else:
                state = state.index_select(0, flat)
            t_idx = t_idx.gather(1, prev_k)
        self.t_idx = t_idx
        self.dec_states = state
        preds, scores = beam.results()
        return {"predictions": preds, "scores": scores}, enc_out
