"""Vectorised restatement of BeamMergeTransducer (decoder/beam_transducer.py:20-243) for B
utterances at once.  Every rule below cites the reference line it reproduces.

State per utterance b and beam slot k (tensors (B,K) on the model's device):
  scores, lm_scores, last symbol `y` (blank / label / eos = -1), partial hypothesis (labels
  without blanks) as a padded (B,K,L) tensor + length, plus the step history (back-pointers and
  symbols) needed for the final trace-back, and the `finished` list as a fixed-capacity buffer.
"""
import torch

EOS = -1          # beam_transducer.py:45
DEAD = -1e20      # beam_transducer.py:104,113


class BeamState(object):
    def __init__(self, batch, beam, blk, n_best, max_len, vocab, device, beam_prune=True,
                 hyp_cap=64):
        B, K = batch, beam
        self.B, self.K, self.V = B, K, vocab
        self.blk, self.n_best, self.beam_prune = blk, n_best, beam_prune
        self.device = device
        self.scores = torch.zeros(B, K, device=device)                       # :34
        self.lm_scores = torch.zeros(B, K, device=device)                    # :71
        self.y = torch.full((B, K), blk, dtype=torch.long, device=device)    # :41-43
        self.max_len = torch.as_tensor(max_len, device=device).long()        # :62 (10000 if unset)
        self.hyp = torch.zeros(B, K, hyp_cap, dtype=torch.long, device=device)   # cur_part_hyp :50
        self.hyp_len = torch.zeros(B, K, dtype=torch.long, device=device)
        self.prev_ks = []    # list of (B,K) long                                :38
        self.next_ys = [self.y.clone()]                                        # :41
        self.eos_top = torch.zeros(B, dtype=torch.bool, device=device)         # :47
        cap = 4 * K + n_best + 8
        self.fin_score = torch.zeros(B, cap, device=device)                    # finished :52
        self.fin_step = torch.zeros(B, cap, dtype=torch.long, device=device)
        self.fin_k = torch.zeros(B, cap, dtype=torch.long, device=device)
        self.fin_n = torch.zeros(B, dtype=torch.long, device=device)
        self.step = 0

    # ---- helpers -------------------------------------------------------------------------
    def _grow(self, name, dim, fill=0):
        t = getattr(self, name)
        shape = list(t.shape)
        shape[dim] = t.shape[dim]
        pad = torch.full(shape, fill, dtype=t.dtype, device=t.device)
        setattr(self, name, torch.cat((t, pad), dim=dim))

    def duplicate_mask(self):
        """beam_transducer.py:106-114: a live slot is disabled if an earlier live slot holds the
        same non-empty partial hypothesis (strings of the blank-free label lists are compared)."""
        B, K = self.B, self.K
        live = self.y.ne(EOS)
        same_len = self.hyp_len.unsqueeze(2).eq(self.hyp_len.unsqueeze(1))            # (B,K,K)
        L = self.hyp.shape[2]
        pos = torch.arange(L, device=self.device)
        valid = pos.view(1, 1, L) < self.hyp_len.unsqueeze(2)                           # (B,K,L)
        h = torch.where(valid, self.hyp, torch.full_like(self.hyp, -7))
        same_tok = h.unsqueeze(2).eq(h.unsqueeze(1)).all(dim=3)                         # (B,K,K)
        nonempty = self.hyp_len.gt(0)
        earlier = torch.tril(torch.ones(K, K, dtype=torch.bool, device=self.device), diagonal=-1)
        # dup[b,i] = any j<i: live[j] & same(i,j)   (i itself must be live and non-empty)
        pair = same_len & same_tok & earlier.unsqueeze(0) & live.unsqueeze(1) & nonempty.unsqueeze(1)
        return pair.any(dim=2) & live & nonempty

    # ---- one search step -----------------------------------------------------------------------
    def advance(self, logp, t_idx, num_frames, lm_scale=1.0, topk_fn=None):
        """logp (B,K,V) log-probs of every live slot; t_idx (B,K) frame index of each slot BEFORE
        re-ordering; num_frames (B,).  Returns prev_k (B,K) for the caller to re-order its state.
        beam_transducer.py:82-187."""
        B, K, V = logp.shape
        if self.prev_ks:                                                      # :93-116
            beam_scores = logp + self.scores.unsqueeze(2) + lm_scale * self.lm_scores.unsqueeze(2)
            dead = self.y.eq(EOS)
            if self.beam_prune:
                dead = dead | self.duplicate_mask()
            beam_scores = torch.where(dead.unsqueeze(2), torch.full_like(beam_scores, DEAD), beam_scores)
            flat = beam_scores.reshape(B, K * V)
        else:
            flat = logp[:, 0, :]                                              # :118
        best, ids = flat.topk(K, dim=1, largest=True, sorted=True)            # :119-121
        prev_k = torch.div(ids, V, rounding_mode="floor")                     # :125 (legacy int '/')
        y = ids - prev_k * V                                                  # :127
        self.prev_ks.append(prev_k)
        new_scores = best - lm_scale * self.lm_scores.gather(1, prev_k)       # :131-132
        self.scores = new_scores
        self.step += 1
        n_ys = self.step + 1                                                  # len(self.next_ys)

        # finish rule :161-164  (A and B) or C
        t_parent = t_idx.gather(1, prev_k)
        fin = (y.eq(self.blk) & t_parent.eq((num_frames - 1).unsqueeze(1))) | \
              (n_ys > self.max_len).unsqueeze(1)
        # partial hypotheses :217-226: slot i inherits its parent's labels (+ y if not blank);
        # finished slots are NOT updated (they keep slot i's previous list, :181-182)
        par_hyp = self.hyp.gather(1, prev_k.unsqueeze(2).expand(-1, -1, self.hyp.shape[2]))
        par_len = self.hyp_len.gather(1, prev_k)
        emit = y.ne(self.blk) & ~fin
        if bool((par_len + 1).max() >= self.hyp.shape[2]):
            extra = torch.zeros_like(self.hyp)
            self.hyp = torch.cat((self.hyp, extra), dim=2)
            par_hyp = torch.cat((par_hyp, extra), dim=2)
        onehot = torch.arange(self.hyp.shape[2], device=self.device).view(1, 1, -1).eq(par_len.unsqueeze(2))
        new_hyp = torch.where(onehot & emit.unsqueeze(2), y.unsqueeze(2), par_hyp)
        new_len = par_len + emit.long()
        self.hyp = torch.where(fin.unsqueeze(2), self.hyp, new_hyp)
        self.hyp_len = torch.where(fin, self.hyp_len, new_len)

        # finished list :165-181, appended in slot order (global scorer returns the scores, :180)
        rank = fin.long().cumsum(1) - 1
        pos = self.fin_n.unsqueeze(1) + rank
        need = int((self.fin_n + fin.sum(1)).max())
        while need > self.fin_score.shape[1]:
            for name in ("fin_score", "fin_step", "fin_k"):
                self._grow(name, 1)
        bidx = torch.arange(B, device=self.device).unsqueeze(1).expand(B, K)
        kidx = torch.arange(K, device=self.device).unsqueeze(0).expand(B, K)
        sel = fin
        self.fin_score[bidx[sel], pos[sel]] = new_scores[sel]
        self.fin_step[bidx[sel], pos[sel]] = n_ys - 1
        self.fin_k[bidx[sel], pos[sel]] = kidx[sel]
        self.fin_n = self.fin_n + fin.sum(1)

        y = torch.where(fin, torch.full_like(y, EOS), y)                      # :166
        self.y = y
        self.next_ys.append(y)
        self.eos_top = self.eos_top | y[:, 0].eq(EOS)                         # :183-186
        return prev_k

    def done(self):
        return self.eos_top & (self.fin_n >= self.n_best)                     # :189-194

    # ---- extraction ------------------------------------------------------------------------
    def results(self):
        """sort_finished(minimum=n_best) + get_hyp for every utterance
        (beam_transducer.py:196-243, transducer_decoder.py:204-217).  Host side: the search is
        over, only K-sized lists remain."""
        ys = torch.stack(self.next_ys).cpu()          # (S+1, B, K)
        ks = torch.stack(self.prev_ks).cpu() if self.prev_ks else torch.zeros(0, self.B, self.K, dtype=torch.long)
        fin_score, fin_step, fin_k = self.fin_score.cpu(), self.fin_step.cpu(), self.fin_k.cpu()
        fin_n, scores = self.fin_n.cpu(), self.scores.cpu()
        preds, out_scores = [], []
        for b in range(self.B):
            n = int(fin_n[b])
            fin = [(fin_score[b, i], int(fin_step[b, i]), int(fin_k[b, i])) for i in range(n)]
            while len(fin) < self.n_best:                                     # :202-210 (i stays 0)
                fin.append((scores[b, 0], len(self.next_ys) - 1, 0))
            fin.sort(key=lambda a: -float(a[0]))                              # :212 (stable)
            hyps = []
            for s, times, k in fin[:self.n_best]:
                hyp = []
                for j in range(times - 1, -1, -1):                            # :238-242
                    hyp.append(ys[j + 1, b, k])
                    k = int(ks[j, b, k])
                hyps.append(hyp[::-1][:-1])                                   # strip trailing eos (:214)
            preds.append(hyps)
            out_scores.append([s for s, _, _ in fin[:self.n_best]])
        return preds, out_scores
