"""Vectorised restatement of BeamMergeTransducer (decoder/beam_transducer.py:20-243) for B
utterances at once.  Every rule below cites the reference line it reproduces.

State per utterance b and beam slot k (tensors (B,K) on the model's device):
  scores, lm_scores, last symbol `y` (blank / label / eos = -1), partial hypothesis (labels
  without blanks) as a padded (B,K,L) tensor + length, plus the step history (back-pointers and
  symbols) needed for the final trace-back, and the `finished` list as a fixed-capacity buffer.

All shapes are FIXED at construction and `advance_steady` contains no host reads, so one search
step can be captured in a hipGraph (torch.cuda.CUDAGraph) and replayed; the host only reads the
`done` flags between replays.
"""
import torch

EOS = -1          # beam_transducer.py:45
DEAD = -1e20      # beam_transducer.py:104,113


_SCALARS = {}


def _scalar_table(n, lo):
    """Object array t with t[i] = numpy.int64(i + lo) for i < n (kept and grown per offset: the symbols of a vocabulary)."""
    import numpy as np
    t = _SCALARS.get(lo)
    if t is None or len(t) < n:
        t = np.empty(n, dtype=object)
        t[:] = [np.int64(v + lo) for v in range(n)]
        _SCALARS[lo] = t
    return t


class BeamState(object):
    def __init__(self, batch, beam, blk, n_best, max_len, vocab, device, beam_prune=True,
                 lm_scorer=None, nonblk_reward=0.0, global_scorer=True):
        B, K = batch, beam
        self.lm_scorer, self.nonblk_reward, self.global_scorer = lm_scorer, nonblk_reward, global_scorer
        # FST shallow fusion keeps, per slot, the set of active LM states with their costs (:64-70);
        # sets are ragged, so this path runs on the host (one device read per step)
        self.state_sets = [[{0: 0.0} for _ in range(K)] for _ in range(B)] if lm_scorer else None
        self.fst_dev = None      # device-resident FST + state sets (include/pika_decode.h), set up below
        self.B, self.K, self.V = B, K, vocab
        self.blk, self.n_best, self.beam_prune = blk, n_best, beam_prune
        self.device = device
        self.max_len = torch.as_tensor(max_len, device=device).long()        # :62 (10000 if unset)
        # once len(next_ys) > max_len every slot finishes (:164), so the search of the slowest
        # utterance ends after at most max(max_len) + 1 steps
        self.s_cap = int(self.max_len.max()) + 2
        S, L = self.s_cap, self.s_cap + 1
        self.scores = torch.zeros(B, K, device=device)                       # :34
        self.lm_scores = torch.zeros(B, K, device=device)                    # :71
        self.y = torch.full((B, K), blk, dtype=torch.long, device=device)    # :41-43
        self.hyp = torch.zeros(B, K, L, dtype=torch.long, device=device)     # cur_part_hyp :50
        self.hyp_len = torch.zeros(B, K, dtype=torch.long, device=device)
        self.ks_hist = torch.zeros(S, B, K, dtype=torch.long, device=device)   # prev_ks :38
        self.ys_hist = torch.full((S + 1, B, K), blk, dtype=torch.long, device=device)  # next_ys :41
        self.eos_top = torch.zeros(B, dtype=torch.bool, device=device)         # :47
        self.fin_cap = K * S + 1                                               # last slot = dump
        self.fin_score = torch.zeros(B, self.fin_cap, device=device)           # finished :52
        self.fin_step = torch.zeros(B, self.fin_cap, dtype=torch.long, device=device)
        self.fin_k = torch.zeros(B, self.fin_cap, dtype=torch.long, device=device)
        self.fin_n = torch.zeros(B, dtype=torch.long, device=device)
        self.step_t = torch.zeros(1, dtype=torch.long, device=device)          # steps taken (device)
        self.steps = 0                                                         # steps taken (host)
        # constants
        self._pos = torch.arange(L, device=device).view(1, 1, L)
        self._earlier = torch.tril(torch.ones(K, K, dtype=torch.bool, device=device), diagonal=-1).unsqueeze(0)
        self._kidx = torch.arange(K, device=device).unsqueeze(0).expand(B, K).contiguous()
        self._brow = (torch.arange(B, device=device) * self.fin_cap).unsqueeze(1)
        if lm_scorer is not None and device.type == "cuda":
            self._init_device_fst()

    def _init_device_fst(self):
        """Upload the matcher's CSR tables and create the per-slot LM state sets on the device, so the FST
        update runs as one HIP launch per step (and the step stays hipGraph-capturable)."""
        import ctypes
        from .. import _lib
        m = self.lm_scorer
        f = getattr(m, "fst", None)
        if f is None or not hasattr(f, "offsets") or len(m.disambig_ids) > 4 or self.K > 64:
            return
        dev, n = self.device, self.B * self.K
        sm = _lib.lib().pika_fst_states_per_slot()
        # the FST's CSR tables are uploaded once per matcher and device (they were converted and copied for every batch:
        # 22 ms of host time at the bigram of bench.py's configs[4] leg), the per-slot state sets are this search's own
        tables = m.__dict__.setdefault("_pika_device_tables", {})
        t = tables.get(str(dev))
        if t is None or t[0] is not f:
            t = tables[str(dev)] = (f, {"off": torch.as_tensor(f.offsets, dtype=torch.int64, device=dev),
                                        "il": torch.as_tensor(f.ilabel, dtype=torch.int32, device=dev),
                                        "wt": torch.as_tensor(f.weight, dtype=torch.float32, device=dev),
                                        "ns": torch.as_tensor(f.nextstate, dtype=torch.int32, device=dev),
                                        "fin": torch.as_tensor(f.final, dtype=torch.float32, device=dev)})
        d = dict(t[1])
        d.update({
             "dis": (ctypes.c_int * max(len(m.disambig_ids), 1))(*m.disambig_ids), "ndis": len(m.disambig_ids),
             "set_n": torch.ones(n, dtype=torch.int32, device=dev),               # every slot starts as {0: 0.0}
             "set_st": torch.zeros(n, sm, dtype=torch.int32, device=dev),
             "set_cs": torch.zeros(n, sm, dtype=torch.float64, device=dev),
             "err": torch.zeros(1, dtype=torch.int32, device=dev),
             "y_raw": torch.zeros(self.B, self.K, dtype=torch.long, device=dev)})
        self.fst_dev = d

    def _fst_advance_device(self, prev_k, lm_scale, skip=None):
        import ctypes
        from .. import _lib
        d, m = self.fst_dev, self.lm_scorer
        with torch.cuda.device(self.device):
            rc = _lib.lib().pika_fst_advance(
                d["off"].data_ptr(), d["il"].data_ptr(), d["wt"].data_ptr(), d["ns"].data_ptr(), d["fin"].data_ptr(),
                int(m.max_num_arcs), int(m.max_id), int(m.backoff_id), ctypes.cast(d["dis"], ctypes.c_void_p), d["ndis"],
                prev_k.data_ptr(), d["y_raw"].data_ptr(), self.y.data_ptr(), self.blk, float(self.nonblk_reward),
                float(lm_scale), d["set_n"].data_ptr(), d["set_st"].data_ptr(), d["set_cs"].data_ptr(),
                self.lm_scores.data_ptr(), self.scores.data_ptr(), self.fin_score.data_ptr(), self.fin_n.data_ptr(),
                self.fin_cap, self.B, self.K, d["err"].data_ptr(), None if skip is None else skip.data_ptr(),
                torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "pika_fst_advance")

    def fst_overflowed(self):
        """After the search: True if a device LM state set overflowed (or a finishing slot had no final LM state,
        where the reference raises); the caller then re-decodes on the host path -- never silently wrong scores."""
        return self.fst_dev is not None and bool(int(self.fst_dev["err"].item()))

    def duplicate_mask(self):
        """beam_transducer.py:106-114: a live slot is disabled if an earlier live slot holds the
        same non-empty partial hypothesis (strings of the blank-free label lists are compared)."""
        live = self.y.ne(EOS)
        same_len = self.hyp_len.unsqueeze(2).eq(self.hyp_len.unsqueeze(1))            # (B,K,K)
        valid = self._pos < self.hyp_len.unsqueeze(2)                                   # (B,K,L)
        h = torch.where(valid, self.hyp, torch.full_like(self.hyp, -7))
        same_tok = h.unsqueeze(2).eq(h.unsqueeze(1)).all(dim=3)                         # (B,K,K)
        nonempty = self.hyp_len.gt(0)
        # dup[b,i] = any j<i: live[j] & same(i,j)   (i itself must be live and non-empty)
        pair = same_len & same_tok & self._earlier & live.unsqueeze(1) & nonempty.unsqueeze(1)
        return pair.any(dim=2) & live & nonempty

    # ---- one search step -----------------------------------------------------------------------
    def advance(self, logp, t_idx, num_frames, lm_scale=1.0):
        """logp (B,K,V) log-probs of every slot; t_idx (B,K) frame index of each slot BEFORE
        re-ordering; num_frames (B,).  Returns prev_k (B,K) for the caller to re-order its state.
        beam_transducer.py:82-187.  The first call is the reference's `else` branch (:117-118)."""
        first = self.steps == 0
        self.steps += 1
        return self._advance(logp, t_idx, num_frames, lm_scale, first)
    # (decode_batch calls _advance directly and counts `steps` itself: under graph replay the
    # Python body does not run)

    def _advance(self, logp, t_idx, num_frames, lm_scale, first):
        B, K, V = logp.shape
        if not first:                                                         # :93-116
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
        self.ks_hist.index_copy_(0, self.step_t, prev_k.unsqueeze(0))
        new_scores = best - lm_scale * self.lm_scores.gather(1, prev_k)       # :131-132
        self.scores.copy_(new_scores)
        if self.lm_scorer is not None:
            self._fst_update(prev_k, y)                                       # :135-159
        self.step_t += 1
        n_ys = self.step_t + 1                                                # len(self.next_ys), (1,)

        # finish rule :161-164  (A and B) or C
        t_parent = t_idx.gather(1, prev_k)
        fin = (y.eq(self.blk) & t_parent.eq((num_frames - 1).unsqueeze(1))) | \
              (n_ys > self.max_len).unsqueeze(1)
        # partial hypotheses :217-226: slot i inherits its parent's labels (+ y if not blank);
        # finished slots are NOT updated (they keep slot i's previous list, :181-182)
        L = self.hyp.shape[2]
        par_hyp = self.hyp.gather(1, prev_k.unsqueeze(2).expand(-1, -1, L))
        par_len = self.hyp_len.gather(1, prev_k)
        emit = y.ne(self.blk) & ~fin
        onehot = self._pos.eq(par_len.unsqueeze(2))
        new_hyp = torch.where(onehot & emit.unsqueeze(2), y.unsqueeze(2), par_hyp)
        new_len = par_len + emit.long()
        self.hyp.copy_(torch.where(fin.unsqueeze(2), self.hyp, new_hyp))
        self.hyp_len.copy_(torch.where(fin, self.hyp_len, new_len))

        # finished list :165-181, appended in slot order (global scorer returns the scores, :180);
        # entries of non-finishing slots go to the dump slot at the end of the buffer
        rank = fin.long().cumsum(1) - 1
        pos = (self.fin_n.unsqueeze(1) + rank).clamp(max=self.fin_cap - 2)
        pos = torch.where(fin, pos, torch.full_like(pos, self.fin_cap - 1))
        idx = (self._brow + pos).reshape(-1)
        fin_value = new_scores
        if self.lm_scorer is not None:
            # :165-180: `s = self.scores[i]` is a VIEW, so `s += scale * final_lm_score` also lands
            # in self.scores, and the global scorer (which returns self.scores) hands it back: the
            # final LM cost is part of the finished score with or without a global scorer
            fin_value = new_scores + lm_scale * self._fst_final(fin)
            self.scores.copy_(fin_value)
        self.fin_score.view(-1).scatter_(0, idx, fin_value.reshape(-1))
        self.fin_step.view(-1).scatter_(0, idx, (n_ys - 1).expand(B * K))
        self.fin_k.view(-1).scatter_(0, idx, self._kidx.reshape(-1))
        self.fin_n += fin.sum(1)

        y = torch.where(fin, torch.full_like(y, EOS), y)                      # :166
        self.y.copy_(y)
        self.ys_hist.index_copy_(0, self.step_t, y.unsqueeze(0))
        self.eos_top |= y[:, 0].eq(EOS)                                       # :183-186
        return prev_k

    def _fst_update(self, prev_k, y):
        """On-the-fly FST scoring of the surviving candidates (beam_transducer.py:135-159)."""
        pk, ys = prev_k.cpu().tolist(), y.cpu().tolist()
        lm = torch.zeros(self.B, self.K)
        for b in range(self.B):
            old, new = self.state_sets[b], []
            for i in range(self.K):
                parent = old[pk[b][i]]
                nxt = {}
                if ys[b][i] != self.blk:
                    ilabel = ys[b][i] + 1                                     # :139
                    for state in list(parent.keys()):
                        scores, states = self.lm_scorer.get_scores(state, ilabel)
                        for ns, cost in zip(states, scores):
                            next_cost = parent[state] + cost
                            if next_cost < nxt.get(ns, float("inf")):         # :147-149 (sic: the
                                nxt[ns] = next_cost - self.nonblk_reward      # reward is not in the test)
                else:
                    nxt = dict(parent)                                        # :151-152
                lm[b, i] = -min(nxt.values()) if nxt else -1e20               # :153-157
                new.append(nxt)
            self.state_sets[b] = new
        self.lm_scores.copy_(lm.to(self.device))

    def _fst_final(self, fin):
        """-min over active states of (cost + final cost) for finishing slots (:167-177), zero for
        the others."""
        out = torch.zeros(self.B, self.K)
        f = fin.cpu()
        for b in range(self.B):
            for i in range(self.K):
                if bool(f[b, i]):
                    best = {}
                    for state, c in self.state_sets[b][i].items():
                        fs, st = self.lm_scorer.final_score(state)
                        for s_, cost in zip(st, fs):
                            if c + cost < best.get(s_, float("inf")):
                                best[s_] = c + cost
                    out[b, i] = -min(best.values())
        return out.to(self.device)

    def fused_ok(self, chain=False):
        """The one-launch advance of this class (pika_beam_advance: a row of logits per wave in LDS, V <= 8192); chain: the
        launch chain of fused_step.FusedSearch, whose advance reads the logits thresholded and has no bound on V of its own
        (pika_beam_advance_logits_lds says what it takes)."""
        return (self.device.type == "cuda" and (self.lm_scorer is None or self.fst_dev is not None) and self.K <= 64 and
                self.K <= self.V and (chain or self.V <= 8192) and self.K * self.hyp.shape[2] * 4 <= 64 * 1024)

    def advance_fused(self, logits, t_idx, num_frames, sm_scale, lm_scale, first):
        """The whole of `_advance` + the frame-index re-ordering in ONE HIP launch
        (include/pika_decode.h).  `logits` are the raw fc2 outputs (B,K,V); `t_idx` is updated in
        place.  Returns prev_k (a persistent buffer)."""
        from .. import _lib
        if not hasattr(self, "_prev_k"):
            self._prev_k = torch.zeros(self.B, self.K, dtype=torch.long, device=self.device)
            self._eos_u8 = torch.zeros(self.B, dtype=torch.uint8, device=self.device)
            self._cand = torch.empty(self.B * self.K * self.K * 8, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            rc = _lib.lib().pika_beam_advance(
                logits.data_ptr(), float(sm_scale), int(bool(first)), self.scores.data_ptr(),
                self.lm_scores.data_ptr(), float(lm_scale), self.y.data_ptr(), t_idx.data_ptr(),
                num_frames.data_ptr(), self.max_len.data_ptr(), self.hyp.data_ptr(),
                self.hyp_len.data_ptr(), self.hyp.shape[2], self.ks_hist.data_ptr(),
                self.ys_hist.data_ptr(), self.step_t.data_ptr(), self._eos_u8.data_ptr(),
                self.fin_score.data_ptr(), self.fin_step.data_ptr(), self.fin_k.data_ptr(),
                self.fin_n.data_ptr(), self.fin_cap, self._prev_k.data_ptr(),
                None if self.fst_dev is None else self.fst_dev["y_raw"].data_ptr(), self._cand.data_ptr(), self.B,
                self.K, self.V, self.blk, int(self.beam_prune), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "pika_beam_advance")
        if self.fst_dev is not None:
            self._fst_advance_device(self._prev_k, lm_scale)
        self.step_t += 1
        self.eos_top.copy_(self._eos_u8.bool())
        return self._prev_k

    def done(self):
        return self.eos_top & (self.fin_n >= self.n_best)                     # :189-194

    # ---- extraction ------------------------------------------------------------------------
    def results(self):
        """sort_finished(minimum=n_best) + get_hyp for every utterance
        (beam_transducer.py:196-243, transducer_decoder.py:204-217).  Host side: the search is
        over, only K-sized lists remain.  All of it on whole arrays (the per-utterance Python loops over the finished
        lists were 13 of the 17 ms this took at B = 64, beam 16)."""
        import numpy as np
        S = self.steps
        on_device = self.ys_hist.is_cuda and self.B > 0      # the back-pointer walk as one launch (pika_beam_backtrack)
        if not on_device:
            ys = self.ys_hist[:S + 1].cpu().numpy()
            ks = self.ks_hist[:S].cpu().numpy()
        fin_n = self.fin_n.cpu().clamp(max=self.fin_cap - 1).numpy()
        nmax = int(fin_n.max()) if self.B else 0
        fin_score, fin_step, fin_k = (t[:, :nmax].cpu().numpy() for t in
                                      (self.fin_score, self.fin_step, self.fin_k))
        scores = self.scores.cpu().numpy()
        B, nb = self.B, self.n_best
        K = self.ys_hist.shape[2]
        # sort_finished: the finished list of an utterance, filled up to n_best with (scores[b, 0], S, 0) (:202-210, i stays 0),
        # stably sorted by descending score (:212).  Columns [0, nmax): the finished slots (valid below fin_n[b]); columns
        # [nmax, nmax + nb): the fill-ups (valid below nb - fin_n[b]); invalid columns sort behind every valid one.
        valid = np.concatenate([np.arange(nmax)[None, :] < fin_n[:, None],
                                np.arange(nb)[None, :] < (nb - fin_n)[:, None]], axis=1)
        all_score = np.concatenate([fin_score.astype(np.float32, copy=False),
                                    np.repeat(scores[:, :1].astype(np.float32, copy=False), nb, axis=1)], axis=1)
        all_step = np.concatenate([fin_step.astype(np.int64, copy=False), np.full((B, nb), S, np.int64)], axis=1)
        all_k = np.concatenate([fin_k.astype(np.int64, copy=False), np.zeros((B, nb), np.int64)], axis=1)
        order = np.lexsort((-all_score, ~valid), axis=1)[:, :nb]                # (stable; last key first)
        sel_score = np.take_along_axis(all_score, order, axis=1)
        sel_step = np.take_along_axis(all_step, order, axis=1)
        sel_k = np.take_along_axis(all_k, order, axis=1)
        # get_hyp (:234-243) for all B*n_best entries at once: walk the back-pointers from each entry's own finishing
        # step down to 0.  Entries ordered by finishing step, so that the ones still walking at step j are a prefix.
        smax = int(sel_step.max()) if B else 0
        n = B * nb
        if on_device:
            from .. import _lib
            dev = self.ys_hist.device
            sel = torch.from_numpy(np.stack((sel_step.reshape(n), sel_k.reshape(n))).astype(np.int32)).to(dev)
            out_d = torch.empty((n, max(smax, 1)), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().pika_beam_backtrack(
                    self.ys_hist.data_ptr(), self.ks_hist.data_ptr(), sel[0].data_ptr(), sel[1].data_ptr(), n, nb, B, K,
                    max(smax, 1), self.blk, out_d.data_ptr(), torch.cuda.current_stream().cuda_stream), "pika_beam_backtrack")
            out = out_d.cpu().numpy()
            return BeamState._lists(out, sel_step.reshape(n), sel_score, B, nb)
        by_len = np.argsort(-sel_step.reshape(n), kind="stable")
        steps_sorted = sel_step.reshape(n)[by_len]
        base = (np.repeat(np.arange(B), nb) * K)[by_len]                          # row b of the (B*K)-wide history rows
        idx = base + sel_k.reshape(n)[by_len]
        ys2, ks2 = ys.reshape(ys.shape[0], -1), ks.reshape(ks.shape[0], -1)
        out_t = np.full((max(smax, 1), n), self.blk, np.int64)                   # [step][entry in by_len order]
        n_act = np.searchsorted(-steps_sorted, -np.arange(smax), side="left")    # entries with step > j
        for j in range(smax - 1, -1, -1):
            m = int(n_act[j])
            cur = idx[:m]
            out_t[j, :m] = ys2[j + 1].take(cur)
            idx[:m] = base[:m] + ks2[j].take(cur)
        out = np.empty((n, max(smax, 1)), np.int64)
        out[by_len] = out_t.T
        return BeamState._lists(out, sel_step.reshape(n), sel_score, B, nb)

    @staticmethod
    def _lists(out, steps, sel_score, B, nb):
        """The (B, n_best) lists of symbols / scores of `results` from the (entries, steps) symbol table."""
        import numpy as np
        n = B * nb
        keep = np.maximum(steps - 1, 0)
        # hyp[:-1]: strip the trailing eos (:214); elements expose .item() like the reference's 0-dim tensors
        # (decode_transducer.py:139): numpy scalars / 0-dim views of one score tensor
        score_elems = torch.from_numpy(np.ascontiguousarray(sel_score, dtype=np.float32).reshape(n)).unbind(0)
        # (the ~300 k list elements are references into ONE table of numpy scalars -- they are immutable -- picked by an
        # object-array gather: creating a scalar per element was 7 of this function's 13 ms)
        lo = min(int(out.min()), 0) if out.size else 0
        elems = _scalar_table(int(out.max()) - lo + 1 if out.size else 1, lo)[out - lo if lo else out]
        preds, out_scores = [], []
        for b in range(B):
            preds.append([elems[b * nb + j, :keep[b * nb + j]].tolist() for j in range(nb)])
            out_scores.append(list(score_elems[b * nb:(b + 1) * nb]))
        return preds, out_scores
