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
from .prednet_cache import IncrementalPredNet


class GlobalScorer(object):
    """decoder/beam_transducer.py:246-258: identity re-scorer."""

    def score(self, beam, logprobs):
        return logprobs


_GC_FROZEN = [False]


def _freeze_gc_once():
    if not _GC_FROZEN[0]:
        import gc
        _GC_FROZEN[0] = True
        gc.collect()
        gc.freeze()


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
        # capture the steady-state step in a hipGraph on the GPU (use_graph = False: eager launches, for counter passes)
        self.use_graph = True
        self.fused_step = True   # fused HIP advance kernel on the GPU (include/pika_decode.h)
        self.incremental = True  # transformer prediction net: one new position per step (cached)
        # GPU, transformer prediction net: the whole step as a fixed launch chain (fused_step.py), replayed
        # `replays_per_sync` x 2 steps per host read of the stop flag.  decode_terms: bf16 terms per GEMM operand
        # (3 = fp32-exact products; 2 = hi.hi + hi.lo + lo.hi, an fp32 product to ~2^-17; 1 = plain bf16 operands)
        import os
        self.fused_search = True        # (False: the stepwise search of the tiny / CPU path on the device)
        # decode_precision "fp32" (default): fp32-grade products everywhere, on two FP16 terms per operand (22 mantissa bits:
        # an fp32 product to ~2^-22 with three MFMA products instead of the six of the exact three-term bf16 split).
        # The step products (prediction network, joint, fc2: 340 steps per batch): include/pika_decode_step.h, terms = 4
        # (second term kept at the first one's magnitude, cross products in their own accumulator).  The encoder and the
        # joint halves (once per batch): pika_amd.gemm precision "fp16x2" (three K-concatenated segments with power-of-two
        # factors, one accumulator) where the direct-to-LDS kernel takes the product, exact products elsewhere
        # (PIKA_DECODE_ENCODER_PRECISION=fp32: exact everywhere).  On the full-width golden every n-best entry separated
        # from its neighbours by > 1e-3 in score sits at its reference rank and scores agree within 3e-4 -- the same as
        # with exact products throughout (decode_precision "fp32-exact": three bf16 terms, 17 % more search time, 16 ms
        # more encoder time per batch); at bench size the encoder output differs from the exact one by the exact mode's
        # own accumulation noise (tests/test_decode_full.py).
        # "bf16x3": two bf16 terms per operand everywhere (an fp32 product to ~2^-17; encoder 20 ms faster per batch, top-1
        # and greedy hypotheses identical to the reference's, scores within 2e-3: tools/decode_two_term_check.py).
        # "bf16": plain bf16 operands.
        self.decode_precision = os.environ.get("PIKA_DECODE_PRECISION", "fp32")
        self.encoder_precision = os.environ.get("PIKA_DECODE_ENCODER_PRECISION", "fp16x2")
        self.replays_per_sync = 4
        self.groups_in_flight = 3       # groups of replays queued ahead of the host's look at the stop flag

    @property
    def decode_terms(self):
        return {"bf16": 1, "bf16x3": 2, "fp32-exact": 3}.get(self.decode_precision, 4)

    # ---- prediction network stepping (fixed shapes: every row is recomputed, rows whose last
    # symbol is not a label keep their state; transducer_decoder.py:139-171) ---------------------
    def _pred_init(self, n, device):
        sos = torch.full((n, 1), self.blk, dtype=torch.long, device=device)
        if self.model.decoder_type == 'rnn':
            _, st = self.model.decoder(self.model.embed(sos))                 # :116-117
            return [st[0].contiguous(), st[1].contiguous()]
        return [self.model.decoder(sos)[:, -1, :].contiguous()]               # :121

    def _pred_step(self, state, inp, beam, L):
        nonblk = inp.gt(self.blk)                                             # labels only (eos = -1)
        if self.model.decoder_type == 'rnn':
            h, c = state
            dec_in = self.model.embed(inp.clamp(min=0).unsqueeze(1))
            _, (h2, c2) = self.model.decoder(dec_in, (h, c))
            m = nonblk.view(1, -1, 1)
            h.copy_(torch.where(m, h2, h))
            c.copy_(torch.where(m, c2, c))
            return
        if self._inc is not None:
            self._inc.step(state[0], inp, beam.hyp_len.view(-1), beam.step_t, L)
            return
        # transformer prediction net: re-run on [blank] + partial hypothesis, padded to the
        # current bucket length L (:153-171); position len(hyp) holds the new state
        hyp = beam.hyp.view(-1, beam.hyp.shape[2])[:, :L - 1]
        ln = beam.hyp_len.view(-1).clamp(max=L - 1)
        pad = self.model.embed.padding_idx
        pos = torch.arange(1, L, device=inp.device).unsqueeze(0)
        body = torch.where(pos <= ln.unsqueeze(1), hyp, torch.full_like(hyp, pad))
        seq = torch.cat((torch.full_like(body[:, :1], self.blk), body), dim=1)
        out = self.model.decoder(seq)
        last = out.gather(1, ln.view(-1, 1, 1).expand(-1, 1, out.shape[2])).squeeze(1)
        state[0].copy_(torch.where(nonblk.unsqueeze(1), last, state[0]))

    # A decode process holds ~1e6 live Python objects (the models' modules and parameters, FST / trie tables, the n-best
    # lists of 0-dim tensors the scripts index); every FULL collection of the interpreter's garbage collector walks them
    # all: a 45 ms stall every fourth batch of decode_transducer.py at B = 64 (profiles/r5_las_pass_overlap.txt).  After the
    # second batch -- models built, tables and caches warm -- the decoder moves what is alive into the collector's permanent
    # generation once per process (gc.freeze()); collections keep running over everything made afterwards.
    # `TransducerDecoder.freeze_gc = False` (class or instance) leaves the collector alone.
    freeze_gc = True

    # ---- the search ---------------------------------------------------------------------------
    @torch.no_grad()
    def decode_batch(self, x, x_len, max_len=None):
        from .. import gemm as G
        old = G.PRECISION
        if x.is_cuda:
            if self.decode_precision not in ("fp32", "fp32-exact", "fp16x2", "bf16x3", "bf16"):
                raise ValueError("unknown decode_precision %r" % (self.decode_precision,))
            # the large products of a decode (encoder, joint halves: once per batch): two bf16 terms / one term in those
            # modes; otherwise fp32-grade -- PIKA_DECODE_ENCODER_PRECISION = "fp32" (three bf16 terms, exact products, six
            # segments) or "fp16x2" (two fp16 terms, ~2^-22, three segments; pika_amd.gemm)
            G.PRECISION = self.decode_precision if self.decode_precision in ("bf16x3", "bf16") else (
                "fp32" if self.decode_precision == "fp32-exact" else self.encoder_precision)
        try:
            ret, enc_out = self._decode_batch(x, x_len, max_len)
        finally:
            G.PRECISION = old
        # what las_rescore / bilas_rescore will be asked about next (batch-ahead scoring, below)
        self._nbest = {"enc_out": enc_out, "hyps": ret["predictions"], "scores": {}}
        self._batches = getattr(self, "_batches", 0) + 1
        if self._batches == 2 and x.is_cuda and self.freeze_gc:
            _freeze_gc_once()
        return ret, enc_out

    def _decode_batch(self, x, x_len, max_len=None):
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
        beam = BeamState(B, K, self.blk, self.n_best, mlen, V, dev, beam_prune=self.beam_prune,
                         lm_scorer=self.lm_scorer,
                         nonblk_reward=getattr(self.args, "nonblk_reward", 0.0),
                         global_scorer=self.global_scorer is not None)
        num_frames = torch.as_tensor(x_len, device=dev).long()
        rnn = model.decoder_type == 'rnn'

        # encoder halves of the joint, once
        w1, wg = model.fc1, model.fc_gate
        e1_all = ops.linear(enc_out, w1.weight[:, :H].contiguous(), w1.bias)
        eg_all = ops.linear(enc_out, wg.weight[:, :H].contiguous(), wg.bias)
        w1p, wgp = w1.weight[:, H:].contiguous(), wg.weight[:, H:].contiguous()
        # per step: both prediction halves of the joint as ONE GEMM, one gather of the encoder halves
        e_all = torch.cat((e1_all, eg_all), dim=2).reshape(B * T, 2 * H)          # row b*T + t = [e1 | eg]
        wp = torch.cat((w1p, wgp), dim=0).contiguous()                            # (2H, H)
        brow = (torch.arange(B, device=dev) * T).unsqueeze(1)

        if enc_out.is_cuda and self.fused_search and self.fused_step:
            from . import fused_step
            if fused_step.supported(model, beam, K) and (self.lm_scorer is None or beam.fst_dev is not None):
                return self._search_fused(beam, e_all, T, num_frames, enc_out, x, x_len, max_len)
        t_idx = torch.full((B, K), -1, dtype=torch.long, device=dev)          # :107
        self._inc = None
        if not rnn and self.incremental:
            self._inc = IncrementalPredNet(model.decoder, B * K, beam.s_cap, beam.hyp.shape[2] + 1,
                                           dev, self.blk)
            state = [self._inc.state0.clone()]
        else:
            state = self._pred_init(B * K, dev)
        bidx = torch.arange(B, device=dev).unsqueeze(1).expand(B, K)
        flags = torch.zeros(2, dtype=torch.long, device=dev)                  # [all done, max hyp len]

        def step(first, L):
            inp = beam.y                                                      # (B,K)  :127
            t_idx.add_(inp.eq(self.blk).long())                               # :129
            tg = t_idx.clamp(0, T - 1)
            if not first:
                self._pred_step(state, inp.reshape(-1), beam, L)
            dec_hid = state[0][-1] if rnn else state[0]                       # (B*K,H)
            z = e_all.index_select(0, (brow + tg).reshape(-1)) + ops.linear(dec_hid, wp)   # (B*K, 2H)
            h = (torch.tanh(z[:, :H]) * torch.sigmoid(z[:, H:])).view(B, K, H)
            logits = ops.linear(h, model.fc2.weight, model.fc2.bias)
            if fused:
                prev_k = beam.advance_fused(logits.contiguous(), t_idx, num_frames, self.sm_scale,
                                            self.lm_scorer_scale, first)
            else:
                logp = F.log_softmax(self.sm_scale * logits, dim=-1)          # :177
                prev_k = beam._advance(logp, t_idx, num_frames, self.lm_scorer_scale, first)   # :182
            # _beam_update :188-202: re-order prediction-net state and frame indices by parent
            flat = (bidx * K + prev_k).reshape(-1)
            for s_ in state:
                s_.copy_(s_.index_select(1 if rnn else 0, flat))
            if self._inc is not None:
                self._inc.reorder(flat)
            if not fused:
                t_idx.copy_(t_idx.gather(1, prev_k))
            flags[0] = beam.done().all().long()
            flags[1] = beam.hyp_len.max()

        def bucket(max_hyp):  # prefix length (SOS + labels) the prediction net attends over
            return 2 if rnn else min(beam.hyp.shape[2] + 1, ((max_hyp + 1 + 1 + 15) // 16) * 16)

        import time as _time
        _t0 = _time.perf_counter()
        fused = self.fused_step and beam.fused_ok()
        use_graph = enc_out.is_cuda and self.use_graph and (self.lm_scorer is None or (fused and beam.fst_dev is not None))
        graphs = {}
        n_eager = 0
        step(True, 2)
        beam.steps += 1
        f = flags.tolist()
        while not f[0]:                                                       # :123
            L = bucket(f[1])
            if use_graph and n_eager >= 2:
                if L not in graphs:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        step(False, L)
                    graphs[L] = g
                graphs[L].replay()
            else:
                step(False, L)
                n_eager += 1
            beam.steps += 1
            f = flags.tolist()
        self.t_idx = t_idx
        self.dec_states = tuple(state) if rnn else state[0]
        _t1 = _time.perf_counter()
        if beam.fst_overflowed():
            # a per-slot LM state set outgrew the device arrays (deep back-off chains x disambiguation arcs):
            # decode this batch again with the host-side state sets, which are unbounded like the reference's
            import warnings
            warnings.warn("pika_amd: device FST state sets overflowed; re-decoding the batch on the host FST path")
            keep = self.fused_step
            self.fused_step = False
            try:
                return self._decode_batch(x, x_len, max_len)
            finally:
                self.fused_step = keep
        preds, scores = beam.results()
        self.timing = {"search_s": _t1 - _t0, "results_s": _time.perf_counter() - _t1,
                       "steps": beam.steps, "graphs": len(graphs)}
        return {"predictions": preds, "scores": scores}, enc_out


def _search_fused(self, beam, e_all, T, num_frames, enc_out, x, x_len, max_len):
    """The search loop on the fixed launch chain of fused_step.FusedSearch: two steps (the double-buffered
    prediction-net state alternates) per hipGraph, `replays_per_sync` replays per host read."""
    import time as _time
    from . import fused_step
    _t0 = _time.perf_counter()
    fs = fused_step.make(self.model, beam, e_all, T, num_frames, self.sm_scale, self.lm_scorer_scale, terms=self.decode_terms)
    fs.step_launches(0)
    fs.step_launches(1)
    graph, n_replays = None, 0
    if self.use_graph:
        # The host never waits for the device while the search runs: groups of `replays_per_sync` replays (2 steps each)
        # are kept `groups_in_flight` deep, each followed by an asynchronous copy of the stop flag into its own pinned
        # word; the host looks at the OLDEST group's word (its event has long completed by the time the queue is full).
        # Launches behind the end of the search are no-ops (every state-mutating kernel returns when `stop` is set).
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fs.step_launches(0)
            fs.step_launches(1)
        depth = max(int(self.groups_in_flight), 1)
        flags = torch.zeros(depth, dtype=torch.int32).pin_memory()
        events = [torch.cuda.Event() for _ in range(depth)]
        queued = 0
        while True:
            slot = queued % depth
            if queued >= depth:                     # the slot's previous group: had the search ended by then?
                events[slot].synchronize()
                if int(flags[slot]):
                    break
            for _ in range(self.replays_per_sync):
                graph.replay()
            n_replays += self.replays_per_sync
            flags[slot:slot + 1].copy_(fs.stop, non_blocking=True)
            events[slot].record()
            queued += 1
    else:
        while not int(fs.stop.item()):
            fs.step_launches(0)
            fs.step_launches(1)
    beam.steps = int(beam.step_t.item())
    self.dec_states, self.t_idx = fs.final_state(beam.steps)
    _t1 = _time.perf_counter()
    if beam.fst_overflowed():
        import warnings
        warnings.warn("pika_amd: device FST state sets overflowed; re-decoding the batch on the host FST path")
        keep = self.fused_step
        self.fused_step = False
        try:
            return self._decode_batch(x, x_len, max_len)
        finally:
            self.fused_step = keep
    preds, scores = beam.results()
    self.timing = {"search_s": _t1 - _t0, "results_s": _time.perf_counter() - _t1, "steps": beam.steps,
                   "graphs": 0 if graph is None else 1, "replays": n_replays, "launches_per_step": fs.launches_per_step(),
                   "terms": self.decode_terms}
    return {"predictions": preds, "scores": scores}, enc_out


TransducerDecoder._search_fused = _search_fused


def _las_scores(rescorer, x, tgt, scale=1.0):
    """Shared body of las_rescore / bilas_rescore (decoder/transducer_decoder.py:219-253)."""
    lens = torch.IntTensor([x.size(0)])
    outputs, _, _, _ = rescorer(x, tgt, lens)
    logp = F.log_softmax(scale * rescorer.dec_proj(outputs), dim=-1).squeeze(1)
    idx = tgt[1:].squeeze(-1).squeeze(-1)
    return logp[torch.arange(idx.size(0)), idx].tolist()


def _batch_ahead(self, which, rescorer, x, tgt, scale):
    """The reference decode script asks for the rescorer's scores ONE hypothesis at a time, 2 x batch x n_best calls per
    decode batch (decode_transducer.py:136-156), each a batch-1 encoder pass and a batch-1 token loop.  The n-best lists
    it will ask about are known since decode_batch returned them: the first call of a batch (per rescorer) scores ALL of
    them in one batched pass (las.Net.score_nbest_batch: one encoder pass, one token loop whose entries share their
    common prefixes) and the calls look their answer up -- same values (tests/test_las.py).  Returns None when the call
    is not about the last decode batch (or PIKA_LAS_BATCH_AHEAD=0): the caller then scores the hypothesis on its own."""
    import os
    nb = getattr(self, "_nbest", None)
    if nb is None or os.environ.get("PIKA_LAS_BATCH_AHEAD", "1") == "0" or not hasattr(rescorer, "score_nbest_batch"):
        return None
    enc = nb["enc_out"]
    if x.dim() != 3 or x.shape[1] != 1 or x.shape[0] != enc.shape[1] or x.device != enc.device or tgt.shape[0] < 2:
        return None
    row = enc.stride(0) * enc.element_size()
    off = x.data_ptr() - enc.data_ptr()
    if off < 0 or off % row or off // row >= enc.shape[0] or x.stride(0) != enc.stride(1):
        return None                                        # not a row of the last batch's encoder output
    i = off // row
    toks = tgt.view(-1).tolist()
    sos, eos, hyp = toks[0], toks[-1], tuple(toks[1:-1])
    key = (which, sos, eos)
    table = nb["scores"].get(key)
    if table is None:
        B = enc.shape[0]

        blk = self.blk
        # (300 k elements per batch of 64 x 16 entries: the interpreter's own filter / map take 6 ms, a comprehension with two
        #  int() calls per element 40)
        labels_of = (lambda h: list(map(int, filter(None, h)))) if blk == 0 else (lambda h: [v for v in map(int, h) if v != blk])

        def lists_of(which_):
            lists = [[labels_of(h) for h in nb["hyps"][b]] for b in range(B)]
            return [[h[::-1] for h in row_] for row_ in lists] if which_ == "bw" else lists
        # The script asks the forward and the backward rescorer about the same batch one after the other
        # (decode_transducer.py:136-156, same SOS / EOS): both passes are run at the first request
        # (las.score_nbest_batch_many; self.las_pair = False: each at its own first request)
        jobs = [(which, rescorer)]
        other = {"fw": "bw", "bw": "fw"}.get(which)
        r2 = {"fw": getattr(self, "las_rescorer", None), "bw": getattr(self, "las_rescorer_bw", None)}.get(other)
        if (r2 is not None and hasattr(r2, "score_nbest_batch") and (other, sos, eos) not in nb["scores"]
                and getattr(self, "las_pair", True)):
            jobs.append((other, r2))
        lists = [lists_of(w) for w, _ in jobs]
        src = enc.transpose(0, 1)
        if all(hasattr(r, "_batch_stages") for _, r in jobs):
            from ..model.las import score_nbest_batch_many
            got = score_nbest_batch_many([(r, src, [enc.shape[1]] * B, ls, sos, eos, scale) for (_, r), ls in zip(jobs, lists)])
        else:       # (a rescorer of another class that offers the batched call)
            got = [r.score_nbest_batch(src, [enc.shape[1]] * B, ls, sos, eos, scale=scale) for (_, r), ls in zip(jobs, lists)]
        for (w, _), ls, g in zip(jobs, lists, got):
            nb["scores"][(w, sos, eos)] = {(b, tuple(h)): sc for b in range(B) for h, sc in zip(ls[b], g[b])}
        table = nb["scores"][key]
    return table.get((i, hyp))


def las_rescore(self, x, tgt, bw=False):
    """x (T,1,C) encoder output of one utterance, tgt (L,1,1) = [SOS] + hyp + [EOS]."""
    with torch.no_grad():
        rescorer = self.las_rescorer_bw if bw else self.las_rescorer
        got = _batch_ahead(self, "bw" if bw else "fw", rescorer, x, tgt, 1.0)
        return got if got is not None else _las_scores(rescorer, x, tgt)


def bilas_rescore(self, x, tgt):
    with torch.no_grad():
        got = _batch_ahead(self, "bi", self.bilas_rescorer, x, tgt, 0.5)
        return got if got is not None else _las_scores(self.bilas_rescorer, x, tgt, scale=0.5)


TransducerDecoder.las_rescore = las_rescore
TransducerDecoder.bilas_rescore = bilas_rescore
