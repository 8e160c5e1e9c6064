"""LAS rescorer, inference scoring path only (reference trainer/model/las.py
:51-90 forward, :522-565 LASRNNEncoder, :600-683 InputFeedRNNDecoder, modules/global_attention.py
:162-248 mlp attention, modules/stacked_rnn.py:20-34).  The module tree and parameter names are the
reference's, so trained rescorers load by state_dict; the configurations the recipes use are
implemented (LSTM, bidirectional encoder, input feeding, mlp/general/dot attention, no coverage /
copy / context gate / downsampler) and anything else raises.

`score_nbest` is the MI355X-shaped entry: all hypotheses of an utterance are scored in ONE batched
decoder pass over ONE encoder pass (the reference re-runs the BLSTM encoder and a batch-1 decoder
loop per hypothesis, decoder/transducer_decoder.py:219-236)."""
import ctypes
import os
import time

import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from . import ops


def _h2d(x, dev, dtype=None):
    """Small host array -> device.  (Measured on this runtime, ROCm 7.2: staging these arrays through pinned memory and
    non-blocking copies -- so that a second scoring pass' host work could run under the first one's kernels -- made the
    kernels that followed slower, 58 -> 76 ms for the two passes of a decode batch; profiles/r5_las_pass_overlap.txt.  The
    plain copy from pageable memory waits for the stream, which paces the host to the device.)"""
    t = torch.from_numpy(x) if not torch.is_tensor(x) else x
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


def _h2d_many(arrays, dev):
    """Several small host arrays -> device as ONE copy (each copy from pageable memory is a staged transfer and a wait of its
    own: ~0.2 ms, 14 of them per scoring pass): {name: ndarray} -> {name: tensor}, views into one uploaded byte buffer."""
    import numpy as np
    arrays = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
    offs, total = {}, 0
    for k, v in arrays.items():
        offs[k] = total
        total += (v.nbytes + 15) & ~15
    blob = np.zeros(max(total, 16), np.uint8)
    for k, v in arrays.items():
        blob[offs[k]:offs[k] + v.nbytes] = v.view(np.uint8).reshape(-1)
    d = torch.from_numpy(blob).to(dev)
    return {k: d[offs[k]:offs[k] + v.nbytes].view(torch.from_numpy(v[:0].reshape(-1)).dtype).view(v.shape)
            for k, v in arrays.items()}


def _phase_timer(net, dev):
    """net.want_phase_times = True: wall time of the phases of a scoring pass (device-synchronised) into net.phase_times."""
    if not getattr(net, "want_phase_times", False) or dev.type != "cuda":
        return lambda name: None
    import time
    torch.cuda.synchronize(dev)
    state = {"t": time.perf_counter()}
    net.phase_times = []

    def tick(name):
        torch.cuda.synchronize(dev)
        now = time.perf_counter()
        net.phase_times.append((name, (now - state["t"]) * 1e3))
        state["t"] = now
    return tick


def scoring_plan(flat, own_h, sos, eos, pad, share):
    """scoring_plan_slices run to its end."""
    it = scoring_plan_slices(flat, own_h, sos, eos, pad, share)
    while True:
        try:
            next(it)
        except StopIteration as done:
            return done.value


def scoring_plan_slices(flat, own_h, sos, eos, pad, share):
    """(A generator: it yields between slices of ~30-300 us of host work -- one trie level, then the remaining array steps --
    so that a caller can spread the plan of the NEXT pass over the launches of this pass' token loop; the plan is the
    generator's return value.)

    Host-side plan of one scoring pass over the hypotheses `flat` (lists of labels; hypothesis i belongs to utterance
    own_h[i]) -- everything `Net._score_flat` hands to the token loop, as whole-array numpy (the per-hypothesis Python loops
    and dict tries this replaces were 5 of the 7 ms of host work per pass at 64 x 16 hypotheses):

    ntok (n,)     decoder steps of a hypothesis = len + 1 (its tokens after <sos>, then <eos>); L = max
    perm, tok     rows of the token loop = hypotheses by descending ntok (stable); tok (L, n): row `col` feeds
                  [sos, h_0, h_1, ...] (padded with `pad`)
    first, end    row `col` takes part in steps first <= t < end.  With `share`: rep[i][t] = the FIRST hypothesis of the
                  utterance whose first t labels are flat[i][:t] (a trie per utterance, built level by level: the class of a
                  prefix of length t+1 = (class of its first t labels, label t)); act[i] = the first t at which rep[i][t] is
                  i itself = the step at which i leaves every earlier entry's prefix and needs a row of its own (L + 1:
                  never -- a duplicate or a prefix of an earlier entry); first = act.  Without: first = 0
    forks         (fork_off (L+1,), dst, src) ordered by step: row dst starts at step t > 0 from the recurrent state of row src
                  = the row that computed its prefix so far; None without `share`
    pair_*        the (step, hypothesis) pairs that exist, hypothesis-major: step, the ROW that computes that step for the
                  hypothesis, and the token it predicts there (h_t, then eos)"""
    import numpy as np
    n = len(flat)
    lens = np.fromiter(map(len, flat), np.int64, n)
    ntok = lens + 1
    L = int(ntok.max())
    labels = np.full((n, L), pad, np.int64)                 # labels[i, t] = h_t; column L - 1 only ever holds padding / eos
    for i, h in enumerate(flat):
        labels[i, :lens[i]] = h
    ar = np.arange(n)
    steps = np.arange(L)[None, :]
    rep = act = None
    if share:
        own_h = np.asarray(own_h, np.int64)
        # rep[i, t] for ALL levels at once (the level-by-level np.unique this replaces was 1.9 of the plan's 3.6 ms): rows
        # [owner, label 0 .. label L-2] sorted as byte strings -- any lexicographic order puts the rows that share a prefix
        # next to each other; a hypothesis that has ended holds `pad` where a longer one holds a label, so the two never
        # share a level beyond its end -- then, per level t, a sorted row starts a new class where it shares fewer than
        # t + 1 leading columns with its predecessor, and a class is represented by its smallest hypothesis index
        M = np.empty((n, L), np.int64)
        M[:, 0] = own_h
        M[:, 1:] = labels[:, :L - 1]
        o = np.argsort(M.view(np.dtype((np.void, 8 * L))).ravel(), kind="stable")
        Ms = M[o]
        neq = Ms[1:] != Ms[:-1]
        lcp = np.where(neq.any(1), neq.argmax(1), L)                    # leading columns shared with the predecessor
        start = np.ones((L, n), bool)
        start[:, 1:] = lcp[None, :] < (np.arange(L)[:, None] + 1)       # [level][sorted position]
        yield
        gid = np.cumsum(start, axis=1) - 1
        mins = np.minimum.reduceat(np.tile(o, L), np.flatnonzero(start.ravel()))
        first_group = np.concatenate(([0], np.cumsum(start.sum(1))[:-1]))
        rep = np.empty((n, L), np.int64)
        rep[o, :] = mins[first_group[:, None] + gid].T
        yield
        own = (rep == ar[:, None]) & (steps <= lens[:, None])
        act = np.where(own.any(1), own.argmax(1), L + 1).astype(np.int64)
    perm = np.argsort(-ntok, kind="stable")
    first = act[perm] if share else np.zeros(n, np.int64)
    end = ntok[perm]
    col_of = np.empty(n, np.int64)
    col_of[perm] = ar
    tok = np.full((L, n), pad, np.int64)
    tok[0, :] = sos
    if L > 1:
        body = labels[perm, :L - 1].T                       # step t + 1 feeds label t ...
        tok[1:, :] = np.where(np.arange(1, L)[:, None] < end[None, :], body, pad)        # ... while the row has steps left
    yield
    forks = None
    if share:
        f_i = np.nonzero((act > 0) & (act <= L))[0]
        f_t = act[f_i]
        order = np.argsort(f_t, kind="stable")
        fork_off = np.searchsorted(f_t[order], np.arange(L + 1)).astype(np.int32)
        forks = (fork_off, col_of[f_i].astype(np.int32)[order], col_of[rep[f_i, f_t - 1]].astype(np.int32)[order])
    yield
    ii, tt = np.nonzero(steps < ntok[:, None])              # hypothesis-major, steps ascending
    rr = col_of[rep[ii, tt]] if share else col_of[ii]
    target = labels.copy()
    target[ar, lens] = eos
    return {"L": L, "ntok": ntok, "perm": perm, "tok": tok, "first": first, "end": end, "forks": forks,
            "row_steps": int(np.maximum(end - first, 0).sum()), "pair_step": tt, "pair_row": rr, "pair_target": target[ii, tt],
            "rep": rep, "act": act}


def _run_stages(gens):
    """Scoring passes (Net._score_stages generators), one after the other, each in its stages: (A) host plan; (B) encoder,
    uploads, loop preparation, then the token loop; (C) the tail queued; (D) the wait and the lists.  What overlaps
    (tools/las_stage_clock.py: the host clock of a call):
      * a token loop is bound by its kernels: the launching thread waits inside every graph launch for room in the queue
        (with the interpreter lock held -- a planner THREAD made no progress until the last launch).  The next pass' plan
        (pure numpy, ~3.6 ms) is cut into slices of 30-300 us and one slice runs between two launches; it used to run
        after the last launch, with the device idle;
      * the next pass' (B) -- its uploads, encoder, loop preparation -- is queued right behind this pass' tail, and its token
        loop too, BEFORE this pass' (D): the wait for the scores and the list building happen under the next loop's kernels.
    Tried on this runtime (ROCm 7.2, graph packet capture off) and measured worse -- profiles/r5_las_pass_overlap.txt:
    the second pass' device work under the first pass' kernels (its replays, launched into a busy stream, cost 0.4-1.3 ms of
    host time each); both passes' token loops as one replayed graph (a pass' weights and projected encoder outputs,
    ~215 MB at B = 64, stay in the Infinity Cache from token to token; two passes' do not: 1.3 x the kernel time); the
    loops launched from Python without a graph (twice the device time)."""
    out = []
    if not gens:
        return out
    next(gens[0])                                   # (A) of the first pass
    loop = next(gens[0])                            # (B) of the first pass
    waiting = None                                  # the pass whose (D) is still to come
    for k, g in enumerate(gens):
        nxt = gens[k + 1] if k + 1 < len(gens) else None
        advance = next(nxt) if nxt is not None else None        # (A) of the next pass, as slices
        drive_token_loops([loop], between=advance)  # this pass' token loop queued
        if waiting is not None:
            out.append(next(waiting))               # (D) of the previous pass, under this pass' token loop
        next(g)                                     # (C)
        if nxt is not None:
            loop = next(nxt)                        # rest of the plan, then (B) of the next pass behind this pass' tail
        waiting = g
    out.append(next(waiting))                       # (D) of the last pass
    return out


def score_nbest_batch_many(jobs):
    """Several rescorers on one decode batch (the forward and the backward LAS of decode_transducer.py:136-156):
    jobs = [(net, src, lengths, hyps, sos, eos, scale), ..] -> [net.score_nbest_batch(src, lengths, hyps, sos, eos, scale), ..].
    The passes run one after the other (_run_stages says what else was tried)."""
    with torch.no_grad():
        preps = [net._batch_prep(src, lengths, hyps) for net, src, lengths, hyps, _, _, _ in jobs]
        for _ in range(2):      # (a second round only if a persistent encoder launch was not resident: status_ok)
            gens = [net._batch_stages(prep, sos, eos, scale, first=(k == 0))
                    for k, ((net, _, _, _, sos, eos, scale), prep) in enumerate(zip(jobs, preps))]
            scores = _run_stages(gens)
            # EVERY rescorer's status word is read (no short-circuit): each encoder whose launch was not resident
            # switches to nn.LSTM for the repeat, so the repeat cannot fail the same way
            ok = [net.encoder.status_ok() for net, _, _, _, _, _, _ in jobs]
            if all(ok):
                break
        else:
            raise RuntimeError("pika_amd LAS rescoring: the BLSTM encoder pass did not complete twice in a row "
                               "(persistent launch not resident, then the repeat); no scores are returned")
        out = []
        for (net, _, _, hyps, _, _, _), sc in zip(jobs, scores):
            res, i = [], 0
            for h in hyps:
                res.append(sc[i:i + len(h)])
                i += len(h)
            out.append(res)
        return out


class TokenLoop(object):
    """A prepared token loop of the fused decoder (InputFeedRNNDecoder._prepare_fused): step() queues one token's launches
    -- every per-token quantity is a device word the step's first launch advances, past the loop's L tokens the launches
    find zero active rows -- and outs (L, N, H) receives the outputs."""

    def __init__(self, step, L, outs, dev, sig=None):
        self.step, self.L, self.outs, self.dev, self.sig = step, L, outs, dev, sig


_warmed = set()     # launch configurations (TokenLoop.sig) whose kernels have run once in this process


def drive_token_loops(loops, between=None):
    """Run prepared token loops (between: host work in slices, one call after every launch of the replayed sequence): ONE captured launch sequence -- a token of every loop -- replayed max(L) times
    (PIKA_LAS_GRAPH=0: the same launches from Python).  The first time a launch configuration is seen its token 0 runs
    eagerly, outside the capture (kernel attributes are set at a kernel's first launch), and the graph takes the rest."""
    loops = [lp for lp in loops if lp is not None]
    if not loops:
        return
    dev = loops[0].dev
    with torch.cuda.device(dev):
        n_more = max(lp.L for lp in loops)
        cold = any(lp.sig is None or lp.sig not in _warmed for lp in loops)
        if cold or n_more <= 1 or os.environ.get("PIKA_LAS_GRAPH", "1") == "0":
            for lp in loops:
                lp.step()
                _warmed.add(lp.sig)
            n_more -= 1
        if n_more <= 0:
            return
        if os.environ.get("PIKA_LAS_GRAPH", "1") != "0":
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream(dev)
            side.wait_stream(cur)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                graph.capture_begin(capture_error_mode="thread_local")
                for lp in loops:
                    lp.step()
                graph.capture_end()
            cur.wait_stream(side)
            for _ in range(n_more):
                graph.replay()
                if between is not None:
                    between()
        else:
            for _ in range(n_more):
                for lp in loops:
                    lp.step()


class LASRNNEncoder(nn.Module):
    def __init__(self, rnn_type, bidirectional, num_layers, hidden_size, dropout, input_dim):
        super().__init__()
        if rnn_type != "LSTM":
            raise NotImplementedError("LAS rescorer: only rnn_type LSTM is on the hot path")
        dirs = 2 if bidirectional else 1
        assert hidden_size % dirs == 0
        self.no_pack_padded_seq = False
        self.rnn = nn.LSTM(input_size=input_dim, hidden_size=hidden_size // dirs, num_layers=num_layers,
                           dropout=dropout, bidirectional=bidirectional)

    def _fused_ok(self, input, lengths, hidden):
        """The persistent-kernel path (include/pika_las.h: pika_blstm_layer): inference on a HIP device, fp32, zero
        initial state, hidden size per direction a multiple of 128 up to 512.  PIKA_LAS_BLSTM=0 keeps nn.LSTM."""
        r = self.rnn
        return (input.is_cuda and lengths is not None and hidden is None and not torch.is_grad_enabled()
                and not self.training and input.dtype == torch.float32 and input.dim() == 3 and r.bias
                and not r.batch_first and r.proj_size == 0 and r.hidden_size % 128 == 0 and r.hidden_size <= 512
                and r.input_size % 8 == 0 and os.environ.get("PIKA_LAS_BLSTM", "1") != "0")

    def _forward_fused(self, input, lengths):
        """nn.LSTM over pack_padded_sequence(input, lengths) -> pad_packed_sequence, per layer: the input projections of
        every position and both directions as ONE GEMM (fp32-grade products), then the recurrence of both directions as ONE
        persistent launch whose workgroups keep their slice of W_hh in registers (pika_amd/csrc/blstm.hip)."""
        import ctypes
        from .. import _lib
        from .. import gemm as G
        lib = _lib.lib()
        r = self.rnn
        dev = input.device
        D, H = (2 if r.bidirectional else 1), r.hidden_size
        S, B, _ = input.shape
        lens_h = torch.as_tensor(lengths).view(-1).to(torch.int64).cpu()
        s_out = int(lens_h.max())
        lens_d = _h2d(lens_h, dev, torch.int32)
        x = input[:s_out].contiguous()
        S = s_out
        hs, cs = [], []
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            wbytes = lib.pika_blstm_work_bytes(S, B, D, H)
            if wbytes < 0:
                return None
            work = torch.empty(int(wbytes), dtype=torch.uint8, device=dev)
            for k in range(r.num_layers):
                sfx = ["", "_reverse"][:D]
                w_ih = torch.cat([getattr(r, "weight_ih_l%d%s" % (k, s)).detach() for s in sfx], 0).float().contiguous()
                w_hh = torch.stack([getattr(r, "weight_hh_l%d%s" % (k, s)).detach() for s in sfx], 0).float().contiguous()
                bias = torch.cat([(getattr(r, "bias_ih_l%d%s" % (k, s)) + getattr(r, "bias_hh_l%d%s" % (k, s))).detach()
                                  for s in sfx], 0).float().contiguous()
                packed = torch.empty(int(lib.pika_blstm_packed_bytes(D, H)), dtype=torch.uint8, device=dev)
                _lib.check(lib.pika_blstm_pack(w_hh.data_ptr(), D, H, packed.data_ptr(), st), "pika_blstm_pack")
                # fp32-grade products (two fp16 terms on the direct-to-LDS kernel; exact products for small shapes)
                gx = G.gemm_nt(x.view(S * B, x.shape[2]), w_ih, bias=bias, precision="fp16x2")     # (S*B, D*4H)
                out = torch.empty((S, B, D * H), device=dev)
                h_n, c_n = torch.empty((D, B, H), device=dev), torch.empty((D, B, H), device=dev)
                rc = lib.pika_blstm_layer(gx.data_ptr(), packed.data_ptr(), lens_d.data_ptr(), out.data_ptr(),
                                          h_n.data_ptr(), c_n.data_ptr(), work.data_ptr(), int(wbytes), S, B, D, H, st)
                if rc == -2 and k == 0:          # PIKA_ETOOBIG: more workgroups than CUs (the grid must be resident)
                    return None
                _lib.check(rc, "pika_blstm_layer(S=%d,B=%d,D=%d,H=%d)" % (S, B, D, H))
                hs.append(h_n)
                cs.append(c_n)
                x = out
                failed = work[:4].view(torch.int32).clone() if k == 0 else failed | work[:4].view(torch.int32)
        # `failed`: the launches' error word (a workgroup gave up waiting for its peers: the grid was not resident), still
        # on the device -- _publish_status() sends the OR over all row blocks of a pass to a pinned word
        return (torch.cat(hs, 0), torch.cat(cs, 0)), x, failed

    _status = None
    _no_fused = False        # set once a pass did not complete: nn.LSTM from then on (status_ok)

    def _publish_status(self, failed):
        """The pass' error word travels to a pinned word behind its last launch; status_ok() -- called where the caller
        synchronises anyway -- reads it."""
        if self._status is None:
            self._status = (torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event())
        word, ev = self._status
        with torch.cuda.device(failed.device):
            word.copy_(failed, non_blocking=True)
            ev.record()

    def status_ok(self):
        """False if the last persistent-kernel pass did not complete (blocks until that pass has finished): some workgroup
        of the grid was not resident -- another process or stream held CUs -- and its peers gave up waiting.  The caller
        repeats the pass; this encoder then runs nn.LSTM (warned once) instead of spinning again."""
        if self._status is None:
            return True
        word, ev = self._status
        ev.synchronize()
        if int(word[0]):
            word.zero_()
            if not self._no_fused:
                import warnings
                warnings.warn("pika_amd LAS encoder: the persistent BLSTM launch was not resident as a whole (CUs held by another "
                              "stream or process); the pass is repeated with nn.LSTM, which this encoder uses from now on")
            self._no_fused = True
            return False
        return True

    def _forward_fused_blocks(self, input, lengths):
        """The persistent kernel needs its whole grid resident -- D x (H/16) x (rows/16) workgroups, i.e. 64 rows on the 256
        CUs of an MI355X at 512 units per direction: larger batches run as consecutive row blocks of that size (rows are
        independent), each block one launch per layer."""
        r = self.rnn
        D, H = (2 if r.bidirectional else 1), r.hidden_size
        cus = torch.cuda.get_device_properties(input.device).multi_processor_count
        rows = 16 * max(1, cus // (D * (H // 16)))
        S, B, _ = input.shape
        if B <= rows:
            res = self._forward_fused(input, lengths)
            if res is None:
                return None
            self._publish_status(res[2])
            return res[0], res[1]
        lens = torch.as_tensor(lengths).view(-1).to(torch.int64).cpu()
        s_out = int(lens.max())
        hs, cs, outs = [], [], []
        failed = None
        for b0 in range(0, B, rows):
            res = self._forward_fused(input[:, b0:b0 + rows].contiguous(), lens[b0:b0 + rows])
            if res is None:
                return None
            (h, c), out, f = res
            failed = f if failed is None else failed | f        # every block's error word counts
            if out.shape[0] < s_out:        # a block whose longest row is shorter than the batch's: zero rows behind it
                out = torch.cat([out, out.new_zeros((s_out - out.shape[0],) + tuple(out.shape[1:]))], 0)
            hs.append(h)
            cs.append(c)
            outs.append(out)
        self._publish_status(failed)
        return (torch.cat(hs, 1), torch.cat(cs, 1)), torch.cat(outs, 1)

    _warned_fallback = False

    def forward(self, input, lengths=None, hidden=None):
        if not self._no_fused and self._fused_ok(input, lengths, hidden):
            res = self._forward_fused_blocks(input, lengths)
            if res is not None:
                return res
            if not LASRNNEncoder._warned_fallback:
                LASRNNEncoder._warned_fallback = True
                import warnings
                warnings.warn("pika_amd LAS encoder: the persistent BLSTM kernel did not take this pass (grid larger than the "
                              "device); running nn.LSTM (MIOpen: ~6x slower per pass)")
        packed = input
        if lengths is not None:
            packed = pack_padded_sequence(input, lengths.view(-1).tolist(), enforce_sorted=False)
        out, hid = self.rnn(packed, hidden)
        if lengths is not None:
            out = pad_packed_sequence(out)[0]
        return hid, out


class LASEmbeddings(nn.Module):
    def __init__(self, opt, output_dim, pad_idx):
        super().__init__()
        self.padding_idx = pad_idx
        self.num_embeddings = output_dim
        self.embedding_size = opt.embd_dim
        self.embeddings = nn.Embedding(output_dim + 1, opt.embd_dim, padding_idx=pad_idx)

    def forward(self, input):
        return self.embeddings(torch.squeeze(input, 2))


class StackedLSTM(nn.Module):
    def __init__(self, num_layers, input_size, rnn_size, dropout):
        super().__init__()
        self.dropout = nn.Dropout(dropout)
        self.num_layers = num_layers
        self.layers = nn.ModuleList()
        for _ in range(num_layers):
            self.layers.append(nn.LSTMCell(input_size, rnn_size))
            input_size = rnn_size

    def forward(self, input, hidden):
        h0, c0 = hidden
        hs, cs = [], []
        for i, cell in enumerate(self.layers):
            g = ops.linear(input, cell.weight_ih, cell.bias_ih) + ops.linear(h0[i], cell.weight_hh, cell.bias_hh)
            gi, gf, gg, go = g.chunk(4, dim=1)                    # LSTMCell gate order i,f,g,o
            c = torch.sigmoid(gf) * c0[i] + torch.sigmoid(gi) * torch.tanh(gg)
            h = torch.sigmoid(go) * torch.tanh(c)
            input = h if i + 1 == self.num_layers else self.dropout(h)
            hs.append(h)
            cs.append(c)
        return input, (torch.stack(hs), torch.stack(cs))


class GlobalAttention(nn.Module):
    def __init__(self, dim, coverage=False, attn_type="dot"):
        super().__init__()
        if coverage:
            raise NotImplementedError("coverage attention is off the rescoring path")
        assert attn_type in ("dot", "general", "mlp")
        self.dim, self.attn_type = dim, attn_type
        if attn_type == "general":
            self.linear_in = nn.Linear(dim, dim, bias=False)
        elif attn_type == "mlp":
            self.linear_context = nn.Linear(dim, dim, bias=False)
            self.linear_query = nn.Linear(dim, dim, bias=True)
            self.v = nn.Linear(dim, 1, bias=False)
        self.linear_out = nn.Linear(dim * 2, dim, bias=attn_type == "mlp")
        self.sm = nn.Softmax(dim=-1)
        self.tanh = nn.Tanh()

    def project_context(self, context):
        """Query-independent half of the score, once per utterance (U_a h_j for mlp)."""
        if self.attn_type == "mlp":
            return ops.linear(context, self.linear_context.weight)
        return context

    def step(self, q, context, ctx_proj, mask=None):
        """q (N,dim) queries, context / ctx_proj (N,S,dim).  Returns (attn_h (N,dim), align (N,S))."""
        if self.attn_type == "mlp":
            wq = ops.linear(q, self.linear_query.weight, self.linear_query.bias)
            align = ops.linear(torch.tanh(wq.unsqueeze(1) + ctx_proj), self.v.weight).squeeze(2)
        else:
            qq = ops.linear(q, self.linear_in.weight) if self.attn_type == "general" else q
            align = torch.bmm(context, qq.unsqueeze(2)).squeeze(2)
        if mask is not None:
            align = align.masked_fill(~mask, float("-inf"))
        a = torch.softmax(align, dim=-1)
        c = torch.bmm(a.unsqueeze(1), context).squeeze(1)
        h = ops.linear(torch.cat([c, q], 1), self.linear_out.weight, self.linear_out.bias)
        if self.attn_type != "mlp":
            h = torch.tanh(h)
        return h, a


class InputFeedRNNDecoder(nn.Module):
    def __init__(self, rnn_type, bidirectional_encoder, num_layers, hidden_size, attn_type="general",
                 coverage_attn=False, context_gate=None, copy_attn=False, dropout=0.0, embeddings=None):
        super().__init__()
        if rnn_type != "LSTM" or coverage_attn or context_gate is not None or copy_attn:
            raise NotImplementedError("LAS rescorer: LSTM input-feed decoder without coverage/copy/context gate")
        self.decoder_type = 'rnn'
        self.bidirectional_encoder = bidirectional_encoder
        self.num_layers, self.hidden_size = num_layers, hidden_size
        self.embeddings = embeddings
        self.dropout = nn.Dropout(dropout)
        self.rnn = StackedLSTM(num_layers, embeddings.embedding_size + hidden_size, hidden_size, dropout)
        self.context_gate = None
        self._coverage = False
        self.attn = GlobalAttention(hidden_size, coverage=False, attn_type=attn_type)
        self._copy = False

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_packed", None)          # the packed copies of the weights (a scoring pass' cache) are not part of a checkpoint
        return state

    # The packed-weight cache of the fused token loop is keyed on (data_ptr, _version) of every weight -- which writes
    # through `p.data` or raw pointers (BMUF's flat vector, `p.data.copy_()`) do not move.  Whatever can change
    # the weights therefore drops the cache: entering training mode, loading a state dict, or an explicit call.
    def invalidate_packed_weights(self):
        self.__dict__.pop("_packed", None)

    def train(self, mode=True):
        if mode:
            self.invalidate_packed_weights()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_packed_weights()
        return super()._load_from_state_dict(*args, **kwargs)

    def _fix_enc_hidden(self, h):
        if self.bidirectional_encoder:
            h = torch.cat([h[0:h.size(0):2], h[1:h.size(0):2]], 2)
        return h

    def _fused_ok(self, context):
        """The per-token HIP kernels (include/pika_las.h) take the scoring pass: inference on a HIP device, mlp
        attention, widths of 16-byte granules.  PIKA_LAS_FUSED=0 keeps the op-by-op path (tests compare the two)."""
        S, _, H = context.shape
        E = self.embeddings.embedding_size
        return (context.is_cuda and not torch.is_grad_enabled() and not self.training and self.attn.attn_type == "mlp"
                and H == self.hidden_size and H % 4 == 0 and H <= 1024 and E % 4 == 0 and S <= 2048
                and context.dtype == torch.float32 and os.environ.get("PIKA_LAS_FUSED", "1") != "0")

    @staticmethod
    def step_lists(own_h, spans, L, N, B):
        """Host side of the token loop's row lists (pure numpy: a scoring pass computes it ahead, Net._score_stages (A)):
        per step t the active rows first[r] <= t < end[r] ordered by utterance, concatenated (qlist) with their counts
        (n_act) and offsets (qoff), and per step where each utterance's rows sit in the step's list (uoff (L, B + 1))."""
        import numpy as np
        by_owner = np.argsort(own_h, kind="stable").astype(np.int32)
        first = np.zeros(N, np.int64) if spans is None else np.asarray(spans[0], np.int64)
        end = np.full(N, L, np.int64) if spans is None else np.asarray(spans[1], np.int64)
        f_o, e_o = first[by_owner], end[by_owner]
        if not (N and L):
            return {"n_act": np.zeros(L, np.int32), "qoff": np.zeros(L + 1, np.int32), "qlist": np.zeros(0, np.int32),
                    "uoff": np.zeros((L, B + 1), np.int32)}
        # all steps at once (a Python loop over the steps -- a mask, a gather and a searchsorted each -- was 1 ms of a plan):
        # cells[t, j] = row by_owner[j] is active at step t; a step's list = its active rows in utterance order; an
        # utterance's offset in it = the active rows of the utterances before it (a running count up to its first row)
        t_col = np.arange(L)[:, None]
        cells = (f_o[None, :] <= t_col) & (t_col < e_o[None, :])
        n_act = cells.sum(1).astype(np.int32)
        qoff = np.concatenate([[0], np.cumsum(n_act)]).astype(np.int32)
        qlist = np.broadcast_to(by_owner, (L, N))[cells].astype(np.int32)
        run = np.concatenate([np.zeros((L, 1), np.int64), np.cumsum(cells, axis=1)], axis=1)      # (L, N + 1)
        uoff = run[:, np.searchsorted(np.asarray(own_h)[by_owner], np.arange(B + 1))].astype(np.int32)
        return {"n_act": n_act, "qoff": qoff, "qlist": qlist, "uoff": uoff}

    @staticmethod
    def loop_arrays(hl, forks):
        """The host arrays a token loop reads on the device (step_lists + the fork lists + the step word's initial value)."""
        import numpy as np
        out = {"qlist": hl["qlist"], "uoff": hl["uoff"], "n_act": hl["n_act"], "qoff": hl["qoff"][:len(hl["n_act"])].copy(),
               "step": np.array([-1, 0, 0, 0], np.int32)}
        if forks is not None and len(forks[1]):
            out.update({"f_off": np.asarray(forks[0], np.int32), "f_dst": np.asarray(forks[1], np.int32),
                        "f_src": np.asarray(forks[2], np.int32)})
        return out

    def _prepare_fused(self, tokens, context, enc_hidden, owner, lens, spans=None, forks=None, owner_host=None,
                       host_lists=None, uploaded=None):
        """The token loop of `run` for N hypotheses of B utterances (owner (N,) -> utterance, lens (B,) valid source
        positions) on un-expanded encoder outputs: per token 2 x (one GEMM over [input | h] with [W_ih | W_hh] + one
        LSTM-cell kernel), the query projection, the attention (a chunk launch + a merge launch, utterance by utterance:
        the (N,S,H) tanh tensor of global_attention.py:218-221 does not exist, an utterance's rows are read once per
        workgroup, not once per hypothesis) and the
        output projection written straight into the result and fed back (input feeding, las.py:649-668).
        The eleven launches of a token are captured ONCE into a hipGraph (drive_token_loops) and replayed per token: the step index, the
        number of active hypotheses and the query-list offset are device words a one-thread kernel advances
        (PIKA_LAS_GRAPH=0: the same launches issued from Python per token).
        spans = (first, end) host int arrays (N,): hypothesis row r takes part in steps first[r] <= t < end[r] (None:
        every row in every step).  The rows of a step form a gather list (ordered by utterance, so that the queries of an
        attention workgroup share the utterance's rows) that EVERY launch of the step runs on; rows outside it are left
        unwritten in the result.  forks = (fork_off, dst, src): rows whose span starts at t > 0 inherit the recurrent
        state of row src at the top of step t (pika_las_fork_rows)."""
        from .. import _lib
        from .. import gemm as G
        lib = _lib.lib()
        dev = context.device
        L, N = tokens.shape
        S, B, H = context.shape
        E = self.embeddings.embedding_size
        nl = self.num_layers
        att = self.attn
        with torch.cuda.device(dev):
            h0, c0 = (self._fix_enc_hidden(e) for e in enc_hidden)                 # (layers, B, H)
            ctx = context.transpose(0, 1).contiguous()                              # (B, S, H)
            # "mixed" is a training arithmetic (bf16 backward): a scoring pass under it runs its forward grade, i.e. exact
            infer = "fp16x2" if G.PRECISION == "mixed" else None
            proj = G.gemm_nt(ctx.view(B * S, H), att.linear_context.weight.detach().contiguous(),
                             precision=infer).view(B, S, H)
            own = owner.to(device=dev, dtype=torch.int32).contiguous()
            ln = lens.to(device=dev, dtype=torch.int32).contiguous()
            # weights packed once per pass into MFMA fragment order (pika_dpack_weight): 1 bf16 term per operand in the
            # bf16 arithmetic mode; otherwise two fp16 terms (terms = 4: 22 mantissa bits per operand, an fp32 product to
            # ~2^-22 with three MFMAs -- the decoder's activations are bounded, far inside fp16's range);
            # self.fused_terms = 3: three bf16 terms, exact fp32 products, six MFMAs
            from ..decoder.fused_step import DGemm, PackedWeight
            terms = 1 if G.PRECISION == "bf16" else int(getattr(self, "fused_terms", 4))
            # (packed once per set of weights: the pack is four kernels over ~90 MB, the weights of a scoring model do not change)
            srcs = [w for c in self.rnn.layers for w in (c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh)] + \
                   [att.linear_query.weight, att.linear_query.bias, att.linear_out.weight, att.linear_out.bias]
            pkey = (terms, str(dev)) + tuple((w.data_ptr(), w._version) for w in srcs)
            if getattr(self, "_packed", (None,))[0] != pkey:
                self._packed = (pkey,
                                [PackedWeight(torch.cat([c.weight_ih, c.weight_hh], 1), terms) for c in self.rnn.layers],
                                [(c.bias_ih + c.bias_hh).detach().float().contiguous() for c in self.rnn.layers],
                                PackedWeight(att.linear_query.weight, terms), att.linear_query.bias.detach().float().contiguous(),
                                PackedWeight(att.linear_out.weight, terms), att.linear_out.bias.detach().float().contiguous())
            _, Wl, bl, Wq, bq, Wo, bo = self._packed
            v = att.v.weight.detach().reshape(-1).contiguous()
            # X[0] = [emb_t | feed | h_0], X[l] = [h_{l-1} | h_l]: a layer's input rows, updated in place
            X = [torch.zeros((N, E + 2 * H), device=dev)] + [torch.zeros((N, 2 * H), device=dev) for _ in range(1, nl)]
            X[0][:, E + H:] = h0[0][owner]
            for l in range(1, nl):
                X[l][:, H:] = h0[l][owner]
            c = [c0[l][owner].contiguous() for l in range(nl)]
            CQ = torch.empty((N, 2 * H), device=dev)                                # [context vector | query h_t]
            gates = torch.empty((N, 4 * H), device=dev)
            wq = torch.empty((N, H), device=dev)
            outs = torch.empty((L, N, H), device=dev)
            # Every per-token quantity lives on the device, so ONE captured launch sequence serves all tokens
            # (include/pika_las.h): step = {t, n, qoff}; the active hypotheses [0, n) of a step ordered by utterance: an
            # attention workgroup takes the queries of one utterance (step_lists)
            import numpy as np
            hl = host_lists if host_lists is not None else self.step_lists(
                owner.cpu().numpy() if owner_host is None else np.asarray(owner_host), spans, L, N, B)
            n_act = hl["n_act"]
            up = uploaded if uploaded is not None else _h2d_many(self.loop_arrays(hl, forks), dev)
            qlist, uoff_d, n_act_d, qoff_d, step = up["qlist"], up["uoff"], up["n_act"], up["qoff"], up["step"]
            # the attention runs utterance by utterance (pika_las_mlp_attention_by_utterance)
            att_work = torch.empty(int(lib.pika_las_attention_work_floats(max(int(n_act.max()) if N > 0 else 1, 1), S, H)),
                                   device=dev)
            n_dev = step[1:2]
            crow = torch.zeros(N, dtype=torch.long, device=dev)
            iden = torch.arange(N, dtype=torch.long, device=dev)
            emb_w = self.embeddings.embeddings.weight.detach().float().contiguous()
            tok = tokens.contiguous()
            n_max = int(n_act.max())

            def dgemm(A, lda, W, bias, C, ldc, crow_=None, C2=None, ldc2=0):
                g = DGemm()
                g.A, g.lda, g.W, g.bias = A.data_ptr(), lda, W.buf.data_ptr(), bias.data_ptr()
                g.C, g.ldc = C.data_ptr(), ldc
                if C2 is not None:
                    g.C2, g.ldc2, g.node = C2.data_ptr(), ldc2, iden.data_ptr()
                g.skip_node = -1
                g.M, g.N, g.K, g.terms, g.flags = n_max, W.N, W.K, W.terms, 0
                g.m_dev = n_dev.data_ptr()
                g.rowlist, g.rowoff_dev = qlist.data_ptr(), step[2:3].data_ptr()      # the step's active rows
                if crow_ is not None:
                    g.crow = crow_.data_ptr()
                _lib.check(lib.pika_dgemm(ctypes.byref(g), torch.cuda.current_stream().cuda_stream),
                           "pika_dgemm(M=%d,N=%d,K=%d)" % (n_max, W.N, W.K))

            fork = None
            if forks is not None and len(forks[1]):
                # rows that leave a shared prefix at step t inherit [feed | h_0], h_l and c_l of the row that computed it
                f_off, f_dst, f_src = up["f_off"], up["f_dst"], up["f_src"]
                segs = [(X[0], E, 2 * H)] + [(X[l], H, H) for l in range(1, nl)] + [(c[l], 0, H) for l in range(nl)]
                ns = len(segs)
                fork = (f_off, f_dst, f_src, int(np.diff(forks[0]).max()), ns,
                        (ctypes.c_void_p * ns)(*[t_.data_ptr() for t_, _, _ in segs]),
                        (ctypes.c_longlong * ns)(*[t_.stride(0) for t_, _, _ in segs]),
                        (ctypes.c_int * ns)(*[c0 for _, c0, _ in segs]), (ctypes.c_int * ns)(*[nc for _, _, nc in segs]))

            def token_step():
                st = torch.cuda.current_stream().cuda_stream
                _lib.check(lib.pika_las_step_advance(step.data_ptr(), n_act_d.data_ptr(), qoff_d.data_ptr(), L, st),
                           "pika_las_step_advance")
                if fork is not None:
                    _lib.check(lib.pika_las_fork_rows(step.data_ptr(), fork[0].data_ptr(), fork[1].data_ptr(),
                                                      fork[2].data_ptr(), fork[3], fork[4], fork[5], fork[6], fork[7],
                                                      fork[8], st), "pika_las_fork_rows")
                _lib.check(lib.pika_las_embed_rows(step.data_ptr(), tok.data_ptr(), emb_w.data_ptr(), X[0].data_ptr(),
                                                   X[0].stride(0), crow.data_ptr(), N, E, qlist.data_ptr(), st),
                           "pika_las_embed_rows")
                for l in range(nl):
                    dgemm(X[l], X[l].stride(0), Wl[l], bl[l], gates, 4 * H)
                    own_block = X[l][:, (E + H if l == 0 else H):]
                    nxt = X[l + 1][:, :H] if l + 1 < nl else CQ[:, H:]
                    _lib.check(lib.pika_lstm_cell(gates.data_ptr(), 4 * H, c[l].data_ptr(), c[l].data_ptr(),
                                                  own_block.data_ptr(), own_block.stride(0), nxt.data_ptr(),
                                                  nxt.stride(0), n_max, H, n_dev.data_ptr(), qlist.data_ptr(),
                                                  step[2:3].data_ptr(), st), "pika_lstm_cell")
                dgemm(CQ[:, H:], 2 * H, Wq, bq, wq, H)
                _lib.check(lib.pika_las_mlp_attention_by_utterance(
                    wq.data_ptr(), H, proj.data_ptr(), ctx.data_ptr(), own.data_ptr(), ln.data_ptr(), qlist.data_ptr(),
                    uoff_d.data_ptr(), v.data_ptr(), CQ.data_ptr(), 2 * H, att_work.data_ptr(), n_max, B, S, H,
                    n_dev.data_ptr(), step[2:3].data_ptr(), step[0:1].data_ptr(), st),
                    "pika_las_mlp_attention_by_utterance")
                # output projection of [context | h_t]: into row (t, r) of the result AND into the feed block of the
                # layer-0 input rows (input feeding, las.py:649-668)
                dgemm(CQ, 2 * H, Wo, bo, outs.view(L * N, H), H, crow_=crow, C2=X[0][:, E:], ldc2=X[0].stride(0))

        return TokenLoop(token_step, L, outs, dev, sig=(str(dev), n_max > 256, terms, H, E, nl, fork is not None))

    def _run_fused(self, tokens, context, enc_hidden, owner, lens, spans=None, forks=None, owner_host=None):
        loop = self._prepare_fused(tokens, context, enc_hidden, owner, lens, spans, forks, owner_host)
        drive_token_loops([loop])
        return loop.outs, None

    def run(self, tokens, context, enc_hidden, mask=None, owner=None, lens=None, spans=None, forks=None, feed=None,
            hidden_is_decoder_state=False, owner_host=None):
        """tokens (L,N) decoder inputs, context (S,N,H).  Returns outputs (L,N,H).
        With `owner` (N,) and `lens` (B,): context / enc_hidden are per UTTERANCE ((S,B,H), (layers,B,H)) and hypothesis
        n reads utterance owner[n], whose valid source positions are [0, lens[owner[n]])."""
        if owner is not None:
            if self._fused_ok(context):
                return self._run_fused(tokens, context, enc_hidden, owner, lens, spans, forks, owner_host)
            assert forks is None, "prefix sharing needs the fused token loop"
            S = context.shape[0]
            context = context[:, owner].contiguous()
            enc_hidden = tuple(e[:, owner].contiguous() for e in enc_hidden)
            mask = torch.arange(S, device=context.device).unsqueeze(0) < lens.to(context.device)[owner].unsqueeze(1)
            mask = None if bool(mask.all()) else mask
        # (a DecoderState's hidden is already in the decoder's layout; an encoder's final state is folded into it)
        hidden = tuple(enc_hidden) if hidden_is_decoder_state else tuple(self._fix_enc_hidden(e) for e in enc_hidden)
        ctx = context.transpose(0, 1).contiguous()
        proj = self.attn.project_context(ctx)
        emb = self.embeddings.embeddings(tokens)
        if feed is None:
            feed = ctx.new_zeros(ctx.shape[0], self.hidden_size)
        outs = []
        for t in range(tokens.shape[0]):
            rnn_out, hidden = self.rnn(torch.cat([emb[t], feed], 1), hidden)
            attn_h, _ = self.attn.step(rnn_out, ctx, proj, mask)
            feed = self.dropout(attn_h)
            outs.append(feed)
        return torch.stack(outs), hidden


class DecoderState(object):
    """What `Net.forward` hands back as its third value (the reference's RNNDecoderState, trainer/model/las.py:560-600):
    the stacked LSTM's (h, c) after the last token and the input-feed vector."""

    def __init__(self, hidden, input_feed):
        self.hidden, self.input_feed = tuple(hidden), input_feed

    def detach(self):
        self.hidden = tuple(h.detach() for h in self.hidden)
        self.input_feed = self.input_feed.detach()


class Net(nn.Module):
    """`las.Net(opt, input_dim, output_dim, pad_idx)`; `forward(src, tgt, lengths)` returns
    (decoder outputs, None, None, enc_out) as the rescoring call sites use it
    (decoder/transducer_decoder.py:227-232)."""

    def __init__(self, opt, input_dim, output_dim, pad_idx):
        super().__init__()
        self.input_dim, self.output_dim, self.hid_dim = input_dim, output_dim, opt.rnn_size
        if opt.encoder_type != "rnn" or getattr(opt, "use_downsampler", False) or \
                getattr(opt, "sampling_decoder", False) or getattr(opt, "num_heads", 1) > 1 or \
                not getattr(opt, "input_feed", 1):
            raise NotImplementedError("LAS rescorer configuration outside the hot path (SURVEY 8a row 16)")
        self.encoder = LASRNNEncoder(opt.rnn_type, opt.brnn, opt.enc_layers, opt.rnn_size, opt.dropout, input_dim)
        self.enc_proj = nn.Linear(opt.rnn_size, output_dim)
        self.downsampler = None
        self.tgt_embeddings = LASEmbeddings(opt, output_dim, pad_idx)
        self.dec_proj = nn.Linear(opt.rnn_size, output_dim)
        self.decoder = InputFeedRNNDecoder(opt.rnn_type, opt.brnn, opt.dec_layers, opt.rnn_size,
                                           opt.global_attention, opt.coverage_attn, opt.context_gate,
                                           opt.copy_attn, opt.dropout, self.tgt_embeddings)

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_enc_cache"] = None          # the memo of the last scoring call is not part of a checkpoint
        return state

    def _pretrain_inputfeed_decoder(self, tgt):
        """Decoder pre-training as a language model (trainer/model/las.py:92-116, `--pretrain_decoder`): the stacked LSTM
        of the input-feed decoder on [emb_t | previous output], zero initial state, no encoder and no attention."""
        emb = self.decoder.embeddings.embeddings(tgt.squeeze(2) if tgt.dim() == 3 else tgt)          # (L, B, E)
        B = emb.shape[1]
        zeros = emb.new_zeros(self.decoder.num_layers, B, self.hid_dim)
        hidden = (zeros, zeros.clone())
        output = emb.new_zeros(B, self.hid_dim)
        outs = []
        for t in range(emb.shape[0]):
            output, hidden = self.decoder.rnn(torch.cat([emb[t], output], 1), hidden)
            outs.append(output)
        return torch.stack(outs)

    def forward(self, src, tgt, lengths, dec_state=None, enable_dec=True, enable_enc=True):
        """trainer/model/las.py:51-90.  enable_enc False: decoder pre-training (outputs, None, None, None); enable_dec False:
        encoder-only training, e.g. on the CTC branch of the script's loss (None, None, None, enc_out); dec_state: the
        `DecoderState` a previous call returned, to continue from instead of the encoder's final state
        (train_las_bmuf_otfaug.py:227-239 always hands None)."""
        tgt = tgt[:-1]                                               # las.py:66 (exclude EOS)
        if not enable_enc:
            return self._pretrain_inputfeed_decoder(tgt), None, None, None
        if not enable_dec:
            return None, None, None, self.encoder(src, lengths)[1]
        if dec_state is not None:
            enc_hidden, enc_out = self.encoder(src, lengths)
            out, hidden = self.decoder.run(tgt.squeeze(2), enc_out, dec_state.hidden, feed=dec_state.input_feed,
                                           hidden_is_decoder_state=True)
            return out, None, DecoderState(hidden, out[-1]), enc_out
        if not torch.is_grad_enabled():
            # the rescoring loop of decode_transducer.py:136-156 calls this for every n-best entry and direction of the
            # SAME utterance: keep the last encoder pass.  The entry HOLDS `src` (its storage cannot be freed and handed to
            # another utterance of the same shape while the entry lives) and is matched by identity of that storage, its
            # version counter and the version counters of the encoder's weights (load_state_dict / optimizer steps)
            try:
                key = (src.data_ptr(), tuple(src.shape), tuple(src.stride()), src._version, src.device,
                       tuple(int(v) for v in torch.as_tensor(lengths).view(-1)),
                       tuple(p._version for p in self.encoder.parameters()))
            except RuntimeError:        # inference-mode tensors carry no version counter: no memo
                key = None
            hit = getattr(self, "_enc_cache", None)
            if (key is not None and hit is not None and hit[0] == key and not self.training
                    and hit[2].untyped_storage().data_ptr() == src.untyped_storage().data_ptr()):
                enc_hidden, enc_out = hit[1]
            else:
                enc_hidden, enc_out = self.encoder(src, lengths)
                self._enc_cache = None if key is None else (key, (enc_hidden, enc_out), src)
        else:
            self._enc_cache = None
            enc_hidden, enc_out = self.encoder(src, lengths)
        if not torch.is_grad_enabled() and enc_out.is_cuda:
            # scoring (decoder/transducer_decoder.py:219-253 calls this once per hypothesis): the per-token kernel chain,
            # every column its own utterance
            n = enc_out.shape[1]
            ln = torch.as_tensor(lengths).to(device=enc_out.device, dtype=torch.long).view(-1)
            out, _ = self.decoder.run(tgt.squeeze(2), enc_out, enc_hidden,
                                      owner=torch.arange(n, device=enc_out.device), lens=ln)
            return out, None, None, enc_out
        out, hidden = self.decoder.run(tgt.squeeze(2), enc_out, enc_hidden)
        return out, None, DecoderState(hidden, out[-1]), enc_out

    def _score_flat(self, enc_out, enc_hidden, owner, lens, flat, sos, eos, scale, _tick=None, owner_host=None):
        """One pass of _score_stages on its own (the encoder has run)."""
        return _run_stages([self._score_stages(lambda: (enc_hidden, enc_out), enc_out.device, enc_out.is_cuda, owner, lens,
                                               flat, sos, eos, scale, _tick, owner_host)])[0]

    def _score_stages(self, encode, dev, fused_hint, owner, lens, flat, sos, eos, scale, _tick=None, owner_host=None,
                      encode_first=False, tick_factory=None):
        """A generator in four stages (_run_stages): (A) the host plan -- pure host work; (B) encode() -> (enc_hidden,
        enc_out), the uploads and the prepared token loop, yielded for the caller to drive; (C) the vocabulary projection of
        the distinct rows, queued; (D) the wait and the lists.  encode_first: encode() is queued in front of (A).
        log P(token_t | prefix) over `hyp + [eos]` for every hypothesis of `flat` (hypothesis i reads utterance
        owner[i] of enc_out (S,B,H), valid positions lens[owner[i]]).

        Decoder step t of a hypothesis is a function of its first t tokens only, and the n-best entries of an utterance
        are near-duplicates: the entries that share a prefix share the decoder rows of that prefix (a trie per utterance;
        the first entry that has a prefix computes it).  A hypothesis gets a row of its own at the step it leaves every
        earlier entry's prefix (`act`), where it inherits the recurrent state of the row that computed the prefix so far
        (pika_las_fork_rows inside the captured token loop), and keeps it until its last token: step t runs on the rows
        with act <= t < steps -- one per DISTINCT prefix of length t (every launch of the token loop takes a gather list of
        rows).  The values are those of scoring every entry from scratch, as the reference does
        (decoder/transducer_decoder.py:219-253).  Without sharing (PIKA_LAS_SHARE_PREFIXES=0, or the op-by-op decoder
        path: training mode, CPU) step t runs on the hypotheses that still have a token.  Only the distinct (step, row)
        pairs are projected onto the vocabulary."""
        import numpy as np
        n = len(flat)
        pad = self.tgt_embeddings.padding_idx
        make_tick = tick_factory            # (a phase clock made when the pass' own device work starts: _phase_timer)
        _tick = _tick or (lambda name: None)
        if make_tick is not None and encode_first:
            _tick = make_tick()
        if owner_host is None:                              # (the batch entry hands over the host copy it built the tensor from)
            owner_host = owner.cpu().numpy() if owner is not None else np.zeros(n, np.int64)
        n_utt = int(lens.numel()) if (owner is not None and torch.is_tensor(lens)) else None

        def plan_slices(share):
            plan = yield from scoring_plan_slices(flat, owner_host, sos, eos, pad, share)
            yield
            # the (step, hypothesis) pairs that exist, the (step, row) each one reads, the token each one predicts; the
            # distinct (step, row) pairs are what the tail projects onto the vocabulary
            # (np.unique over the ~48 k pairs without its sort: the distinct (step, row) pairs are the cells first[r] <= t <
            #  end[r] of the (step, row) table -- every cell is read by the hypothesis that owns the row -- and a pair finds its
            #  cell's rank by a running count)
            t_col = np.arange(plan["L"])[:, None]
            cells = (t_col >= plan["first"][None, :]) & (t_col < plan["end"][None, :])
            plan["key"] = np.flatnonzero(cells.ravel())
            rank = np.cumsum(cells.ravel()) - 1
            plan["inv"] = rank[plan["pair_step"] * n + plan["pair_row"]]
            yield
            if n_utt is not None:                           # the token loop's row lists (the fused loop takes them)
                plan["lists"] = InputFeedRNNDecoder.step_lists(owner_host[plan["perm"]], (plan["first"], plan["end"]),
                                                               plan["L"], n, n_utt)
            return plan

        def make_plan(share):
            it = plan_slices(share)
            while True:
                try:
                    next(it)
                except StopIteration as done:
                    return done.value
        # (whether the fused token loop takes the pass is known for sure once the encoder has run: fused_hint says what to plan for)
        share = os.environ.get("PIKA_LAS_SHARE_PREFIXES", "1") != "0" and owner is not None and bool(fused_hint)
        if encode_first:                                    # (the first pass of a call: its plan runs under its encoder's kernels)
            enc_hidden, enc_out = encode()
            plan = make_plan(share)
            yield None                                      # ---- (A) done
        else:
            # a later pass: the plan in slices, handed to the caller, who runs one between two launches of the previous
            # pass' token loop (that loop is bound by its kernels: the launching thread has ~250 us to spare per token) and
            # the rest when the loop is queued
            it, box = plan_slices(share), {}

            def advance(finish=False, budget=120e-6):
                # (between two launches: slices until ~120 us are spent -- a token of the loop keeps the device busy for ~250)
                t_end = time.perf_counter() + budget
                while "plan" not in box:
                    try:
                        next(it)
                    except StopIteration as done:
                        box["plan"] = done.value
                    if not finish and time.perf_counter() >= t_end:
                        break
            yield advance                                   # ---- (A) handed over
            advance(finish=True)
            plan = box["plan"]
        def upload(plan, with_lists):
            # ONE upload: the tokens, the permutation, the tail's index arrays (once the token loop is queued a copy waits for
            # all of it) and the token loop's row lists
            key = plan["key"]
            arrays = {"tok": plan["tok"], "perm": plan["perm"], "key_t": key // n, "key_r": key % n, "inv": plan["inv"],
                      "tgt": plan["pair_target"]}
            if with_lists:
                arrays.update(InputFeedRNNDecoder.loop_arrays(plan["lists"], plan["forks"]))
            return _h2d_many(arrays, dev)
        up = None
        if not encode_first:
            if make_tick is not None:                       # (a later pass' plan ran under the previous pass' token loop: its
                _tick = make_tick()                         #  phase clock starts with its encoder)
            # the upload BEFORE the encoder is queued: the copy waits for what the stream holds -- the previous pass' tail --
            # instead of this pass' encoder, and the loop preparation that follows runs under the encoder's kernels
            if share and "lists" in plan:
                up = upload(plan, True)
            enc_hidden, enc_out = encode()
        _tick("encoder + host plan" if encode_first else "encoder")
        fused = owner is not None and self.decoder._fused_ok(enc_out)
        if share and not fused:
            plan, share, up = make_plan(False), False, None
        L, ntok, perm, tok, first, end, forks, row_steps = (plan[k] for k in
                                                            ("L", "ntok", "perm", "tok", "first", "end", "forks", "row_steps"))
        key = plan["key"]
        lists = plan.get("lists") if (fused and enc_out.shape[1] == n_utt) else None
        if up is None or lists is None:
            up = upload(plan, lists is not None)
        tok_d, key_t, key_r, inv_d, tgt_d = up["tok"], up["key_t"], up["key_r"], up["inv"], up["tgt"]
        own = owner[up["perm"].to(owner.device)]
        _tick("host prep + uploads")
        if fused:
            loop = self.decoder._prepare_fused(tok_d, enc_out, enc_hidden, own, lens, (first, end), forks, owner_host[perm],
                                               host_lists=lists, uploaded=up if lists is not None else None)
            yield loop                                      # ---- (B) done: the caller drives the loop
            out = loop.outs
        else:
            yield None
            out, _ = self.decoder.run(tok_d, enc_out, enc_hidden, owner=own, lens=lens, spans=(first, end), forks=forks)
        _tick("token loop (%d tokens, %d hypotheses, %d pairs, %d row steps)" % (L, n, int(ntok.sum()), row_steps))
        self.last_pass = {"pairs": int(ntok.sum()), "row_steps": row_steps, "shared": bool(share)}
        rows = out[key_t, key_r]                            # (R, H)
        logp = torch.log_softmax(scale * ops.linear(rows, self.dec_proj.weight, self.dec_proj.bias), dim=-1)
        picked_d = logp[inv_d, tgt_d.clamp(max=logp.shape[1] - 1)]
        if dev.type == "cuda":
            host = torch.empty(picked_d.shape, dtype=picked_d.dtype, pin_memory=True)
            host.copy_(picked_d, non_blocking=True)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
        else:
            host, done = picked_d, None
        yield None                                          # ---- (C) done: everything is queued
        if done is not None:
            done.synchronize()
        picked = host.numpy()
        _tick("vocabulary projection + log-softmax + gather (%d distinct rows)" % len(key))
        res, o = [None] * n, 0
        for i in range(n):
            res[i] = picked[o:o + ntok[i]].tolist()
            o += ntok[i]
        _tick("host lists")
        yield res

    @torch.no_grad()
    def score_nbest(self, src, hyps, sos, eos, scale=1.0):
        """src (T,1,C) one utterance; hyps: list of label lists.  Returns, per hypothesis, the list
        of log P(token_t | prefix) over `hyp + [eos]` -- what `las_rescore` returns one by one."""
        lens = torch.tensor([src.shape[0]], dtype=torch.int32)
        for _ in range(2):      # (a second pass only if the persistent encoder launch was not resident: status_ok)
            enc_hidden, enc_out = self.encoder(src, lens)
            owner = torch.zeros(len(hyps), dtype=torch.long, device=src.device)
            res = self._score_flat(enc_out, enc_hidden, owner, torch.tensor([enc_out.shape[0]], device=src.device),
                                   [list(h) for h in hyps], sos, eos, scale)
            if self.encoder.status_ok():
                break
        else:
            raise RuntimeError("pika_amd LAS rescoring: the BLSTM encoder pass did not complete twice in a row")
        return res

    def score_nbest_batch(self, src, lengths, hyps, sos, eos, scale=1.0):
        """All utterances of a decode batch at once: src (S,B,C) padded encoder outputs, lengths (B,), hyps[b] =
        list of label lists.  Returns out[b][j] = what score_nbest(src[:len_b, b:b+1], hyps[b])[j] returns.
        One batched encoder pass and ONE pass of the input-feed decoder over all sum_b len(hyps[b]) hypotheses
        (decode_transducer.py:136-156 scores them one by one, re-encoding the utterance every time).
        Two rescorers on the same batch: score_nbest_batch_many."""
        return score_nbest_batch_many([(self, src, lengths, hyps, sos, eos, scale)])[0]

    def _batch_prep(self, src, lengths, hyps):
        dev = src.device
        B = src.shape[1]
        lens = torch.as_tensor(lengths).to(torch.int64).cpu()
        order = torch.argsort(lens, descending=True, stable=True)          # packed sequences want sorted lengths
        inv = torch.empty_like(order)
        inv[order] = torch.arange(B)
        import numpy as np
        owner_h = np.repeat(inv.numpy(), [len(hyps[b]) for b in range(B)]).astype(np.int64)
        return {"src": src, "dev": dev, "lens": lens, "order": order, "owner_h": owner_h, "owner": _h2d(owner_h, dev),
                "flat": [list(h) for b in range(B) for h in hyps[b]]}

    def _batch_stages(self, prep, sos, eos, scale, first=False):
        dev, lens, order = prep["dev"], prep["lens"], prep["order"]

        def encode():
            return self.encoder(prep["src"][:, _h2d(order, dev)], lens[order].to(torch.int32))
        fused_hint = prep["src"].is_cuda and not self.training and not torch.is_grad_enabled()
        return self._score_stages(encode, dev, fused_hint, prep["owner"], _h2d(lens[order], dev), prep["flat"], sos, eos, scale,
                                  owner_host=prep["owner_h"], encode_first=first,
                                  tick_factory=lambda: _phase_timer(self, dev))
