"""One beam-search step as a fixed chain of HIP launches (include/pika_decode_step.h), captured in ONE hipGraph.

Reference loop body: decoder/transducer_decoder.py:123-186 -- prediction-network step for the rows whose last symbol
is a label (:139-171; conv-transformer prediction net, trainer/model/rnnt_conv_transformer_lm.py:59-80), joint
(:173-175), log-softmax (:177), `advance` per utterance (:182, decoder/beam_transducer.py:82-187), `_beam_update`
(:188-202).  Here, per step and for all B*beam rows at once (row = b*beam + k):

    prep -> [conv -> LN -> qkv -> attention -> out-proj(+res) -> LN -> w1(relu) -> w2(+res)] x layers -> LN -> linear_out
         -> prediction halves of fc1/fc_gate with the gate in the epilogue -> fc2 with log-sum-exp + top-K partials
         -> advance from the partials (+ device FST advance)

22 launches (2 layers), no torch ops, no host reads: the (B*beam, V) logits never exist, nothing is cast or
concatenated per step (weights are packed once per `decode_batch`), and because every buffer has a fixed shape the
host can replay the graph several times per read of the `stop` flag (state-mutating launches are no-ops once it is set).

Arithmetic: `terms` = 4 (default): two fp16 terms per operand, fp32-grade products; 3: three bf16 terms, exact fp32 products;
2: two bf16 terms; 1: plain bf16 operands.  Against the reference's fp32 CPU decoder (tests/test_decode_full.py): greedy
hypotheses and every top-1 identical, n-best entries separated by > 1e-3 in score at their reference rank.
"""
import ctypes

import torch

from .. import _lib

_vp, _ll, _i = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
DG_RELU, DG_GATE, DG_ROWMASK, DG_FEW_ROWS = 1, 2, 4, 8
MAX_LAYERS = 4


class DGemm(ctypes.Structure):
    _fields_ = [("A", _vp), ("lda", _ll), ("W", _vp), ("bias", _vp), ("res", _vp), ("ldr", _ll), ("C", _vp), ("ldc", _ll),
                ("C2", _vp), ("ldc2", _ll), ("node", _vp), ("skip_node", _ll), ("e_all", _vp), ("t_idx", _vp),
                ("T", _i), ("beam", _i), ("M", _i), ("N", _i), ("K", _i), ("terms", _i), ("flags", _i),
                ("m_dev", _vp), ("crow", _vp), ("rowlist", _vp), ("rowoff_dev", _vp),
                ("ln_gamma", _vp), ("ln_beta", _vp), ("ln_eps", ctypes.c_float)]


class DJoint(ctypes.Structure):
    _fields_ = [("pj", _vp * 2), ("h", _vp), ("e_all", _vp), ("rowmap32", _vp), ("T", _i), ("JH", _i)]


class DPrep(ctypes.Structure):
    _fields_ = [("prev_k", _vp), ("y", _vp), ("hyp_len", _vp), ("step_t", _vp), ("t_idx", _vp),
                ("state", _vp * 2), ("anc", _vp * 2), ("emb", _vp), ("X", _vp * MAX_LAYERS), ("A", _vp * MAX_LAYERS),
                ("C", _i * MAX_LAYERS), ("lda", _ll * MAX_LAYERS), ("node", _vp), ("pos", _vp), ("rowmap", _vp),
                ("count", _vp), ("dump_node", _ll),
                ("zero_node", _ll), ("layers", _i), ("rows", _i), ("beam", _i), ("H", _i), ("L", _i), ("blk", _i),
                ("stop", _vp), ("joint", DJoint)]


class DPrepLSTM(ctypes.Structure):
    _fields_ = [("prev_k", _vp), ("y", _vp), ("step_t", _vp), ("t_idx", _vp), ("state", _vp * 2), ("emb", _vp),
                ("A", _vp * MAX_LAYERS), ("lda", _ll * MAX_LAYERS), ("rowmap", _vp), ("count", _vp),
                ("layers", _i), ("rows", _i), ("beam", _i), ("H", _i), ("E", _i), ("blk", _i), ("stop", _vp),
                ("joint", DJoint)]


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def supported(model, beam, K):
    """Conv-transformer or LSTM prediction net on a HIP device, shapes the kernels take."""
    dec = getattr(model, "decoder", None)
    if dec is None or not beam.fused_ok(chain=True):
        return False
    splits = _lib.lib().pika_dfc2_splits(model.output_dim)
    # the advance's LDS budget, as the library states it (include/pika_decode_step.h; 0 = the shape is not taken)
    if not _lib.lib().pika_beam_advance_logits_lds(K, int(beam.hyp.shape[2]), splits) or model.hid_dim % 4:
        return False
    if model.decoder_type == "rnn":
        return (isinstance(dec, torch.nn.LSTM) and not dec.bidirectional and dec.batch_first and dec.bias
                and getattr(dec, "proj_size", 0) == 0 and dec.num_layers <= MAX_LAYERS and dec.hidden_size == model.hid_dim
                and model.hid_dim % 32 == 0)      # the joint product reads h in place: whole 32-column K blocks
    if not hasattr(dec, "conv"):
        return False
    d = dec.layer_norm.normalized_shape[0]
    heads = dec.transformer[0].self_attn.head_count
    dh = d // heads
    g = dh // 4
    return (len(dec.conv) <= MAX_LAYERS and d % 32 == 0 and d <= 1024 and 256 % (d // 4) == 0 and dh % 4 == 0
            and 1 <= g <= 64 and (g & (g - 1)) == 0)


def make(model, beam, e_all, T, num_frames, sm_scale, lm_scale, terms=3):
    cls = FusedSearchLSTM if model.decoder_type == "rnn" else FusedSearch
    return cls(model, beam, e_all, T, num_frames, sm_scale, lm_scale, terms)


class PackedWeight(object):
    def __init__(self, w, terms, interleave2=False):
        w = w.detach().float().contiguous()
        self.N, self.K, self.terms = w.shape[0], w.shape[1], terms
        lib = _lib.lib()
        self.buf = torch.empty(lib.pika_dpack_bytes(self.N, self.K, terms), dtype=torch.uint8, device=w.device)
        _lib.check(lib.pika_dpack_weight(w.data_ptr(), w.stride(0), self.N, self.K, terms, int(interleave2),
                                         self.buf.data_ptr(), _stream()), "pika_dpack_weight")
        self._keep = w      # stream-ordered use: keep the source alive until the pack kernel has run


def _ceil(a, b):
    return (a + b - 1) // b * b


class FusedSearch(object):
    """Device state + the launch chain of one step for a conv-transformer prediction net."""

    def _init_common(self, model, beam, e_all, T, num_frames, sm_scale, lm_scale, terms):
        """What every prediction net shares: beam bookkeeping buffers, the joint's prediction halves and fc2 (packed
        once), the partial buffers of the fc2 / advance pair."""
        self.model, self.beam = model, beam
        dev = e_all.device
        self.dev = dev
        B, K = beam.B, beam.K
        self.B, self.K, self.rows = B, K, B * K
        R = self.rows
        self.T, self.H, self.V = T, model.hid_dim, model.output_dim
        self.terms = int(terms)
        self.sm_scale, self.lm_scale = float(sm_scale), float(lm_scale)
        self.e_all, self.num_frames = e_all.contiguous(), num_frames
        f32 = dict(dtype=torch.float32, device=dev)
        i64 = dict(dtype=torch.long, device=dev)
        self.node = torch.zeros(R, **i64)          # per SLOT of the compact list of rows that emitted a label
        self.rowmap = torch.arange(R, device=dev)  # slot -> row
        self.t_idx = torch.full((B, K), -1, **i64)                                  # :107
        self.prev_k = torch.arange(K, device=dev).repeat(B).contiguous()            # identity parents for step 0
        self.stop = torch.zeros(1, dtype=torch.int32, device=dev)
        self.sync = torch.zeros(8, dtype=torch.int32, device=dev)
        self.max_hyp = torch.zeros(1, **i64)
        self.eos_u8 = torch.zeros(B, dtype=torch.uint8, device=dev)
        self.h = torch.empty(R, self.H, **f32)
        # the joint's prediction half of every row (raw, fc1 / fc_gate interleaved), double-buffered like the state: computed
        # when a row's state is, carried along by `prep` otherwise (pika_dstep_joint_t)
        self.pj = [torch.zeros(R, 2 * self.H, **f32) for _ in range(2)]
        self.rowmap32 = torch.arange(R, dtype=torch.int32, device=dev)
        lib = _lib.lib()
        self.splits = lib.pika_dfc2_splits(self.V)
        self.pmax = torch.empty(R * self.splits, **f32)
        self.psum = torch.empty(R * self.splits, **f32)
        # the scaled logits of a step, (rows, splits * columns per split): the advance reads each row once, thresholded
        self.ldl = self.splits * lib.pika_dfc2_cols_per_split()
        self.logits = torch.empty(R, self.ldl, **f32)
        H = self.H
        self.w2 = PackedWeight(model.fc2.weight, self.terms)
        self.b2 = model.fc2.bias.detach().float().contiguous()
        self.dump_node = 0
        self.graph = None

    def __init__(self, model, beam, e_all, T, num_frames, sm_scale, lm_scale, terms=3):
        self._init_common(model, beam, e_all, T, num_frames, sm_scale, lm_scale, terms)
        net = model.decoder
        dev = self.dev
        B, K, R = self.B, self.K, self.rows
        self.nl = len(net.conv)
        d = net.layer_norm.normalized_shape[0]
        self.d, self.heads = d, net.transformer[0].self_attn.head_count
        self.L = beam.hyp.shape[2] + 1                     # SOS + labels
        max_steps = beam.s_cap
        cap = (max_steps + 2) * R + 3
        self.zero_node, self.dump_node = cap - 2, cap - 1
        f32 = dict(dtype=torch.float32, device=dev)
        i64 = dict(dtype=torch.long, device=dev)
        self.Cin = [c.weight.shape[1] for c in net.conv]
        self.lda = [_ceil(5 * c, 32) for c in self.Cin]
        self.X = [torch.zeros(cap, c, **f32) for c in self.Cin]
        self.Kc = [torch.zeros(cap, d, **f32) for _ in range(self.nl)]
        self.Vc = [torch.zeros(cap, d, **f32) for _ in range(self.nl)]
        self.A = [torch.zeros(R, w, **f32) for w in self.lda]
        # A row's carried "state" is the last transformer layer's output at its last position (d wide, before the final
        # LayerNorm): linear_out and the prediction halves of fc1 / fc_gate are ONE product per step (folded below), and the
        # decoder's own state -- linear_out(LayerNorm(.)) -- is only formed for the rows that are left at the end
        self.state = [torch.zeros(R, d, **f32) for _ in range(2)]
        self.anc = [torch.full((R, self.L), self.dump_node, **i64) for _ in range(2)]
        self.pos = torch.zeros(R, **i64)
        # activations of one step
        self.y_conv = torch.empty(R, d, **f32)
        self.kvq = torch.empty(R, 3 * d, **f32)
        self.ctx = torch.empty(R, d, **f32)
        self.o = torch.empty(R, d, **f32)
        dff = net.transformer[0].feed_forward.w_1.weight.shape[0]
        self.hmid = torch.empty(R, dff, **f32)
        # weights, packed once
        t = self.terms
        self.layers = []
        for l in range(self.nl):
            conv, layer = net.conv[l], net.transformer[l]
            att, ff = layer.self_attn, layer.feed_forward
            wconv = conv.weight.permute(0, 2, 1).reshape(conv.weight.shape[0], -1)   # tap-major (N, 5*C)
            wqkv = torch.cat([att.linear_keys.weight, att.linear_values.weight, att.linear_query.weight], 0)
            bqkv = torch.cat([att.linear_keys.bias, att.linear_values.bias, att.linear_query.bias], 0).detach().float().contiguous()
            self.layers.append(dict(
                conv=PackedWeight(wconv, t), bconv=conv.bias.detach().float().contiguous(),
                ln1=layer.layer_norm, qkv=PackedWeight(wqkv, t), bqkv=bqkv,
                fin=PackedWeight(att.final_linear.weight, t), bfin=att.final_linear.bias.detach().float().contiguous(),
                ln2=ff.layer_norm, w1=PackedWeight(ff.w_1.weight, t), b1=ff.w_1.bias.detach().float().contiguous(),
                w2=PackedWeight(ff.w_2.weight, t), b2=ff.w_2.bias.detach().float().contiguous()))
        self.wout = PackedWeight(net.linear_out.weight, t)
        self.bout = net.linear_out.bias.detach().float().contiguous()
        # No nonlinearity sits between linear_out (rnnt_conv_transformer_lm.py:80) and the prediction halves of fc1 / fc_gate
        # (transducer.py:107-109): per step the two products are one, W' = [fc1_p; fc_gate_p] . W_out (2H x d, formed in
        # float64, rounded once), and the constant [fc1_p; fc_gate_p] . b_out joins the encoder halves (which already hold
        # the biases of fc1 / fc_gate).  One K = d product instead of a K = d and a K = H one, one launch less per step.
        wp64 = torch.cat((model.fc1.weight[:, self.H:], model.fc_gate.weight[:, self.H:]), dim=0).detach().double()
        self.wp = PackedWeight((wp64 @ net.linear_out.weight.detach().double()).float(), t, interleave2=True)
        self.e_all = (self.e_all + (wp64 @ net.linear_out.bias.detach().double()).float().unsqueeze(0)).contiguous()
        self.emb = net.embeddings.weight.detach().float().contiguous()
        self._init_sos()

    # ---- launches ------------------------------------------------------------------------------------------
    def _gemm(self, A, lda, W, bias, C, ldc, M, relu=False, res=None, ldr=0, C2=None, ldc2=0, rowmask=False, gate=False,
              m_dev=None, crow=None, ln=None, rowlist=None):
        g = DGemm()
        g.m_dev, g.crow, g.rowlist = _ptr(m_dev), _ptr(crow), _ptr(rowlist)
        if ln is not None:      # LayerNorm of the A rows inside the launch
            g.ln_gamma, g.ln_beta, g.ln_eps = ln.weight.data_ptr(), ln.bias.data_ptr(), float(ln.eps)
        g.A, g.lda, g.W, g.bias = _ptr(A), lda, W.buf.data_ptr(), _ptr(bias)
        g.res, g.ldr, g.C, g.ldc = _ptr(res), ldr, _ptr(C), ldc
        g.C2, g.ldc2 = _ptr(C2), ldc2
        g.node, g.skip_node = self.node.data_ptr(), self.dump_node
        g.e_all, g.t_idx, g.T, g.beam = self.e_all.data_ptr(), self.t_idx.data_ptr(), self.T, self.K
        g.M, g.N, g.K, g.terms = M, W.N, W.K, W.terms
        g.flags = ((DG_RELU if relu else 0) | (DG_GATE if gate else 0) | (DG_ROWMASK if rowmask else 0) |
                   (DG_FEW_ROWS if m_dev is not None else 0))      # compact rows: ~1/6 of the beam rows emit a label in a step
        _lib.check(_lib.lib().pika_dgemm(ctypes.byref(g), _stream()), "pika_dgemm(M=%d,N=%d,K=%d)" % (M, W.N, W.K))

    def _prednet(self, anc_dst, state_dst, count):
        """All layers at the new position of the `count` (device int32) rows of the compact list; `rowmap` / `node` /
        `pos` / A[l] were prepared (slot order).  Only these rows' states change (:139-171)."""
        lib = _lib.lib()
        R, d = self.rows, self.d
        for l, w in enumerate(self.layers):
            # every LayerNorm rides in the A-load of the product that consumes it (pika_dgemm ln_gamma / ln_beta)
            self._gemm(self.A[l], self.lda[l], w["conv"], w["bconv"], self.y_conv, d, R, relu=True, m_dev=count)
            self._gemm(self.y_conv, d, w["qkv"], w["bqkv"], self.kvq, 3 * d, R, m_dev=count, ln=w["ln1"])
            _lib.check(lib.pika_dstep_attention(self.kvq.data_ptr(), 3 * d, self.Kc[l].data_ptr(), self.Vc[l].data_ptr(),
                                                anc_dst.data_ptr(), self.L, self.pos.data_ptr(), self.node.data_ptr(),
                                                self.rowmap.data_ptr(), count.data_ptr(), R, self.L, d, self.heads,
                                                self.ctx.data_ptr(), _stream()), "pika_dstep_attention")
            self._gemm(self.ctx, d, w["fin"], w["bfin"], self.o, d, R, res=self.y_conv, ldr=d, m_dev=count)
            self._gemm(self.o, d, w["w1"], w["b1"], self.hmid, self.hmid.shape[1], R, relu=True, m_dev=count, ln=w["ln2"])
            if l + 1 < self.nl:
                # the next layer's input: fifth tap block of its conv matrix + its cache row
                nxt = self.A[l + 1][:, 4 * self.Cin[l + 1]:]
                self._gemm(self.hmid, self.hmid.shape[1], w["w2"], w["b2"], nxt, self.lda[l + 1], R, res=self.o, ldr=d,
                           C2=self.X[l + 1], ldc2=self.Cin[l + 1], m_dev=count)
            else:
                self._gemm(self.hmid, self.hmid.shape[1], w["w2"], w["b2"], state_dst, d, R, res=self.o, ldr=d, m_dev=count,
                           crow=self.rowmap)

    def _prep(self, parity):
        p = DPrep()
        b = self.beam
        p.prev_k, p.y, p.hyp_len, p.step_t = self.prev_k.data_ptr(), b.y.data_ptr(), b.hyp_len.data_ptr(), b.step_t.data_ptr()
        p.t_idx = self.t_idx.data_ptr()
        for i in range(2):
            p.state[i], p.anc[i] = self.state[i].data_ptr(), self.anc[i].data_ptr()
        p.emb = self.emb.data_ptr()
        for l in range(self.nl):
            p.X[l], p.A[l], p.C[l], p.lda[l] = self.X[l].data_ptr(), self.A[l].data_ptr(), self.Cin[l], self.lda[l]
        p.node, p.pos, p.rowmap = self.node.data_ptr(), self.pos.data_ptr(), self.rowmap.data_ptr()
        p.count = self.sync[5:7].data_ptr()
        p.dump_node, p.zero_node = self.dump_node, self.zero_node
        p.layers, p.rows, p.beam, p.H, p.L, p.blk = self.nl, self.rows, self.K, self.d, self.L, b.blk     # (H: state width)
        p.stop = self.stop.data_ptr()
        self._fill_joint(p.joint)
        _lib.check(_lib.lib().pika_dstep_prep(ctypes.byref(p), _stream()), "pika_dstep_prep")

    def _init_sos(self):
        """Position 0 (the shared SOS / blank token, transducer_decoder.py:121) for every row: node 0."""
        R = self.rows
        with torch.cuda.device(self.dev):
            self.node.zero_()
            self.pos.zero_()
            x0 = self.emb[self.beam.blk].unsqueeze(0)
            self.X[0][0] = x0[0]
            self.A[0].zero_()
            self.A[0][:, 4 * self.Cin[0]:5 * self.Cin[0]] = x0
            for l in range(1, self.nl):
                self.A[l].zero_()
            self.anc[0][:, 0] = 0
            self.sync[5] = R                # every row is in the compact list (identity rowmap) for this one pass
            self._prednet(self.anc[0], self.state[0], self.sync[5:6])
            self._joint_rows(self.state[0], self.d, self.pj[0], self.sync[5:6], ln=self.model.decoder.layer_norm)
            self.sync[5] = 0

    def step_launches(self, parity):
        """Enqueue one search step that reads state/anc buffer `parity` (= steps taken & 1)."""
        dst = parity ^ 1
        with torch.cuda.device(self.dev):
            self._prep(parity)
            self._prednet(self.anc[dst], self.state[dst], self.sync[5 + parity:6 + parity])
            self._joint_rows(self.state[dst], self.d, self.pj[dst], self.sync[5 + parity:6 + parity],
                             ln=self.model.decoder.layer_norm)
            self._joint_and_advance()

    def _fill_joint(self, j):
        for i in range(2):
            j.pj[i] = self.pj[i].data_ptr()
        j.h, j.e_all, j.rowmap32 = self.h.data_ptr(), self.e_all.data_ptr(), self.rowmap32.data_ptr()
        j.T, j.JH = self.T, self.H

    def _joint_rows(self, dec_hid, lda, pj_dst, count, ln=None):
        """The joint's prediction half + this step's joint hidden for the rows of the compact list (their state is new):
        dec_hid (rows, .) with pitch lda, gathered through rowmap32; the other rows got theirs in `prep`."""
        self._gemm(dec_hid, lda, self.wp, None, self.h, self.H, self.rows, gate=True, m_dev=count, rowlist=self.rowmap32,
                   C2=pj_dst, ldc2=2 * self.H, ln=ln)

    def _joint_and_advance(self):
        """fc2 with the row statistics + scaled logits -> advance (+ device FST advance)."""
        lib = _lib.lib()
        b = self.beam
        _lib.check(lib.pika_dfc2_logits(self.h.data_ptr(), self.H, self.w2.buf.data_ptr(), self.b2.data_ptr(), self.rows,
                                        self.V, self.H, self.terms, self.sm_scale, self.pmax.data_ptr(),
                                        self.psum.data_ptr(), self.logits.data_ptr(), self.ldl, _stream()), "pika_dfc2_logits")
        fst = b.fst_dev
        _lib.check(lib.pika_beam_advance_logits(
            self.pmax.data_ptr(), self.psum.data_ptr(), self.logits.data_ptr(), self.ldl, self.splits, b.scores.data_ptr(),
            b.lm_scores.data_ptr(), self.lm_scale, b.y.data_ptr(), self.t_idx.data_ptr(), self.num_frames.data_ptr(),
            b.max_len.data_ptr(), b.hyp.data_ptr(), b.hyp_len.data_ptr(), b.hyp.shape[2], b.ks_hist.data_ptr(),
            b.ys_hist.data_ptr(), b.step_t.data_ptr(), self.eos_u8.data_ptr(), b.fin_score.data_ptr(),
            b.fin_step.data_ptr(), b.fin_k.data_ptr(), b.fin_n.data_ptr(), b.fin_cap, self.prev_k.data_ptr(),
            None if fst is None else fst["y_raw"].data_ptr(), self.B, self.K, self.V, b.blk, int(b.beam_prune),
            b.n_best, self.stop.data_ptr(), self.max_hyp.data_ptr(), self.sync.data_ptr(), _stream()),
            "pika_beam_advance_logits")
        if fst is not None:
            b._fst_advance_device(self.prev_k.view(self.B, self.K), self.lm_scale, skip=self.sync[4:5])

    def launches_per_step(self):
        return 1 + 6 * self.nl + 1 + 1 + 1 + (1 if self.beam.fst_dev is not None else 0)

    def final_state(self, steps):
        """Prediction-net states / frame indices in beam order after the last step (the attributes the reference
        decoder leaves behind, transducer_decoder.py:107,121,188-202)."""
        flat = (torch.arange(self.B, device=self.dev).unsqueeze(1) * self.K + self.prev_k.view(self.B, self.K)).reshape(-1)
        src = self.state[steps & 1].index_select(0, flat)
        out = torch.empty(self.rows, self.H, dtype=torch.float32, device=self.dev)
        with torch.cuda.device(self.dev):
            self._gemm(src, self.d, self.wout, self.bout, out, self.H, self.rows, ln=self.model.decoder.layer_norm)
        return out, self.t_idx


class FusedSearchLSTM(FusedSearch):
    """The same chain for the LSTM prediction net of the shipped recipes (`dec_type=rnn`; trainer/model/transducer.py
    :55-61, stepped by decoder/transducer_decoder.py:139-148):

        prep (state follows the parent; rows that emitted a label get a slot and their [emb | h] input rows)
          -> per layer: gates = [x | h] . [W_ih | W_hh]^T + (b_ih + b_hh) on the compact rows -> cell kernel (c', h' into
             the rows' state; h' is also the first block of the next layer's input row)
          -> prediction halves of fc1 / fc_gate with the gate in the epilogue (reads h of the last layer in place)
          -> fc2 with log-sum-exp + top-K partials -> advance

    4 + 2 * layers launches per step (8 for the recipe's two layers) against 22 for the conv-transformer net."""

    def __init__(self, model, beam, e_all, T, num_frames, sm_scale, lm_scale, terms=3):
        self._init_common(model, beam, e_all, T, num_frames, sm_scale, lm_scale, terms)
        rnn = model.decoder
        dev, R, H = self.dev, self.rows, self.H
        self.nl = rnn.num_layers
        self.E = model.embed.embedding_dim
        self.SP = self.nl * 2 * H
        f32 = dict(dtype=torch.float32, device=dev)
        self.state = [torch.zeros(R, self.SP, **f32) for _ in range(2)]
        self.lda = [_ceil(self.E + H, 32)] + [_ceil(2 * H, 32)] * (self.nl - 1)
        self.A = [torch.zeros(R, w, **f32) for w in self.lda]          # pad columns stay zero
        self.gates = torch.empty(R, 4 * H, **f32)
        self.emb = model.embed.weight.detach().float().contiguous()
        wp = torch.cat((model.fc1.weight[:, H:], model.fc_gate.weight[:, H:]), dim=0)   # (2H, H) prediction halves
        self.wp = PackedWeight(wp, self.terms, interleave2=True)
        self.Wl, self.bl = [], []
        for l in range(self.nl):
            w = torch.cat([getattr(rnn, "weight_ih_l%d" % l), getattr(rnn, "weight_hh_l%d" % l)], 1)
            self.Wl.append(PackedWeight(w, self.terms))
            self.bl.append((getattr(rnn, "bias_ih_l%d" % l) + getattr(rnn, "bias_hh_l%d" % l)).detach().float().contiguous())
        self._init_sos()

    def _layers(self, state_dst, count):
        lib = _lib.lib()
        R, H = self.rows, self.H
        for l in range(self.nl):
            self._gemm(self.A[l], self.lda[l], self.Wl[l], self.bl[l], self.gates, 4 * H, R, m_dev=count)
            nxt = self.A[l + 1] if l + 1 < self.nl else None
            _lib.check(lib.pika_dstep_lstm_cell(self.gates.data_ptr(), 4 * H, state_dst.data_ptr(), self.SP, l,
                                                self.rowmap.data_ptr(), count.data_ptr(), _ptr(nxt),
                                                0 if nxt is None else self.lda[l + 1], R, H, _stream()),
                       "pika_dstep_lstm_cell")

    def _init_sos(self):
        """Every row starts from the state the LSTM reaches on the SOS / blank token from zeros
        (transducer_decoder.py:116-117)."""
        R = self.rows
        with torch.cuda.device(self.dev):
            self.A[0][:, :self.E] = self.emb[self.beam.blk].unsqueeze(0)
            self.sync[5] = R                # every row is in the compact list (identity rowmap) for this one pass
            self._layers(self.state[0], self.sync[5:6])
            self._joint_rows(self.state[0][:, (self.nl - 1) * 2 * self.H:], self.SP, self.pj[0], self.sync[5:6])
            self.sync[5] = 0

    def _prep(self, parity):
        p = DPrepLSTM()
        b = self.beam
        p.prev_k, p.y, p.step_t = self.prev_k.data_ptr(), b.y.data_ptr(), b.step_t.data_ptr()
        p.t_idx = self.t_idx.data_ptr()
        for i in range(2):
            p.state[i] = self.state[i].data_ptr()
        p.emb = self.emb.data_ptr()
        for l in range(self.nl):
            p.A[l], p.lda[l] = self.A[l].data_ptr(), self.lda[l]
        p.rowmap = self.rowmap.data_ptr()
        p.count = self.sync[5:7].data_ptr()
        p.layers, p.rows, p.beam, p.H, p.E, p.blk = self.nl, self.rows, self.K, self.H, self.E, b.blk
        p.stop = self.stop.data_ptr()
        self._fill_joint(p.joint)
        _lib.check(_lib.lib().pika_dstep_prep_lstm(ctypes.byref(p), _stream()), "pika_dstep_prep_lstm")

    def step_launches(self, parity):
        dst = parity ^ 1
        with torch.cuda.device(self.dev):
            self._prep(parity)
            self._layers(self.state[dst], self.sync[5 + parity:6 + parity])
            top = self.state[dst][:, (self.nl - 1) * 2 * self.H:]                 # h of the last layer, in place
            self._joint_rows(top, self.SP, self.pj[dst], self.sync[5 + parity:6 + parity])
            self._joint_and_advance()

    def launches_per_step(self):
        return 1 + 2 * self.nl + 1 + 1 + 1 + (1 if self.beam.fst_dev is not None else 0)

    def final_state(self, steps):
        """(h, c) as nn.LSTM keeps them, (layers, rows, H) each, in beam order after the last step, and the frame
        indices (transducer_decoder.py:107,116-117,188-202)."""
        src = self.state[steps & 1]
        flat = (torch.arange(self.B, device=self.dev).unsqueeze(1) * self.K + self.prev_k.view(self.B, self.K)).reshape(-1)
        st = src.index_select(0, flat).view(self.rows, self.nl, 2, self.H)
        return (st[:, :, 0].transpose(0, 1).contiguous(), st[:, :, 1].transpose(0, 1).contiguous()), self.t_idx
