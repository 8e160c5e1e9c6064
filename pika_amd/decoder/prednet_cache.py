"""Incremental conv-transformer prediction network for beam search.

The reference re-runs the whole prediction network on `[blank] + partial_hyp` for every row that
emitted a label (decoder/transducer_decoder.py:153-171): O(L) tokens per step, O(L^2) per
utterance.  Here every emitted label is processed ONCE: its per-layer input, key and value
vectors are stored in flat caches under a node id, and each beam slot keeps the list of node ids
of its ancestors (`anc`, re-ordered with the beam by parent index -- 8 bytes per position
instead of copying K/V).  The causal Conv1d(k=5) reads the previous four layer inputs and the
attention reads all previous keys/values through that ancestry list.  Same arithmetic as
trainer/model/rnnt_conv_transformer_lm.py:59-80 + modules/transformer.py:85-100 at one position
(causal + padding masks reduce to "positions <= p").  All shapes are fixed for a given bucket
length, so the step is hipGraph-capturable.
"""
import math

import torch

from .. import _lib
from ..model import ops


class IncrementalPredNet(object):
    def __init__(self, net, rows, max_steps, max_pos, device, blk):
        self.net, self.rows = net, rows
        self.nl = len(net.conv)
        d = net.layer_norm.normalized_shape[0]
        self.d = d
        self.heads = net.transformer[0].self_attn.head_count
        cap = (max_steps + 1) * rows + 3
        self.zero_node, self.dump_node = cap - 2, cap - 1
        cin = [c.weight.shape[1] for c in net.conv]
        self.X = [torch.zeros(cap, c, device=device) for c in cin]     # layer inputs per node
        self.Kc = [torch.zeros(cap, d, device=device) for _ in range(self.nl)]
        self.Vc = [torch.zeros(cap, d, device=device) for _ in range(self.nl)]
        self.anc = torch.full((rows, max_pos), self.dump_node, dtype=torch.long, device=device)
        self.wconv = [c.weight.permute(0, 2, 1).reshape(c.weight.shape[0], -1).contiguous()
                      for c in net.conv]                                # tap-major (N, 5*C)
        self.row_ids = torch.arange(rows, device=device)
        self.fused = True     # HIP incremental-attention kernel on the GPU (include/pika_decode.h)
        # weights are constant while decoding: the three attention projections of a layer become ONE GEMM
        self.wqkv, self.bqkv = [], []
        for layer in net.transformer:
            a = layer.self_attn
            self.wqkv.append(torch.cat([a.linear_keys.weight, a.linear_values.weight, a.linear_query.weight], 0)
                             .detach().contiguous())
            self.bqkv.append(torch.cat([a.linear_keys.bias, a.linear_values.bias, a.linear_query.bias], 0)
                             .detach().contiguous())
        self._tap_off = torch.arange(-4, 0, device=device).unsqueeze(0)          # conv taps p-4 .. p-1
        # node 0 = the shared SOS position: process it once
        sos = torch.full((rows,), blk, dtype=torch.long, device=device)
        p0 = torch.zeros(rows, dtype=torch.long, device=device)
        node0 = torch.zeros(rows, dtype=torch.long, device=device)
        self.state0 = self._process(sos, p0, node0, commit=torch.ones(rows, dtype=torch.bool, device=device), L=1)

    def _process(self, tok, p, node, commit, L):
        """Run one new position per row.  tok (rows,) token, p (rows,) its position, node (rows,)
        node id to store under; rows with commit=False compute into the dump node."""
        net = self.net
        node = torch.where(commit, node, torch.full_like(node, self.dump_node))
        anc = self.anc[:, :L]
        # ancestry including the new position (for committed rows)
        pos = torch.arange(L, device=tok.device).unsqueeze(0)
        anc_now = torch.where(pos.eq(p.unsqueeze(1)), node.unsqueeze(1), anc)
        valid = pos <= p.unsqueeze(1)                                             # keys <= p
        x = net.embeddings(tok.clamp(min=0))
        for l in range(self.nl):
            self.X[l].index_copy_(0, node, x)
            # causal conv: taps at positions p-4 .. p (zeros left of position 0), gathered in one go
            if l == 0:
                tq = p.unsqueeze(1) + self._tap_off                                 # (rows, 4)
                tap_idx = torch.where(tq >= 0, anc_now.gather(1, tq.clamp(min=0)),
                                      torch.full_like(tq, self.zero_node)).reshape(-1)
            taps = self.X[l].index_select(0, tap_idx).view(self.rows, -1)           # (rows, 4*C)
            conv = net.conv[l]
            y = ops.linear(torch.cat((taps, x), dim=1), self.wconv[l], conv.bias, relu=1)
            layer = net.transformer[l]
            att = layer.self_attn
            n = ops.layer_norm(y, layer.layer_norm)
            kvq = ops.linear(n, self.wqkv[l], self.bqkv[l])
            k, v, q_ = kvq[:, :self.d], kvq[:, self.d:2 * self.d], kvq[:, 2 * self.d:].contiguous()
            self.Kc[l].index_copy_(0, node, k)
            self.Vc[l].index_copy_(0, node, v)
            H, dh = self.heads, self.d // self.heads
            if self._fused_attention(q_, L):
                # one HIP launch reads the prefix keys / values straight through the ancestry list
                ctx = torch.empty_like(q_)
                with torch.cuda.device(q_.device):
                    _lib.check(_lib.lib().pika_incremental_attention(
                        q_.data_ptr(), self.Kc[l].data_ptr(), self.Vc[l].data_ptr(), self.anc.data_ptr(),
                        self.anc.stride(0), p.data_ptr(), node.data_ptr(), self.rows, L, self.d, H, ctx.data_ptr(),
                        torch.cuda.current_stream().cuda_stream), "pika_incremental_attention")
                o = ops.linear(ctx, att.final_linear.weight, att.final_linear.bias) + y
                ff = layer.feed_forward
                hmid = ops.linear(ops.layer_norm(o, ff.layer_norm), ff.w_1.weight, ff.w_1.bias, relu=1)
                x = ops.linear(hmid, ff.w_2.weight, ff.w_2.bias) + o
                continue
            Kp = self.Kc[l].index_select(0, anc_now.reshape(-1)).view(self.rows, L, H, dh)
            Vp = self.Vc[l].index_select(0, anc_now.reshape(-1)).view(self.rows, L, H, dh)
            qh = (q_ / math.sqrt(dh)).view(self.rows, H, 1, dh)
            sc = torch.matmul(qh, Kp.permute(0, 2, 3, 1)).float()                  # (rows,H,1,L)
            sc = sc.masked_fill(~valid.view(self.rows, 1, 1, L), -1e18)
            ctx = torch.matmul(torch.softmax(sc, dim=-1), Vp.permute(0, 2, 1, 3))  # (rows,H,1,dh)
            ctx = ctx.reshape(self.rows, self.d)
            o = ops.linear(ctx, att.final_linear.weight, att.final_linear.bias) + y
            ff = layer.feed_forward
            hmid = ops.relu(ops.linear(ops.layer_norm(o, ff.layer_norm), ff.w_1.weight, ff.w_1.bias))
            x = ops.linear(hmid, ff.w_2.weight, ff.w_2.bias) + o
        # commit the ancestry of rows that really emitted
        self.anc[:, :L].copy_(torch.where(commit.unsqueeze(1), anc_now, anc))
        out = ops.linear(ops.layer_norm(x, net.layer_norm), net.linear_out.weight, net.linear_out.bias)
        return out

    def _fused_attention(self, q, L):
        dh = self.d // self.heads
        g = dh // 4
        return (q.is_cuda and q.dtype == torch.float32 and self.d % 4 == 0 and dh % 4 == 0 and g >= 1
                and g <= 64 and (g & (g - 1)) == 0 and self.d <= 2048 and self.heads * L * 4 <= 64 * 1024
                and self.fused)

    def step(self, state, tok, hyp_len, step_t, L):
        """tok (rows,) last symbols; rows with a label (> blank) append it at position hyp_len
        (the beam already counts it) and get a new state; others keep theirs."""
        commit = tok.gt(0)
        node = 1 + step_t * self.rows + self.row_ids
        new = self._process(tok, hyp_len.clamp(max=L - 1), node, commit, L)
        state.copy_(torch.where(commit.unsqueeze(1), new, state))

    def reorder(self, flat_parent):
        self.anc.copy_(self.anc.index_select(0, flat_parent))
