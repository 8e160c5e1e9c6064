"""LAS rescorer, inference scoring path only (reference trainer/model/las.py
:51-90 forward, :522-565 LASRNNEncoder, :600-683 InputFeedRNNDecoder, modules/global_attention.py
:162-248 mlp attention, modules/stacked_rnn.py:20-34).  The module tree and parameter names are the
reference's, so trained rescorers load by state_dict; the configurations the recipes use are
implemented (LSTM, bidirectional encoder, input feeding, mlp/general/dot attention, no coverage /
copy / context gate / downsampler) and anything else raises.

`score_nbest` is the MI355X-shaped entry: all hypotheses of an utterance are scored in ONE batched
decoder pass over ONE encoder pass (the reference re-runs the BLSTM encoder and a batch-1 decoder
loop per hypothesis, decoder/transducer_decoder.py:219-236)."""
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from . import ops


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

    def forward(self, input, lengths=None, hidden=None):
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

    def _fix_enc_hidden(self, h):
        if self.bidirectional_encoder:
            h = torch.cat([h[0:h.size(0):2], h[1:h.size(0):2]], 2)
        return h

    def run(self, tokens, context, enc_hidden, mask=None):
        """tokens (L,N) decoder inputs, context (S,N,H).  Returns outputs (L,N,H)."""
        hidden = tuple(self._fix_enc_hidden(e) for e in enc_hidden)
        ctx = context.transpose(0, 1).contiguous()
        proj = self.attn.project_context(ctx)
        emb = self.embeddings.embeddings(tokens)
        feed = ctx.new_zeros(ctx.shape[0], self.hidden_size)
        outs = []
        for t in range(tokens.shape[0]):
            rnn_out, hidden = self.rnn(torch.cat([emb[t], feed], 1), hidden)
            attn_h, _ = self.attn.step(rnn_out, ctx, proj, mask)
            feed = self.dropout(attn_h)
            outs.append(feed)
        return torch.stack(outs), hidden


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

    def forward(self, src, tgt, lengths, dec_state=None, enable_dec=True, enable_enc=True):
        if not enable_enc or not enable_dec or dec_state is not None:
            raise NotImplementedError("LAS training/pre-training paths are out of scope (SURVEY 2.1)")
        tgt = tgt[:-1]                                               # las.py:66 (exclude EOS)
        enc_hidden, enc_out = self.encoder(src, lengths)
        out, _ = self.decoder.run(tgt.squeeze(2), enc_out, enc_hidden)
        return out, None, None, enc_out

    @torch.no_grad()
    def score_nbest(self, src, hyps, sos, eos, scale=1.0):
        """src (T,1,C) one utterance; hyps: list of label lists.  Returns, per hypothesis, the list
        of log P(token_t | prefix) over `hyp + [eos]` -- what `las_rescore` returns one by one."""
        n = len(hyps)
        lens = torch.tensor([src.shape[0]], dtype=torch.int32)
        enc_hidden, enc_out = self.encoder(src, lens)
        L = max(len(h) for h in hyps) + 1
        pad = self.tgt_embeddings.padding_idx
        tok = torch.full((L, n), pad, dtype=torch.long, device=src.device)
        tgt = torch.full((L, n), pad, dtype=torch.long, device=src.device)
        for i, h in enumerate(hyps):
            seq = [sos] + list(h) + [eos]
            tok[:len(seq) - 1, i] = torch.tensor(seq[:-1], device=src.device)
            tgt[:len(seq) - 1, i] = torch.tensor(seq[1:], device=src.device)
        ctx = enc_out.expand(-1, n, -1).contiguous()
        hid = tuple(e.expand(-1, n, -1).contiguous() for e in enc_hidden)
        out, _ = self.decoder.run(tok, ctx, hid)
        logp = torch.log_softmax(scale * ops.linear(out, self.dec_proj.weight, self.dec_proj.bias), dim=-1)
        picked = logp.gather(2, tgt.clamp(max=logp.shape[2] - 1).unsqueeze(2)).squeeze(2).cpu()
        return [picked[:len(h) + 1, i].tolist() for i, h in enumerate(hyps)]

    @torch.no_grad()
    def score_nbest_batch(self, src, lengths, hyps, sos, eos, scale=1.0):
        """All utterances of a decode batch at once: src (S,B,C) padded encoder outputs, lengths (B,), hyps[b] =
        list of label lists.  Returns out[b][j] = what score_nbest(src[:len_b, b:b+1], hyps[b])[j] returns.
        One batched encoder pass and ONE pass of the input-feed decoder over all sum_b len(hyps[b]) hypotheses
        (decode_transducer.py:136-156 scores them one by one, re-encoding the utterance every time)."""
        dev = src.device
        B = src.shape[1]
        lens = torch.as_tensor(lengths).to(torch.int64).cpu()
        order = torch.argsort(lens, descending=True, stable=True)          # packed sequences want sorted lengths
        inv = torch.empty_like(order)
        inv[order] = torch.arange(B)
        enc_hidden, enc_out = self.encoder(src[:, order.to(dev)], lens[order].to(torch.int32))
        S = enc_out.shape[0]
        owner = torch.tensor([inv[b].item() for b in range(B) for _ in hyps[b]], dtype=torch.long, device=dev)
        flat = [h for b in range(B) for h in hyps[b]]
        n = len(flat)
        L = max(len(h) for h in flat) + 1
        pad = self.tgt_embeddings.padding_idx
        tok = torch.full((L, n), pad, dtype=torch.long)
        tgt = torch.full((L, n), pad, dtype=torch.long)
        for i, h in enumerate(flat):
            seq = [sos] + list(h) + [eos]
            tok[:len(seq) - 1, i] = torch.tensor(seq[:-1])
            tgt[:len(seq) - 1, i] = torch.tensor(seq[1:])
        tok, tgt = tok.to(dev), tgt.to(dev)
        ctx = enc_out[:, owner].contiguous()
        hid = tuple(e[:, owner].contiguous() for e in enc_hidden)
        mask = torch.arange(S, device=dev).unsqueeze(0) < lens[order].to(dev)[owner].unsqueeze(1)      # (n,S)
        out, _ = self.decoder.run(tok, ctx, hid, mask=None if bool(mask.all()) else mask)
        logp = torch.log_softmax(scale * ops.linear(out, self.dec_proj.weight, self.dec_proj.bias), dim=-1)
        picked = logp.gather(2, tgt.clamp(max=logp.shape[2] - 1).unsqueeze(2)).squeeze(2).cpu()
        res, i = [], 0
        for b in range(B):
            row = []
            for h in hyps[b]:
                row.append(picked[:len(h) + 1, i].tolist())
                i += 1
            res.append(row)
        return res

