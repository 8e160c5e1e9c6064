"""nn.LSTM as the reference's prediction network runs it in training (trainer/model/transducer.py:55-61,93-96: unidirectional,
batch_first, zero initial state, dropout between the layers), with the recurrence of every layer as ONE persistent HIP
launch per direction of time (include/pika_lstm.h, csrc/lstm_train.hip) instead of the library's two launches per step:

    gx = x W_ih^T + (b_ih + b_hh)          all steps at once: ops.linear, the arithmetic of the step (pika_amd.gemm)
    h  = recurrence(gx, W_hh)              LstmRecurrenceFn: forward keeps the activated gates and the cell states,
                                           backward returns d(gx) and dW_hh = sum_t dgates_t^T h_{t-1} (one product)

Same parameters, same function as nn.LSTM; the module stays an nn.LSTM (checkpoints, the decoder's own stepping)."""
import torch

from .. import _lib
from .. import gemm as G
from . import ops

# False: the library's recurrence (MIOpen), layer by layer (pika_amd/model/transducer.py::_lstm_forward)
PERSISTENT = True
_WORK = {}
_RETIRED = []      # outgrown scratch buffers (kept: _work)
_CUS = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def applies(rnn, x):
    """The persistent recurrence takes this call: a HIP device, fp32, H a multiple of 256 up to 1024, all workgroups
    resident at once, and an arithmetic whose products are not exact fp32 anyway (the recurrence multiplies two bf16 terms
    per operand: ~2^-17 per product)."""
    if not (PERSISTENT and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and rnn.batch_first
            and not rnn.bidirectional and getattr(rnn, "proj_size", 0) == 0 and rnn.bias):
        return False
    if rnn._forward_pre_hooks or rnn._backward_hooks or getattr(rnn, "_backward_pre_hooks", None):
        return False        # (forward hooks are honoured by `forward`; the others belong to the module call)
    H = rnn.hidden_size
    if H % 256 or H > 1024 or G.PRECISION not in ("mixed", "bf16", "bf16x3"):
        return False
    cus = _CUS.get(x.device.index)
    if cus is None:
        cus = _CUS[x.device.index] = torch.cuda.get_device_properties(x.device).multi_processor_count
    if ((x.shape[0] + 15) // 16) * (H // 16) > cus or x.shape[1] < 1:
        return False
    # the scratch is allocated (and the backward's filled) OUTSIDE a capture: inside one it would belong to the graph's pool
    # and be filled again by every replay
    return not torch.cuda.is_current_stream_capturing() or _work_ready(x.device, x.shape[1], x.shape[0], H)


def _work(device, nbytes, backward):
    """Exchange scratch of the recurrences of this device, one per direction of time (the launches of a stream are ordered;
    the layers use it one after the other).  The backward's is filled with 0xff once, here: every completed launch leaves
    it so (include/pika_lstm.h `armed`); `status` refills it after a launch that gave up."""
    key = (device.index, backward)
    w = _WORK.get(key)
    if w is None or w.numel() < nbytes:
        if w is not None:
            # a captured training step (pika_amd/train_graph.py) may hold this buffer's address: it stays alive, and the
            # next one is sized so that this does not happen often
            _RETIRED.append(w)
            nbytes = max(nbytes, w.numel() + w.numel() // 2)
        w = _WORK[key] = torch.full((nbytes,), 255, dtype=torch.uint8, device=device)
    return w


def _work_ready(device, S, B, H):
    lib = _lib.lib()
    fw, bw = _WORK.get((device.index, False)), _WORK.get((device.index, True))
    return (fw is not None and bw is not None and fw.numel() >= lib.pika_lstm_train_fwd_work_bytes(S, B, H)
            and bw.numel() >= lib.pika_lstm_train_bwd_work_bytes(S, B, H))


def reserve(rnn, B, S, device):
    """Make the scratch of both directions large enough for a (B, S) batch NOW -- called in front of a stream capture
    (pika_amd/train_graph.py, pika_amd/mbr.py: a graph captured at a bucket's boundary runs more steps than any batch so far),
    since inside one the scratch can neither be allocated nor filled."""
    H = getattr(rnn, "hidden_size", 0)
    if not (PERSISTENT and isinstance(rnn, torch.nn.LSTM) and device.type == "cuda" and H % 256 == 0 and 0 < H <= 1024):
        return
    lib = _lib.lib()
    nf, nb = lib.pika_lstm_train_fwd_work_bytes(int(S), int(B), H), lib.pika_lstm_train_bwd_work_bytes(int(S), int(B), H)
    if nf > 0 and nb > 0:
        with torch.cuda.device(device):
            _work(device, nf, False)
            _work(device, nb, True)


class LstmRecurrenceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gx, w_hh):
        lib = _lib.lib()
        B, S, H4 = gx.shape
        H = H4 // 4
        gx = gx.contiguous()
        with torch.cuda.device(gx.device):
            packed = torch.empty(lib.pika_lstm_train_packed_bytes(H), dtype=torch.uint8, device=gx.device)
            _lib.check(lib.pika_lstm_train_pack(w_hh.detach().contiguous().data_ptr(), H, packed.data_ptr(), _stream()),
                       "pika_lstm_train_pack")
            work = _work(gx.device, lib.pika_lstm_train_fwd_work_bytes(S, B, H), False)
            _work(gx.device, lib.pika_lstm_train_bwd_work_bytes(S, B, H), True)         # (never first inside a capture)
            out = torch.empty((B, S, H), dtype=torch.float32, device=gx.device)
            gates = torch.empty((B, S, H4), dtype=torch.float32, device=gx.device)
            cells = torch.empty((B, S, H), dtype=torch.float32, device=gx.device)
            _lib.check(lib.pika_lstm_train_fwd(gx.data_ptr(), packed.data_ptr(), out.data_ptr(), gates.data_ptr(),
                                               cells.data_ptr(), work.data_ptr(), work.numel(), S, B, H, _stream()),
                       "pika_lstm_train_fwd")
        ctx.save_for_backward(out, gates, cells, packed)
        ctx.dims = (B, S, H)
        c_n = cells[:, -1].clone()
        ctx.mark_non_differentiable(c_n)
        return out, c_n

    @staticmethod
    def backward(ctx, dy, _dc=None):
        from . import hipops as HO
        out, gates, cells, packed = ctx.saved_tensors
        B, S, H = ctx.dims
        lib = _lib.lib()
        dy = dy.float().contiguous()
        with torch.cuda.device(dy.device):
            dgates = torch.empty((B, S, 4 * H), dtype=torch.float32, device=dy.device)
            work = _work(dy.device, lib.pika_lstm_train_bwd_work_bytes(S, B, H), True)
            _lib.check(lib.pika_lstm_train_bwd(dy.data_ptr(), packed.data_ptr(), gates.data_ptr(), cells.data_ptr(),
                                               dgates.data_ptr(), work.data_ptr(), work.numel(), 1, S, B, H, _stream()),
                       "pika_lstm_train_bwd")
            dw = None
            if ctx.needs_input_grad[1]:
                # dW_hh[k, j] = sum over rows and steps t >= 1 of dgates[b, t, k] h[b, t-1, j]
                hprev = torch.zeros_like(out)
                hprev[:, 1:] = out[:, :-1]
                d2, h2 = HO._bf16_operand(dgates.view(B * S, 4 * H)), HO._bf16_operand(hprev.view(B * S, H))
                dw = HO._grad_weight(d2, G.matrix(h2)[0], HO._g(h2), B * S, H, 4 * H)
        return (dgates if ctx.needs_input_grad[0] else None), dw


def status(device=None):
    """Error words of the last recurrences launched on the device (0: fine; 1: a workgroup gave up waiting for a peer).
    Synchronises the stream: tests and diagnostics.  A backward launch that gave up left its scratch half used: refilled."""
    import ctypes
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    bad = 0
    for backward in (False, True):
        w = _WORK.get((device.index, backward))
        if w is None:
            continue
        word = ctypes.c_int(0)
        with torch.cuda.device(device):
            _lib.check(_lib.lib().pika_lstm_train_status(w.data_ptr(), ctypes.byref(word), _stream()), "pika_lstm_train_status")
        if word.value and backward:
            w.fill_(255)
        bad |= int(word.value)
    return bad


def forward(rnn, x):
    """nn.LSTM(x)[0] for a call `applies` admits, as (output, None) -- or, when the module carries forward hooks,
    (output, (h_n, c_n)) after the hooks have seen the call as they would see the library's.  Under the "mixed" arithmetic of the train step the network is an island of the two-term mode,
    backward included, like the conv-transformer prediction network (ops.precision_island)."""
    h_n, c_n = [], []
    with ops.precision_island(rnn.weight_hh_l0) as isl:
        out = isl.inp(x)
        for l in range(rnn.num_layers):
            w_ih, w_hh, b_ih, b_hh = (getattr(rnn, n % l) for n in ("weight_ih_l%d", "weight_hh_l%d", "bias_ih_l%d", "bias_hh_l%d"))
            out, c_last = LstmRecurrenceFn.apply(ops.linear(out, w_ih, b_ih + b_hh), w_hh)
            h_n.append(out[:, -1])
            c_n.append(c_last)
            if l + 1 < rnn.num_layers and rnn.training and rnn.dropout > 0.0:
                out = torch.nn.functional.dropout(out, rnn.dropout, True)
        out = isl.out(out)
    if not rnn._forward_hooks:
        return out, None                # (nobody looks at the final states: pika_amd/model/transducer.py takes [0])
    res = (out, (torch.stack(h_n).detach(), torch.stack(c_n)))
    for hook in list(rnn._forward_hooks.values()):
        got = hook(rnn, (x,), res)
        if got is not None:
            res = got
    return res
