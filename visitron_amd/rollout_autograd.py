"""Autograd nodes of the rollout caller (training): the reference back-propagates the rollout loss
through `OscarEncoder` and `AttnDecoderLSTM` (tasks/viewpoint_select/agent.py:493-518: `self.loss.backward()`, then Adam
on encoder and decoder).  Each node's forward is the HIP kernel the inference path uses (plus the saves its backward
needs) and its backward is HIP kernels too:

* `dense`        nn.Linear (+ tanh): vt_skinny_linear_f32 / the NT GEMM forward; dX by the same kernels on W^T, dW / db by
                 vt_wgrad_bf16
* `softdot`      the dot / mask / softmax / weighted-sum block of SoftDotAttention (agent_models.py:336-352):
                 vt_softdot_attention_f32 / vt_softdot_attention_bwd_f32
* `lstm_cell`    nn.LSTMCell (agent_models.py:417): input projection + vt_lstm_step_train_f32; vt_lstm_step_bwd_f32, dX / dh
                 by dense products of the gate gradients, dW_ih / dW_hh / db by one grouped vt_wgrad_bf16
* `lstm_sequence` one nn.LSTM direction over the padded trunk output (agent_models.py:285-303): the input projection as one
                 GEMM, vt_lstm_sequence_train_f32; vt_lstm_sequence_bwd_f32 (T launches), then dX = dgates . W_ih (NT GEMM),
                 dW_ih / dW_hh / db by one grouped vt_wgrad_bf16 over all B*S rows

No CPU fallback: these raise like every other op when the library or a HIP tensor is missing.
"""
import torch

from . import ops
from .ops import ACT_NONE, ACT_TANH, BF16, round_up


def _pad_cols(w_bf16, mult=64):
    N, K = w_bf16.shape
    if K % mult == 0 and w_bf16.is_contiguous():
        return w_bf16
    out = torch.zeros((N, round_up(K, mult)), dtype=BF16, device=w_bf16.device)
    out[:, :K] = w_bf16
    return out


def packed_linear(weight):
    """bf16 operands of an nn.Linear for both directions: (W padded [N, Kpad], W^T padded [K, Npad])."""
    w = weight.detach().to(BF16)
    return _pad_cols(w), _pad_cols(w.t().contiguous())


def _bf16_rows(x, mult=8):
    """fp32 / bf16 [M, K] -> bf16 [M, round_up(K, mult)] zero-padded (vt_wgrad_bf16 wants row strides in multiples of 8)."""
    M, K = x.shape
    if K % mult == 0:
        return x.to(BF16).contiguous()
    out = torch.zeros((M, round_up(K, mult)), dtype=BF16, device=x.device)
    out[:, :K] = x
    return out


def _wgrad_problem(dy16, x16, N, K):
    if K % 4:
        raise NotImplementedError("weight gradients are served for input widths that are multiples of 4 (got %d)" % K)
    dw = torch.empty((N, K), dtype=torch.float32, device=dy16.device)
    db = torch.empty((N,), dtype=torch.float32, device=dy16.device)
    return dict(dy=dy16[:, :N], x=x16[:, :K], dw=dw, db=db)


class _Dense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, packs):
        from .rollout import _dense, _f32c

        w_pad, wt_pad = packs
        xf = _f32c(x)
        y = _dense((xf,), w_pad, None if bias is None else _f32c(bias), act)
        ctx.act, ctx.wt_pad, ctx.has_bias = act, wt_pad, bias is not None
        ctx.save_for_backward(xf, y if act == ACT_TANH else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .rollout import _dense

        x, y = ctx.saved_tensors
        dz = dy.float()
        if ctx.act == ACT_TANH:
            dz = dz * (1.0 - y * y)
        dz = dz.contiguous()
        M, K = x.shape
        N = dz.shape[1]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _dense((dz,), ctx.wt_pad)
            dx = dx if dx.shape[1] == K else dx[:, :K]
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            pr = _wgrad_problem(_bf16_rows(dz), _bf16_rows(x), N, K)
            ops.wgrad([pr], M)
            dw, db = pr["dw"], (pr["db"] if ctx.has_bias else None)
        return dx, dw, db, None, None


def dense(x, weight, bias, act, packs):
    """act(x @ weight.T + bias) as an autograd node; x fp32 [M, K]."""
    return _Dense.apply(x, weight, bias, act, packs)


class _SoftDot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, target, context, mask, output_prob):
        from .rollout import _ctx_f32, _f32c

        t, c = _f32c(target), _ctx_f32(context)
        weighted, attn = ops.softdot_attention(t, c, mask, True, True, output_prob)
        ctx.mask, ctx.output_prob = mask, bool(output_prob)
        ctx.save_for_backward(t, c)
        ctx.set_materialize_grads(False)
        return weighted, attn

    @staticmethod
    def backward(ctx, d_weighted, d_attn):
        t, c = ctx.saved_tensors
        if d_weighted is None and d_attn is None:
            return None, None, None, None
        d_t, d_c = ops.softdot_attention_bwd(t, c, ctx.mask, d_weighted, d_attn, ctx.output_prob, ctx.needs_input_grad[1])
        return d_t, d_c, None, None


def softdot(target, context, mask, output_prob):
    """(weighted context, probabilities | masked logits) of SoftDotAttention after linear_in, as an autograd node."""
    return _SoftDot.apply(target, context, mask, output_prob)


def _lstm_weight_grads(dg16, x16, in_dim, h16, hs, M, needs):
    """dW_ih = dgates^T x, dW_hh = dgates^T h_prev, db = column sums: one grouped launch."""
    G = 4 * hs
    p_ih = _wgrad_problem(dg16, x16, G, in_dim)
    p_hh = _wgrad_problem(dg16, h16, G, hs)
    p_hh["db"] = None
    ops.wgrad([p_ih, p_hh], M)
    return p_ih["dw"], p_hh["dw"], p_ih["db"]


class _LSTMCell(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h_prev, c_prev, w_ih, w_hh, b_ih, b_hh, packs):
        from .rollout import _dense, _f32c

        w_ih_pad, w_ih_t, w_hh16, w_hh_t = packs
        xf, hp = _f32c(x), _f32c(h_prev)
        bias = (b_ih.detach().float() + b_hh.detach().float()).contiguous()
        xproj = _dense((xf,), w_ih_pad, bias)
        h1, c1, saved = ops.lstm_step_train(xproj.contiguous(), hp, c_prev.detach(), w_hh16)
        ctx.packs = (w_ih_t, w_hh_t)
        ctx.save_for_backward(xf, *saved)
        ctx.set_materialize_grads(False)
        return h1, c1

    @staticmethod
    def backward(ctx, dh, dc):
        from .rollout import _dense

        xf, sv_g, sv_c, sv_h = ctx.saved_tensors
        w_ih_t, w_hh_t = ctx.packs
        B, hs = sv_c.shape
        if dh is None and dc is None:
            return (None,) * 8
        dg32, dg16, dc_prev = ops.lstm_step_bwd(dh, dc, (sv_g, sv_c, sv_h), w_hh_t)
        in_dim = xf.shape[1]
        dx = dh_prev = None
        if ctx.needs_input_grad[0]:
            dx = _dense((dg32,), w_ih_t)
            dx = dx if dx.shape[1] == in_dim else dx[:, :in_dim]
        if ctx.needs_input_grad[1]:
            dh_prev = _dense((dg32,), w_hh_t)
        dw_ih, dw_hh, db = _lstm_weight_grads(dg16, _bf16_rows(xf), in_dim, sv_h, hs, B, ctx.needs_input_grad)
        return dx, dh_prev, (dc_prev if ctx.needs_input_grad[2] else None), dw_ih, dw_hh, db, db, None


def lstm_cell(x, h_prev, c_prev, w_ih, w_hh, b_ih, b_hh, packs):
    """nn.LSTMCell(x, (h_prev, c_prev)) -> (h_1, c_1) as an autograd node; packs = (W_ih padded, W_ih^T padded, W_hh bf16,
    W_hh^T bf16)."""
    return _LSTMCell.apply(x, h_prev, c_prev, w_ih, w_hh, b_ih, b_hh, packs)


class _LSTMSequence(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, lengths, T, reverse, packs):
        w_ih_pad, w_ih_t, w_hh16, w_hh_t = packs
        B, S, H = x.shape
        hs = w_hh.shape[1]
        x16 = _bf16_rows(x.detach().reshape(B * S, H), 64)               # the GEMM's K granule
        bias = (b_ih.detach().float() + b_hh.detach().float()).contiguous()
        xproj = torch.empty((B * S, 4 * hs), dtype=torch.float32, device=x.device)
        ops.linear(x16, w_ih_pad, bias, out=xproj, out_f32=True)
        seq_out, h_T, c_T, saved = ops.lstm_sequence_train(xproj.view(B, S, 4 * hs), w_hh16, T, lengths, reverse)
        ctx.packs, ctx.dims, ctx.lengths = (w_ih_t, w_hh_t), (B, S, H, hs, int(T), bool(reverse)), lengths
        ctx.save_for_backward(x16, *saved)
        ctx.set_materialize_grads(False)
        return seq_out, h_T, c_T

    @staticmethod
    def backward(ctx, d_seq, dh_T, dc_T):
        x16, sv_g, sv_c, sv_h = ctx.saved_tensors
        w_ih_t, w_hh_t = ctx.packs
        B, S, H, hs, T, reverse = ctx.dims
        if d_seq is None and dh_T is None and dc_T is None:
            return (None,) * 9
        dg = ops.lstm_sequence_bwd(d_seq, dh_T, dc_T, (sv_g, sv_c, sv_h), w_hh_t, T, ctx.lengths, reverse)
        dg2 = dg.view(B * S, 4 * hs)
        dx = None
        if ctx.needs_input_grad[0]:
            Hp = w_ih_t.shape[0]
            dxb = torch.empty((B * S, round_up(Hp, 4)), dtype=torch.float32, device=dg.device)
            ops.linear(dg2, w_ih_t, None, out=dxb, out_f32=True)
            dx = dxb[:, :H].reshape(B, S, H)
        dw_ih, dw_hh, db = _lstm_weight_grads(dg2, x16, H, sv_h.view(B * S, hs), hs, B * S, ctx.needs_input_grad)
        return dx, dw_ih, dw_hh, db, db, None, None, None, None


def lstm_sequence(x, w_ih, w_hh, b_ih, b_hh, lengths, T, reverse, packs):
    """One nn.LSTM direction (zero initial state) over x fp32 [B, S, H] with the packed-sequence rule -> (padded output
    fp32 [B, T, hs], h_T, c_T) as an autograd node."""
    return _LSTMSequence.apply(x, w_ih, w_hh, b_ih, b_hh, lengths, T, reverse, packs)


def packed_lstm(w_ih, w_hh):
    """(W_ih padded [4hs, Kpad], W_ih^T padded [in, 4hs], W_hh bf16 [4hs, hs], W_hh^T bf16 [hs, 4hs])."""
    w_ih_pad, w_ih_t = packed_linear(w_ih)
    w16 = w_hh.detach().to(BF16).contiguous()
    return w_ih_pad, w_ih_t, w16, w16.t().contiguous()
