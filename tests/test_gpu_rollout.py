"""Parity of the rollout caller's modules (tasks/viewpoint_select/agent_models.py OscarEncoder,
SoftDotAttention, AttnDecoderLSTM) against the CPU oracle on identical weights and inputs, through the C-ABI
(vt_lstm_step_f32 / vt_lstm_sequence_f32 / vt_softdot_attention_f32 + the NT GEMM).  bf16 operands in the dense
products: the 5e-2 tolerance of the bf16 path."""
import pytest
import torch

from helpers import check_close, maxabs, model_pair

pytestmark = pytest.mark.gpu

TOL = 5e-2


def _pair(ocls, pcls, args, kwargs, seed, dev):
    torch.manual_seed(seed)
    ref = ocls(*args, **kwargs).eval()
    prod = pcls(*args, **kwargs).eval()
    prod.load_state_dict(ref.state_dict())
    return ref, prod.to(dev)


def test_lstm_step_and_sequence_ops(dev):
    """The recurrent kernel against torch.nn.LSTMCell / nn.LSTM arithmetic written out: one step (ragged batch, no
    lengths), and a packed sequence in both directions (states kept past a row's length, zero padding)."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(0)
    B, hs, S = 21, 128, 9
    w_hh = (torch.randn(4 * hs, hs, generator=g) * 0.08).to(torch.bfloat16)
    xp = torch.randn(B, S, 4 * hs, generator=g)
    lens = torch.tensor([9, 9, 8, 8, 7, 7, 6, 5, 5, 5, 4, 4, 3, 3, 3, 2, 2, 1, 1, 1, 1], dtype=torch.int32)

    def cell(x, h, c):
        gates = x + h.to(torch.bfloat16).float() @ w_hh.float().t()
        i, f, gg, o = gates.chunk(4, 1)
        c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        return torch.sigmoid(o) * torch.tanh(c2), c2

    h0, c0 = torch.randn(B, hs, generator=g) * 0.5, torch.randn(B, hs, generator=g) * 0.5
    wh, wc = cell(xp[:, 0], h0, c0)
    h_out = torch.empty(B, hs, device=dev)
    c_dev = c0.to(dev).clone()
    ops.lstm_step(xp.to(dev)[:, 0], h0.to(dev), h_out, c_dev, w_hh.to(dev))
    assert maxabs(h_out, wh) < 2e-3 and maxabs(c_dev, wc) < 2e-3
    for reverse in (False, True):
        h, c = torch.zeros(B, hs), torch.zeros(B, hs)
        want = torch.zeros(B, S, hs)
        for t in (range(S - 1, -1, -1) if reverse else range(S)):
            hn, cn = cell(xp[:, t], h, c)
            act = (t < lens)[:, None]
            h, c = torch.where(act, hn, h), torch.where(act, cn, c)
            want[:, t] = torch.where(act, hn, torch.zeros_like(hn))
        h2 = (torch.zeros(B, hs, device=dev), torch.empty(B, hs, device=dev))
        cd = torch.zeros(B, hs, device=dev)
        seq = torch.full((B, S, 2 * hs), 7.0, device=dev)
        ops.lstm_sequence(xp.to(dev), h2, cd, w_hh.to(dev), S, lens.to(dev), seq[:, :, hs:], reverse=reverse)
        assert maxabs(seq[:, :, hs:], want) < 5e-3 and maxabs(h2[0], h) < 5e-3 and maxabs(cd, c) < 5e-3
        assert float(seq[:, :, :hs].min()) == 7.0          # the strided view's neighbours are untouched
    with pytest.raises(RuntimeError):
        ops.lstm_step(xp.to(dev)[:, 0, : 4 * 96], h0.to(dev)[:, :96].contiguous(), torch.empty(B, 96, device=dev),
                      torch.zeros(B, 96, device=dev), w_hh.to(dev)[: 4 * 96, :96].contiguous())   # hs % 128 != 0


@pytest.mark.parametrize("D,L", [(2052, 36), (512, 80), (130, 7)])
def test_softdot_attention_module(dev, D, L):
    """SoftDotAttention.forward in its four output modes, with and without a mask (agent_models.py:328-357); D = 130
    takes the scalar (unaligned) path of the kernel."""
    from oracle.rollout import SoftDotAttention as OSoft
    from visitron_amd.rollout import SoftDotAttention

    Q, B = 128, 5
    ref, prod = _pair(OSoft, SoftDotAttention, (Q, D), {}, 1, dev)
    g = torch.Generator().manual_seed(D + L)
    h = torch.randn(B, Q, generator=g)
    ctx = torch.randn(B, L, D, generator=g) * 0.5
    mask = torch.zeros(B, L, dtype=torch.bool)
    mask[1, L // 2:] = True
    mask[3, :2] = True
    for m in (None, mask):
        for tilde in (True, False):
            for prob in (True, False):
                with torch.no_grad():
                    w0, w1 = ref(h, ctx, None if m is None else m.clone(), output_tilde=tilde, output_prob=prob)
                    g0, g1 = prod(h.to(dev), ctx.to(dev), None if m is None else m.to(dev), output_tilde=tilde,
                                  output_prob=prob)
                assert g0.shape == w0.shape and g1.shape == w1.shape
                assert maxabs(g0, w0) < TOL
                if prob:
                    assert maxabs(g1, w1) < 2e-2
                else:
                    fin = torch.isfinite(w1)
                    assert bool((torch.isfinite(g1.cpu()) == fin).all())          # -inf exactly where masked
                    # logits are O(sqrt(D)) dot products of bf16-projected targets: relative tolerance
                    assert float(((g1.cpu() - w1)[fin]).abs().max()) < 2e-2 * max(1.0, float(w1[fin].abs().max()))


def test_decoder_step_matches_oracle(dev):
    """AttnDecoderLSTM.forward (agent_models.py:384-428) at the reference's sizes (angle 4, embedding 64, hidden 512,
    features 2048 + 4, 36 views), ragged candidate count, an instruction mask; then fed back for a second step."""
    from oracle.rollout import AttnDecoderLSTM as ODec
    from visitron_amd.rollout import AttnDecoderLSTM

    ang, emb, hs, feat = 4, 64, 512, 2048 + 4
    ref, prod = _pair(ODec, AttnDecoderLSTM, (ang, emb, hs, 0.5), dict(feature_size=feat), 2, dev)
    B, L, C = 7, 45, 11
    g = torch.Generator().manual_seed(5)
    action = torch.randn(B, ang, generator=g)
    feature = torch.randn(B, 36, feat, generator=g).abs() * 0.3
    cand = torch.randn(B, C, feat, generator=g).abs() * 0.3
    h1, c0 = torch.randn(B, hs, generator=g) * 0.3, torch.randn(B, hs, generator=g) * 0.3
    ctx = torch.randn(B, L, hs, generator=g) * 0.5
    cmask = torch.zeros(B, L, dtype=torch.bool)
    cmask[2, 30:] = True
    cmask[6, 5:] = True
    want_state, got_state = (h1, c0), (h1.to(dev), c0.to(dev))
    for _ in range(2):
        with torch.no_grad():
            want = ref(action, feature, cand, None, want_state[0], want_state[1], ctx, cmask.clone())
            got = prod(action.to(dev), feature.to(dev), cand.to(dev), None, got_state[0], got_state[1], ctx.to(dev),
                       cmask.to(dev))
        assert len(got) == len(want) == 4
        for gi, wi in zip(got, want):
            assert gi.shape == wi.shape
        assert maxabs(got[0], want[0]) < TOL and maxabs(got[1], want[1]) < TOL and maxabs(got[3], want[3]) < TOL
        assert maxabs(got[2], want[2]) < 2e-2 * max(1.0, float(want[2].abs().max()))
        want_state, got_state = (want[3], want[1]), (got[3], got[1])      # agent.py:383: h1 <- h_tilde, c_t <- c_1
    assert float(c0.to(dev).sub(got_state[1]).abs().max()) > 0        # the caller's c_0 was not updated in place
    # train() with dropout 0.5: the step runs as autograd nodes (tests/test_gpu_rollout_train.py) and nn.Dropout does its
    # work -- with a graph when grad is enabled, forward only under no_grad (agent.py:476-489, test(use_dropout=True))
    prod.train()
    args = (action.to(dev), feature.to(dev), cand.to(dev), None, h1.to(dev), c0.to(dev), ctx.to(dev))
    with torch.no_grad():
        o1, o2 = prod(*args), prod(*args)
    assert not o1[2].requires_grad and float((o1[2] - o2[2]).abs().max()) > 0     # two dropout draws
    out = prod(*args)
    assert out[2].requires_grad and out[0].requires_grad


@pytest.mark.parametrize("bidir,dec_hidden", [(False, 128), (True, 128), (False, 96)])
def test_oscar_encoder_matches_oracle(dev, bidir, dec_hidden):
    """OscarEncoder.forward (agent_models.py:256-310): text-only trunk call with the inverted uint8 mask (the
    reference quirk), nn.LSTM over the packed output, padded context of length max(lengths), decoder init."""
    from oracle.modeling import BertImgModelwithLocationEmbeds as OTrunk
    from oracle.rollout import OscarEncoder as OEnc
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.rollout import OscarEncoder

    cfg = mini_config()
    rb, pb = model_pair(OTrunk, BertImgModelwithLocationEmbeds, cfg, seed=21, device=dev)
    hs = 128
    torch.manual_seed(3)
    ref = OEnc(None, rb, hs, dec_hidden, 0.5, bidirectional=bidir).eval()
    prod = OscarEncoder(None, pb, hs, dec_hidden, 0.5, bidirectional=bidir).eval()
    prod.load_state_dict(ref.state_dict())
    prod = prod.to(dev)
    B, S = 6, 40
    g = torch.Generator().manual_seed(9)
    lengths = torch.tensor([37, 33, 33, 20, 9, 1])
    ids = torch.randint(5, cfg.vocab_size, (B, S), generator=g)
    pos = torch.arange(S)[None, :]
    pad = pos >= lengths[:, None]
    ids[pad] = 0
    mask = pad.byte()                                         # agent.py:181: uint8, 1 = padding
    with torch.no_grad():
        want = ref(ids, lengths, mask)
        got = prod(ids.to(dev), lengths, mask.to(dev))
    D = 2 if bidir else 1
    assert got[0].shape == want[0].shape == (B, 37, D * hs)
    assert got[1].shape == want[1].shape == (B, dec_hidden)
    assert got[2].shape == want[2].shape == (B, dec_hidden if D * hs != dec_hidden else D * hs)
    for gi, wi in zip(got, want):
        assert maxabs(gi, wi) < TOL
    assert float(got[0][5, 1:].abs().max()) == 0.0            # padded positions are exact zeros
    with pytest.raises(RuntimeError):
        prod(ids.to(dev), torch.tensor([1, 2, 3, 4, 5, 6]), mask.to(dev))      # not sorted in decreasing order


@pytest.mark.parametrize("B,hs,S", [(21, 128, 9), (64, 512, 40), (5, 256, 33), (48, 512, 17)])
def test_persistent_lstm_recurrence_equals_step_launches(dev, B, hs, S):
    """vt_lstm_sequence_persistent_f32 (one resident launch, W_hh in registers, hidden state exchanged per step through
    write-through stores and an arrival counter) against the one-launch-per-position form: same arithmetic, so the
    sequences, the final states and the untouched neighbours must agree to fp32 rounding of a different summation order
    -- padded and compacted input projections, both directions, ragged lengths, non-zero initial state."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(B + hs + S)
    # (recurrent gain below 1: with a gain of 1.5 the dynamics are chaotic and the two kernels' last-bit differences -- the
    # compiler contracts c' = f c + i g into fmas differently -- grow to 5e-4 over 40 steps)
    w_hh = (torch.randn(4 * hs, hs, generator=g) * (0.6 / hs ** 0.5)).to(torch.bfloat16).to(dev)
    xp = torch.randn(B, S, 4 * hs, generator=g).to(dev)
    lens = torch.randint(1, S + 1, (B,), generator=g).sort(descending=True).values.to(torch.int32)
    lens[0] = S
    lens_d = lens.to(dev)
    h0, c0 = (torch.randn(B, hs, generator=g) * 0.5).to(dev), (torch.randn(B, hs, generator=g) * 0.5).to(dev)
    # compacted copy of the projections: only the rows below each length
    start = (torch.cumsum(lens, 0) - lens).to(torch.int32)
    rows = torch.cat([xp[b, : int(lens[b])] for b in range(B)], 0).contiguous()
    for reverse in (False, True):
        outs = {}
        for mode in ("steps", "persistent"):
            ops.LSTM_PERSISTENT = mode == "persistent"
            try:
                for layout in ("padded", "rows"):
                    h2 = (h0.clone(), torch.empty(B, hs, device=dev))
                    c = c0.clone()
                    seq = torch.full((B, S, 2 * hs), 7.0, device=dev)
                    if layout == "padded":
                        ops.lstm_sequence(xp, h2, c, w_hh, S, lens_d, seq[:, :, :hs], reverse=reverse)
                    else:
                        ops.lstm_sequence_rows(rows, start.to(dev), h2, c, w_hh, S, lens_d, seq[:, :, :hs], reverse=reverse)
                    torch.cuda.synchronize()
                    assert float(seq[:, :, hs:].min()) == 7.0
                    outs[(mode, layout)] = (seq[:, :, :hs].clone(), h2[0].clone(), c.clone())
            finally:
                ops.LSTM_PERSISTENT = True
        # the recurrence written out in torch (bf16-rounded hidden state into the recurrent product, as both kernels do)
        wf = w_hh.float().cpu()
        h, c = h0.cpu().clone(), c0.cpu().clone()
        want = torch.zeros(B, S, hs)
        xc = xp.cpu()
        for t in (range(S - 1, -1, -1) if reverse else range(S)):
            gates = xc[:, t] + h.to(torch.bfloat16).float() @ wf.t()
            i_, f_, g_, o_ = gates.chunk(4, 1)
            cn = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
            hn = torch.sigmoid(o_) * torch.tanh(cn)
            act = (t < lens)[:, None]
            h, c = torch.where(act, hn, h), torch.where(act, cn, c)
            want[:, t] = torch.where(act, hn, torch.zeros_like(hn))
        for layout in ("padded", "rows"):
            for k, (name, ref) in enumerate((("sequence", want), ("final h", h), ("final c", c))):
                tag = "persistent LSTM B=%d hs=%d S=%d rev=%d %s %s" % (B, hs, S, reverse, layout, name)
                # a last-bit difference in a hidden value can flip its bf16 rounding (2^-9 relative) for the next step: the
                # two kernels (same arithmetic, different fma contraction) drift apart by 1e-5 .. 1e-4 over tens of steps,
                # each staying as close to the written-out recurrence as the other
                check_close(tag + " vs steps", outs[("persistent", layout)][k], outs[("steps", layout)][k], 1e-3)
                check_close(tag + " vs torch", outs[("persistent", layout)][k], ref, 5e-3)
        # and the zero padding past each length
        seqp = outs[("persistent", "padded")][0]
        for b in range(B):
            assert float(seqp[b, int(lens[b]):].abs().max() if int(lens[b]) < S else 0.0) == 0.0


def test_decoder_step_replayed_from_a_graph_equals_the_direct_step(dev):
    """AttnDecoderLSTM.use_graph: the inference step captured as a HIP graph per input geometry and replayed -- same four
    results as the direct launches, bit for bit, over steps that feed each other (agent.py:383), a changed geometry (a new
    capture) and a weight update (captures are keyed on the parameters' versions)."""
    from visitron_amd.rollout import AttnDecoderLSTM

    ang, emb, hs, feat = 4, 64, 512, 2048 + 4
    torch.manual_seed(4)
    dec = AttnDecoderLSTM(ang, emb, hs, 0.5, feature_size=feat).eval().to(dev)
    g = torch.Generator().manual_seed(8)

    def inputs(B, L, C):
        action = torch.randn(B, ang, generator=g).to(dev)
        feature = (torch.randn(B, 36, feat, generator=g).abs() * 0.3).to(dev)
        cand = (torch.randn(B, C, feat, generator=g).abs() * 0.3).to(dev)
        h1, c0 = (torch.randn(B, hs, generator=g) * 0.3).to(dev), (torch.randn(B, hs, generator=g) * 0.3).to(dev)
        ctx = (torch.randn(B, L, hs, generator=g) * 0.5).to(dev)
        mask = torch.zeros(B, L, dtype=torch.bool)
        mask[:, L - 5:] = True
        return action, feature, cand, h1, c0, ctx, mask.to(dev)

    for B, L, C in ((7, 45, 11), (3, 20, 5), (7, 45, 11)):
        action, feature, cand, h1, c0, ctx, mask = inputs(B, L, C)
        hd, cd, hg, cg = h1, c0, h1, c0
        for _ in range(3):
            with torch.no_grad():
                dec.use_graph = False
                want = dec(action, feature, cand, None, hd, cd, ctx, mask)
                dec.use_graph = True
                got = dec(action, feature, cand, None, hg, cg, ctx, mask)
            for gi, wi in zip(got, want):
                assert gi.shape == wi.shape and torch.equal(gi, wi)
            hd, cd, hg, cg = want[3], want[1], got[3], got[1]
    assert len(dec._graphs) == 2
    with torch.no_grad():
        dec.embedding[0].weight.mul_(1.5)          # a real update bumps the parameter's version: a fresh capture
        dec.use_graph = False
        want = dec(action, feature, cand, None, h1, c0, ctx, mask)
        dec.use_graph = True
        got = dec(action, feature, cand, None, h1, c0, ctx, mask)
    assert all(torch.equal(gi, wi) for gi, wi in zip(got, want)) and len(dec._graphs) == 3


def test_persistent_lstm_timeout_restores_the_state_and_falls_back(dev):
    """The persistent recurrence's bounded waits: on a time-out some workgroups may already have written their final
    state.  The wrapper restores the caller's h / c from its own copies and lets the step launches run -- forced here by
    the test hook, the result must equal the step-launch path bit for bit."""
    from visitron_amd import ops

    B, hs, S = 16, 256, 12
    g = torch.Generator().manual_seed(3)
    xp = (torch.randn(B, S, 4 * hs, generator=g) * 0.5).to(dev)
    w_hh = (torch.randn(4 * hs, hs, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    lens = torch.tensor([12] * 6 + [9] * 5 + [3] * 5, dtype=torch.int32, device=dev)

    def run(force, persistent):
        ops.LSTM_PERSISTENT, ops._LSTM_FORCE_TIMEOUT[0] = persistent, force
        try:
            h2 = (torch.full((B, hs), 0.25, device=dev), torch.empty(B, hs, device=dev))
            c = torch.full((B, hs), -0.5, device=dev)
            seq = torch.zeros(B, S, hs, device=dev)
            ops.lstm_sequence(xp, h2, c, w_hh, S, lens, seq)
            torch.cuda.synchronize()
            return h2[0].clone(), c.clone(), seq
        finally:
            ops.LSTM_PERSISTENT, ops._LSTM_FORCE_TIMEOUT[0] = True, False

    want = run(False, False)       # the step launches
    got = run(True, True)          # the persistent launch "times out": restore, then the step launches
    for a, b in zip(got, want):
        assert torch.equal(a, b)
