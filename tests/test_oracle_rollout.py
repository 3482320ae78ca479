"""CPU known-answer checks of the rollout oracle (oracle/rollout.py).  The reference holds no fixtures for this path:
the checks are analytic, plus the packed-sequence rule of torch's own nn.LSTM against the step-by-step recurrence the
HIP kernels implement (rows past their length keep their state, padded outputs are zero)."""
import torch

from oracle.rollout import AttnDecoderLSTM, SoftDotAttention


def test_softdot_uniform_attention_is_masked_mean():
    att = SoftDotAttention(8, 6).eval()
    with torch.no_grad():
        att.linear_in.weight.zero_()                       # every logit 0 -> uniform over the unmasked keys
    ctx = torch.arange(2 * 5 * 6, dtype=torch.float32).view(2, 5, 6)
    mask = torch.tensor([[0, 0, 0, 0, 0], [0, 1, 1, 0, 0]], dtype=torch.bool)
    with torch.no_grad():
        w, p = att(torch.ones(2, 8), ctx, mask.clone(), output_tilde=False)
        _, logit = att(torch.ones(2, 8), ctx, mask.clone(), output_tilde=False, output_prob=False)
    assert torch.allclose(w[0], ctx[0].mean(0)) and torch.allclose(w[1], ctx[1][[0, 3, 4]].mean(0))
    assert torch.allclose(p[1], torch.tensor([1 / 3, 0, 0, 1 / 3, 1 / 3]))
    assert torch.isinf(logit[1, 1]) and float(logit[0, 0]) == 0.0   # the returned logits alias the masked tensor


def test_packed_lstm_equals_stepwise_recurrence():
    torch.manual_seed(0)
    B, S, I, hs = 4, 6, 5, 8
    lstm = torch.nn.LSTM(I, hs, 1, batch_first=True, bidirectional=True).eval()
    x = torch.randn(B, S, I)
    lens = torch.tensor([6, 4, 4, 1])
    with torch.no_grad():
        packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True)
        out, (hn, cn) = lstm(packed)
        out, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True)
        for d, sfx in enumerate(["", "_reverse"]):
            wih, whh = getattr(lstm, "weight_ih_l0" + sfx), getattr(lstm, "weight_hh_l0" + sfx)
            b = getattr(lstm, "bias_ih_l0" + sfx) + getattr(lstm, "bias_hh_l0" + sfx)
            h, c = torch.zeros(B, hs), torch.zeros(B, hs)
            seq = torch.zeros(B, S, hs)
            for t in (range(S - 1, -1, -1) if d else range(S)):
                i, f, g, o = (x[:, t] @ wih.t() + b + h @ whh.t()).chunk(4, 1)
                c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                h2 = torch.sigmoid(o) * torch.tanh(c2)
                act = (t < lens)[:, None]
                h, c = torch.where(act, h2, h), torch.where(act, c2, c)
                seq[:, t] = torch.where(act, h2, torch.zeros_like(h2))
            assert torch.allclose(seq, out[:, :, d * hs:(d + 1) * hs], atol=1e-6)
            assert torch.allclose(h, hn[d], atol=1e-6) and torch.allclose(c, cn[d], atol=1e-6)


def test_decoder_step_ignores_h0_and_returns_unmasked_candidate_logits():
    torch.manual_seed(1)
    dec = AttnDecoderLSTM(4, 8, 16, 0.5, feature_size=12).eval()
    B = 3
    a, f, cf = torch.randn(B, 4), torch.randn(B, 36, 12), torch.randn(B, 5, 12)
    h1, c0, ctx = torch.randn(B, 16), torch.randn(B, 16), torch.randn(B, 7, 16)
    with torch.no_grad():
        o1 = dec(a, f, cf, torch.zeros(B, 16), h1, c0, ctx)
        o2 = dec(a, f, cf, torch.full((B, 16), 9.0), h1, c0, ctx)
        want_logit = torch.bmm(cf, dec.candidate_att_layer.linear_in(o1[3]).unsqueeze(2)).squeeze(2)
    assert all(torch.equal(x, y) for x, y in zip(o1, o2))
    assert o1[2].shape == (B, 5) and torch.allclose(o1[2], want_logit, atol=1e-6)
