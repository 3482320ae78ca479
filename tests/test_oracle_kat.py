"""CPU: analytic known-answer tests of the oracle and fp64 gradcheck."""
import math

import pytest
import torch

from oracle.config import TINY, make_config
from oracle.modeling import (
    BertImgModelwithLocationEmbeds, CaptionBertEncoder, CaptionBertLayer, CaptionBertSelfAttention, PreTrainOscar,
)
from oracle.optim import AdamW, WarmupConstantSchedule, WarmupLinearSchedule, grouped_parameters


def test_zero_query_gives_masked_mean_of_values():
    cfg = make_config(TINY)
    att = CaptionBertSelfAttention(cfg).double().eval()
    with torch.no_grad():
        att.query.weight.zero_(); att.query.bias.zero_()
        att.value.weight.copy_(torch.eye(cfg.hidden_size)); att.value.bias.zero_()
    x = torch.randn(2, 9, cfg.hidden_size, dtype=torch.float64)
    keep = torch.ones(2, 9, dtype=torch.float64)
    keep[0, 5:] = 0
    ext = (1.0 - keep)[:, None, None, :] * -10000.0
    ctx = att(x, ext)[0]
    want0 = x[0, :5].mean(0)
    assert torch.allclose(ctx[0], want0.expand(9, -1), atol=1e-12)
    assert torch.allclose(ctx[1], x[1].mean(0).expand(9, -1), atol=1e-12)


def test_masked_key_weight_is_exactly_zero_in_fp32():
    cfg = make_config(TINY, output_attentions=True)
    att = CaptionBertSelfAttention(cfg).eval()
    x = torch.randn(1, 6, cfg.hidden_size)
    ext = torch.zeros(1, 1, 1, 6)
    ext[..., 4] = -10000.0
    probs = att(x, ext)[1]
    assert float(probs[..., 4].abs().max()) == 0.0  # exp(-10000) underflows
    assert torch.allclose(probs.sum(-1), torch.ones_like(probs.sum(-1)), atol=1e-6)


def test_all_zero_weights_layer_outputs_layernorm_bias():
    cfg = make_config(TINY)
    layer = CaptionBertLayer(cfg).eval()
    with torch.no_grad():
        for n, p in layer.named_parameters():
            p.zero_()
        layer.output.LayerNorm.bias.copy_(torch.arange(cfg.hidden_size, dtype=torch.float32))
    y = layer(torch.zeros(1, 3, cfg.hidden_size), torch.zeros(1, 1, 1, 3))[0]
    assert torch.equal(y[0, 0], torch.arange(cfg.hidden_size, dtype=torch.float32))


def test_scale_is_applied_after_the_product_and_mask_is_additive():
    cfg = make_config(TINY, num_attention_heads=1, output_attentions=True)
    att = CaptionBertSelfAttention(cfg).double().eval()
    with torch.no_grad():
        for lin in (att.query, att.key):
            lin.weight.copy_(torch.eye(cfg.hidden_size)); lin.bias.zero_()
    x = torch.randn(1, 4, cfg.hidden_size, dtype=torch.float64)
    ext = torch.tensor([0.0, -1.5, 0.0, 2.0], dtype=torch.float64).view(1, 1, 1, 4)
    probs = att(x, ext)[1][0, 0]
    want = torch.softmax(x[0] @ x[0].t() / math.sqrt(cfg.hidden_size) + ext.view(1, 4), -1)
    assert torch.allclose(probs, want, atol=1e-12)


def test_token_loss_is_cross_entropy_of_softmax_probabilities():
    """encoder.py:323-326,380-385: CE is applied to already-softmaxed probabilities."""
    p = torch.tensor([[0.7, 0.2, 0.1], [0.1, 0.1, 0.8]])
    labels = torch.tensor([0, -1])
    got = torch.nn.CrossEntropyLoss(ignore_index=-1)(p, labels)
    want = -(0.7 - math.log(math.exp(0.7) + math.exp(0.2) + math.exp(0.1)))
    assert abs(float(got) - want) < 1e-6


def test_accuracy_formulas_on_hand_made_labels():
    """encoder.py:402-431 with the ignore-count correction."""
    cfg = make_config(TINY)
    m = PreTrainOscar(cfg).eval()
    B, T = 2, 6
    ids = torch.randint(1, cfg.vocab_size, (B, T))
    with torch.no_grad():
        seq, pooled = m.bert(ids)[:2]
        scores, tokp, act = m.heads(seq, pooled)
    pw, pt, pa = scores.argmax(2), tokp.argmax(2), act.argmax(1)
    labels = torch.full((B, T), -1)
    labels[0, 1], labels[0, 2], labels[1, 3] = pw[0, 1], (pw[0, 2] + 1) % cfg.vocab_size, pw[1, 3]
    tl = torch.full((B, T), -1)
    tl[1, 1], tl[1, 2] = pt[1, 1], (pt[1, 2] + 1) % cfg.detector_classes
    na = torch.stack([pa[0], (pa[1] + 1) % cfg.action_space])
    with torch.no_grad():
        out = m(ids, labels=labels, token_labels=tl, next_action=na)
    assert abs(float(out[4]) - 2.0 / 3.0) < 1e-6
    assert abs(float(out[5]) - 0.5) < 1e-6
    assert abs(float(out[6]) - 0.5) < 1e-6
    assert abs(float(out[0]) - float(out[1] + out[2] + out[3])) < 1e-5


def test_token_labels_none_raises_like_the_reference():
    cfg = make_config(TINY)
    m = PreTrainOscar(cfg).eval()
    ids = torch.randint(1, cfg.vocab_size, (1, 4))
    with pytest.raises((NameError, UnboundLocalError)):
        m(ids, labels=torch.full((1, 4), -1))


def test_trunk_mask_rank_and_region_concat():
    cfg = make_config(TINY)
    t = BertImgModelwithLocationEmbeds(cfg).eval()
    ids = torch.randint(1, cfg.vocab_size, (2, 5))
    with pytest.raises(NotImplementedError):
        t(ids, attention_mask=torch.ones(2, 1, 1, 5))
    feats = torch.rand(2, 3, cfg.img_feature_dim)
    loc = torch.rand(2, 3, 128)
    with torch.no_grad():
        seq, pooled = t(ids, attention_mask=torch.ones(2, 8), img_feats=feats, img_location_embeddings=loc)
        seq3 = t(ids, attention_mask=torch.ones(2, 8, 8), img_feats=feats, img_location_embeddings=loc)[0]
    assert seq.shape == (2, 8, cfg.hidden_size) and pooled.shape == (2, cfg.hidden_size)
    assert torch.allclose(seq, seq3, atol=1e-6)
    with pytest.raises(AssertionError):
        t(ids, img_feats=feats, img_location_embeddings=loc, encoder_history_states=[torch.zeros(2, 1, cfg.hidden_size)] * 2)


def test_history_state_extends_keys_and_values():
    cfg = make_config(TINY)
    enc = CaptionBertEncoder(cfg).double().eval()
    x = torch.randn(1, 4, cfg.hidden_size, dtype=torch.float64)
    hist = [torch.randn(1, 2, cfg.hidden_size, dtype=torch.float64) for _ in range(cfg.num_hidden_layers)]
    y = enc(x, torch.zeros(1, 1, 1, 6, dtype=torch.float64), head_mask=[None] * 2, encoder_history_states=hist)[0]
    assert y.shape == (1, 4, cfg.hidden_size)
    # with the history keys masked out the result equals the plain run
    ext = torch.zeros(1, 1, 1, 6, dtype=torch.float64)
    ext[..., :2] = -10000.0
    y_masked = enc(x, ext, head_mask=[None] * 2, encoder_history_states=hist)[0]
    y_plain = enc(x, torch.zeros(1, 1, 1, 4, dtype=torch.float64), head_mask=[None] * 2)[0]
    assert torch.allclose(y_masked, y_plain, atol=1e-10)


def test_resize_embeddings_keeps_rows_and_unties_decoder():
    cfg = make_config(TINY)
    m = PreTrainOscar(cfg)
    old = m.bert.embeddings.word_embeddings.weight.detach().clone()
    assert m.mlmhead.predictions.decoder.weight is m.bert.embeddings.word_embeddings.weight
    m.resize_embeddings({"word_embeddings": cfg.vocab_size + 3, "position_embeddings": 40, "token_type_embeddings": 6})
    e = m.bert.embeddings
    assert e.word_embeddings.weight.shape[0] == cfg.vocab_size + 3 and e.position_embeddings.weight.shape[0] == 40
    assert torch.equal(e.word_embeddings.weight[: cfg.vocab_size], old)
    assert m.mlmhead.predictions.decoder.weight.shape[0] == cfg.vocab_size  # reference quirk: not re-tied


def test_adamw_rule_first_two_steps_by_hand():
    p = torch.nn.Parameter(torch.tensor([1.0, -2.0]))
    opt = AdamW([{"params": [p], "weight_decay": 0.1}], lr=0.01, eps=1e-8)
    g1 = torch.tensor([0.5, -0.25])
    p.grad = g1.clone()
    opt.step()
    m = 0.1 * g1
    v = 0.001 * g1 * g1
    step = 0.01 * math.sqrt(1 - 0.999) / (1 - 0.9)
    want = torch.tensor([1.0, -2.0]) - step * m / (v.sqrt() + 1e-8)
    want = want - 0.01 * 0.1 * want  # decay AFTER the Adam move, on the moved parameter
    assert torch.allclose(p.detach(), want, atol=1e-7)
    # eps placement: with a tiny gradient eps dominates the UN-corrected sqrt(v)
    q = torch.nn.Parameter(torch.tensor([0.0]))
    o2 = AdamW([q], lr=1.0, eps=1e-3)
    q.grad = torch.tensor([1e-6])
    o2.step()
    mm, vv = 0.1 * 1e-6, 0.001 * 1e-12
    want_q = -(math.sqrt(1 - 0.999) / (1 - 0.9)) * mm / (math.sqrt(vv) + 1e-3)
    assert abs(float(q) - want_q) < 1e-9


def test_schedules_and_no_decay_split():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = AdamW([p], lr=1.0)
    s = WarmupLinearSchedule(opt, warmup_steps=4, t_total=10)
    lrs = []
    for _ in range(12):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step(); s.step()
    assert lrs[:5] == [0.0, 0.25, 0.5, 0.75, 1.0]
    assert abs(lrs[7] - 0.5) < 1e-12 and lrs[10] == 0.0 and lrs[11] == 0.0
    opt2 = AdamW([p], lr=2.0)
    s2 = WarmupConstantSchedule(opt2, warmup_steps=2)
    got = []
    for _ in range(4):
        got.append(opt2.param_groups[0]["lr"]); opt2.step(); s2.step()
    assert got == [0.0, 1.0, 2.0, 2.0]
    m = PreTrainOscar(make_config(TINY))
    groups = grouped_parameters(m, 0.05)
    named = dict(m.named_parameters())
    nodecay = {id(q) for q in groups[1]["params"]}
    for n, q in named.items():
        assert (id(q) in nodecay) == ("bias" in n or "LayerNorm.weight" in n), n


def test_gradcheck_encoder_layer_fp64():
    cfg = make_config(TINY, hidden_size=16, num_attention_heads=2, intermediate_size=24)
    layer = CaptionBertLayer(cfg).double().eval()
    x = torch.randn(1, 3, 16, dtype=torch.float64, requires_grad=True)
    ext = torch.zeros(1, 1, 1, 3, dtype=torch.float64)
    ext[..., 2] = -3.0
    assert torch.autograd.gradcheck(lambda t: layer(t, ext)[0], (x,), eps=1e-6, atol=1e-5)
