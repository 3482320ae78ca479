"""End-to-end parity of the drop-in modules against the CPU oracle on identical synthetic inputs
and weights.  Tolerance from BASELINE.json north_star: 5e-2 for the bf16 path."""
import pytest
import torch

import visitron_amd
from helpers import check_close, maxabs, model_pair

pytestmark = pytest.mark.gpu

TOL_BF16 = 5e-2


def _to(batch, dev):
    return {k: v.to(dev) for k, v in batch.items()}


def test_encoder_stack_matches_oracle(dev):
    from oracle.modeling import CaptionBertEncoder as OEnc
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import CaptionBertEncoder

    cfg = mini_config(num_hidden_layers=3, output_hidden_states=True)
    ref, prod = model_pair(OEnc, CaptionBertEncoder, cfg, seed=1, device=dev)
    g = torch.Generator().manual_seed(0)
    B, S, H = 3, 45, cfg.hidden_size
    x = torch.randn(B, S, H, generator=g)
    keep = (torch.rand(B, S, generator=g) > 0.3).float()
    keep[:, 0] = 1
    ext = (1.0 - keep)[:, None, None, :] * -10000.0
    with torch.no_grad():
        want = ref(x, ext, head_mask=[None] * 3)
        got = prod(x.to(dev), ext.to(dev), head_mask=[None] * 3)
    check_close("mini encoder stack last hidden", got[0], want[0], TOL_BF16)
    assert len(got[1]) == len(want[1]) == 4
    for i, (a, b) in enumerate(zip(got[1], want[1])):
        check_close("mini encoder stack hidden[%d]" % i, a, b, TOL_BF16)


def test_sublayer_forwards_match_oracle(dev):
    """CaptionBertLayer / CaptionBertAttention / CaptionBertSelfAttention are callable on their own."""
    from oracle.modeling import CaptionBertLayer as OLayer
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import CaptionBertLayer

    cfg = mini_config()
    ref, prod = model_pair(OLayer, CaptionBertLayer, cfg, seed=2, device=dev)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 19, cfg.hidden_size, generator=g)
    ext = torch.zeros(2, 1, 1, 19)
    ext[1, 0, 0, 10:] = -10000.0
    hm = torch.tensor([1.0, 0.5]).view(1, 2, 1, 1)
    with torch.no_grad():
        assert maxabs(prod(x.to(dev), ext.to(dev))[0], ref(x, ext)[0]) < TOL_BF16
        assert maxabs(prod.attention(x.to(dev), ext.to(dev))[0], ref.attention(x, ext)[0]) < TOL_BF16
        assert maxabs(prod.attention.self(x.to(dev), ext.to(dev), hm.to(dev))[0], ref.attention.self(x, ext, hm)[0]) < TOL_BF16


@pytest.mark.parametrize("text_only", [False, True])
def test_trunk_matches_oracle_mini(dev, text_only):
    from oracle.modeling import BertImgModelwithLocationEmbeds as OTrunk
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.synth import make_batch

    cfg = mini_config()
    ref, prod = model_pair(OTrunk, BertImgModelwithLocationEmbeds, cfg, seed=4, device=dev)
    b = make_batch(cfg, 4, text_len=21, region_len=0 if text_only else 13, seed=7, with_labels=False)
    with torch.no_grad():
        want = ref(**b)
        got = prod(**_to(b, dev))
    assert got[0].shape == want[0].shape and got[1].shape == want[1].shape
    check_close("mini trunk sequence_output (text_only=%s)" % text_only, got[0], want[0], TOL_BF16)
    check_close("mini trunk pooled_output (text_only=%s)" % text_only, got[1], want[1], TOL_BF16)


def test_trunk_defaults_and_errors(dev):
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds

    cfg = mini_config()
    prod = BertImgModelwithLocationEmbeds(cfg).eval().to(dev)
    ids = torch.randint(1, cfg.vocab_size, (2, 8), device=dev)
    with torch.no_grad():
        a = prod(ids)  # mask defaults to ones, types to zeros (encoder.py:215-219)
        b = prod(ids, token_type_ids=torch.zeros_like(ids), attention_mask=torch.ones_like(ids))
    assert torch.equal(a[0], b[0])
    with pytest.raises(NotImplementedError):
        prod(ids, attention_mask=torch.ones(2, 1, 1, 8, device=dev))  # rank not in {2,3}: encoder.py:230-231
    with pytest.raises(IndexError):   # reported like a GPU nn.Embedding's device-side assert: asynchronously
        prod(torch.full((2, 8), cfg.vocab_size + 1, device=dev))
        visitron_amd.check_errors()
    prod(ids)                          # the flag was consumed: the module is usable afterwards
    visitron_amd.check_errors()
    with pytest.raises(RuntimeError):
        prod(ids.cpu())  # no CPU fallback


def test_trunk_uint8_inverted_mask_quirk(dev):
    """the rollout caller passes ~mask of a uint8 tensor (values 255/254);
    the trunk must do (1 - m) * -10000 literally."""
    from oracle.modeling import BertImgModelwithLocationEmbeds as OTrunk
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds

    cfg = mini_config()
    ref, prod = model_pair(OTrunk, BertImgModelwithLocationEmbeds, cfg, seed=5, device=dev)
    ids = torch.randint(1, cfg.vocab_size, (2, 12))
    pad = torch.zeros(2, 12, dtype=torch.uint8)
    pad[1, 7:] = 1
    m = ~pad
    with torch.no_grad():
        want = ref(ids, attention_mask=m)
        got = prod(ids.to(dev), attention_mask=m.to(dev))
    assert maxabs(got[0], want[0]) < TOL_BF16


def test_pretrain_heads_and_losses_match_oracle_mini(dev):
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch

    cfg = mini_config(use_img_layernorm=True, img_layer_norm_eps=1e-12)
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=6, device=dev)
    b = make_batch(cfg, 5, text_len=24, region_len=11, seed=3)
    with torch.no_grad():
        want = ref(**b)
        got = prod(**_to(b, dev))
        seq, pooled = ref.bert(**{k: v for k, v in b.items() if k in ("input_ids", "attention_mask", "img_feats", "img_location_embeddings")})[:2]
        w_scores, w_tok, w_act = ref.heads(seq, pooled)
        outs, p_pooled, _, B, S = prod.bert.run_trunk(
            b["input_ids"].to(dev), attention_mask=b["attention_mask"].to(dev), img_feats=b["img_feats"].to(dev),
            img_location_embeddings=b["img_location_embeddings"].to(dev))
        g_scores, g_tok, g_act = prod.head_outputs(outs[-1], p_pooled)
    check_close("mini heads prediction_scores", g_scores, w_scores.view(B * S, -1), TOL_BF16)
    check_close("mini heads token_probs", g_tok, w_tok.view(B * S, -1), TOL_BF16)
    check_close("mini heads action_scores", g_act, w_act, TOL_BF16)
    for i, n in enumerate(("loss", "mask_loss", "next_loss", "token_loss")):
        check_close("mini heads " + n, float(got[i]), float(want[i]), TOL_BF16)
    assert len(got) == 7 and all(torch.is_tensor(t) and t.dim() == 0 for t in got)


def test_base_config_cfg1_matches_oracle(dev):
    """BASELINE config 1/2 shape: 12L/768d, 128 text + 100 region tokens, B=2."""
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch

    cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=0, device=dev, weight_std=0.03)
    b = make_batch(cfg, 2, seed=1234)
    trunk_keys = ("input_ids", "attention_mask", "img_feats", "img_location_embeddings")
    with torch.no_grad():
        want = ref(**b)
        got = prod(**_to(b, dev))
        w_seq, w_pool = ref.bert(**{k: b[k] for k in trunk_keys})[:2]
        g_seq, g_pool = prod.bert(**{k: b[k].to(dev) for k in trunk_keys})[:2]
        w_scores, _, w_act = ref.heads(w_seq, w_pool)
        g_scores = prod.mlmhead(g_seq)
        g_act = prod.next_action(g_pool)
    # north_star: 5e-2 ABSOLUTE for the bf16 path -- asserted as it stands on every output, these harsher-than-init weights
    # included (hash weights of std 0.03, LayerNorm gains 1 +- 0.1, non-zero biases; the reference's own init is the case of
    # tests/test_gpu_round3.py).  Round 2 missed it on the two large tensors (5.9e-2 / 6.6e-2 with the bf16 residual stream
    # of the seven-launch layer); the deferred-LayerNorm path with its fp16 stream is what closed it.
    check_close("base cfg1 sequence_output", g_seq, w_seq, TOL_BF16)
    check_close("base cfg1 prediction_scores", g_scores, w_scores, TOL_BF16)
    # the seven-launch layer (VT_DEFERRED_LN=0: what training-mode forwards and compacted rows run) at the same flat 5e-2:
    # since round 4 it keeps its residual stream at fp16 precision (fp16 pre-LayerNorm sums, fp16 copies of the LayerNorm
    # outputs for the residual adds; 5.8e-2 with the bf16 stream of rounds 1-3); and its opt-in fp32 last layer
    prod.bert.encoder.deferred_ln = False
    with torch.no_grad():
        g_seq7 = prod.bert(**{k: b[k].to(dev) for k in trunk_keys})[0]
        prod.bert.encoder.precise_final = True
        g_seq2 = prod.bert(**{k: b[k].to(dev) for k in trunk_keys})[0]
    prod.bert.encoder.precise_final = False
    prod.bert.encoder.deferred_ln = True
    check_close("base cfg1 sequence_output (seven-launch layer, fp16 residual stream)", g_seq7, w_seq, TOL_BF16)
    check_close("base cfg1 sequence_output (seven-launch layer, fp16 stream, precise_final)", g_seq2, w_seq, TOL_BF16)
    check_close("base cfg1 pooled_output", g_pool, w_pool, TOL_BF16)
    check_close("base cfg1 action_scores", g_act, w_act, TOL_BF16)
    for i, n in enumerate(("loss", "mask_loss", "next_loss", "token_loss")):
        check_close("base cfg1 " + n, float(got[i]), float(want[i]), TOL_BF16)


def test_output_attentions_matches_oracle(dev):
    """config.output_attentions (oscar/modeling_bert.py:74-79, 157-168; encoder.py:299-303): the per-layer attention
    probabilities [B, heads, S, S] behind the hidden states in the encoder / trunk outputs, and as the second output of
    the attention sub-modules (with head_mask applied)."""
    from oracle.modeling import BertImgModelwithLocationEmbeds as OTrunk
    from oracle.modeling import CaptionBertLayer as OLayer
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds, CaptionBertLayer
    from visitron_amd.synth import make_batch

    cfg = mini_config(output_attentions=True, output_hidden_states=True)
    ref, prod = model_pair(OTrunk, BertImgModelwithLocationEmbeds, cfg, seed=6, device=dev)
    b = make_batch(cfg, 3, text_len=21, region_len=13, seed=9, with_labels=False)
    with torch.no_grad():
        want = ref(**b)
        got = prod(**_to(b, dev))
    assert len(got) == len(want) == 4 and len(got[3]) == len(want[3]) == cfg.num_hidden_layers
    assert maxabs(got[0], want[0]) < TOL_BF16
    for pg, pw in zip(got[3], want[3]):
        assert pg.shape == pw.shape == (3, cfg.num_attention_heads, 34, 34)
        assert maxabs(pg, pw) < 2e-2                      # probabilities are <= 1
        assert maxabs(pg.sum(-1), torch.ones(3, cfg.num_attention_heads, 34)) < 1e-2
    # sub-module call with a head mask
    refl, prodl = model_pair(OLayer, CaptionBertLayer, cfg, seed=7, device=dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 19, cfg.hidden_size, generator=g)
    ext = torch.zeros(2, 1, 1, 19)
    ext[1, 0, 0, 10:] = -10000.0
    hm = torch.tensor([1.0, 0.5]).view(1, 2, 1, 1)
    with torch.no_grad():
        w = refl.attention.self(x, ext, hm)
        o = prodl.attention.self(x.to(dev), ext.to(dev), hm.to(dev))
        wl, ol = refl(x, ext), prodl(x.to(dev), ext.to(dev))
    assert len(o) == len(w) == 2 and maxabs(o[1], w[1]) < 2e-2 and maxabs(o[0], w[0]) < TOL_BF16
    assert len(ol) == len(wl) == 2 and maxabs(ol[1], wl[1]) < 2e-2


def test_three_dimensional_attention_mask(dev):
    """attention_mask of rank 3 ([B, S, S], encoder.py:226-229): a per-query mask -> [B,1,S,S] additive bias; trunk
    outputs, attention probabilities and the encoder called directly with the extended mask."""
    from oracle.modeling import BertImgModelwithLocationEmbeds as OTrunk
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.synth import make_batch

    cfg = mini_config(output_attentions=True)
    ref, prod = model_pair(OTrunk, BertImgModelwithLocationEmbeds, cfg, seed=8, device=dev)
    b = make_batch(cfg, 3, text_len=21, region_len=13, seed=5, with_labels=False)
    S = 34
    g = torch.Generator().manual_seed(4)
    m3 = (torch.rand(3, S, S, generator=g) > 0.3).float()
    m3[:, :, 0] = 1.0                       # every query keeps at least one key
    m3 = torch.tril(m3)                     # a causal-like pattern on top of the random one
    m3[:, :, 0] = 1.0
    b["attention_mask"] = m3
    with torch.no_grad():
        want = ref(**b)
        got = prod(**_to(b, dev))
    assert maxabs(got[0], want[0]) < TOL_BF16 and maxabs(got[1], want[1]) < TOL_BF16
    for pg, pw in zip(got[2], want[2]):
        assert maxabs(pg, pw) < 2e-2
        assert float(pg.cpu()[m3[:, None].expand_as(pg) == 0].abs().max()) < 1e-6     # masked pairs get no weight
    # the encoder on its own, extended mask [B,1,S,S]
    x = torch.randn(3, S, cfg.hidden_size, generator=g)
    ext = (1.0 - m3)[:, None] * -10000.0
    with torch.no_grad():
        we = ref.encoder(x, ext, head_mask=[None] * cfg.num_hidden_layers)
        ge = prod.encoder(x.to(dev), ext.to(dev), head_mask=[None] * cfg.num_hidden_layers)
    assert maxabs(ge[0], we[0]) < TOL_BF16


def test_history_states_match_oracle(dev):
    """history_state / encoder_history_states (oscar/modeling_bert.py:37-41, 148-155; encoder.py:271-274): each layer's
    keys and values run over cat([history_i, hidden], 1); text-only trunk call, the encoder called directly, one
    attention sub-module with the probabilities, and the image-features assertion."""
    from oracle.modeling import BertImgModelwithLocationEmbeds as OTrunk
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.synth import make_batch

    cfg = mini_config(output_attentions=True, output_hidden_states=True)
    ref, prod = model_pair(OTrunk, BertImgModelwithLocationEmbeds, cfg, seed=12, device=dev)
    B, T, Sh, H = 3, 18, 11, cfg.hidden_size
    b = make_batch(cfg, B, text_len=T, region_len=4, seed=4, with_labels=False)
    g = torch.Generator().manual_seed(8)
    hist = [torch.randn(B, Sh, H, generator=g) for _ in range(cfg.num_hidden_layers)]
    am = torch.ones(B, Sh + T, dtype=torch.long)
    am[1, 3:7] = 0
    am[2, Sh + 12:] = 0
    args = dict(input_ids=b["input_ids"], attention_mask=am)
    with torch.no_grad():
        want = ref(encoder_history_states=hist, **args)
        got = prod(encoder_history_states=[h.to(dev) for h in hist], **_to(args, dev))
    assert len(got) == len(want) == 4
    assert maxabs(got[0], want[0]) < TOL_BF16 and maxabs(got[1], want[1]) < TOL_BF16
    for hg, hw in zip(got[2], want[2]):
        assert hg.shape == hw.shape == (B, T, H) and maxabs(hg, hw) < TOL_BF16
    for pg, pw in zip(got[3], want[3]):
        assert pg.shape == pw.shape == (B, cfg.num_attention_heads, T, Sh + T)
        assert maxabs(pg, pw) < 2e-2
    # the encoder directly with the additive extended mask, and one attention sub-module
    x = torch.randn(B, T, H, generator=g)
    ext = ((1.0 - am.float()) * -10000.0).view(B, 1, 1, Sh + T)
    with torch.no_grad():
        we = ref.encoder(x, ext, head_mask=[None] * cfg.num_hidden_layers, encoder_history_states=hist)
        ge = prod.encoder(x.to(dev), ext.to(dev), head_mask=[None] * cfg.num_hidden_layers,
                          encoder_history_states=[h.to(dev) for h in hist])
        wa = ref.encoder.layer[0].attention.self(x, ext, None, hist[0])
        ga = prod.encoder.layer[0].attention.self(x.to(dev), ext.to(dev), None, hist[0].to(dev))
    assert maxabs(ge[0], we[0]) < TOL_BF16
    assert maxabs(ga[0], wa[0]) < TOL_BF16 and ga[1].shape == wa[1].shape and maxabs(ga[1], wa[1]) < 2e-2
    with pytest.raises(AssertionError):
        prod(encoder_history_states=[h.to(dev) for h in hist], **_to(b, dev))
