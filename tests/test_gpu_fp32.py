"""GPU: the fp32 parity path (visitron_amd.set_precision(model, "fp32"); csrc/fp32_path.hip) -- BASELINE north_star:
outputs within 1e-3 of the reference's fp32 CPU forward (the reference computes in fp32: encoder.py:238-240)."""
import os

import numpy as np
import pytest
import torch

from helpers import check_close, model_pair

pytestmark = pytest.mark.gpu
TOL_FP32 = 1e-3
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRUNK_KEYS = ("input_ids", "attention_mask", "img_feats", "img_location_embeddings")


def _to(b, dev):
    return {k: v.to(dev) for k, v in b.items()}


@pytest.mark.parametrize("M,N,K,act,res,kn", [(300, 200, 2182, 0, False, False), (77, 768, 768, 1, False, False),
                                               (513, 130, 64, 2, True, False), (228, 64, 228, 0, False, True),
                                               (37, 37, 64, 0, True, False), (1000, 3072, 768, 1, True, False)])
def test_linear_f32_matches_fp64(dev, M, N, K, act, res, kn):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn((K, N) if kn else (N, K), generator=g) * 0.05
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g) if res else None
    z = a.double() @ (w.double() if kn else w.double().t()) * 0.5 + b.double()
    if act == 1:
        z = torch.nn.functional.gelu(z)
    elif act == 2:
        z = torch.tanh(z)
    if res:
        z = z + r.double()
    got = ops.linear_f32(a.to(dev), w.to(dev), b.to(dev), residual=None if r is None else r.to(dev), act=act, w_is_kn=kn,
                         alpha=0.5)
    torch.cuda.synchronize()
    check_close("linear_f32 M%d N%d K%d act%d" % (M, N, K, act), got, z.float(), 2e-5 * (1 + float(z.abs().max())))


def test_linear_f32_row_remap_and_strided_input(dev):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(3)
    B, R, S, H, K = 3, 5, 12, 64, 70
    a = torch.randn(B * R, K, generator=g)
    w = torch.randn(H, K, generator=g) * 0.1
    out = torch.zeros(B * S, H, device=dev)
    ops.linear_f32(a.to(dev), w.to(dev), None, out=out[7:], ldc=H, grp_rows=R, grp_stride=S)
    want = torch.zeros(B, S, H)
    want[:, 7:] = (a.double() @ w.double().t()).float().view(B, R, H)
    check_close("linear_f32 row remap", out.view(B, S, H), want, 1e-5)
    # token 0 of every sequence through the row stride (the pooler's read)
    seq = torch.randn(B * S, H, generator=g)
    got = ops.linear_f32(seq.to(dev), w[:, :H].contiguous().to(dev), None, act=2, M=B, lda=S * H)
    check_close("linear_f32 strided rows", got, torch.tanh(seq.view(B, S, H)[:, 0].double() @ w[:, :H].double().t()).float(), 1e-5)


@pytest.mark.parametrize("S,mode", [(37, "raw"), (228, "raw"), (300, "additive"), (45, "per_query"), (64, "none")])
def test_attention_f32_matches_reference_arithmetic(dev, S, mode):
    """oscar/modeling_bert.py:47-72 on the packed projection: scores / sqrt(64) + mask, softmax, * head_mask, probs v."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(S)
    B, nh = 2, 3
    H = nh * 64
    qkv = torch.randn(B * S, 3 * H, generator=g)
    keep = (torch.rand(B, S, generator=g) > 0.3).float()
    keep[:, 0] = 1
    hm = torch.tensor([1.0, 0.0, 0.5])
    if mode == "raw":
        mask, add, ext = keep, False, ((1.0 - keep) * -10000.0)[:, None, None, :]
    elif mode == "additive":
        bias = (1.0 - keep) * -10000.0
        mask, add, ext = bias, True, bias[:, None, None, :]
    elif mode == "per_query":
        m3 = (torch.rand(B, S, S, generator=g) > 0.3).float()
        m3[:, :, 0] = 1
        bias = (1.0 - m3) * -10000.0
        mask, add, ext = bias, True, bias[:, None]
    else:
        mask, add, ext = None, False, 0.0
    sp = lambda t: t.double().view(B, S, nh, 64).permute(0, 2, 1, 3)
    q, k, v = sp(qkv[:, :H]), sp(qkv[:, H:2 * H]), sp(qkv[:, 2 * H:])
    sc = q @ k.transpose(-1, -2) / 8.0 + (ext if torch.is_tensor(ext) else 0.0)
    pr = torch.softmax(sc, -1) * hm.double().view(1, nh, 1, 1)
    want = (pr @ v).permute(0, 2, 1, 3).reshape(B * S, H)
    ctx, probs = ops.attention_f32(qkv.to(dev), B, S, nh, mask=None if mask is None else mask.contiguous().to(dev),
                                   mask_additive=add, head_scale=hm.to(dev), want_probs=True)
    torch.cuda.synchronize()
    check_close("attention_f32 S=%d %s ctx" % (S, mode), ctx, want.float(), 2e-5)
    check_close("attention_f32 S=%d %s probs" % (S, mode), probs, pr.float(), 2e-6)


@pytest.mark.parametrize("H", [128, 768, 3072])
def test_layernorm_rows_all_type_combinations(dev, H):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(H)
    x = torch.randn(33, H, generator=g) * 2 + 0.5
    gam, bet = torch.rand(H, generator=g) + 0.5, torch.randn(H, generator=g) * 0.1
    for xin in (torch.float32, torch.bfloat16):
        xs = x.to(xin)
        want = torch.nn.functional.layer_norm(xs.double(), (H,), gam.double(), bet.double(), 1e-12)
        for out32 in (True, False):
            got = ops.layernorm_rows(xs.to(dev), gam.to(dev), bet.to(dev), 1e-12, out_f32=out32)
            tol = 2e-5 if out32 else 2e-2
            check_close("layernorm_rows H=%d in=%s out32=%s" % (H, str(xin)[6:], out32), got, want.float(), tol)


def test_fp32_mode_mini_model_and_golden(dev):
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd import set_precision
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch

    cfg = mini_config(use_img_layernorm=True, img_layer_norm_eps=1e-12, output_hidden_states=True)
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=6, device=dev)
    set_precision(prod, "fp32")
    b = make_batch(cfg, 5, text_len=24, region_len=11, seed=3)
    with torch.no_grad():
        want = ref(**b)
        got = prod(**_to(b, dev))
        w_tr = ref.bert(**{k: b[k] for k in TRUNK_KEYS})
        g_tr = prod.bert(**{k: b[k].to(dev) for k in TRUNK_KEYS})
    for i in range(4):
        check_close("fp32 mini tuple7[%d]" % i, float(got[i]), float(want[i]), TOL_FP32)
    for i in range(4, 7):
        check_close("fp32 mini tuple7[%d]" % i, float(got[i]), float(want[i]), 1e-6)
    check_close("fp32 mini sequence_output", g_tr[0], w_tr[0], TOL_FP32)
    check_close("fp32 mini pooled_output", g_tr[1], w_tr[1], TOL_FP32)
    assert len(g_tr[2]) == len(w_tr[2]) == cfg.num_hidden_layers + 1
    for i, (a, c) in enumerate(zip(g_tr[2], w_tr[2])):
        check_close("fp32 mini hidden[%d]" % i, a, c, TOL_FP32)
    prod.train()
    with pytest.raises(NotImplementedError):
        prod(**_to(b, dev))                     # fp32 serves inference; training stays on the bf16 kernels
    # golden fixture: the REFERENCE's own outputs (tests/golden/make_golden_from_reference.py), no oracle call
    g = np.load(os.path.join(GOLD, "ref_mini.npz"))
    from visitron_amd.synth import deterministic_state_dict

    cfg2 = mini_config()
    m = PreTrainOscar(cfg2).eval()
    m.load_state_dict(deterministic_state_dict(m, seed=3, weight_std=0.05))
    m.tie_weights()
    m = set_precision(m.to(dev), "fp32")
    gb = {k: torch.from_numpy(g["in_" + k]).to(dev) for k in TRUNK_KEYS}
    with torch.no_grad():
        outs, pooled, _, B, S = m.bert.run_trunk(gb["input_ids"], attention_mask=gb["attention_mask"], img_feats=gb["img_feats"],
                                                 img_location_embeddings=gb["img_location_embeddings"])
        scores, tokp, act = m.head_outputs(outs[-1], pooled)
    check_close("fp32 golden mini sequence_output", outs[-1], g["sequence_output"], TOL_FP32)
    check_close("fp32 golden mini prediction_scores", scores, g["prediction_scores"], TOL_FP32)
    check_close("fp32 golden mini token_probs", tokp, g["token_probs"], TOL_FP32)
    check_close("fp32 golden mini action_scores", act, g["action_scores"], TOL_FP32)


def test_fp32_mode_base_config_cfg0_within_1e_3(dev):
    """BASELINE configs[0]: base config, B = 2, 128 text + 100 region tokens, against the CPU fp32 oracle: action logits,
    MLM logits and the sequence output within 1e-3 (north_star)."""
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd import set_precision
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch

    cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=0, device=dev, weight_std=0.03)
    set_precision(prod, "fp32")
    b = make_batch(cfg, 2, seed=1234)
    with torch.no_grad():
        want = ref(**b)
        got = prod(**_to(b, dev))
        w_seq, w_pool = ref.bert(**{k: b[k] for k in TRUNK_KEYS})[:2]
        g_seq, g_pool = prod.bert(**{k: b[k].to(dev) for k in TRUNK_KEYS})[:2]
        w_scores, w_tok, w_act = ref.heads(w_seq, w_pool)
        g_scores, g_tok, g_act = prod.head_outputs(g_seq.reshape(-1, cfg.hidden_size), g_pool)
    check_close("fp32 base cfg0 sequence_output", g_seq, w_seq, TOL_FP32)
    check_close("fp32 base cfg0 pooled_output", g_pool, w_pool, TOL_FP32)
    check_close("fp32 base cfg0 prediction_scores", g_scores, w_scores, TOL_FP32)
    check_close("fp32 base cfg0 token_probs", g_tok, w_tok, TOL_FP32)
    check_close("fp32 base cfg0 action_scores", g_act, w_act, TOL_FP32)
    for i in range(4):
        check_close("fp32 base cfg0 tuple7[%d]" % i, float(got[i]), float(want[i]), TOL_FP32)
    g = np.load(os.path.join(GOLD, "ref_base_cfg0.npz"))
    check_close("fp32 golden base cfg1 sequence_output slice", g_seq.cpu()[:, ::19, ::31], g["sequence_output_slice"], TOL_FP32)
    check_close("fp32 golden base cfg1 prediction_scores slice", g_scores.cpu().view(2, 228, -1)[:, ::19, ::1009],
                g["prediction_scores_slice"], TOL_FP32)
    check_close("fp32 golden base cfg1 action_scores", g_act, g["action_scores"], TOL_FP32)


def test_fp32_mode_rollout_caller_text_only_with_history_and_head_mask(dev):
    """The text-only call of the rollout caller (agent_models.py:270-275), history states and head_mask in fp32 mode."""
    from oracle.modeling import BertImgModelwithLocationEmbeds as OTrunk
    from visitron_amd import set_precision
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds

    cfg = mini_config(output_attentions=True)
    ref, prod = model_pair(OTrunk, BertImgModelwithLocationEmbeds, cfg, seed=5, device=dev)
    set_precision(prod, "fp32")
    g = torch.Generator().manual_seed(2)
    B, T, Sh = 3, 14, 6
    ids = torch.randint(1, cfg.vocab_size, (B, T), generator=g)
    pad = torch.zeros(B, T, dtype=torch.uint8)
    pad[1, 9:] = 1
    hm = torch.tensor([[1.0, 0.5], [0.0, 1.0]])
    with torch.no_grad():
        want = ref(ids, attention_mask=~pad, head_mask=hm)       # the uint8 ~mask quirk: values 255 / 254
        got = prod(ids.to(dev), attention_mask=(~pad).to(dev), head_mask=hm.to(dev))
    check_close("fp32 text-only uint8-mask sequence_output", got[0], want[0], TOL_FP32)
    for i, (a, c) in enumerate(zip(got[2], want[2])):
        check_close("fp32 text-only attentions[%d]" % i, a, c, 1e-4)
    hist = [torch.randn(B, Sh, cfg.hidden_size, generator=g) for _ in range(cfg.num_hidden_layers)]
    m = torch.ones(B, Sh + T)
    m[2, 3] = 0
    with torch.no_grad():
        want = ref(ids, attention_mask=m, encoder_history_states=hist)
        got = prod(ids.to(dev), attention_mask=m.to(dev), encoder_history_states=[h.to(dev) for h in hist])
    check_close("fp32 history sequence_output", got[0], want[0], TOL_FP32)
