"""GPU: the HIP path against the outputs of the REFERENCE's own source (tests/golden/ref_*.npz, written in the build
container by tests/golden/make_golden_from_reference.py from oscar/modeling_bert.py, tasks/viewpoint_select/encoder.py
and agent_models.py).  No oracle call: inputs come from the seeded generators (checked against the fixture's checksums),
weights from the integer-hash state dict, expected values from the fixture.

Covers the trunk's edge cases at mini size (head_mask 1-D / 2-D, 3-D mask, float and uint8 ~mask, history states, token
types / positions, text_only, loss corners, hidden states / attentions, image LayerNorm) and the reference's SHIPPED
shapes through the model and the engine: pretrain 511 text + 256 regions, B = 2 (data_loader_pretrain.py:618-626) and
the rollout's text-only T = 511, B = 8 (agent_models.py:270-275); plus the rollout modules."""
import os

import numpy as np
import pytest
import torch

from helpers import check_close
from test_gpu_train import LOSS_TOL, _check_grad_slices

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 5e-2   # bf16 path, BASELINE.json north_star
TRUNK_KEYS = ("input_ids", "attention_mask", "img_feats", "img_location_embeddings")


def _product(cls, cfg, seed, std, dev):
    from visitron_amd.synth import deterministic_state_dict

    m = cls(cfg).eval()
    m.load_state_dict(deterministic_state_dict(m, seed=seed, weight_std=std))
    if hasattr(m, "tie_weights"):
        m.tie_weights()
    return m.to(dev)


def _d(x, dev):
    return x.to(dev) if isinstance(x, torch.Tensor) else torch.from_numpy(x).to(dev)


def test_mini_edge_cases_against_the_reference_outputs(dev):
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds, PreTrainOscar

    g = np.load(os.path.join(GOLD, "ref_mini.npz"))
    cfg = mini_config()
    m = _product(PreTrainOscar, cfg, 3, 0.05, dev)
    b = {k: _d(g["in_" + k], dev) for k in TRUNK_KEYS + ("labels", "token_labels", "next_action")}
    tk = {k: b[k] for k in TRUNK_KEYS}
    B, T, R = 3, 20, 17

    def two(tag, out):
        check_close("ref mini %s sequence_output" % tag, out[0], g[tag + "_0"], TOL)
        check_close("ref mini %s pooled_output" % tag, out[1], g[tag + "_1"], TOL)

    with torch.no_grad():
        two("headmask1d", m.bert(head_mask=_d(g["in_head_mask_1d"], dev), **tk))
        two("headmask2d", m.bert(head_mask=_d(g["in_head_mask_2d"], dev), **tk))
        two("mask3d", m.bert(b["input_ids"], attention_mask=_d(g["in_mask3d"], dev), img_feats=b["img_feats"],
                             img_location_embeddings=b["img_location_embeddings"]))
        two("maskfloat", m.bert(b["input_ids"], attention_mask=b["attention_mask"].float() * 0.5 + 0.25,
                                img_feats=b["img_feats"], img_location_embeddings=b["img_location_embeddings"]))
        two("masku8", m.bert(b["input_ids"], attention_mask=_d(g["in_mask_u8"], dev)))
        hist = [_d(g["in_history_%d" % i], dev) for i in range(cfg.num_hidden_layers)]
        mh = torch.cat([torch.ones(B, hist[0].shape[1], dtype=torch.long, device=dev), b["attention_mask"][:, :T]], 1)
        two("history", m.bert(b["input_ids"], attention_mask=mh, encoder_history_states=hist))
        tt = (torch.arange(T, device=dev)[None, :] >= 9).long().expand(B, T).contiguous()
        pid = torch.arange(T - 1, -1, -1, device=dev)[None, :].expand(B, T).contiguous()
        two("types_positions", m.bert(b["input_ids"], token_type_ids=tt, position_ids=pid,
                                      attention_mask=b["attention_mask"][:, :T]))
        two("text_only", m(b["input_ids"], attention_mask=b["attention_mask"], img_feats=b["img_feats"],
                           img_location_embeddings=b["img_location_embeddings"], text_only=True))

        def seven(tag, **over):
            bb = dict(b)
            bb.update(over)
            got = [float(x) for x in m(**bb)]
            want = g[tag]
            for i in range(7):
                if np.isnan(want[i]):
                    assert np.isnan(got[i]), (tag, i, got)       # the reference's own NaN corner (mean over no element)
                else:
                    check_close("ref mini %s[%d]" % (tag, i), got[i], float(want[i]), TOL if i < 4 else 1e-6)

        seven("tuple7_no_labels", labels=torch.full_like(b["labels"], -1))
        seven("tuple7_no_token_labels", token_labels=torch.full_like(b["token_labels"], -1))
        na = b["next_action"].clone()
        na[1] = -1
        seven("tuple7_one_action_ignored", next_action=na)
        seven("tuple7_all_actions_ignored", next_action=torch.full_like(na, -1))

        cfg2 = mini_config(output_hidden_states=True, output_attentions=True, use_img_layernorm=True, img_layer_norm_eps=1e-5)
        m2 = _product(BertImgModelwithLocationEmbeds, cfg2, 4, 0.05, dev)
        out = m2(**tk)
        assert len(out) == 4 and len(out[2]) == cfg.num_hidden_layers + 1 and len(out[3]) == cfg.num_hidden_layers
        check_close("ref mini imgln sequence_output", out[0], g["imgln_seq"], TOL)
        check_close("ref mini imgln pooled_output", out[1], g["imgln_pooled"], TOL)
        for i, h in enumerate(out[2]):
            check_close("ref mini hidden_states[%d]" % i, h, g["hidden_states_%d" % i], TOL)
        for i, a in enumerate(out[3]):
            check_close("ref mini attentions[%d]" % i, a, g["attentions_%d" % i], 2e-2)


def _base(dev, fname, B, T, R, seed):
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch

    g = np.load(os.path.join(GOLD, fname))
    cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    b = make_batch(cfg, B, text_len=T, region_len=R, seed=seed, with_labels=R > 0)
    for k, v in b.items():                                         # the generator reproduces the fixture's inputs
        if v.dtype.is_floating_point:
            c = g["in_%s_checksum" % k]
            assert abs(float(v.double().sum()) - c[0]) <= 1e-9 * abs(c[0]) + 1e-9, k
        else:
            assert np.array_equal(g["in_" + k], v.numpy()), k
    m = _product(PreTrainOscar, cfg, 0, 0.03, dev)
    return g, cfg, {k: v.to(dev) for k, v in b.items()}, m


@pytest.mark.parametrize("fname,T,R,seed,tag", [("ref_shipped_s767.npz", 511, 256, 767, "shipped S=767"),
                                                 ("ref_base_long.npz", 512, 144, 77, "S=656")])
def test_shipped_pretrain_shape_forward_and_one_step(dev, fname, T, R, seed, tag):
    """511 text + 256 regions, B = 2 (the reference's shipped pretrain batch; S = 767 is odd and > 3 key blocks of 256)
    and configs[4]'s 512 + 144: inference outputs, then one engine step's losses and every gradient against the
    reference's own backward."""
    from visitron_amd.training import PretrainEngine

    g, cfg, b, m = _base(dev, fname, 2, T, R, seed)
    S, st = T + R, int(g["seq_stride"][0])
    with torch.no_grad():
        outs, pooled, _, B, S_ = m.bert.run_trunk(b["input_ids"], attention_mask=b["attention_mask"], img_feats=b["img_feats"],
                                                  img_location_embeddings=b["img_location_embeddings"])
        scores, tokp, act = m.head_outputs(outs[-1], pooled)
        out7 = m(**b)
    seq = outs[-1].float().cpu().view(2, S, -1)
    check_close("ref %s sequence_output slice" % tag, seq[:, ::st, ::31], g["sequence_output_slice"], TOL)
    check_close("ref %s pooled_output" % tag, pooled, g["pooled_output"], TOL)
    check_close("ref %s prediction_scores slice" % tag, scores.float().cpu().view(2, S, -1)[:, ::st, ::1009],
                g["prediction_scores_slice"], TOL)
    check_close("ref %s token_probs slice" % tag, tokp.float().cpu().view(2, S, -1)[:, ::st, ::97], g["token_probs_slice"], TOL)
    check_close("ref %s action_scores" % tag, act, g["action_scores"], TOL)
    for i in range(4):
        check_close("ref %s eval tuple7[%d]" % (tag, i), float(out7[i]), float(g["tuple7"][i]), LOSS_TOL)
    del seq, scores, tokp, outs
    m.train()
    eng = PretrainEngine(m)
    eng.compact_min_rows = 0
    got = eng.forward_backward(b)
    torch.cuda.synchronize()
    for i in range(4):
        # (B = 2: a mean over ~150 supervised words; measured 4.7e-3 on the sum of the three losses at S = 767)
        check_close("ref %s train tuple7[%d]" % (tag, i), float(got[i]), float(g["tuple7"][i]), 2 * LOSS_TOL)
    for i in range(4, 7):
        check_close("ref %s train tuple7[%d]" % (tag, i), float(got[i]), float(g["tuple7"][i]), 1e-6)
    _check_grad_slices("ref %s train" % tag, m, g, bound=0.035)     # S = 656 measured 1.6 % worst (round 2)


def test_shipped_rollout_shape_text_only_t511_b8(dev):
    """Text-only T = 511, B = 8, base config: the uint8 ~mask the rollout caller passes (254 / 255) and the 0 / 1 mask."""
    g, cfg, b, m = _base(dev, "ref_text511.npz", 8, 511, 0, 511)
    with torch.no_grad():
        a = m.bert(b["input_ids"], attention_mask=~(b["attention_mask"] == 0).byte())
        a01 = m.bert(b["input_ids"], attention_mask=b["attention_mask"])
    check_close("ref text511 u8 sequence_output slice", a[0].float().cpu()[:, ::37, ::31], g["sequence_output_slice"], TOL)
    check_close("ref text511 u8 pooled_output", a[1], g["pooled_output"], TOL)
    check_close("ref text511 0/1 sequence_output slice", a01[0].float().cpu()[:, ::37, ::31], g["sequence_output_01mask_slice"], TOL)
    check_close("ref text511 0/1 pooled_output", a01[1], g["pooled_output_01mask"], TOL)


def test_rollout_modules_against_the_reference_outputs(dev):
    """agent_models.py's SoftDotAttention (four output modes, with / without mask), one AttnDecoderLSTM step and
    OscarEncoder (bool and uint8 padding masks, ragged lengths, 768-wide 2-layer trunk) against the reference's outputs."""
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.rollout import AttnDecoderLSTM, OscarEncoder, SoftDotAttention
    from visitron_amd.synth import deterministic_state_dict

    g = np.load(os.path.join(GOLD, "ref_rollout.npz"))
    att = _product(lambda _: SoftDotAttention(128, 132), None, 7, 0.08, dev)
    h, ctx, mask = (_d(g[k], dev) for k in ("sda_h", "sda_ctx", "sda_mask"))
    with torch.no_grad():
        for mi, mk in enumerate((None, mask)):
            for tilde in (True, False):
                for prob in (True, False):
                    out = att(h, ctx, None if mk is None else mk.clone(), output_tilde=tilde, output_prob=prob)
                    tag = "sda_m%d_t%d_p%d" % (mi, tilde, prob)
                    check_close("ref rollout %s[0]" % tag, out[0], g[tag + "_0"], TOL)
                    want = g[tag + "_1"]
                    fin = np.isfinite(want)
                    got = out[1].float().cpu().numpy()
                    assert np.array_equal(fin, np.isfinite(got)), tag                 # -inf exactly where masked
                    scale = max(1.0, float(np.abs(want[fin]).max()))
                    check_close("ref rollout %s[1]" % tag, np.where(fin, got, 0) / scale, np.where(fin, want, 0) / scale, 2e-2)
        dec = _product(lambda _: AttnDecoderLSTM(4, 64, 128, 0.5, feature_size=132), None, 8, 0.06, dev)
        ins = {k[len("dec_in_"):]: _d(g[k], dev) for k in g.files if k.startswith("dec_in_")}
        out = dec(**ins)
        for i, n in enumerate(("h_1", "c_1", "logit", "h_tilde")):
            scale = max(1.0, float(np.abs(g["dec_" + n]).max()))
            check_close("ref rollout decoder %s" % n, out[i].float().cpu() / scale, g["dec_" + n] / scale, TOL if n != "logit" else 2e-2)
    cfg = BertConfig(num_hidden_layers=2, vocab_size=600, max_position_embeddings=64, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0, detector_classes=40)
    bert = BertImgModelwithLocationEmbeds(cfg).eval()
    bert.load_state_dict(deterministic_state_dict(bert, seed=9, weight_std=0.03))
    enc = OscarEncoder(None, bert, 128, 96, 0.5).eval()
    sd = deterministic_state_dict(enc, seed=10, weight_std=0.03)
    sd.update({k: v for k, v in enc.state_dict().items() if k.startswith("bert.")})
    enc.load_state_dict(sd)
    enc = enc.to(dev)
    ids, lengths = _d(g["enc_in_ids"], dev), torch.tensor([int(x) for x in g["enc_in_lengths"]])
    pad = torch.zeros(ids.shape, dtype=torch.bool)
    for i, n in enumerate(lengths.tolist()):
        pad[i, n:] = True
    with torch.no_grad():
        for tag, mk in (("bool", pad), ("u8", pad.byte())):
            out = enc(ids, lengths, mk.to(dev))
            for i, n in enumerate(("ctx", "decoder_init", "c_t")):
                check_close("ref rollout OscarEncoder %s %s" % (tag, n), out[i], g["enc_%s_%s" % (tag, n)], TOL)


@pytest.mark.parametrize("name,kw", [("rev", dict(reverse_input=True)), ("l2", dict(num_layers=2)),
                                     ("l2bi_rev", dict(num_layers=2, bidirectional=True, reverse_input=True))])
def test_oscar_encoder_reverse_input_and_stacked_lstm_against_the_reference(dev, name, kw):
    """Round 6: OscarEncoder(reverse_input=True) (agent_models.py:277-282) and stacked / bidirectional encoder LSTMs
    (num_layers = 2, :223-230) -- constructor arguments the earlier rounds refused -- against outputs and gradients of the
    reference's own class run with them (ref_rollout2.npz): inference forward with bool and uint8 padding masks (the uint8
    `~mask` indexes every position: a whole-row reversal, padding included), then the training path's autograd nodes."""
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.rollout import OscarEncoder
    from visitron_amd.synth import deterministic_state_dict

    g = np.load(os.path.join(GOLD, "ref_rollout2.npz"))
    cfg = BertConfig(num_hidden_layers=2, vocab_size=600, max_position_embeddings=64, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0, detector_classes=40)
    bert = BertImgModelwithLocationEmbeds(cfg).eval()
    bert.load_state_dict(deterministic_state_dict(bert, seed=9, weight_std=0.03))
    enc = OscarEncoder(None, bert, 128, 96, 0.5, **kw).eval()
    sd = deterministic_state_dict(enc, seed=12, weight_std=0.03)
    sd.update({k: v for k, v in enc.state_dict().items() if k.startswith("bert.")})
    enc.load_state_dict(sd)
    enc = enc.to(dev)
    ids, lengths = _d(g["enc_in_ids"], dev), torch.tensor([int(x) for x in g["enc_in_lengths"]])
    pad = torch.zeros(ids.shape, dtype=torch.bool)
    for i, n in enumerate(lengths.tolist()):
        pad[i, n:] = True
    with torch.no_grad():
        for tag, mk in (("bool", pad), ("u8", pad.byte())):
            out = enc(ids, lengths, mk.to(dev))
            for i, n in enumerate(("ctx", "decoder_init", "c_t")):
                check_close("ref rollout OscarEncoder(%s) %s %s" % (name, tag, n), out[i], g["enc_%s_%s_%s" % (name, tag, n)], TOL)
    # the training path (autograd nodes, the trunk node on the pretrain engine): train() with every dropout probability at
    # zero is the fixture's eval() arithmetic with a graph -- outputs and gradients
    enc.drop.p = 0.0
    enc.lstm.dropout = 0.0
    enc.train()
    with torch.enable_grad():
        out = enc(ids, lengths, pad.to(dev))
        for i, n in enumerate(("ctx", "decoder_init", "c_t")):
            check_close("ref rollout OscarEncoder(%s) autograd %s" % (name, n), out[i], g["enc_%s_bool_%s" % (name, n)], TOL)
        (out[0].sum() + out[1].sum() + out[2].sum()).backward()
    # (attention key biases: their true gradient is exactly zero -- the softmax does not see a per-query shift -- and under
    # this sum-of-outputs loss the rounding noise left on both sides is above the absolute floor sized for the pretrain loss)
    _check_grad_slices("ref rollout OscarEncoder(%s)" % name, enc, g, bound=0.04, prefix="enc_%s_" % name,
                       skip=("attention.self.key.bias",))
