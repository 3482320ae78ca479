"""CPU: the oracle reproduces every fixture written by EXECUTING the reference's own source
(tests/golden/make_golden_from_reference.py -> tests/golden/ref_*.npz): oscar/modeling_bert.py:26-169,
tasks/viewpoint_select/encoder.py:142-441, agent_models.py:192-428, data_loader_pretrain.py:25-49,549-712.

The generator records bitwise equality in its own process (ref_pin_report.json: every max |oracle - reference| is 0.0).
Here the oracle is re-run, possibly with another thread count or on another CPU, so floating-point outputs are compared
at a few fp32 ulps of their magnitude; integer outputs must be equal.  What these fixtures do NOT pin is the arithmetic
inside the un-vendored pytorch-transformers blocks (the generator ran the reference over oracle.bert_blocks as their
stand-in): that rests on test_oracle_golden.py's cross-check and test_oracle_kat.py."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRUNK_KEYS = ("input_ids", "attention_mask", "img_feats", "img_location_embeddings")


def _load(name):
    return np.load(os.path.join(GOLD, name))


def _close(got, want, atol, what=""):
    got = got.detach().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    nan = np.isnan(want)
    assert np.array_equal(nan, np.isnan(got)), what
    np.testing.assert_allclose(got[~nan], want[~nan], atol=atol, rtol=1e-5, err_msg=what)


def _oracle(cls, cfg, seed, std):
    from visitron_amd.synth import deterministic_state_dict

    m = cls(cfg).eval()
    m.load_state_dict(deterministic_state_dict(m, seed=seed, weight_std=std))
    return m


def test_generator_recorded_bitwise_equality_for_every_output():
    r = json.load(open(os.path.join(GOLD, "ref_pin_report.json")))
    cases = r["max_abs_oracle_minus_reference"]
    assert set(cases) == {"mini", "base_cfg0", "base_long", "shipped_pretrain", "rollout", "text511", "data"}
    worst = {c: max(v.values()) for c, v in cases.items()}
    assert all(w == 0.0 for w in worst.values()), worst
    assert any("oracle.bert_blocks" in s for s in r["standins"])     # the stand-in is declared with the fixtures


def test_mini_fixture_every_output_gradient_and_edge_case():
    from oracle.modeling import BertImgModelwithLocationEmbeds, PreTrainOscar
    from visitron_amd.config import mini_config
    from visitron_amd.synth import make_batch

    g = _load("ref_mini.npz")
    cfg = mini_config()
    b = make_batch(cfg, 3, text_len=20, region_len=17, seed=11)
    for k, v in b.items():
        assert np.array_equal(g["in_" + k], v.numpy()), k
    m = _oracle(PreTrainOscar, cfg, 3, 0.05)
    tk = {k: b[k] for k in TRUNK_KEYS}
    with torch.no_grad():
        seq, pooled = m.bert(**tk)[:2]
        scores, tokp, act = m.heads(seq, pooled)
    _close(seq, g["sequence_output"], 2e-6), _close(pooled, g["pooled_output"], 2e-6)
    _close(scores, g["prediction_scores"], 5e-6), _close(tokp, g["token_probs"], 1e-7), _close(act, g["action_scores"], 2e-6)
    out7 = m(**b)
    _close(torch.stack([x.detach().double() for x in out7]), g["tuple7"], 1e-5)
    out7[0].backward()
    grads = {n: p.grad for n, p in m.named_parameters()}
    names = list(g["grad_names"])
    assert names == sorted(grads)
    for i, n in enumerate(names):
        w = g["grad_%03d" % i]
        _close(grads[n], w, 1e-6 * max(1.0, float(np.abs(w).max())), n)

    B, T, R = 3, 20, 17
    with torch.no_grad():
        def two(tag, out, atol=2e-6):
            for i in range(2):
                _close(out[i], g["%s_%d" % (tag, i)], atol, tag)

        two("headmask1d", m.bert(head_mask=torch.from_numpy(g["in_head_mask_1d"]), **tk))
        two("headmask2d", m.bert(head_mask=torch.from_numpy(g["in_head_mask_2d"]), **tk))
        two("mask3d", m.bert(b["input_ids"], attention_mask=torch.from_numpy(g["in_mask3d"]), img_feats=b["img_feats"],
                             img_location_embeddings=b["img_location_embeddings"]))
        two("maskfloat", m.bert(b["input_ids"], attention_mask=b["attention_mask"].float() * 0.5 + 0.25,
                                img_feats=b["img_feats"], img_location_embeddings=b["img_location_embeddings"]))
        two("masku8", m.bert(b["input_ids"], attention_mask=torch.from_numpy(g["in_mask_u8"])), atol=2e-5)
        hist = [torch.from_numpy(g["in_history_%d" % i]) for i in range(cfg.num_hidden_layers)]
        mh = torch.cat([torch.ones(B, hist[0].shape[1], dtype=torch.long), b["attention_mask"][:, :T]], 1)
        two("history", m.bert(b["input_ids"], attention_mask=mh, encoder_history_states=hist))
        tt = (torch.arange(T)[None, :] >= 9).long().expand(B, T).contiguous()
        pid = torch.arange(T - 1, -1, -1)[None, :].expand(B, T).contiguous()
        two("types_positions", m.bert(b["input_ids"], token_type_ids=tt, position_ids=pid,
                                      attention_mask=b["attention_mask"][:, :T]))
        two("text_only", m(b["input_ids"], attention_mask=b["attention_mask"], img_feats=b["img_feats"],
                           img_location_embeddings=b["img_location_embeddings"], text_only=True))

        def seven(tag, **over):
            bb = dict(b)
            bb.update(over)
            _close(torch.stack([x.double() for x in m(**bb)]), g[tag], 1e-5, tag)

        seven("tuple7_no_labels", labels=torch.full_like(b["labels"], -1))
        seven("tuple7_no_token_labels", token_labels=torch.full_like(b["token_labels"], -1))
        na = b["next_action"].clone()
        na[1] = -1
        seven("tuple7_one_action_ignored", next_action=na)
        seven("tuple7_all_actions_ignored", next_action=torch.full_like(na, -1))
        assert np.isnan(g["tuple7_no_labels"][1]) and np.isnan(g["tuple7_no_labels"][0])   # the reference's own NaN corner

        cfg2 = mini_config(output_hidden_states=True, output_attentions=True, use_img_layernorm=True, img_layer_norm_eps=1e-5)
        m2 = _oracle(BertImgModelwithLocationEmbeds, cfg2, 4, 0.05)
        out = m2(**tk)
        assert len(out) == 4
        _close(out[0], g["imgln_seq"], 2e-6), _close(out[1], g["imgln_pooled"], 2e-6)
        for i, h in enumerate(out[2]):
            _close(h, g["hidden_states_%d" % i], 2e-6)
        for i, a in enumerate(out[3]):
            _close(a, g["attentions_%d" % i], 1e-7)


def check_base_fixture(g, seq, pooled, scores, tokp, act, out7, atol, scores_atol):
    st = int(g["seq_stride"][0])
    _close(seq[:, ::st, ::31], g["sequence_output_slice"], atol)
    _close(pooled, g["pooled_output"], atol)
    _close(scores[:, ::st, ::1009], g["prediction_scores_slice"], scores_atol)
    _close(tokp[:, ::st, ::97], g["token_probs_slice"], 1e-6)
    _close(act, g["action_scores"], atol)
    _close(np.array([float(x) for x in out7]), g["tuple7"], 2e-4)


def grad_slice(t, n=2048):
    flat = t.detach().reshape(-1)
    step = max(1, flat.numel() // n)
    return flat[::step][:n]


def check_input_checksums(g, b):
    for k, v in b.items():
        if v.dtype.is_floating_point:
            c = g["in_%s_checksum" % k]
            assert abs(float(v.double().sum()) - c[0]) <= 1e-9 * abs(c[0]) + 1e-9 and float(v.double().abs().max()) == c[1], k
        else:
            assert np.array_equal(g["in_" + k], v.numpy()), k


@pytest.mark.parametrize("fname,B,T,R,seed", [("ref_base_cfg0.npz", 2, 128, 100, 1234),
                                              ("ref_base_long.npz", 2, 512, 144, 77),
                                              ("ref_shipped_s767.npz", 2, 511, 256, 767)])
def test_base_fixtures_forward_and_gradients(fname, B, T, R, seed):
    """configs[0] (B = 2, 128 + 100), configs[4]'s S = 656 and the reference's shipped pretrain shape 511 + 256:
    outputs, 7-tuple and every parameter's gradient (norm + a 2048-element strided slice) of the 12-layer model."""
    from oracle.modeling import PreTrainOscar
    from visitron_amd.config import BertConfig
    from visitron_amd.synth import make_batch

    torch.set_num_threads(min(8, os.cpu_count() or 1))
    g = _load(fname)
    cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    b = make_batch(cfg, B, text_len=T, region_len=R, seed=seed)
    check_input_checksums(g, b)
    m = _oracle(PreTrainOscar, cfg, 0, 0.03)
    with torch.no_grad():
        seq, pooled = m.bert(**{k: b[k] for k in TRUNK_KEYS})[:2]
        scores, tokp, act = m.heads(seq, pooled)
    out7 = m(**b)
    check_base_fixture(g, seq, pooled, scores, tokp, act, out7, 2e-5, 1e-4)
    del scores, tokp
    out7[0].backward()
    grads = {n: p.grad for n, p in m.named_parameters()}
    names = list(g["grad_names"])
    assert names == sorted(grads)
    off = g["grad_slice_offsets"]
    for i, n in enumerate(names):
        want = g["grad_slices"][off[i]:off[i + 1]]
        got = grad_slice(grads[n]).numpy()
        scale = max(float(np.abs(want).max()), 1e-6)
        np.testing.assert_allclose(got, want, atol=2e-5 * scale + 1e-9, rtol=1e-4, err_msg=n)
        assert abs(float(grads[n].double().norm()) - g["grad_norms"][i]) <= 1e-4 * g["grad_norms"][i] + 1e-9, n


def test_text_only_shipped_rollout_shape():
    """Text-only T = 511, B = 8 (agent_models.py:270-275) with the uint8 ~mask the rollout caller passes, and with 0/1."""
    from oracle.modeling import PreTrainOscar
    from visitron_amd.config import BertConfig
    from visitron_amd.synth import make_batch

    torch.set_num_threads(min(8, os.cpu_count() or 1))
    g = _load("ref_text511.npz")
    cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    b = make_batch(cfg, 8, text_len=511, region_len=0, seed=511, with_labels=False)
    check_input_checksums(g, b)
    m = _oracle(PreTrainOscar, cfg, 0, 0.03)
    with torch.no_grad():
        a = m.bert(b["input_ids"], attention_mask=~(b["attention_mask"] == 0).byte())
        a01 = m.bert(b["input_ids"], attention_mask=b["attention_mask"])
    # the 254 / 255 mask puts biases of -2.5e6 on the scores: fp32 rounding of the sums is coarser there
    _close(a[0][:, ::37, ::31], g["sequence_output_slice"], 2e-4), _close(a[1], g["pooled_output"], 2e-4)
    _close(a01[0][:, ::37, ::31], g["sequence_output_01mask_slice"], 2e-5), _close(a01[1], g["pooled_output_01mask"], 2e-5)


class _Args:
    device = torch.device("cpu")


def test_rollout_modules():
    from oracle import rollout as orollout
    from oracle.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.config import BertConfig
    from visitron_amd.synth import deterministic_state_dict

    g = _load("ref_rollout.npz")
    h, ctx, mask = (torch.from_numpy(g[k]) for k in ("sda_h", "sda_ctx", "sda_mask"))
    att = _oracle(lambda _: orollout.SoftDotAttention(128, 132), None, 7, 0.08)
    with torch.no_grad():
        for mi, m in enumerate((None, mask)):
            for tilde in (True, False):
                for prob in (True, False):
                    out = att(h, ctx, None if m is None else m.clone(), output_tilde=tilde, output_prob=prob)
                    for i in range(2):
                        want = g["sda_m%d_t%d_p%d_%d" % (mi, tilde, prob, i)]
                        fin = np.isfinite(want)
                        assert np.array_equal(fin, np.isfinite(out[i].numpy()))
                        _close(torch.where(torch.from_numpy(fin), out[i], torch.zeros(())), np.where(fin, want, 0), 2e-6)
        dec = _oracle(lambda _: orollout.AttnDecoderLSTM(4, 64, 128, 0.5, feature_size=132), None, 8, 0.06)
        ins = {k[len("dec_in_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("dec_in_")}
        out = dec(**ins)
        for i, n in enumerate(("h_1", "c_1", "logit", "h_tilde")):
            _close(out[i], g["dec_" + n], 5e-6, n)
    cfg = BertConfig(num_hidden_layers=2, vocab_size=600, max_position_embeddings=64, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0, detector_classes=40)
    bert = _oracle(BertImgModelwithLocationEmbeds, cfg, 9, 0.03)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc = orollout.OscarEncoder(_Args(), bert, 128, 96, 0.5).eval()
    sd = deterministic_state_dict(enc, seed=10, weight_std=0.03)
    sd.update({k: v for k, v in enc.state_dict().items() if k.startswith("bert.")})
    enc.load_state_dict(sd)
    ids, lengths = torch.from_numpy(g["enc_in_ids"]), [int(x) for x in g["enc_in_lengths"]]
    pad = torch.zeros(ids.shape, dtype=torch.bool)
    for i, n in enumerate(lengths):
        pad[i, n:] = True
    for tag, m in (("bool", pad), ("u8", pad.byte())):
        out = enc(ids, lengths, m)
        for i, n in enumerate(("ctx", "decoder_init", "c_t")):
            _close(out[i], g["enc_%s_%s" % (tag, n)], 2e-4 if tag == "u8" else 2e-5, n)
    (out[0].sum() + out[1].sum() + out[2].sum()).backward()
    gr = {n: p.grad for n, p in enc.named_parameters() if p.grad is not None}
    names = list(g["enc_grad_names"])
    assert names == sorted(gr)
    off = g["enc_grad_slice_offsets"]
    for i, n in enumerate(names):
        want = g["enc_grad_slices"][off[i]:off[i + 1]]
        scale = max(float(np.abs(want).max()), 1e-6)
        np.testing.assert_allclose(grad_slice(gr[n]).numpy(), want, atol=5e-4 * scale, rtol=1e-3, err_msg=n)


def replay_data_items(g):
    """The draws of make_golden_from_reference.case_data, reproduced from the same seeded generator in the same order."""
    rec = json.loads(str(g["items_json"]))
    gen = torch.Generator().manual_seed(33)
    T = 40
    for mtp in (False, True):
        for it in range(6):
            tag = "mtp%d_item%d" % (mtp, it)
            ids = torch.randint(200, 997, (T,), generator=gen)
            ids[0] = 101
            n_real = T - 3 * it
            ids[n_real - 1] = 102
            ids[n_real:] = 0
            tc = torch.full((T,), -1, dtype=torch.long)
            tc[3 + it] = 7 + it
            tc[10] = 2
            u = [torch.rand(T, generator=gen) for _ in range(3)]
            words = torch.randint(997, (T,), generator=gen)
            feats = torch.cat([torch.rand(5, 2054, generator=gen) for _ in range(36)], 0)
            assert np.array_equal(g[tag + "_in_ids"], ids.numpy()) and np.array_equal(g[tag + "_in_token_classes"], tc.numpy())
            assert np.array_equal(g[tag + "_in_u"], torch.stack(u).numpy()) and np.array_equal(g[tag + "_in_words"], words.numpy())
            assert np.array_equal(g[tag + "_in_feats_first"], feats[:5, :8].numpy())
            yield tag, mtp, rec[tag], ids, tc, u, words, feats


def test_pretrain_input_preparation_items():
    """build_viewpoint_loc_embedding and PretrainDataset._preprocess_item (mask tokens -> region features -> padding /
    truncation -> labels) on the recorded draws: integer outputs equal, float outputs equal (they are gathers)."""
    from oracle import data as odata

    g = _load("ref_data.npz")
    assert np.array_equal(g["loc_tables"], np.stack(odata.STATIC)) and np.array_equal(g["static_tables"], g["loc_tables"])
    view_ids = [v for v in range(36) for _ in range(5)]
    n = 0
    for tag, mtp, meta, ids, tc, u, words, feats in replay_data_items(g):
        w_in, w_lab, w_att = odata.mask_tokens_item(ids.clone(), {0, 100, 101, 102, 103}, 0, 103, 0.15, tc if mtp else None,
                                                    u[0], u[1], u[2], words)
        out = odata.preprocess_item_tail(w_in, w_lab, w_att, feats.numpy(), view_ids, meta["current_view_index"],
                                         meta["target_rel_view_index"], meta["max_img"], token_classes=tc if mtp else None,
                                         no_action_grounding=meta["no_action_grounding"])
        for k in ("input_ids", "labels", "attention_mask", "img_location_embeddings", "next_action"):
            assert np.array_equal(np.asarray(out[k]), g["%s_%s" % (tag, k)]), (tag, k)
        if mtp:
            assert np.array_equal(out["token_labels"].numpy(), g[tag + "_token_labels"]), tag
        else:
            assert out["token_labels"] is None and (tag + "_token_labels") not in g.files
        c = g[tag + "_img_feats"]
        assert out["img_feats"].shape[0] == int(c[1])     # == max_img, except max_img = 0: `x[-0:]` keeps all 180 (:659-662)
        assert abs(float(out["img_feats"].double().sum()) - c[0]) <= 1e-9 * abs(c[0])
        n += 1
    assert n == 12


def test_oscar_encoder_reverse_input_and_stacked_lstm():
    """Round 6: the constructor arguments no reference caller sets -- reverse_input (agent_models.py:277-282; a uint8 mask
    indexes as 'every position') and num_layers = 2 / bidirectional -- against the reference's own outputs and gradients
    (ref_rollout2.npz, written by running agent_models.OscarEncoder with those arguments)."""
    from oracle import rollout as orollout
    from oracle.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.config import BertConfig
    from visitron_amd.synth import deterministic_state_dict

    g = _load("ref_rollout2.npz")
    cfg = BertConfig(num_hidden_layers=2, vocab_size=600, max_position_embeddings=64, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0, detector_classes=40)
    bert = _oracle(BertImgModelwithLocationEmbeds, cfg, 9, 0.03)
    ids, lengths = torch.from_numpy(g["enc_in_ids"]), [int(x) for x in g["enc_in_lengths"]]
    pad = torch.zeros(ids.shape, dtype=torch.bool)
    for i, n in enumerate(lengths):
        pad[i, n:] = True
    for name, kw in (("rev", dict(reverse_input=True)), ("l2", dict(num_layers=2)),
                     ("l2bi_rev", dict(num_layers=2, bidirectional=True, reverse_input=True))):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            enc = orollout.OscarEncoder(_Args(), bert, 128, 96, 0.5, **kw).eval()
            sd = deterministic_state_dict(enc, seed=12, weight_std=0.03)
            sd.update({k: v for k, v in enc.state_dict().items() if k.startswith("bert.")})
            enc.load_state_dict(sd)
            for tag, m in (("bool", pad), ("u8", pad.byte())):
                with torch.no_grad():
                    out = enc(ids, lengths, m)
                for i, n in enumerate(("ctx", "decoder_init", "c_t")):
                    _close(out[i], g["enc_%s_%s_%s" % (name, tag, n)], 2e-4 if tag == "u8" else 2e-5, "%s %s %s" % (name, tag, n))
            enc.zero_grad()
            out = enc(ids, lengths, pad)
        (out[0].sum() + out[1].sum() + out[2].sum()).backward()
        gr = {n: p.grad for n, p in enc.named_parameters() if p.grad is not None}
        names = list(g["enc_%s_grad_names" % name])
        assert names == sorted(gr)
        off = g["enc_%s_grad_slice_offsets" % name]
        for i, n in enumerate(names):
            want = g["enc_%s_grad_slices" % name][off[i]:off[i + 1]]
            scale = max(float(np.abs(want).max()), 1e-6)
            np.testing.assert_allclose(grad_slice(gr[n]).numpy(), want, atol=5e-4 * scale, rtol=1e-3, err_msg=name + " " + n)
