"""GPU: the HIP path against the committed golden fixtures (no oracle call at test time)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 5e-2  # bf16 path tolerance, BASELINE.json north_star


def _product(cfg, seed, std, dev):
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import deterministic_state_dict

    m = PreTrainOscar(cfg).eval()
    m.load_state_dict(deterministic_state_dict(m, seed=seed, weight_std=std))
    m.tie_weights()
    return m.to(dev)


def test_mini_fixture(dev):
    from visitron_amd.config import mini_config

    g = np.load(os.path.join(GOLD, "mini_pretrain.npz"))
    cfg = mini_config()
    m = _product(cfg, 3, 0.05, dev)
    b = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("in_")}
    with torch.no_grad():
        outs, pooled, _, B, S = m.bert.run_trunk(b["input_ids"], attention_mask=b["attention_mask"],
                                                 img_feats=b["img_feats"], img_location_embeddings=b["img_location_embeddings"])
        scores, tokp, act = m.head_outputs(outs[-1], pooled)
        out7 = m(**b)
    d = lambda t, ref: float(np.abs(t.float().cpu().numpy().reshape(ref.shape) - ref).max())
    assert d(outs[-1], g["sequence_output"]) < TOL
    assert d(pooled, g["pooled_output"]) < TOL
    assert d(scores, g["prediction_scores"]) < TOL
    assert d(tokp, g["token_probs"]) < TOL
    assert d(act, g["action_scores"]) < TOL
    for i in range(4):
        assert abs(float(out7[i]) - g["tuple7"][i]) < TOL


def test_base_cfg1_fixture(dev):
    from visitron_amd.config import BertConfig
    from visitron_amd.synth import make_batch

    g = np.load(os.path.join(GOLD, "base_cfg1.npz"))
    cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    b = make_batch(cfg, 2, seed=1234)
    assert np.array_equal(g["in_input_ids"], b["input_ids"].numpy())
    b = {k: v.to(dev) for k, v in b.items()}
    m = _product(cfg, 0, 0.03, dev)
    with torch.no_grad():
        outs, pooled, _, B, S = m.bert.run_trunk(b["input_ids"], attention_mask=b["attention_mask"],
                                                 img_feats=b["img_feats"], img_location_embeddings=b["img_location_embeddings"])
        scores, tokp, act = m.head_outputs(outs[-1], pooled)
        out7 = m(**b)
    seq = outs[-1].float().cpu().view(B, S, -1)
    assert float(np.abs(seq[:, ::19, ::31].numpy() - g["sequence_output_slice"]).max()) < 2 * TOL
    assert float(np.abs(pooled.cpu().numpy() - g["pooled_output"]).max()) < TOL
    sc = scores.float().cpu().view(B, S, -1)[:, ::23, ::1009].numpy()
    assert float(np.abs(sc - g["prediction_scores_slice"]).max()) < TOL * float(g["prediction_scores_absmax"][0])
    assert float(np.abs(act.cpu().numpy() - g["action_scores"]).max()) < TOL
    for i in range(4):
        assert abs(float(out7[i]) - g["tuple7"][i]) < TOL * max(1.0, abs(g["tuple7"][i]))
