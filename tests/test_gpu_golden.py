"""GPU: the HIP path against the committed golden fixtures (no oracle call at test time).  The fixtures are the outputs
of the REFERENCE's own source, executed in the build container by tests/golden/make_golden_from_reference.py
(oscar/modeling_bert.py, tasks/viewpoint_select/encoder.py over a flagged stand-in for the un-vendored
pytorch-transformers blocks) -- not the oracle's."""
import os

import numpy as np
import pytest
import torch

from helpers import check_close

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

    g = np.load(os.path.join(GOLD, "ref_mini.npz"))
    cfg = mini_config()
    m = _product(cfg, 3, 0.05, dev)
    b = {k: torch.from_numpy(g["in_" + k]).to(dev) for k in ("input_ids", "attention_mask", "img_feats", "img_location_embeddings",
                                                               "labels", "token_labels", "next_action")}
    with torch.no_grad():
        outs, pooled, _, B, S = m.bert.run_trunk(b["input_ids"], attention_mask=b["attention_mask"],
                                                 img_feats=b["img_feats"], img_location_embeddings=b["img_location_embeddings"])
        scores, tokp, act = m.head_outputs(outs[-1], pooled)
        out7 = m(**b)
    check_close("golden mini sequence_output", outs[-1], g["sequence_output"], TOL)
    check_close("golden mini pooled_output", pooled, g["pooled_output"], TOL)
    check_close("golden mini prediction_scores", scores, g["prediction_scores"], TOL)
    check_close("golden mini token_probs", tokp, g["token_probs"], TOL)
    check_close("golden mini action_scores", act, g["action_scores"], TOL)
    for i in range(4):
        check_close("golden mini tuple7[%d]" % i, float(out7[i]), float(g["tuple7"][i]), TOL)


def test_base_cfg1_fixture(dev):
    from visitron_amd.config import BertConfig
    from visitron_amd.synth import make_batch

    g = np.load(os.path.join(GOLD, "ref_base_cfg0.npz"))
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
    check_close("golden base cfg1 sequence_output slice", seq[:, ::19, ::31], g["sequence_output_slice"], TOL)
    check_close("golden base cfg1 pooled_output", pooled, g["pooled_output"], TOL)
    check_close("golden base cfg1 prediction_scores slice", scores.float().cpu().view(B, S, -1)[:, ::19, ::1009],
                g["prediction_scores_slice"], TOL)
    check_close("golden base cfg1 token_probs slice", tokp.float().cpu().view(B, S, -1)[:, ::19, ::97], g["token_probs_slice"], TOL)
    check_close("golden base cfg1 action_scores", act, g["action_scores"], TOL)
    for i in range(4):
        check_close("golden base cfg1 tuple7[%d]" % i, float(out7[i]), float(g["tuple7"][i]), TOL)
