"""Generate the golden fixtures from the CPU oracle (run in the build container):

    python tests/golden/make_golden.py

SUPERSEDED for outputs and gradients by make_golden_from_reference.py (ref_*.npz: the reference's
own source, executed).  What is left here is what the reference cannot give -- the AdamW step
deltas of the un-vendored pytorch-transformers optimizer, restated in oracle/optim.py -- with
weights from the hash-based deterministic generator (visitron_amd.synth.deterministic_state_dict), i.e. no RNG state is involved.
Fixtures hold inputs and expected outputs only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.modeling import PreTrainOscar  # noqa: E402
from oracle.optim import AdamW, grouped_parameters  # noqa: E402
from visitron_amd.config import mini_config  # noqa: E402
from visitron_amd.synth import deterministic_state_dict, make_batch  # noqa: E402

TRUNK_KEYS = ("input_ids", "attention_mask", "img_feats", "img_location_embeddings")


def run(cfg, batch, seed, weight_std):
    torch.set_num_threads(8)
    m = PreTrainOscar(cfg).eval()
    m.load_state_dict(deterministic_state_dict(m, seed=seed, weight_std=weight_std))
    with torch.no_grad():
        seq, pooled = m.bert(**{k: batch[k] for k in TRUNK_KEYS})[:2]
        scores, tokp, act = m.heads(seq, pooled)
    out7 = m(**batch)
    out7[0].backward()
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    return m, seq, pooled, scores, tokp, act, [float(x) for x in out7], grads


def mini():
    cfg = mini_config()
    b = make_batch(cfg, 3, text_len=20, region_len=17, seed=11)
    m, seq, pooled, scores, tokp, act, out7, grads = run(cfg, b, seed=3, weight_std=0.05)
    # one AdamW step with the pretrain settings (pretrain.py:109-130: lr 5e-5, eps 1e-8, wd 0.05)
    opt = AdamW(grouped_parameters(m, 0.05), lr=5e-5, eps=1e-8)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt.step()
    after = dict(m.named_parameters())
    names = sorted(grads)
    q0 = "bert.encoder.layer.0.attention.self.query.weight"
    np.savez_compressed(
        os.path.join(HERE, "mini_pretrain.npz"),
        **{"in_" + k: v.numpy() for k, v in b.items()},
        sequence_output=seq.numpy(), pooled_output=pooled.numpy(), prediction_scores=scores.numpy(),
        token_probs=tokp.numpy(), action_scores=act.numpy(), tuple7=np.array(out7, dtype=np.float64),
        grad_names=np.array(names),
        grad_norms=np.array([float(grads[n].norm()) for n in names]),
        grad_query0=grads[q0].numpy(),
        grad_ln_last=grads["bert.encoder.layer.1.output.LayerNorm.weight"].numpy(),
        grad_img=grads["bert.img_embedding.weight"].numpy(),
        adamw_delta_norms=np.array([float((after[n].detach() - before[n]).norm()) for n in names]),
        adamw_delta_query0=(after[q0].detach() - before[q0]).numpy(),
    )
    print("mini 7-tuple", out7)


if __name__ == "__main__":
    mini()
