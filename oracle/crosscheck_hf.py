"""Pin the oracle's un-vendored blocks against an INDEPENDENT implementation.

TEST INFRASTRUCTURE.  The reference's encoder cannot be imported here (its
``transformers.pytorch_transformers`` submodule is empty), so the strongest
available pin is the third-party ``transformers`` package that happens to be
installed in this image (5.x; eager attention).  It is NOT the reference, and
it does not exist on the GPU box -- this script runs in the build container
only and writes a small report (``tests/golden/hf_crosscheck.json``) which the
CPU test-suite re-checks when ``transformers`` is importable.

Checked (same weights via state_dict, fp64, dropout 0):
  encoder stack      oracle.CaptionBertEncoder   vs  hf BertEncoder
  text embeddings    oracle.BertEmbeddings       vs  hf BertEmbeddings
  pooler             oracle.BertPooler           vs  hf BertPooler
  MLM head           oracle.BertOnlyMLMHead      vs  hf BertOnlyMLMHead
  gradients of the encoder stack w.r.t. every parameter (fp64)

Usage:  python -m oracle.crosscheck_hf [--write]
"""
import argparse
import json
import os
import sys

import torch

from .bert_blocks import BertEmbeddings, BertOnlyMLMHead, BertPooler
from .config import BASE, make_config
from .modeling import CaptionBertEncoder

REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hf_crosscheck.json")


def _hf(cfg):
    from transformers import BertConfig as HFConfig

    c = HFConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
        hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
        max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
        layer_norm_eps=cfg.layer_norm_eps,
    )
    c._attn_implementation = "eager"
    return c


def _maxdiff(a, b):
    return float((a - b).abs().max())


def run(dtype=torch.float64, seed=0):
    from transformers.models.bert import modeling_bert as hf

    torch.manual_seed(seed)
    cfg = make_config(BASE, num_hidden_layers=3, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hcfg = _hf(cfg)
    B, S = 2, 37
    res = {}

    # --- encoder stack ---
    mine = CaptionBertEncoder(cfg).to(dtype).eval()
    for p in mine.parameters():
        p.data.normal_(0, 0.05)
    theirs = hf.BertEncoder(hcfg).to(dtype).eval()
    missing = theirs.load_state_dict(mine.state_dict(), strict=True)
    x = torch.randn(B, S, cfg.hidden_size, dtype=dtype)
    keep = (torch.rand(B, S) > 0.2).to(dtype)
    keep[:, 0] = 1
    ext = (1.0 - keep[:, None, None, :]) * -10000.0
    xm = x.clone().requires_grad_(True)
    xt = x.clone().requires_grad_(True)
    ym = mine(xm, ext, head_mask=[None] * cfg.num_hidden_layers)[0]
    yt = theirs(xt, attention_mask=ext)
    yt = yt[0] if isinstance(yt, tuple) else yt.last_hidden_state
    res["encoder_fwd_maxabs"] = _maxdiff(ym, yt)
    w = torch.randn_like(ym)
    (ym * w).sum().backward()
    (yt * w).sum().backward()
    res["encoder_dx_maxabs"] = _maxdiff(xm.grad, xt.grad)
    gm = dict(mine.named_parameters())
    gt = dict(theirs.named_parameters())
    res["encoder_dparam_maxabs"] = max(_maxdiff(gm[k].grad, gt[k].grad) for k in gm)
    res["encoder_param_names_equal"] = sorted(gm) == sorted(gt)

    # --- embeddings ---
    me = BertEmbeddings(cfg).to(dtype).eval()
    for p in me.parameters():
        p.data.normal_(0, 0.05)
    te = hf.BertEmbeddings(hcfg).to(dtype).eval()
    sd = {k: v for k, v in me.state_dict().items()}
    te.load_state_dict(sd, strict=False)
    ids = torch.randint(0, cfg.vocab_size, (B, S))
    tt = torch.randint(0, cfg.type_vocab_size, (B, S))
    res["embeddings_maxabs"] = _maxdiff(me(ids, tt), te(input_ids=ids, token_type_ids=tt))

    # --- pooler ---
    mp = BertPooler(cfg).to(dtype)
    tp = hf.BertPooler(hcfg).to(dtype)
    tp.load_state_dict(mp.state_dict())
    res["pooler_maxabs"] = _maxdiff(mp(x), tp(x))

    # --- MLM head ---
    mh = BertOnlyMLMHead(cfg).to(dtype)
    for p in mh.parameters():
        p.data.normal_(0, 0.05)
    th = hf.BertOnlyMLMHead(hcfg).to(dtype)
    sd = mh.state_dict()
    tsd = th.state_dict()
    for k in tsd:
        if k in sd:
            tsd[k] = sd[k]
        elif k.endswith("decoder.bias"):
            tsd[k] = sd["predictions.bias"]
    th.load_state_dict(tsd)
    res["mlmhead_maxabs"] = _maxdiff(mh(x), th(x))
    res.update(run_optim(dtype, seed))
    res["transformers_version"] = __import__("transformers").__version__
    res["torch_version"] = torch.__version__
    return res


def run_optim(dtype=torch.float64, seed=0):
    """AdamW and the linear schedule of oracle/optim.py against two implementations by other hands.

    * ``torch.optim.AdamW(eps=0, weight_decay=0)``: at eps = 0 the pytorch-transformers rule (eps on the UN-corrected sqrt(v),
      corrections folded into the step size) and torch's (eps on the corrected one) are the same function -- ten steps pin
      the two moments and both bias corrections.  eps > 0 and the decay placement differ BY DESIGN (oracle/optim.py header)
      and stay with the hand KAT (tests/test_oracle_kat.py); here their difference from torch's rule is recorded with its
      closed form: one step from zero moments moves p by lr * g / (|g| + eps_pt) against lr * g / (|g| + eps_torch), and a
      decayed step differs by lr * wd * (p_moved - p_before).
    * ``transformers.get_linear_schedule_with_warmup`` / ``get_constant_schedule_with_warmup``: the published successors of
      WarmupLinearSchedule / WarmupConstantSchedule, same lambda by definition."""
    from .optim import AdamW, WarmupConstantSchedule, WarmupLinearSchedule

    res = {}
    gen = torch.Generator().manual_seed(seed + 17)
    shapes = [(7, 5), (11,), (3, 4, 2)]
    p0 = [torch.randn(s, generator=gen, dtype=dtype) for s in shapes]
    grads = [[torch.randn(s, generator=gen, dtype=dtype) * (0.1 + 0.3 * t) for s in shapes] for t in range(10)]
    mine = [p.clone().requires_grad_(True) for p in p0]
    theirs = [p.clone().requires_grad_(True) for p in p0]
    om = AdamW(mine, lr=3e-3, betas=(0.9, 0.999), eps=0.0, weight_decay=0.0)
    ot = torch.optim.AdamW(theirs, lr=3e-3, betas=(0.9, 0.999), eps=0.0, weight_decay=0.0)
    worst = 0.0
    for t in range(10):
        for a_, b_, g_ in zip(mine, theirs, grads[t]):
            a_.grad = g_.clone()
            b_.grad = g_.clone()
        om.step()
        ot.step()
        worst = max(worst, max(_maxdiff(a_.detach(), b_.detach()) for a_, b_ in zip(mine, theirs)))
    res["adamw_eps0_10steps_vs_torch_maxabs"] = worst
    res["adamw_eps0_moments_vs_torch_maxabs"] = max(
        max(_maxdiff(om.state[a_]["exp_avg"], ot.state[b_]["exp_avg"]),
            _maxdiff(om.state[a_]["exp_avg_sq"], ot.state[b_]["exp_avg_sq"])) for a_, b_ in zip(mine, theirs))
    # eps placement: first step from zero moments, closed form of the pytorch-transformers rule
    eps, lr = 1e-3, 1e-2
    q = p0[0].clone().requires_grad_(True)
    q.grad = grads[0][0].clone()
    AdamW([q], lr=lr, eps=eps, weight_decay=0.0).step()
    g0 = grads[0][0]
    b1, b2 = 0.9, 0.999
    want = p0[0] - lr * (1 - b2) ** 0.5 / (1 - b1) * ((1 - b1) * g0) / (((1 - b2) * g0 * g0).sqrt() + eps)
    res["adamw_eps_on_uncorrected_sqrt_closed_form_maxabs"] = _maxdiff(q.detach(), want)
    # decay placement: after the move, on the moved parameter
    q2 = p0[0].clone().requires_grad_(True)
    q2.grad = grads[0][0].clone()
    AdamW([q2], lr=lr, eps=eps, weight_decay=0.05).step()
    res["adamw_decay_after_move_closed_form_maxabs"] = _maxdiff(q2.detach(), want * (1.0 - lr * 0.05))
    # schedules
    from transformers.optimization import get_constant_schedule_with_warmup, get_linear_schedule_with_warmup

    def lrs(make, n=60):
        w = torch.zeros(1, requires_grad=True)
        opt = torch.optim.SGD([w], lr=5e-5)
        sch = make(opt)
        out = []
        for _ in range(n):
            out.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        return torch.tensor(out, dtype=torch.float64)

    for name, ws, tt in (("w0", 0, 50), ("w7", 7, 50), ("w10_short", 10, 30)):
        a_ = lrs(lambda o: WarmupLinearSchedule(o, warmup_steps=ws, t_total=tt))
        b_ = lrs(lambda o: get_linear_schedule_with_warmup(o, num_warmup_steps=ws, num_training_steps=tt))
        res["linear_schedule_%s_vs_hf_maxabs" % name] = _maxdiff(a_, b_)
    a_ = lrs(lambda o: WarmupConstantSchedule(o, warmup_steps=7))
    b_ = lrs(lambda o: get_constant_schedule_with_warmup(o, num_warmup_steps=7))
    res["constant_schedule_w7_vs_hf_maxabs"] = _maxdiff(a_, b_)
    return res


TOL = 1e-9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    a = ap.parse_args()
    r = run()
    print(json.dumps(r, indent=1))
    bad = [k for k, v in r.items() if k.endswith("maxabs") and not v < TOL]
    if not r["encoder_param_names_equal"]:
        bad.append("encoder_param_names_equal")
    if a.write:
        with open(REPORT, "w") as f:
            json.dump(r, f, indent=1, sort_keys=True)
            f.write("\n")
    if bad:
        print("CROSSCHECK FAILED:", bad)
        sys.exit(1)
    print("crosscheck ok (tol %g)" % TOL)


if __name__ == "__main__":
    main()
