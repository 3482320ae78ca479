"""Round 4 GPU tests: the fp16 copies of the residual stream in the seven-launch (training) layer -- GEMM epilogues that
write the pre-LayerNorm sum as fp16 and read an fp16 residual, LayerNorm over fp16 rows with its second (fp16) output, its
backward -- against fp32 / fp64 arithmetic on the same operands (BertSelfOutput / BertOutput, called at
oscar/modeling_bert.py:94,120; their autograd inside loss.backward(), tasks/viewpoint_select/pretrain.py:191)."""
import pytest
import torch

from helpers import maxabs

BF16 = torch.bfloat16

pytestmark = pytest.mark.gpu
F16 = torch.float16


def _rand(shape, g, std=1.0):
    return torch.randn(shape, generator=g) * std


@pytest.mark.parametrize("variant", [1, 14, 11, 15, 16, 18, 19, 22, 23, 9])
@pytest.mark.parametrize("M,N,K", [(1000, 768, 768), (777, 768, 3072), (300, 200, 128)])
def test_linear_fp16_residual_in_and_fp16_sum_out(dev, M, N, K, variant):
    """C (fp16, saturating) = A W^T + b + R with R fp16: every kernel variant's epilogue (variant 9's grouped epilogue is
    bf16-only: the library substitutes variant 1), full tiles and tails; against the fp32 product on the same operands."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M + N + K + variant)
    a = _rand((M, K), g).to(BF16)
    w = _rand((N, K), g, 0.05).to(BF16)
    b = _rand((N,), g, 0.1)
    r = (_rand((M, N), g) * 3.0).to(F16)
    want = a.float() @ w.float().t() + b + r.float()
    ops.set_gemm_variant(variant)
    try:
        out = torch.empty((M, N), dtype=F16, device=dev)
        ops.linear(a.to(dev), w.to(dev), b.to(dev), residual=r.to(dev), out=out)
        # bf16 residual with fp16 output, and fp16 residual with bf16 output: the two flags are independent
        out2 = torch.empty((M, N), dtype=F16, device=dev)
        ops.linear(a.to(dev), w.to(dev), b.to(dev), residual=r.to(BF16).to(dev), out=out2)
        out3 = ops.linear(a.to(dev), w.to(dev), b.to(dev), residual=r.to(dev))
        torch.cuda.synchronize()
    finally:
        ops.set_gemm_variant(-1)
    scale = float(want.abs().max())
    assert maxabs(out, want) <= scale * 2.0 ** -10            # one fp16 rounding of the result
    assert maxabs(out2, a.float() @ w.float().t() + b + r.to(BF16).float()) <= scale * 2.0 ** -10
    assert out3.dtype == BF16 and maxabs(out3, want) <= scale * 2.0 ** -7


def test_linear_fp16_output_saturates_instead_of_overflowing(dev):
    from visitron_amd import ops

    a = torch.full((64, 64), 40.0).to(BF16)
    w = torch.full((64, 64), 40.0).to(BF16)        # every element 102 400 > 65 504
    out = torch.empty((64, 64), dtype=F16, device=dev)
    ops.linear(a.to(dev), w.to(dev), out=out)
    ops.linear(a.to(dev), (-w).to(dev), out=out[:32], M=32)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all())
    assert float(out[32:].float().min()) == 65504.0 and float(out[:32].float().max()) == -65504.0


@pytest.mark.parametrize("M,H", [(1000, 768), (37, 128), (515, 1024)])
def test_layernorm_over_fp16_rows_with_its_fp16_copy_and_backward(dev, M, H):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M + H)
    x = (_rand((M, H), g) * 4.0 + 0.5).to(F16)
    gam, bet = 1.0 + 0.1 * _rand((H,), g), 0.1 * _rand((H,), g)
    dy = _rand((M, H), g).to(BF16)
    eps = 1e-12
    out_h = torch.empty((M, H), dtype=F16, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    y = ops.layernorm(x.to(dev), gam.to(dev), bet.to(dev), eps, out_h=out_h, mean=mean, rstd=rstd)
    y_only = ops.layernorm(x.to(dev), gam.to(dev), bet.to(dev), eps)
    xd = x.double().requires_grad_(True)
    gd, bd = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    want = torch.nn.functional.layer_norm(xd, (H,), gd, bd, eps)
    want.backward(dy.double())
    dgam, dbet = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    dx_d = torch.empty((M, H), dtype=BF16, device=dev)
    drop = (0.25, 99, 5)
    dx = ops.layernorm_bwd(x.to(dev), dy.to(dev), gam.to(dev), eps, dgam, dbet, dx_dropped=dx_d, drop=drop)
    torch.cuda.synchronize()
    assert y.dtype == BF16 and torch.equal(y, y_only)
    wy = want.detach().float()
    assert maxabs(out_h, wy) <= float(wy.abs().max()) * 2.0 ** -10
    assert maxabs(y, wy) <= float(wy.abs().max()) * 2.0 ** -7
    assert maxabs(mean, x.double().mean(1).float()) <= 1e-5 and maxabs(rstd * x.double().std(1, unbiased=False).float().to(dev), torch.ones(M)) <= 1e-4
    wdx = xd.grad.float()
    assert maxabs(dx, wdx) <= float(wdx.abs().max()) * 2.0 ** -7 + 1e-6
    assert maxabs(dgam, gd.grad.float()) <= 2e-3 * (1.0 + float(gd.grad.abs().max()))
    assert maxabs(dbet, bd.grad.float()) <= 2e-3 * (1.0 + float(bd.grad.abs().max()))
    keep = ops.dropout_mask(M * H, drop, device=dev).view(M, H).float().cpu()
    assert maxabs(dx_d, wdx * keep / 0.75) <= float(wdx.abs().max()) / 0.75 * 2.0 ** -7 + 1e-6


def test_seven_launch_layer_keeps_the_stream_at_fp16_precision(dev):
    """The layer loop with the fp16 copies against the same loop with every tensor bf16 (VT_F16_STREAM=0 = table fields left
    NULL) and against the oracle: the fp16 stream must be the more accurate of the two on the base layer shape."""
    from oracle.modeling import BertImgModelwithLocationEmbeds as OTrunk
    from helpers import model_pair
    from visitron_amd import ops
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.synth import make_batch

    cfg = BertConfig(num_hidden_layers=4, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ref, prod = model_pair(OTrunk, BertImgModelwithLocationEmbeds, cfg, seed=5, device=dev, weight_std=0.03)
    keys = ("input_ids", "attention_mask", "img_feats", "img_location_embeddings")
    b = {k: v for k, v in make_batch(cfg, 2, seed=77).items() if k in keys}
    _to = lambda batch, d: {k: v.to(d) for k, v in batch.items()}
    prod.encoder.deferred_ln = False
    errs = {}
    with torch.no_grad():
        want = ref(**b)[0]
        for flag in (True, False):
            ops.F16_STREAM = flag
            prod.encoder._ws.clear()
            try:
                got = prod(**_to(b, dev))[0]
            finally:
                ops.F16_STREAM = True
            errs[flag] = maxabs(got, want)
    prod.encoder._ws.clear()
    print("seven-launch layer, 4 base layers: max error fp16 stream %.3e, bf16 stream %.3e" % (errs[True], errs[False]))
    assert errs[True] <= 5e-2 and errs[True] < errs[False]


def test_compacted_rows_run_the_deferred_layernorm_loop(dev):
    """run_trunk(keep=...) -- the rollout's eval forward on the rows that exist (agent_models.py:256-277) -- goes through the
    deferred-LayerNorm loop over compacted rows (vt_encoder_forward_ln_seq_bf16): against the oracle's masked forward at the
    kept positions (flat 5e-2), against the padded deferred loop (same arithmetic, masked keys instead of absent ones) and
    not less accurate than the seven-launch layer it replaces there; text + regions, ragged lengths down to one token."""
    from oracle.modeling import BertImgModelwithLocationEmbeds as OTrunk
    from helpers import model_pair
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.synth import make_batch

    cfg = BertConfig(num_hidden_layers=3, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ref, prod = model_pair(OTrunk, BertImgModelwithLocationEmbeds, cfg, seed=8, device=dev, weight_std=0.03)
    B, T, R = 5, 40, 12
    b = make_batch(cfg, B, text_len=T, region_len=R, seed=31)
    lens_t = torch.tensor([40, 33, 17, 2, 1])
    lens_r = torch.tensor([12, 0, 5, 12, 1])
    keep = torch.cat([torch.arange(T)[None, :] < lens_t[:, None], torch.arange(R)[None, :] < lens_r[:, None]], 1)
    mask = keep.to(torch.int64)
    args = dict(input_ids=b["input_ids"], img_feats=b["img_feats"], img_location_embeddings=b["img_location_embeddings"])
    with torch.no_grad():
        want, want_pooled = ref(attention_mask=mask, **args)[:2]
        dargs = {k: v.to(dev) for k, v in args.items()}
        outs, pooled, _, _, _ = prod.run_trunk(dargs["input_ids"], img_feats=dargs["img_feats"],
                                               img_location_embeddings=dargs["img_location_embeddings"], keep=keep.to(dev))
        lay = prod._last_layout
        got_c = outs[-1][:lay.rows].float().cpu()
        pooled_c = pooled.float().cpu()
        padded = prod(attention_mask=mask.to(dev), **dargs)
        prod.encoder.deferred_ln = False
        outs7, _, _, _, _ = prod.run_trunk(dargs["input_ids"], img_feats=dargs["img_feats"],
                                           img_location_embeddings=dargs["img_location_embeddings"], keep=keep.to(dev))
        got_7 = outs7[-1][:lay.rows].float().cpu()
        prod.encoder.deferred_ln = True
    assert lay.rows == int(keep.sum())
    H = cfg.hidden_size
    want_rows = want.reshape(-1, H)[keep.reshape(-1)]
    e_c = maxabs(got_c, want_rows)
    e_7 = maxabs(got_7, want_rows)
    e_p = maxabs(padded[0].reshape(-1, H).float().cpu()[keep.reshape(-1)], got_c)
    print("compacted deferred-LN loop: max error %.3e (seven-launch layer %.3e); against the padded deferred loop %.3e"
          % (e_c, e_7, e_p))
    assert e_c <= 5e-2 and e_c <= e_7 + 5e-3
    # absent keys against keys at -10000: the same stream arithmetic, one bf16 rounding step of the largest outputs apart
    assert e_p <= max(2e-2, float(want_rows.abs().max()) * 2.0 ** -7)
    assert maxabs(pooled_c, want_pooled) <= 5e-2


def test_two_rank_loss_curve_bf16_exchange_follows_fp32_exchange(dev, tmp_path):
    """The data-parallel gradient exchange moves a bf16 copy of the gradient slab by default where the reference's DDP
    (tasks/viewpoint_select/pretrain.py:96-102,191) reduces fp32 buckets: six optimizer steps under two ranks with either
    exchange, same kernels, same data -- the loss curves must stay together (the bf16 sum over ranks rounds each gradient to 8
    significant bits once) and both must fall."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "dp_curve_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=root)
    res = {}
    for comm, port in (("fp32", "29681"), ("bf16", "29683")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", port, script, str(tmp_path), comm]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
        res[comm] = torch.load(os.path.join(str(tmp_path), "curve_%s.pt" % comm))
    a, b = torch.tensor(res["fp32"]["curve"]), torch.tensor(res["bf16"]["curve"])
    print("two-rank loss curve, fp32 exchange:", [round(float(v), 4) for v in a[:, 0]])
    print("two-rank loss curve, bf16 exchange:", [round(float(v), 4) for v in b[:, 0]])
    assert float(a[-2:, 0].mean()) < float(a[:2, 0].mean()) and float(b[-2:, 0].mean()) < float(b[:2, 0].mean())
    assert float((a[:, :4] - b[:, :4]).abs().max()) <= 2e-2 * float(a[:, 0].abs().max())
    pa, pb = res["fp32"]["p"], res["bf16"]["p"]
    assert float((pa - pb).norm() / pa.norm()) <= 2e-3


def test_attention_dropout_mask_is_bernoulli_per_key_and_decorrelated(dev):
    """The attention sites' keep function (csrc/common.hpp, vt_keep_attn: one hash word per four neighbouring keys, a byte
    each against an 8-bit threshold): keep rate 1 - p_q for the quantised probability, the four bytes of a word and
    neighbouring words uncorrelated, different (batch, head) streams and different sites uncorrelated, rows decorrelated."""
    import math

    from visitron_amd import ops

    n = 1024
    for p in (0.1, 0.5, 0.2):
        pq = ops.attn_drop_p(p)
        assert abs(pq - p) <= 1.0 / 512 + 1e-9
        m = ops.attn_dropout_mask(n, (p, 4321, ops.site_attn(3)), 5, device=dev).float().cpu()
        N = m.numel()
        tol = 4.5 * math.sqrt(pq * (1 - pq) / N)
        assert abs(float(m.mean()) - (1.0 - pq)) < tol, (p, float(m.mean()))
        c = m - m.mean()
        var = float((c * c).mean())
        for lag in (1, 2, 3, 4, 5, 8, 32):           # along the keys: inside a hash word (1..3) and across words
            r = float((c[:, :-lag] * c[:, lag:]).mean()) / var
            assert abs(r) < 5.0 / math.sqrt(N), (p, "key lag", lag, r)
        for lag in (1, 2, 7):                         # along the queries
            r = float((c[:-lag] * c[lag:]).mean()) / var
            assert abs(r) < 5.0 / math.sqrt(N), (p, "query lag", lag, r)
        for other in (dict(head=6), dict(site=ops.site_attn(4)), dict(seed=4322)):
            m2 = ops.attn_dropout_mask(n, (p, other.get("seed", 4321), other.get("site", ops.site_attn(3))), other.get("head", 5),
                                       device=dev).float().cpu()
            r = float(((m2 - m2.mean()) * c).mean()) / var
            assert abs(r) < 5.0 / math.sqrt(N), (p, other, r)
        for e in range(4):                            # every byte position of the word has the same rate
            assert abs(float(m[:, e::4].mean()) - (1.0 - pq)) < 2.0 * tol * 2.0, (p, e)
