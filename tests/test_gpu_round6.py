"""Round 6 GPU tests.

* The straight-line epilogue of the 256x256-tile GEMM kernels with a residual / factor operand on column counts whose last
  column tile is partial (N = 64 mod 128 and N = 128 mod 256), short K (two K-steps: the next tile's origin is computed
  right behind the epilogue) and several tiles per persistent workgroup -- the shape class of round 5's process abort
  (gemm_v7_kernels.hpp, V7_HALF's drain).  BertSelfOutput / BertOutput and their dgrads:
  oscar/modeling_bert.py:94,120.
* The persistent kernel's XCD chunks sized by workgroups (tile counts that are not multiples of 8).
"""
import os

import pytest
import torch

from helpers import maxabs

BF16 = torch.bfloat16
F16 = torch.float16

pytestmark = pytest.mark.gpu

TILE256_VARIANTS = [15, 16, 18, 19, 20, 21, 22, 23]


def _rand(shape, g, std=1.0):
    return torch.randn(shape, generator=g) * std


@pytest.mark.parametrize("variant", TILE256_VARIANTS)
@pytest.mark.parametrize("M,N,K", [(600, 832, 128), (900, 384, 128), (2100, 640, 128), (70000, 320, 128), (1500, 640, 256)])
def test_residual_epilogues_on_partial_last_column_tiles(dev, M, N, K, variant):
    """fp16 residual -> fp16 sum, bf16 residual -> bf16, ACT_MUL factor and the rebuilt-LayerNorm residual, all on shapes
    whose last 256-column tile holds 64 or 128 columns.  M = 70 000 x N = 320 gives every persistent workgroup several
    tiles (the ring registers of a skipped column half are then the next tile's registers); K = 128 is two K-steps."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M + N + K + variant)
    a = _rand((M, K), g).to(BF16)
    w = _rand((N, K), g, 0.05).to(BF16)
    b = _rand((N,), g, 0.1)
    r = (_rand((M, N), g) * 3.0).to(F16)
    f = _rand((M, N), g).to(BF16)
    gamma, beta = 1 + 0.2 * _rand((N,), g), 0.3 * _rand((N,), g)
    mean = r.float().mean(-1)
    rstd = 1.0 / torch.sqrt((r.float() - mean[:, None]).pow(2).mean(-1) + 1e-12)
    prod = a.float() @ w.float().t()
    ad, wd, bd = a.to(dev), w.to(dev), b.to(dev)
    ops.set_gemm_variant(variant)
    try:
        o_sum = torch.empty((M, N), dtype=F16, device=dev)
        ops.linear(ad, wd, bd, residual=r.to(dev), out=o_sum)
        o_bf = ops.linear(ad, wd, bd, residual=r.to(BF16).to(dev))
        o_mul = ops.linear(ad, wd, None, residual=f.to(dev), act=ops.ACT_MUL)
        o_ln = torch.empty((M, N), dtype=F16, device=dev)
        ops.linear(ad, wd, bd, out=o_ln, residual=r.to(dev),
                   residual_ln=(mean.to(dev), rstd.to(dev), gamma.to(dev), beta.to(dev)))
        torch.cuda.synchronize()
    finally:
        ops.set_gemm_variant(-1)
    want = prod + b + r.float()
    scale = float(want.abs().max())
    assert maxabs(o_sum, want) <= scale * 2.0 ** -10
    assert maxabs(o_bf, prod + b + r.to(BF16).float()) <= scale * 2.0 ** -7
    want_mul = prod * f.float()
    assert maxabs(o_mul, want_mul) <= float(want_mul.abs().max()) * 2.0 ** -7
    want_ln = prod + b + (r.float() - mean[:, None]) * rstd[:, None] * gamma + beta
    assert maxabs(o_ln, want_ln) <= float(want_ln.abs().max()) * 2.0 ** -10


@pytest.mark.parametrize("variant", [16, 18, 19, 20, 21])
@pytest.mark.parametrize("M,N,K", [(8208, 768, 768), (8208, 768, 3072), (7150, 2304, 768), (3000, 3072, 768), (33000, 768, 768)])
def test_persistent_gemm_tile_counts_that_do_not_divide_over_the_xcds(dev, M, N, K, variant):
    """gemm_nt_bf16_v8's chunk of the tile order per XCD follows the XCD's share of the workgroups (round 6): every tile is
    computed exactly once whatever T mod 8 and grid mod 8 are (the output is pre-filled with NaN: a tile left out shows)."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M + N + K + variant)
    a = _rand((M, K), g).to(BF16).to(dev)
    w = _rand((N, K), g, 0.05).to(BF16).to(dev)
    b = _rand((N,), g, 0.1).to(dev)
    want = a.float() @ w.float().t() + b
    out = torch.full((M, N), float("nan"), dtype=BF16, device=dev)
    ops.set_gemm_variant(variant)
    try:
        ops.linear(a, w, b, out=out)
        torch.cuda.synchronize()
    finally:
        ops.set_gemm_variant(-1)
    assert not bool(torch.isnan(out).any())
    assert float((out.float() - want).abs().max()) <= float(want.abs().max()) * 2.0 ** -7


# ------------------------------------------------------------------------------------------------
# The ENGINE at the BASELINE batches (configs[2]: B = 256, configs[3]'s per-GPU share: B = 36; S = 128 + 100, base config)
# ------------------------------------------------------------------------------------------------
def _base_engine(dev, dropout, seed=0):
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.training import PretrainEngine

    cfg = BertConfig(hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
    torch.manual_seed(seed)
    m = PreTrainOscar(cfg).to(dev).train()
    return cfg, m, PretrainEngine(m, lr=5e-5, weight_decay=0.05, eps=1e-8, schedule="linear", warmup_steps=0, t_total=20000)


@pytest.mark.parametrize("B", [36, 256])
def test_engine_at_the_baseline_batches_compacted_equals_padded(dev, B):
    """PretrainEngine.forward_backward at B x 228 on the base config, dropout 0: the step on the real rows only (what
    bench.py times) against the step over every padded row -- the four losses to 2e-3 and the whole 113 M-element gradient
    slab to 1 % relative L2 (the two runs take different GEMM tiles and a different summation order, nothing else); the
    first two sequences' hidden states against the CPU oracle on the same weights within north_star's 5e-2
    (encoder.py:204-303 through pretrain.py:150-193)."""
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd.synth import make_batch

    from helpers import check_close

    cfg, m, eng = _base_engine(dev, 0.0)
    batch = make_batch(cfg, B, 128, 100, seed=1234, device=dev, with_labels=True)
    eng.compact_min_rows = 0
    eng.compact_rows = True
    out_c = [float(v) for v in eng.forward_backward(batch)[:4]]
    assert eng.last_layout is not None and eng.last_rows < B * 228
    g_c = eng.flat.g.detach().clone()
    eng.compact_rows = False
    out_p = [float(v) for v in eng.forward_backward(batch)[:4]]
    assert eng.last_layout is None and eng.last_rows == B * 228
    g_p = eng.flat.g.detach()
    for n, a_, b_ in zip(("loss", "mask_loss", "next_loss", "token_loss"), out_c, out_p):
        assert abs(a_ - b_) <= 2e-3 * max(1.0, abs(b_)), (n, a_, b_)
        assert a_ == a_ and abs(a_) < 1e4
    rel = float((g_c - g_p).norm() / g_p.norm())
    print("B=%d compacted vs padded: gradient slab rel-L2 %.3e, |g| %.4e, checksum %.6e / %.6e" % (
        B, rel, float(g_p.norm()), float(g_c.double().sum()), float(g_p.double().sum())))
    assert rel <= 1e-2
    # hidden states of the first two sequences inside the full batch vs the oracle
    ref = OModel(cfg).eval()
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    sub = {k: v[:2].cpu() for k, v in batch.items()}
    with torch.no_grad():
        want = ref.bert(input_ids=sub["input_ids"], attention_mask=sub["attention_mask"], img_feats=sub["img_feats"],
                        img_location_embeddings=sub["img_location_embeddings"])[0]
    eng.compact_rows = True
    m.eval()
    got = eng.trunk_forward({k: batch[k] for k in ("input_ids", "attention_mask", "img_feats", "img_location_embeddings")},
                            training=False)[0]
    m.train()
    keep = sub["attention_mask"].bool()
    got2 = got.float().cpu().view(B, 228, -1)[:2]
    check_close("engine B=%d base: sequence_output of the first two sequences (real rows) vs oracle" % B,
                got2[keep], want[keep], 5e-2)


@pytest.mark.parametrize("B", [36, 256])
def test_engine_at_the_baseline_batches_trains(dev, B):
    """Six pretrain steps (dropout 0.1, lr 3e-4 so that six steps show) at B x 228: every loss finite, the total loss of
    the last two steps below the first on the same batch, weights and Adam moments moved (pretrain.py:150-193)."""
    from visitron_amd.synth import make_batch

    cfg, m, eng = _base_engine(dev, 0.1)
    eng.lr = 3e-4
    batch = make_batch(cfg, B, 128, 100, seed=1234, device=dev, with_labels=True)
    p0 = eng.flat.p.detach().clone()
    losses = []
    for _ in range(6):
        out = eng.train_step(batch)
        losses.append([float(v) for v in out[:4]])
    torch.cuda.synchronize()
    print("B=%d losses over six steps: %s" % (B, [round(r_[0], 4) for r_ in losses]))
    assert all(v == v and abs(v) < 1e4 for row in losses for v in row)
    assert 0.5 * (losses[4][0] + losses[5][0]) < losses[0][0] - 0.05
    assert float((eng.flat.p - p0).abs().max()) > 0 and float(eng.flat.v.abs().max()) > 0
    assert eng.step_count == 6
    # the default engine compacts at both batches (round 6: B = 36 too)
    assert eng.last_layout is not None


def test_two_ranks_at_36_sequences_each_match_one_rank_accumulating_both_shards(dev, tmp_path):
    """configs[3]'s per-GPU shape: two processes (both on this GPU, gloo) run PretrainEngine.train_step on 36 x 228 each
    (base config, 12 layers, dropout 0): chunked backward, bucketed all-reduce of the 113 M-element slab in fp32,
    loss /= world (pretrain.py:170,191).  What AdamW is handed on both ranks against ONE rank accumulating the two shards
    with grad_scale 1/2 -- per parameter relative L2 with a floor; the ranks agree bitwise with each other."""
    import os
    import subprocess
    import sys

    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import deterministic_state_dict, make_batch
    from visitron_amd.training import PretrainEngine

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "dp_engine_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29687", script, str(tmp_path), "fp32", "b36"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    got = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    other = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    assert torch.equal(got["g"], other["g"]), "ranks disagree on the all-reduced gradients"
    assert torch.equal(got["p"], other["p"]), "ranks diverged after the optimizer step"

    cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = PreTrainOscar(cfg)
    m.load_state_dict(deterministic_state_dict(m, seed=5, weight_std=0.03))
    m.tie_weights()
    m = m.to(dev).eval()
    eng = PretrainEngine(m, lr=1e-3, weight_decay=0.05, schedule="constant", warmup_steps=0)
    outs = []
    for r_ in range(2):
        shard = {k: v.to(dev) for k, v in make_batch(cfg, 36, 128, 100, seed=100 + r_, with_labels=True).items()}
        outs.append([float(v) for v in eng.forward_backward(shard, grad_scale=0.5, accumulate=(r_ == 1))[:4]])
    want = eng.flat.g.detach().float().cpu()
    for r_, rec in enumerate((got, other)):
        for i in range(4):
            assert abs(rec["out"][i] - outs[r_][i]) <= 5e-3 * max(1.0, abs(outs[r_][i])), (r_, i, rec["out"], outs[r_])
    norms = sorted(float(want[s_:e_].norm()) for s_, e_ in got["ranges"])
    floor = 0.03 * norms[len(norms) // 2]
    worst, worst_name = 0.0, None
    for n, (s_, e_) in zip(got["names"], got["ranges"]):
        rel = float((got["g"][s_:e_] - want[s_:e_]).norm() / max(float(want[s_:e_].norm()), floor))
        if rel > worst:
            worst, worst_name = rel, n
    print("2 ranks x 36 vs one rank accumulating: worst parameter %s rel-L2 %.3e" % (worst_name, worst))
    assert worst <= 2e-2


# ------------------------------------------------------------------------------------------------
# Shared tiles of the persistent GEMM (kernel variants 28 .. 32): the left-over tiles of a launch cut along K
# ------------------------------------------------------------------------------------------------
SHARED = [28, 29, 30, 31, 32]


@pytest.mark.parametrize("variant", SHARED)
@pytest.mark.parametrize("M,N,K", [(7150, 768, 3072), (8208, 768, 2304), (14592, 2304, 768), (14592, 3072, 768), (1534, 768, 3072),
                                   (50845, 768, 768), (4088, 768, 768), (300, 256, 192), (33000, 832, 128)])
def test_shared_tiles_equal_the_fp32_product(dev, M, N, K, variant):
    """C = A W^T + b with the left-over tiles of the persistent kernel shared along K (GemmArgs::sk_parts): tile counts
    below one round (every tile shared: 84 tiles of 256 rows at M = 7 150), one tile over two rounds (513 tiles at
    M = 14 592, N = 2 304), tiny M, K too short to share (192, 128), several epilogues.  Against the fp32 product; the two
    runs of a launch agree BITWISE (the finisher adds the parts in part order); no bounded wait ran out."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M + N + K + variant)
    a = _rand((M, K), g).to(BF16).to(dev)
    w = _rand((N, K), g, 0.05).to(BF16).to(dev)
    b = _rand((N,), g, 0.1).to(dev)
    r = (_rand((M, N), g) * 3.0).to(F16).to(dev)
    want = a.float() @ w.float().t() + b
    ops.set_gemm_variant(variant)
    try:
        out = torch.full((M, N), float("nan"), dtype=BF16, device=dev)
        ops.linear(a, w, b, out=out)
        out_b = torch.full((M, N), float("nan"), dtype=BF16, device=dev)
        ops.linear(a, w, b, out=out_b)
        o_sum = torch.empty((M, N), dtype=F16, device=dev)
        ops.linear(a, w, b, residual=r, out=o_sum)
        o32 = ops.linear(a, w, b, out_f32=True)
        o_gelu = ops.linear(a, w, b, act=ops.ACT_GELU)
        torch.cuda.synchronize()
    finally:
        ops.set_gemm_variant(-1)
    assert ops.gemm_shared_tile_timeouts() == 0
    scale = float(want.abs().max())
    assert not bool(torch.isnan(out).any())
    assert torch.equal(out, out_b)
    assert float((out.float() - want).abs().max()) <= scale * 2.0 ** -7
    assert float((o32 - want).abs().max()) <= scale * 2.0 ** -15 * (K / 64) ** 0.5 + 1e-4
    ws = want + r.float()
    assert float((o_sum.float() - ws).abs().max()) <= float(ws.abs().max()) * 2.0 ** -10
    wg = torch.nn.functional.gelu(want)
    assert float((o_gelu.float() - wg).abs().max()) <= max(1.0, float(wg.abs().max())) * 2.0 ** -7


@pytest.mark.parametrize("variant", SHARED)
def test_shared_tiles_in_the_deferred_layernorm_gemms(dev, variant):
    """vt_linear_ln_bf16 (the inference path's GEMMs, LayerNorm applied in the epilogue) on variants 28 .. 32 against the
    same call on the unshared persistent kernel of the same tile height: mode 1 (QKV shape at B = 64: 513 tiles of 256 rows)
    and mode 2 (FFN-down shape) -- equal to a bf16 / fp16 rounding step (the K sum is taken in another order)."""
    from visitron_amd import ops

    M, H, I = 14592, 768, 3072
    g = torch.Generator().manual_seed(variant)
    rows = M
    v16 = (_rand((M, H), g) * 2.0 + 0.3).to(F16)
    vb = v16.float().to(BF16).to(dev)
    np_ = H // 128
    st = torch.zeros((np_, rows, 2))
    vf = v16.float().view(M, np_, 128)
    st[:, :, 0] = vf.sum(-1).t()
    st[:, :, 1] = (vf * vf).sum(-1).t()
    st = st.to(dev)
    w1 = _rand((3 * H, H), g, 0.05).to(BF16).to(dev)
    b1, c1 = _rand((3 * H,), g, 0.1).to(dev), _rand((3 * H,), g, 0.1).to(dev)
    a2 = _rand((M, I), g).to(BF16).to(dev)
    w2 = _rand((H, I), g, 0.03).to(BF16).to(dev)
    b2, c2 = _rand((H,), g, 0.1).to(dev), (1 + 0.1 * _rand((H,), g)).to(dev)
    plain = 16 if variant == 28 else variant - 11
    res = {}
    for v in (plain, variant):
        ops.set_gemm_variant(v)
        try:
            o1 = ops.linear_ln(vb, w1, b1, c1, st, 1e-12, 1)
            o2, s2, so2 = ops.linear_ln(a2, w2, b2, c2, st, 1e-12, 2, rs=v16.to(dev))
            torch.cuda.synchronize()
        finally:
            ops.set_gemm_variant(-1)
        res[v] = (o1.float(), o2.float(), s2.float(), so2)
    assert ops.gemm_shared_tile_timeouts() == 0
    for i, tol in ((0, 2.0 ** -7), (1, 2.0 ** -7), (2, 2.0 ** -9)):
        x, y = res[variant][i], res[plain][i]
        assert float((x - y).abs().max()) <= float(y.abs().max()) * tol, i
    assert float((res[variant][3] - res[plain][3]).abs().max()) <= 1e-2 * float(res[plain][3].abs().max())


# ------------------------------------------------------------------------------------------------
# Variant 33: split-K of the one-tile kernel, the whole epilogue behind the ordered plane sum
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(912, 768, 3072), (1534, 768, 2304), (4088, 768, 3072), (300, 200, 1024), (1000, 2304, 768),
                                   (640, 768, 768)])
def test_splitk_with_the_whole_epilogue_equals_the_fp32_product(dev, M, N, K):
    """Small M, long K (the reference's own per-GPU batches: 2 x 767, 8 x 511 rows): ksplit copies of the tile list write fp32
    planes into the GEMM workspace, one elementwise kernel sums them in order and runs the register epilogue.  Every epilogue
    kind the training layer uses -- bias, dropout, fp16 / bf16 residual, rebuilt-LayerNorm residual, ACT_MUL, GELU with its
    derivative output, fp32 output, a ragged N -- against fp32 on the same operands; two launches agree bitwise."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M + N + K)
    a = _rand((M, K), g).to(BF16).to(dev)
    w = _rand((N, K), g, 0.05).to(BF16).to(dev)
    b = _rand((N,), g, 0.1).to(dev)
    r = (_rand((M, N), g) * 3.0).to(F16).to(dev)
    f = _rand((M, N), g).to(BF16).to(dev)
    gamma, beta = (1 + 0.2 * _rand((N,), g)).to(dev), (0.3 * _rand((N,), g)).to(dev)
    mean = r.float().mean(-1)
    rstd = 1.0 / torch.sqrt((r.float() - mean[:, None]).pow(2).mean(-1) + 1e-12)
    prod = a.float() @ w.float().t()
    drop = (0.1, 77, 5)
    ops.set_gemm_variant(33)
    try:
        out = torch.full((M, N), float("nan"), dtype=BF16, device=dev)
        ops.linear(a, w, b, out=out)
        out_b = ops.linear(a, w, b)
        o32 = ops.linear(a, w, b, out_f32=True)
        o_sum = torch.empty((M, N), dtype=F16, device=dev)
        ops.linear(a, w, b, residual=r, out=o_sum, drop=drop)
        o_mul = ops.linear(a, w, None, residual=f, act=ops.ACT_MUL)
        pre = torch.empty((M, N), dtype=BF16, device=dev)
        o_gelu = ops.linear(a, w, b, act=ops.ACT_GELU, pre_act_out=pre)
        o_ln = None
        if N % 16 == 0:
            o_ln = torch.empty((M, N), dtype=F16, device=dev)
            ops.linear(a, w, b, out=o_ln, residual=r, residual_ln=(mean, rstd, gamma, beta))
        torch.cuda.synchronize()
    finally:
        ops.set_gemm_variant(-1)
    want = prod + b
    scale = float(want.abs().max())
    assert not bool(torch.isnan(out).any()) and torch.equal(out, out_b)
    assert float((out.float() - want).abs().max()) <= scale * 2.0 ** -7
    assert float((o32 - want).abs().max()) <= scale * 2.0 ** -15 * (K / 64) ** 0.5 + 1e-4
    keep = ops.dropout_mask(M * N, drop, device=dev).view(M, N).float()
    ws = want * keep / 0.9 + r.float()
    assert float((o_sum.float() - ws).abs().max()) <= float(ws.abs().max()) * 2.0 ** -10
    wm = prod * f.float()
    assert float((o_mul.float() - wm).abs().max()) <= float(wm.abs().max()) * 2.0 ** -7
    wg = torch.nn.functional.gelu(want)
    assert float((o_gelu.float() - wg).abs().max()) <= max(1.0, float(wg.abs().max())) * 2.0 ** -7
    x = want.double()
    dg = (0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5).float()
    assert float((pre.float() - dg).abs().max()) <= 2.0 ** -6
    if o_ln is not None:
        wl = want + (r.float() - mean[:, None]) * rstd[:, None] * gamma + beta
        assert float((o_ln.float() - wl).abs().max()) <= float(wl.abs().max()) * 2.0 ** -10


def test_splitk_variant_in_a_tune_table_falls_back_where_nothing_can_be_split(dev):
    """A table entry naming variant 33 that the library applies to a neighbouring row count with too many tiles to split:
    the default kernel runs instead of an error."""
    from visitron_amd import _lib, ops

    ops.ensure_gemm_workspace()
    M, N, K = 40000, 768, 768
    _lib.load().vt_gemm_tune(M, N, K, ops.tune_kind(ops.ACT_NONE), 33)
    g = torch.Generator().manual_seed(1)
    a = _rand((M, K), g).to(BF16).to(dev)
    w = _rand((N, K), g, 0.05).to(BF16).to(dev)
    out = ops.linear(a, w)
    want = a.float() @ w.float().t()
    assert float((out.float() - want).abs().max()) <= float(want.abs().max()) * 2.0 ** -7
    _lib.load().vt_gemm_tune(M, N, K, ops.tune_kind(ops.ACT_NONE), 16)


@pytest.mark.parametrize("M,N,K", [(912, 768, 3072), (4088, 768, 2304), (300, 200, 1024), (640, 768, 128)])
def test_three_stage_ring_of_the_128_tile_kernel(dev, M, N, K):
    """Variant 35 (gemm_nt_bf16_v2 on a ring of three stages; K of one, two and many K-steps): bias, fp16 residual with dropout,
    fp32 output and GELU against fp32; bitwise equal to the two-stage kernel (the same sums in the same order)."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M + N + K + 35)
    a = _rand((M, K), g).to(BF16).to(dev)
    w = _rand((N, K), g, 0.05).to(BF16).to(dev)
    b = _rand((N,), g, 0.1).to(dev)
    r = (_rand((M, N), g) * 3.0).to(F16).to(dev)
    drop = (0.1, 5, 2)
    res = {}
    for v in (1, 35):
        ops.set_gemm_variant(v)
        try:
            o = ops.linear(a, w, b)
            o32 = ops.linear(a, w, b, out_f32=True)
            os_ = torch.empty((M, N), dtype=F16, device=dev)
            ops.linear(a, w, b, residual=r, out=os_, drop=drop)
            og = ops.linear(a, w, b, act=ops.ACT_GELU)
            torch.cuda.synchronize()
        finally:
            ops.set_gemm_variant(-1)
        res[v] = (o, o32, os_, og)
    for x, y in zip(res[1], res[35]):
        assert torch.equal(x, y)
    want = a.float() @ w.float().t() + b
    assert float((res[35][1] - want).abs().max()) <= float(want.abs().max()) * 2.0 ** -15 * (K / 64) ** 0.5 + 1e-4


# ---- attention-probability dropout, both resolutions (vt_set_attn_dropout_bits): 16-bit fields = the default since round 6
# (oscar/modeling_bert.py:62's nn.Dropout(0.1) runs as 0.100006), 8-bit fields = the faster form of rounds 4-5 (0.1016) ------
@pytest.fixture(params=[16, 8])
def attn_bits(request):
    from visitron_amd import ops

    before = ops.attn_dropout_bits()
    ops.set_attn_dropout_bits(request.param)
    try:
        yield request.param
    finally:
        ops.set_attn_dropout_bits(before)


def test_attention_dropout_resolution_default_and_keep_function(dev, attn_bits):
    """The default is the exact-p form: p resolved to 1/65536 (0.1 -> 6554/65536); the 8-bit form resolves 1/256 (0.1 ->
    26/256).  Either way: the keep rate of the masks the kernels derive, independence between heads and between neighbouring
    keys (fields of one hash word), a p below half a step runs as one step, a p that rounds to 1 is refused."""
    from visitron_amd import ops

    n_steps = float(1 << attn_bits)
    assert ops.attn_dropout_bits() == attn_bits
    assert ops.attn_drop_p(0.1) == (6554.0 / 65536.0 if attn_bits == 16 else 26.0 / 256.0) and ops.attn_drop_p(0.0) == 0.0
    assert ops.attn_drop_p(1e-6) == 1.0 / n_steps                  # below half a step: one step, never silently off
    with pytest.raises(ValueError):
        ops.attn_drop_p(1.0 - 0.25 / n_steps)
    n = 512
    for p in (0.1, 0.25, 0.003):
        pe = ops.attn_drop_p(p)
        m = ops.attn_dropout_mask(n, (p, 4321, ops.site_attn(3)), 5, device=dev).float().cpu()
        sd = (pe * (1 - pe) / (n * n)) ** 0.5
        assert abs(float(m.mean()) - (1 - pe)) < 5 * sd, (p, float(m.mean()))
        m2 = ops.attn_dropout_mask(n, (p, 4321, ops.site_attn(3)), 6, device=dev).float().cpu()
        agree = float((m == m2).float().mean())
        want = pe * pe + (1 - pe) * (1 - pe)
        assert abs(agree - want) < 6 * (want * (1 - want) / (n * n)) ** 0.5 + 1e-4, (p, agree, want)
        a, b = m[:, 0::2].flatten(), m[:, 1::2].flatten()
        cov = float(((a - a.mean()) * (b - b.mean())).mean()) / max(float(a.std() * b.std()), 1e-9)
        assert abs(cov) < 6.0 / (a.numel() ** 0.5), (p, cov)
    with pytest.raises(RuntimeError):
        ops.set_attn_dropout_bits(12)


def test_attention_dropout_default_is_the_exact_p_form(dev):
    from visitron_amd import ops

    if os.environ.get("VT_ATTN_DROPOUT_BITS") in (None, "", "16"):
        assert ops.attn_dropout_bits() == 16 and abs(ops.attn_drop_p(0.1) - 0.1) < 1e-5


@pytest.mark.parametrize("B,S,nh", [(2, 228, 3), (1, 37, 2), (2, 300, 2)])
def test_attention_dropout_words_hash_backward_and_fp32_at_both_resolutions(dev, attn_bits, B, S, nh):
    """The forward's keep words are the mask of the resolution in force (vt_debug_dropout_mask), the context does not depend on writing them, the
    backward that re-derives the mask from the hash (8-wave and 4-wave kernels) equals the backward reading the words, and
    context / gradient agree with fp32 autograd under the same mask and 1 / (1 - p_effective)."""
    from test_gpu_round3 import _unpack_keep_words
    from visitron_amd import ops

    g = torch.Generator().manual_seed(11 * S + nh)
    H = nh * 64
    drop = (0.1, 4242, ops.site_attn(1))
    qkv = (torch.randn(B * S, 3 * H, generator=g) * 0.9).to(BF16)
    dctx = (torch.randn(B * S, H, generator=g) * 0.6).to(BF16)
    mask = (torch.rand(B, S, generator=g) > 0.2).float()
    mask[:, 0] = 1.0
    lse0 = torch.zeros((B, nh, S), dtype=torch.float32, device=dev)
    lse1 = torch.zeros_like(lse0)
    words = torch.full((ops.keep_words(B, nh, S),), -1, dtype=torch.int32, device=dev)
    q_d, d_d, m_d = qkv.to(dev), dctx.to(dev), mask.to(dev)
    ctx0 = ops.attention_fwd(q_d, B, S, nh, mask=m_d, lse=lse0, drop=drop)
    ctx1 = ops.attention_fwd(q_d, B, S, nh, mask=m_d, lse=lse1, drop=drop, keep_bits=words)
    torch.cuda.synchronize()
    assert torch.equal(ctx0, ctx1) and torch.equal(lse0, lse1)
    got = _unpack_keep_words(words, B, nh, S)[:, :S, :S]
    want = torch.stack([ops.attn_dropout_mask(S, drop, i, device=dev) for i in range(B * nh)]).bool().cpu()
    assert torch.equal(got, want)
    assert abs(float(want.float().mean()) - (1 - ops.attn_drop_p(0.1))) < 0.01
    d_words = ops.attention_bwd(q_d, d_d, ctx1, lse1, B, S, nh, mask=m_d, drop=drop, keep_bits=words)
    outs = {}
    for waves in (8, 4):
        ops.set_attn_bwd_waves(waves)
        try:
            outs[waves] = ops.attention_bwd(q_d, d_d, ctx0, lse0, B, S, nh, mask=m_d, drop=drop)      # hash path
            if waves == 8:
                outs["8w"] = ops.attention_bwd(q_d, d_d, ctx1, lse1, B, S, nh, mask=m_d, drop=drop, keep_bits=words)
            torch.cuda.synchronize()
        finally:
            ops.set_attn_bwd_waves(0)
    if S <= 256:
        assert torch.equal(outs[8], outs["8w"])             # the same kernel, hash against words: bit for bit
    top = float(d_words.float().abs().max())
    for k, v in outs.items():
        assert float((v.float() - d_words.float()).abs().max()) <= 2.0 ** -6 * top, k
    # fp32 autograd under the same mask
    x = qkv.float().requires_grad_(True)
    t = x.view(B, S, 3, nh, 64).permute(2, 0, 3, 1, 4)
    bias = ((1.0 - mask) * -10000.0)[:, None, None, :]
    p = torch.softmax(t[0] @ t[1].transpose(-1, -2) / 8.0 + bias, -1)
    km = want.view(B, nh, S, S).float()
    o = ((p * km / (1.0 - ops.attn_drop_p(0.1))) @ t[2]).permute(0, 2, 1, 3).reshape(B * S, H)
    o.backward(dctx.float())
    assert maxabs(ctx1, o.detach()) < 3e-2
    assert maxabs(d_words, x.grad) < 3e-2 * (1 + float(x.grad.abs().max()))


def test_attention_dropout_on_compacted_rows_at_both_resolutions(dev, attn_bits):
    """The same on the training step's compacted layout: the hash index runs over each sequence's own length."""
    from test_gpu_round3 import _unpack_keep_words
    from visitron_amd import ops

    g = torch.Generator().manual_seed(5)
    B, S, nh = 3, 96, 2
    H = nh * 64
    lens = torch.tensor([96, 50, 33])
    seq = ops.SeqLayout((torch.arange(S)[None, :] < lens[:, None]).to(dev))
    drop = (0.1, 77, ops.site_attn(0))
    qkv = (torch.randn(seq.rows, 3 * H, generator=g) * 0.8).to(dev, BF16)
    dctx = torch.randn(seq.rows, H, generator=g).to(dev, BF16)
    lse = torch.zeros((B, nh, S), dtype=torch.float32, device=dev)
    words = torch.zeros(ops.keep_words(B, nh, S), dtype=torch.int32, device=dev)
    ctx = ops.attention_fwd(qkv, B, S, nh, lse=lse, drop=drop, seq=seq, keep_bits=words)
    d1 = ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh, drop=drop, seq=seq, keep_bits=words)
    ops.set_attn_bwd_waves(8)
    try:
        d0 = ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh, drop=drop, seq=seq)
        d2 = ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh, drop=drop, seq=seq, keep_bits=words)
        torch.cuda.synchronize()
    finally:
        ops.set_attn_bwd_waves(0)
    assert torch.equal(d0, d2)
    assert float((d0.float() - d1.float()).abs().max()) <= 2.0 ** -7 * float(d0.float().abs().max())
    got = _unpack_keep_words(words, B, nh, S)
    for b in range(B):
        n = int(lens[b])
        for h in range(nh):
            want = ops.attn_dropout_mask(n, drop, b * nh + h, device=dev).bool().cpu()
            assert torch.equal(got[b * nh + h, :n, :n], want), (b, h)


@pytest.mark.parametrize("compact", [True, False])
def test_training_matches_oracle_with_same_masks_at_both_resolutions(dev, attn_bits, compact):
    """A whole training step at either resolution against the oracle running under the masks the kernels derived (losses and
    every gradient: tests/test_gpu_train.py's dropout test), and the engine reports the probability it ran."""
    import test_gpu_train as t
    from visitron_amd import ops

    t.test_dropout_training_matches_oracle_with_same_masks(dev, 0.1, 0.1, compact)
    cfg = t._dropout_cfg(0.1, 0.1)
    _, _, eng = t._engine_pair(cfg, 5, dev)
    assert eng.attention_dropout_effective == ops.attn_drop_p(0.1) == (6554.0 / 65536.0 if attn_bits == 16 else 26.0 / 256.0)


# ---- visitron_amd.parallel.DataParallel: the reference's multi-gpu-dp mode (pretrain.py:93-94), rehearsed on one GPU ----------
def _dp_models(dev, cfg=None, seed=3):
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.parallel import DataParallel

    cfg = cfg or mini_config()
    cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = 0.0
    torch.manual_seed(seed)
    master = PreTrainOscar(cfg).to(dev)
    single = PreTrainOscar(cfg).to(dev)
    single.load_state_dict(master.state_dict())
    return cfg, master, single, DataParallel(master, device_ids=[0, 0])


def _halves(batch, n=2):
    parts = [dict() for _ in range(n)]
    for k, v in batch.items():
        for i, c in enumerate(v.chunk(n, 0)):
            parts[i][k] = c
    return parts


@pytest.mark.parametrize("B", [6, 5])
def test_data_parallel_training_step_equals_the_per_replica_steps(dev, B):
    """`model = DataParallel(model)`; `loss = model(**batch)[0].mean()`; `loss.backward()` (pretrain.py:93-94,169-193): every
    entry of the 7-tuple is the vector of the replicas' values, and the wrapped module's gradients are those of the mean of
    the per-chunk losses computed by ONE module on the chunks (torch.chunk: 3 + 3 and 3 + 2 sequences) -- the replicas'
    own gradients are gone afterwards, parameters() and state_dict() show the wrapped module only."""
    from visitron_amd.synth import make_batch

    cfg, master, single, dp = _dp_models(dev)
    dp.train(); single.train()
    batch = {k: v.to(dev) for k, v in make_batch(cfg, B, text_len=24, region_len=12, seed=9).items()}
    out = dp(**batch)
    assert len(out) == 7 and all(o.shape == (2,) for o in out)
    out[0].mean().backward()
    ref = []
    for p in _halves(batch):                      # (one forward / backward pair at a time: the HIP step keeps one gradient set)
        ref.append(single(**p))
        (ref[-1][0] / 2).backward()
    torch.cuda.synchronize()
    for i in range(7):
        want = torch.stack([torch.as_tensor(r[i], dtype=torch.float32, device=dev) for r in ref])
        assert torch.allclose(out[i].float(), want, rtol=1e-4, atol=1e-5, equal_nan=True), (i, out[i], want)
    worst = 0.0
    for (n, pm), (_, ps) in zip(master.named_parameters(), single.named_parameters()):
        assert (pm.grad is None) == (ps.grad is None), n
        if ps.grad is not None:
            err = float((pm.grad - ps.grad).norm() / (ps.grad.norm() + 1e-6 * ps.grad.numel() ** 0.5))
            worst = max(worst, err)
            assert err < 2e-3, (n, err)
    assert all(p.grad is None for r in dp.replicas()[1:] for p in r.parameters())
    assert len(list(dp.parameters())) == len(list(master.parameters()))
    assert all(k.startswith("module.") for k in dp.state_dict()) and len(dp.state_dict()) == len(master.state_dict())
    assert dp.module is master


def test_data_parallel_replicas_follow_the_optimizer(dev):
    """Two steps of the reference's loop with a torch optimizer on `model.parameters()`: the second forward must see the
    updated weights on every replica (they are refreshed from the wrapped module before each forward), i.e. equal the
    single module stepping on the same chunks' mean loss."""
    from visitron_amd.synth import make_batch

    cfg, master, single, dp = _dp_models(dev, seed=5)
    dp.train(); single.train()
    opt_dp = torch.optim.SGD(dp.parameters(), lr=0.05)
    opt_s = torch.optim.SGD(single.parameters(), lr=0.05)
    losses = []
    for step in range(3):
        batch = {k: v.to(dev) for k, v in make_batch(cfg, 4, text_len=20, region_len=8, seed=30 + step).items()}
        opt_dp.zero_grad(); opt_s.zero_grad()
        l_dp = dp(**batch)[0].mean()
        l_dp.backward()
        opt_dp.step()
        l_s = 0.0
        for p in _halves(batch):
            l = single(**p)[0] / 2
            l.backward()
            l_s += float(l)
        opt_s.step()
        losses.append((float(l_dp), l_s))
    torch.cuda.synchronize()
    for a, b in losses:
        assert abs(a - b) < 2e-3 * max(1.0, abs(b)), losses
    assert losses[0][0] != losses[2][0]
    for (n, pm), (_, ps) in zip(master.named_parameters(), single.named_parameters()):
        assert float((pm - ps).abs().max()) < 2e-3 * (1.0 + float(ps.abs().max())), n


def test_data_parallel_inference_gathers_along_the_batch(dev):
    """eval mode, torch.no_grad(): the trunk's outputs of a scattered batch are the per-chunk outputs concatenated; one
    device id is the wrapped module itself; a module on another device than device_ids[0] and torch.nn.DataParallel's own
    replicas are refused with the way out."""
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.parallel import DataParallel
    from visitron_amd.synth import make_batch

    cfg = mini_config()
    torch.manual_seed(1)
    trunk = BertImgModelwithLocationEmbeds(cfg).to(dev).eval()
    dp = DataParallel(trunk, device_ids=[0, 0]).eval()
    b = make_batch(cfg, 5, text_len=16, region_len=8, seed=2, with_labels=False)
    kw = {k: b[k].to(dev) for k in ("input_ids", "token_type_ids", "attention_mask", "img_feats", "img_location_embeddings")
          if k in b}
    with torch.no_grad():
        seq, pooled = dp(**kw)[:2]
        parts = [trunk(**p)[:2] for p in _halves(kw)]
    torch.cuda.synchronize()
    assert seq.shape[0] == 5 and pooled.shape[0] == 5
    assert torch.equal(seq, torch.cat([p[0] for p in parts])) and torch.equal(pooled, torch.cat([p[1] for p in parts]))
    one = DataParallel(trunk, device_ids=[0])
    with torch.no_grad():
        s1 = one(**kw)[0]
        s0 = trunk(**kw)[0]
    assert torch.equal(s1, s0)
    with pytest.raises(RuntimeError, match="device_ids"):
        DataParallel(BertImgModelwithLocationEmbeds(cfg), device_ids=[0, 0])(**kw)      # parameters still on the CPU
    replica = trunk._replicate_for_data_parallel()
    replica._former_parameters = {}
    with pytest.raises(NotImplementedError, match="visitron_amd.parallel.DataParallel"):
        replica(**kw)


def test_backward_after_a_later_forward_of_the_same_module_is_refused(dev):
    """The fused step computes the gradients together with the forward and keeps one set per module: `l1 = model(a)[0];
    l2 = model(b)[0]; l1.backward()` cannot be served and must say so instead of handing l2's gradients to l1."""
    from visitron_amd.synth import make_batch

    cfg, master, single, _ = _dp_models(dev)
    single.train()
    a = {k: v.to(dev) for k, v in make_batch(cfg, 2, text_len=16, region_len=8, seed=1).items()}
    b = {k: v.to(dev) for k, v in make_batch(cfg, 2, text_len=16, region_len=8, seed=2).items()}
    l1 = single(**a)[0]
    l2 = single(**b)[0]
    with pytest.raises(RuntimeError, match="before the next forward"):
        l1.backward()
    l2.backward()                                   # the latest pair is served
    assert all(p.grad is not None for p in single.parameters() if p.requires_grad)


# ---- weight prefetch in the layer loops (vt_set_weight_prefetch): reads only, so every mode must return the same bits -------------
def test_weight_prefetch_modes_do_not_change_a_training_step(dev):
    """PretrainEngine.forward_backward on the base hidden size with each training prefetch mode (off, per-layer launch, per-GEMM
    launches, side stream, riding in the LayerNorm kernels): losses and the whole gradient slab bit for bit the same (the prefetch
    reads weights and drops them; the spare workgroups of the LayerNorm forward / the LayerNorm backward's reduction must not
    change what the row workgroups compute -- including the grid-stride of the rows)."""
    from visitron_amd import ops
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch
    from visitron_amd.training import PretrainEngine

    cfg = BertConfig(num_hidden_layers=3, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    torch.manual_seed(0)
    model = PreTrainOscar(cfg).to(dev).train()
    eng = PretrainEngine(model)
    batch = {k: v.to(dev) for k, v in make_batch(cfg, 5, text_len=40, region_len=23, seed=3).items()}
    before = ops.weight_prefetch()
    outs = {}
    try:
        for mode in (0, 1, 2, 3, 4):
            ops.set_weight_prefetch(training=mode)
            assert ops.weight_prefetch()[0] == mode
            eng.fb_count = 7                       # the same dropout masks every time
            t = eng.forward_backward(batch)
            torch.cuda.synchronize()
            outs[mode] = ([float(x) for x in t], eng.flat.g.clone())
    finally:
        ops.set_weight_prefetch(*before)
    for mode in (1, 2, 3, 4):
        assert outs[mode][0] == outs[0][0], (mode, outs[mode][0], outs[0][0])
        assert torch.equal(outs[mode][1], outs[0][1]), mode
    with pytest.raises(RuntimeError):
        ops.set_weight_prefetch(training=9)


@pytest.mark.parametrize("B,T,R", [(3, 40, 23), (2, 511, 0)])
def test_weight_prefetch_modes_do_not_change_an_inference_forward(dev, B, T, R):
    """BertImgModelwithLocationEmbeds in eval mode (the deferred-LayerNorm layer loop) under each inference prefetch mode: the
    attention kernel's spare z-slices read the next GEMMs' weights, everything it returns stays bit for bit what it was."""
    from visitron_amd import ops
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.synth import make_batch

    cfg = BertConfig(num_hidden_layers=2)
    torch.manual_seed(1)
    trunk = BertImgModelwithLocationEmbeds(cfg).to(dev).eval()
    b = make_batch(cfg, B, text_len=T, region_len=max(R, 1), seed=4, with_labels=False)
    keys = ("input_ids", "token_type_ids", "attention_mask") + (("img_feats", "img_location_embeddings") if R else ())
    kw = {k: b[k].to(dev) for k in keys if k in b}
    if not R:
        kw["attention_mask"] = kw["attention_mask"][:, :T]
    before = ops.weight_prefetch()
    outs = {}
    try:
        for mode in (0, 1, 2, 3):
            ops.set_weight_prefetch(inference=mode)
            with torch.no_grad():
                seq, pooled = trunk(**kw)[:2]
            torch.cuda.synchronize()
            outs[mode] = (seq.clone(), pooled.clone())
    finally:
        ops.set_weight_prefetch(*before)
    for mode in (1, 2, 3):
        assert torch.equal(outs[mode][0], outs[0][0]) and torch.equal(outs[mode][1], outs[0][1]), mode


def test_step_counters_ride_with_the_row_counts(dev):
    """ops.batch_row_counts: five batch numbers and the two bounded-wait counters (vt_step_counters) in ONE read-back that
    travels on a side stream; the counts against torch, the counters 0 in a healthy process and equal to what the blocking
    entry points report; two read-backs in flight at once keep their own values."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(4)
    B, S = 5, 37
    labels = torch.randint(-1, 3, (B * S,), generator=g).to(dev)
    tl = torch.randint(-1, 2, (B * S,), generator=g).to(dev)
    mask = (torch.rand(B, S, generator=g) > 0.3).float()
    mask[:, 0] = 1.0
    mask_d = mask.to(dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    h1 = ops.batch_row_counts_begin(labels, tl, mask_d, err, B, S)
    h2 = ops.batch_row_counts_begin(labels, None, None, err, B, S)      # (same pinned buffer: must be read in order)
    vals, tiles = ops.batch_row_counts_end(h1)
    assert len(vals) == 7 and vals[0] == 0
    assert vals[1] == int((labels != -1).sum()) and vals[2] == int((tl != -1).sum()) and vals[3] == int((mask != 0).sum())
    assert vals[5] == 0 and vals[6] == 0
    assert ops.wgrad_turn_timeouts() == 0 and ops.gemm_shared_tile_timeouts() == 0
    vals2, _ = ops.batch_row_counts_end(h2)
    assert vals2[1] == vals[1] and vals2[2] == 0


def test_small_inference_forwards_take_the_seven_launch_layer(dev):
    """The product's rule (CaptionBertEncoder.serves_deferred_ln): an eval forward below deferred_ln_min_rows (2 800) padded
    token rows runs the seven-launch layer -- bit for bit what `deferred_ln = False` returns -- and the deferred-LayerNorm
    loop from there on; both inside 5e-2 of the oracle on the base configuration's width (B = 2 and B = 13 sequences of 228)."""
    from oracle.modeling import BertImgModelwithLocationEmbeds as OTrunk
    from helpers import check_close, model_pair
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.synth import make_batch

    cfg = BertConfig(num_hidden_layers=3, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ref, prod = model_pair(OTrunk, BertImgModelwithLocationEmbeds, cfg, seed=2, device=dev)
    enc = prod.encoder
    enc.deferred_ln_min_rows = 2800                       # the product's default (the test session sets 0: tests/conftest.py)
    keys = ("input_ids", "token_type_ids", "attention_mask", "img_feats", "img_location_embeddings")
    for B, deferred in ((2, False), (13, True)):
        b = make_batch(cfg, B, seed=5, with_labels=False)
        kw = {k: b[k] for k in keys if k in b}
        assert enc.serves_deferred_ln(rows=B * 228) is deferred
        with torch.no_grad():
            want = ref(**kw)[0]
            got = prod(**{k: v.to(dev) for k, v in kw.items()})[0]
            enc.deferred_ln = False
            seven = prod(**{k: v.to(dev) for k, v in kw.items()})[0]
            enc.deferred_ln = True
        check_close("inference B=%d x 228 under the row rule (%s)" % (B, "deferred loop" if deferred else "seven-launch layer"),
                    got, want, 5e-2)
        assert torch.equal(got, seven) is (not deferred)
