"""Round 6 GPU tests.

* The straight-line epilogue of the 256x256-tile GEMM kernels with a residual / factor operand on column counts whose last
  column tile is partial (N = 64 mod 128 and N = 128 mod 256), short K (two K-steps: the next tile's origin is computed
  right behind the epilogue) and several tiles per persistent workgroup -- the shape class of round 5's process abort
  (gemm_v7_kernels.hpp, V7_HALF's drain).  BertSelfOutput / BertOutput and their dgrads:
  oscar/modeling_bert.py:94,120.
* The persistent kernel's XCD chunks sized by workgroups (tile counts that are not multiples of 8).
"""
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
