"""One smoke parametrisation per experimental GEMM variant (24: gemm_v10.hip two 256 x 128 workgroups per CU, 25: gemm_v11.hip
eight waves on shared stages, 26 / 27: gemm_v12.hip short tiles on three stages) against the fp32 product.  Not part of the
product suite (tests/): build the library with `make -C tools/experiments gemmlab` and run on a GPU box

    VT_HIP_LIB=$PWD/tools/experiments/bin/libvisitron_hip_gemmlab.so python -m pytest tools/experiments/test_experimental_gemms.py -q
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pytestmark = pytest.mark.skipif("gemmlab" not in os.environ.get("VT_HIP_LIB", ""), reason="needs the gemmlab build under VT_HIP_LIB")


@pytest.mark.parametrize("variant", [24, 25, 26, 27])
@pytest.mark.parametrize("res", [False, True])
def test_experimental_variant_matches_fp32(variant, res):
    from visitron_amd import ops

    dev = "cuda:0"
    g = torch.Generator().manual_seed(variant)
    M, N, K = 4100, 768, 768
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g).to(torch.bfloat16) if res else None
    want = a.float() @ w.float().t() + b + (r.float() if res else 0.0)
    ops.set_gemm_variant(variant)
    try:
        got = ops.linear(a.to(dev), w.to(dev), b.to(dev), residual=None if r is None else r.to(dev))
        torch.cuda.synchronize()
    finally:
        ops.set_gemm_variant(-1)
    assert float((got.float().cpu() - want).abs().max()) <= 2e-2 * (1 + float(want.abs().max()))
