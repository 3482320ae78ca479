"""Time one GEMM + epilogue (ops.linear) at a training shape with a fixed kernel variant; meant to be run twice in one
gpurun call with VT_HIP_LIB pointing at two builds of the library (A/B of an epilogue change on the same box).
    python tools/epilogue_ab.py [M] [N] [K] [act] [pre_act 0/1] [variant]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops  # noqa: E402


def main():
    arg = [int(v) for v in sys.argv[1:]]
    M, N, K, act, pre, variant = (arg + [51200, 3072, 768, 1, 1, 16][len(arg):])[:6]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.03).to(dev, torch.bfloat16)
    b = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    p_out = torch.empty(M, N, dtype=torch.bfloat16, device=dev) if pre else None
    ops.set_gemm_variant(variant)
    ts = []
    for _ in range(7):
        ops.linear(a, w, b, act=act, pre_act_out=p_out, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.linear(a, w, b, act=act, pre_act_out=p_out, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    ts.sort()
    print("%s  M=%d N=%d K=%d act=%d pre=%d variant %d: median %.1f us (min %.1f)  %.0f TF/s" % (
        os.environ.get("VT_HIP_LIB", "default lib"), M, N, K, act, pre, variant, ts[3], ts[0], 2.0 * M * N * K / ts[3] / 1e6))


if __name__ == "__main__":
    main()
