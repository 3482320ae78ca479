"""What the residual epilogues cost at small M: python tools/r6/epi_cost.py [M ...]
Per shape (out-projection K = 768, FFN-down K = 3072; N = 768) and variant: plain | + bf16 residual | + fp16 residual that is a
LayerNorm rebuilt in the epilogue, fp16 out (the training layer's form) | the same + dropout 0.1."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from visitron_amd import ops

dev = "cuda:0"
BF16, F16 = torch.bfloat16, torch.float16


def timeit(fn, iters=30, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


ops.ensure_gemm_workspace()
for M in [int(x) for x in sys.argv[1:]] or [7091]:
    for N, K in ((768, 768), (768, 3072), (768, 2304)):
        x = torch.randn(M, K, device=dev).to(BF16)
        w = (torch.randn(N, K, device=dev) * 0.03).to(BF16)
        b = torch.zeros(N, device=dev)
        rb = torch.randn(M, N, device=dev).to(BF16)
        rh = torch.randn(M, N, device=dev).to(F16)
        mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
        gamma, beta = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        yb = torch.empty(M, N, device=dev, dtype=BF16)
        yh = torch.empty(M, N, device=dev, dtype=F16)
        drop = (0.1, 1234, ops.site_out(0))
        modes = (
            ("plain", lambda: ops.linear(x, w, b, out=yb)),
            ("+res bf16", lambda: ops.linear(x, w, b, residual=rb, out=yb)),
            ("+res fp16", lambda: ops.linear(x, w, b, residual=rh, out=yh)),
            ("+lnres", lambda: ops.linear(x, w, b, residual=rh, out=yh, residual_ln=(mean, rstd, gamma, beta))),
            ("+lnres+drop", lambda: ops.linear(x, w, b, residual=rh, out=yh, residual_ln=(mean, rstd, gamma, beta), drop=drop)),
        )
        print("M=%d N=%d K=%d" % (M, N, K))
        for v in ([int(t) for t in os.environ["VARIANTS"].split(",")] if os.environ.get("VARIANTS") else (1, 14, 35, 15, 22, 23, 20, 21, 33, -1)):
            ops.set_gemm_variant(v)
            row = []
            for name, fn in modes:
                try:
                    row.append("%s %6.1f" % (name, timeit(fn)))
                except RuntimeError:
                    row.append("%s    n/a" % name)
            print("   v%-3d %s" % (v, " | ".join(row)))
        ops.set_gemm_variant(-1)
