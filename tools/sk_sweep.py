"""Stream-K variants (31, 32) against the best plain variants over row counts: python tools/sk_sweep.py M1 M2 ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visitron_amd import ops

dev = "cuda:0"
BF16 = torch.bfloat16


def timeit(fn, iters=20, warm=5):
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
for M in [int(x) for x in sys.argv[1:]]:
    for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)):
        x = torch.randn(M, K, device=dev).to(BF16)
        w = torch.randn(N, K, device=dev).to(BF16)
        b = torch.zeros(N, device=dev)
        y = torch.empty(M, N, device=dev, dtype=BF16)
        res = {}
        for v in (1, 35, 14, 15, 23, 20, 21, 31, 32, 33):
            ops.set_gemm_variant(v)
            try:
                res[v] = timeit(lambda: ops.linear(x, w, b, out=y))
            except RuntimeError:
                pass
        ops.set_gemm_variant(-1)
        plain = min((t, v) for v, t in res.items() if v < 28)
        deep = min((t, v) for v, t in res.items() if v in (35,))
        sk = min((t, v) for v, t in res.items() if 28 <= v <= 32)
        s33 = res.get(33)
        print("M=%6d N=%5d K=%5d  best plain v%-2d %6.1f us | best stream-K v%-2d %6.1f us (%+5.1f %%) | split-K v33 %s   %s" % (
            M, N, K, plain[1], plain[0], sk[1], sk[0], 100 * (sk[0] / plain[0] - 1),
            "   n/a" if s33 is None else "%6.1f us (%+5.1f %%)" % (s33, 100 * (s33 / plain[0] - 1)),
            "deep v%d %.1f (%+.1f %%)" % (deep[1], deep[0], 100 * (deep[0] / plain[0] - 1))))
