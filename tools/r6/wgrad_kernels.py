"""The layer's grouped weight gradient at small M: persistent kernel (auto / split forced) against the one-tile-per-workgroup
kernels (vt_debug_set_wgrad_kernel 128 / 256 / -8).  python tools/r6/wgrad_kernels.py [M ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from visitron_amd import ops

dev = "cuda:0"
H, I = 768, 3072
for M in [int(x) for x in sys.argv[1:]] or [912, 1822, 3580, 5440, 7091]:
    shapes = [(3 * H, H), (H, H), (I, H), (H, I)]
    mk = lambda n: (torch.randn(M, n, device=dev) * 0.1).to(torch.bfloat16)
    probs = [dict(dy=mk(N), x=mk(K), dw=torch.zeros(N, K, device=dev), db=torch.zeros(N, device=dev)) for N, K in shapes]
    res = []
    for mode in (0, 128, 256, -8):
        ops.set_wgrad_kernel(mode)
        try:
            for _ in range(5):
                ops.wgrad(probs, M)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                ops.wgrad(probs, M)
            e1.record()
            torch.cuda.synchronize()
            res.append("mode %4d: %6.1f us" % (mode, e0.elapsed_time(e1) / 30 * 1e3))
        except RuntimeError as e:
            res.append("mode %4d: n/a" % mode)
    ops.set_wgrad_kernel(0)
    print("M=%5d  %s" % (M, " | ".join(res)))
