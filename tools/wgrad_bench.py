"""Weight-gradient GEMM group of one encoder layer (q/k/v packed, attention output, FFN up, FFN down) at a given token
count: time per launch and TF/s, HIP events.  python tools/wgrad_bench.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops  # noqa: E402

dev = "cuda:0"
H, I = 768, 3072
for M in [int(v) for v in sys.argv[1:]] or [8208, 58368]:
    g = torch.Generator().manual_seed(0)
    mk = lambda n: (torch.randn(M, n, generator=g) * 0.1).to(dev, torch.bfloat16)
    shapes = [(3 * H, H), (H, H), (I, H), (H, I)]
    probs = [dict(dy=mk(N), x=mk(K), dw=torch.zeros(N, K, device=dev), db=torch.zeros(N, device=dev)) for N, K in shapes]
    flops = sum(2.0 * M * N * K for N, K in shapes)
    for _ in range(3):
        ops.wgrad(probs, M)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.wgrad(probs, M)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = probs[1]["dy"].float().t() @ probs[1]["x"].float()
    err = float((probs[1]["dw"] - ref).abs().max() / ref.abs().max())
    print("M=%6d: %8.1f us per layer group  %6.0f TF/s   (attn-out dW rel err %.2e)" % (M, us, flops / us / 1e6, err))
