"""The layer's grouped weight-gradient launch on one reused operand set (Infinity-Cache warm) against L distinct sets in turn
(what the step does): python tools/r6/wgrad_cold.py [M] [L]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from visitron_amd import ops

dev = "cuda:0"
H, I = 768, 3072
M = int(sys.argv[1]) if len(sys.argv) > 1 else 7091
L = int(sys.argv[2]) if len(sys.argv) > 2 else 12
shapes = [(3 * H, H), (H, H), (I, H), (H, I)]
mk = lambda n: (torch.randn(M, n, device=dev) * 0.1).to(torch.bfloat16)
sets = [[dict(dy=mk(N), x=mk(K), dw=torch.zeros(N, K, device=dev), db=torch.zeros(N, device=dev)) for N, K in shapes] for _ in range(L)]


def timed(order, reps=5):
    for i in order:
        ops.wgrad(sets[i], M)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for i in order:
            ops.wgrad(sets[i], M)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(order)) * 1e3


print("M=%d: one set reused %.1f us per launch | %d distinct sets in turn %.1f us per launch" % (M, timed([0] * L), L, timed(list(range(L)))))
