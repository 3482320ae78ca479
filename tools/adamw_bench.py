"""adamw_flat over a slab of n parameters: us per launch and TB/s on its 30 (28 with bf16 gradients) bytes per parameter.
   python tools/adamw_bench.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visitron_amd import ops

dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 113_000_000 // 16 * 16
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 0.01
m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
mir = torch.empty(n, dtype=torch.bfloat16, device=dev)
run = lambda: ops.adamw_flat(p, g, m, v, mir, 5e-5, 5e-5, 0.9, 0.999, 1e-8, 0.05, 1.0)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print("adamw_flat n=%d: %.1f us per launch, %.2f TB/s on 30 B per parameter" % (n, us, n * 30 / us / 1e6))
