"""LayerNorm forward / backward at the bench shape: us per call and effective TB/s (HIP events)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops  # noqa: E402

dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 58368
H = 768
x = torch.randn(M, H, device=dev).to(torch.bfloat16)
dy = torch.randn(M, H, device=dev).to(torch.bfloat16)
g, b = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev)
y = torch.empty_like(x)
dx = torch.empty_like(x)
dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
ws = torch.empty(ops.LN_BWD_WS_ROWS * 2 * H, device=dev)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


t = timeit(lambda: ops.layernorm(x, g, b, 1e-12, out=y))
print("layernorm fwd  M=%d: %6.1f us  %.2f TB/s" % (M, t, 4.0 * M * H / t / 1e6))
t = timeit(lambda: ops.layernorm_bwd(x, dy, g, 1e-12, dg, db, dx=dx, ws=ws))
print("layernorm bwd  M=%d: %6.1f us  %.2f TB/s" % (M, t, 6.0 * M * H / t / 1e6))
# the training step's form: fp16 pre-LayerNorm sums in, dx written twice (plain + dropout-masked for the dense branch)
xh = x.to(torch.float16)
dxd = torch.empty_like(dx)
t = timeit(lambda: ops.layernorm_bwd(xh, dy, g, 1e-12, dg, db, dx=dx, ws=ws, dx_dropped=dxd, drop=(0.1, 4242, ops.site_out(1))))
print("layernorm bwd  M=%d, fp16 x, dx + dropped dx: %6.1f us  %.2f TB/s" % (M, t, 8.0 * M * H / t / 1e6))
