"""Times vt_pack_concat_bf16 at the region-projection shapes (B x 100 rows of 2054 + 128 floats -> 2240 bf16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops
dev = torch.device("cuda:0")
for B in (64, 256):
    a = torch.randn(B * 100, 2054, device=dev); b = torch.randn(B * 100, 128, device=dev)
    out = ops.pack_concat(a, b, 2240)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): ops.pack_concat(a, b, 2240, out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e3
    print("pack_concat B=%d: %.1f us  %.2f TB/s" % (B, t, B * 100 * (4 * 2182 + 2 * 2240) / t / 1e6))
