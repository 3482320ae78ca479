"""Timeline of the persistent GEMM with shared tiles (variants 28 .. 32): per-workgroup realtime stamps.
   python tools/sk_trace.py M N K [variant]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visitron_amd import ops, _lib

dev = "cuda:0"
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 28
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = torch.randn(N, K, device=dev).to(torch.bfloat16)
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
bias = torch.zeros(N, device=dev)
ops.set_gemm_variant(variant)
for _ in range(3):
    ops.linear(x, w, bias, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.linear(x, w, bias, out=y)
e1.record()
torch.cuda.synchronize()
print("variant %d  M=%d N=%d K=%d: %.1f us per launch" % (variant, M, N, K, e0.elapsed_time(e1) * 100))
buf = torch.zeros(256 * 64 + 8, dtype=torch.int64, device=dev)
_lib.load().vt_debug_set_gemm_trace(buf.data_ptr())
ops.linear(x, w, bias, out=y)
torch.cuda.synchronize()
_lib.load().vt_debug_set_gemm_trace(None)
t = buf[:256 * 64].view(256, 64).cpu()
rt0 = int(t[:, 0][t[:, 0] > 0].min())
print("kernel span: %.1f us" % ((int(t[:, 62].max()) - rt0) / 100.0))
for wg in list(range(0, 256, 8))[:32]:
    r = t[wg]
    if int(r[0]) == 0:
        continue
    ev = [(int(v) - rt0) / 100.0 for v in r[2:40] if int(v) != 0]
    print("wg %3d: start %.2f | " % (wg, (int(r[0]) - rt0) / 100.0) + " ".join("%.2f" % v for v in ev) + " | end %.2f" % ((int(r[62]) - rt0) / 100.0))
