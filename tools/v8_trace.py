"""Phase timeline of the persistent GEMM (variant 16): per-workgroup timestamps -> per-tile K-loop / epilogue time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visitron_amd import ops, _lib

dev = "cuda:0"
M = 58368
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (768, 768)
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = torch.randn(N, K, device=dev).to(torch.bfloat16)
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
MODE = sys.argv[4] if len(sys.argv) > 4 else "plain"      # plain | gelu_pre | res | bias
bias = torch.zeros(N, device=dev) if MODE != "plain" else None
pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if MODE == "gelu_pre" else None
res = torch.randn(M, N, device=dev).to(torch.bfloat16) if MODE == "res" else None
kw = dict(bias=bias, residual=res, act=1 if MODE == "gelu_pre" else 0, pre_act_out=pre)
ops.set_gemm_variant(int(os.environ.get("VARIANT", "16")))
for _ in range(3):
    ops.linear(x, w, out=y, **kw)
torch.cuda.synchronize()
buf = torch.zeros(256 * 64 + 8, dtype=torch.int64, device=dev)
buf[256 * 64] = int(float(sys.argv[3]) * 100) if len(sys.argv) > 3 else 0   # stagger window in us
buf[256 * 64 + 1] = int(sys.argv[5]) if len(sys.argv) > 5 else 0             # 1: every tile fetches tile (0,0)'s operands
buf[256 * 64 + 2] = int(os.environ.get("V11_DBG", "0"))   # variant 25 only: group B paused (1), + 288 / 768 fmas per step (2 / 3)
_lib.load().vt_debug_set_gemm_trace(buf.data_ptr())
ops.linear(x, w, out=y, **kw)
torch.cuda.synchronize()
_lib.load().vt_debug_set_gemm_trace(None)
t = buf[:256 * 64].view(256, 64).cpu()
rt0 = int(t[:, 0].min())
clk = (t[:, 63] - t[:, 1]).double() / ((t[:, 62] - t[:, 0]).double() / 100.0)   # shader cycles per us
print("shader clock during the kernel: %.0f MHz (min %.0f max %.0f)" % (clk.mean(), clk.min(), clk.max()))
print("kernel span: %.1f us" % ((int(t[:, 62].max()) - rt0) / 100.0))
for wg in (0, 1, 8, 100, 255):
    r = t[wg]
    ev = [(int(v) - rt0) / 100.0 for v in r[2:40] if int(v) != 0]
    print("wg %3d: start %.2f us; " % (wg, (int(r[0]) - rt0) / 100.0) + " | ".join(
        "loop %.2f epi %.2f" % (ev[i + 1] - ev[i], ev[i + 2] - ev[i + 1]) for i in range(0, len(ev) - 2, 3))
        + "; end %.2f" % ((int(r[62]) - rt0) / 100.0))
ev = t[:, 2:40].double()
n_t = (ev != 0).sum(1) // 3
loops = torch.cat([(ev[i, 1:3 * n:3] - ev[i, 0:3 * n:3]) for i, n in enumerate(n_t.tolist())]) / 100.0
epis = torch.cat([(ev[i, 2:3 * n:3] - ev[i, 1:3 * n:3]) for i, n in enumerate(n_t.tolist())]) / 100.0
print("tiles per wg: min %d max %d; K loop mean %.2f us (%.3f us per K-step); epilogue mean %.2f us" % (
    int(n_t.min()), int(n_t.max()), loops.mean(), loops.mean() / (K // 64), epis.mean()))

