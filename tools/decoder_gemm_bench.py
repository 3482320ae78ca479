"""The MLM decoder's forward GEMM ([Ml, 768] x [30528, 768]^T -> fp32 logits, encoder.py:377) under every kernel variant."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import _lib, ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4272
N, K = 30528, 768
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
w = (torch.randn(N, K, generator=g) * 0.02).to(dev, torch.bfloat16)
b = torch.randn(N, generator=g).to(dev)
out = torch.empty(M, N, dtype=torch.float32, device=dev)
lib = _lib.load()
for v in (-1, 1, 14, 9, 10, 11, 15, 22, 23, 16, 18, 19, 20, 21):
    lib.vt_debug_set_gemm_variant(v)
    try:
        ops.linear(a, w, b, out=out, out_f32=True)
        torch.cuda.synchronize()
    except Exception as e:
        print("variant %3d: %s" % (v, str(e)[:60])); continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.linear(a, w, b, out=out, out_f32=True)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e3
    print("variant %3d: %7.1f us  %5.0f TF/s  %.2f TB/s of logits written" % (v, t, 2.0 * M * N * K / t / 1e6, M * N * 4 / t / 1e6))
lib.vt_debug_set_gemm_variant(-1)
