"""The region projection of the trunk (img_embedding + location_embeds as one K-concatenated GEMM writing rows b*S + T + r of
the [B*S, H] embedding output, encoder.py:277-287) under every kernel variant: python tools/region_gemm_bench.py [B] [out_f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
f32 = (int(sys.argv[2]) if len(sys.argv) > 2 else 1) != 0
R, T, H, K = 100, 128, 768, 2240
S = T + R
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
a = torch.randn(B * R, K, generator=g).to(dev, torch.bfloat16)
w = (torch.randn(H, K, generator=g) * 0.02).to(dev, torch.bfloat16)
b = torch.randn(H, generator=g).to(dev)
x = torch.zeros(B * S, H, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
lib = _lib.load()
ref = None
for v in (-1, 1, 14, 9, 10, 11, 15, 22, 23, 16, 18, 19, 20, 21):
    lib.vt_debug_set_gemm_variant(v)
    try:
        x.zero_()
        ops.linear(a, w, b, out=x[T:], ldc=H, grp_rows=R, grp_stride=S, out_f32=f32)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        print("variant %3d: %s" % (v, str(e)[:80]))
        continue
    if ref is None:
        ref = x.clone()
    err = float((x.float() - ref.float()).abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ops.linear(a, w, b, out=x[T:], ldc=H, grp_rows=R, grp_stride=S, out_f32=f32)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 30 * 1e3
    print("variant %3d: %7.1f us  %5.0f TF/s   max |diff to the first| %.2e" % (v, t, 2.0 * B * R * H * K / t / 1e6, err))
lib.vt_debug_set_gemm_variant(-1)
