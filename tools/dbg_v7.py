import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visitron_amd import ops
dev = "cuda:0"; BF16 = torch.bfloat16
def run(M, N, K, act, res, variant):
    g = torch.Generator().manual_seed(M * 7 + N)
    a = (torch.randn((M, K), generator=g)).to(BF16).float()
    w = (torch.randn((N, K), generator=g) * 0.05).to(BF16).float()
    b = torch.randn((N,), generator=g) * 0.1
    r = torch.randn((M, N), generator=g).to(BF16).float() if res else None
    want = a @ w.t() + b
    if act == 1:
        want = want * 0.5 * (1.0 + torch.erf(want / math.sqrt(2.0)))
    if res:
        want = want + r
    ops.set_gemm_variant(variant)
    out = ops.linear(a.to(dev, BF16), w.to(dev, BF16), b.to(dev), residual=None if r is None else r.to(dev, BF16), act=act)
    torch.cuda.synchronize()
    ops.set_gemm_variant(-1)
    err = ((out.float().cpu() - want).abs() / (1 + want.abs()))
    bad = (err > 0.02).nonzero()
    print("M%d N%d K%d act%d res%d v%d: max err %.4f, bad %d" % (M, N, K, act, res, variant, err.max(), bad.shape[0]))
    if bad.shape[0]:
        rows = bad[:, 0].unique(); cols = bad[:, 1].unique()
        print("   bad rows: n=%d min %d max %d  mod16 set %s | bad cols: n=%d min %d max %d, mod64 set size %d" % (
            rows.numel(), rows.min(), rows.max(), sorted(set((rows % 16).tolist()))[:16], cols.numel(), cols.min(), cols.max(), len(set((cols % 64).tolist()))))
        print("   rows//16 set:", sorted(set((rows // 16).tolist()))[:40])
        i, j = bad[0].tolist()
        print("   first bad [%d,%d]: got %.4f want %.4f  (want-r %.4f)" % (i, j, out[i, j].item(), want[i, j].item(), (want[i, j] - (r[i, j] if res else 0)).item()))
run(512, 512, 128, 0, True, 15)
run(512, 512, 128, 1, False, 15)
run(300, 3072, 768, 1, False, 15)
run(700, 768, 192, 0, True, 16)
