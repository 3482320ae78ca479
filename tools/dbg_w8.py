import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visitron_amd import ops
dev = "cuda:0"; BF16 = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 14592
specs = [(2304, 768), (768, 768), (3072, 768), (768, 3072)] if len(sys.argv) < 3 else [(int(a.split("x")[0]), int(a.split("x")[1])) for a in sys.argv[2:]]
g = torch.Generator().manual_seed(M)
probs, wants = [], []
for N, K in specs:
    dy = (torch.randn((M, N), generator=g) * 0.5).to(BF16); x = torch.randn((M, K), generator=g).to(BF16)
    probs.append(dict(dy=dy.to(dev), x=x.to(dev), dw=torch.full((N, K), 3.0, device=dev), db=torch.full((N,), 3.0, device=dev)))
    wants.append((dy.double().t() @ x.double(), dy.double().sum(0)))
ops.wgrad(probs, M)
torch.cuda.synchronize()
nk = (M + 63) // 64
tiles = sum((N // 256) * (K // 256) for N, K in specs)
print("M", M, "nk", nk, "tiles", tiles, "steps", tiles * nk, "share(256 wgs)", -(-tiles * nk // 256))
t0 = 0
for (N, K), p, (w, wb) in zip(specs, probs, wants):
    err = (p["dw"].cpu().double() - w).abs()
    eb = (p["db"].cpu().double() - wb).abs()
    print("problem %dx%d: max err %.3g (scale %.3g)  bias err %.3g" % (N, K, err.max(), w.abs().max(), eb.max()))
    for bn in range(N // 256):
        for bk in range(K // 256):
            e = err[bn * 256:(bn + 1) * 256, bk * 256:(bk + 1) * 256]
            if e.max() > 1e-2:
                quad = [[float(e[128 * a:128 * a + 128, 128 * b:128 * b + 128].max()) for b in range(2)] for a in range(2)]
                rows = (e.max(1).values > 1e-2).nonzero().flatten(); cols = (e.max(0).values > 1e-2).nonzero().flatten()
                t = t0 + bn * (K // 256) + bk
                print("   tile %d (bn %d bk %d) linear steps [%d,%d): max %.3g quadrants(n,k) %s rows %d..%d (%d) cols %d..%d (%d)" % (
                    t, bn, bk, t * nk, (t + 1) * nk, e.max(), [["%.2g" % q for q in r] for r in quad], rows.min(), rows.max(), rows.numel(), cols.min(), cols.max(), cols.numel()))
    t0 += (N // 256) * (K // 256)
