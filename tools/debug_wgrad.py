import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
def mk(M, N, K, g, pad=False):
    ldy = (N + 7) // 8 * 8 if pad else N
    dyf = torch.zeros(M, ldy); dyf[:, :N] = (torch.randn(M, N, generator=g) * 0.5).to(BF16).float()
    x = torch.randn(M, K, generator=g).to(BF16).float()
    return dict(dy=dyf.to(dev, BF16)[:, :N], x=x.to(dev, BF16), dw=torch.zeros(N, K, device=dev), db=torch.zeros(N, device=dev)), dyf[:, :N], x
def check(tag, p, dy, x):
    want = dy.double().t() @ x.double()
    err = (p["dw"].cpu().double() - want).abs()
    bad = err > 1e-3 * want.abs().max()
    N, K = want.shape
    print("%s N=%d K=%d maxerr %.4g bad %d/%d" % (tag, N, K, err.max(), int(bad.sum()), bad.numel()))
    if bad.any():
        rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
        print("  bad rows(n):", rows[:12], "...", rows[-4:], "count", len(rows))
        print("  bad cols(k):", cols[:12], "...", cols[-4:], "count", len(cols))
    print("  bias err %.4g" % (p["db"].cpu().double() - dy.double().sum(0)).abs().max())
M = 456
g = torch.Generator().manual_seed(0)
specs = [(256, 128), (768, 768), (36, 128), (130, 2240), (1601, 64)]
for N, K in specs:
    p, dy, x = mk(M, N, K, g, pad=True)
    ops.wgrad([p], M); check("single", p, dy, x)
items = [mk(M, N, K, g, pad=True) for N, K in specs]
ops.wgrad([it[0] for it in items], M)
for it in items: check("grouped", *it)
