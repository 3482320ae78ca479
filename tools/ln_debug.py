"""Debug driver of the deferred-LayerNorm GEMM epilogues: error pattern per variant (which rows / columns are off)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visitron_amd import ops

dev = torch.device("cuda:0")


def row_stats(v, rows):
    M, H = v.shape
    parts = v.view(M, H // 128, 128)
    st = torch.zeros(H // 128, rows, 2)
    st[:, :M, 0] = parts.sum(-1).t()
    st[:, :M, 1] = (parts * parts).sum(-1).t()
    return st


def pattern(name, got, want, tol):
    d = (got.float().cpu() - want).abs()
    bad = ~(d <= tol)
    nbad = int(bad.sum())
    print("%-50s max %.3e  bad %d / %d  nan %d" % (name, float(d[~torch.isnan(d)].max()) if (~torch.isnan(d)).any() else -1, nbad, d.numel(), int(torch.isnan(d).sum())))
    if nbad:
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("    bad rows: n=%d first %s last %s | rows mod 256: %s" % (len(rows), rows[:8].tolist(), rows[-4:].tolist(), sorted(set((rows % 256).tolist()))[:40]))
        print("    bad cols: n=%d first %s last %s | cols mod 128: %s" % (len(cols), cols[:8].tolist(), cols[-4:].tolist(), sorted(set((cols % 128).tolist()))[:40]))


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 700
    variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [15, 22, 23, 16, 18, 19]
    H, I, eps = 768, 1024, 1e-12
    g = torch.Generator().manual_seed(1)
    rows = ops.round_up(M, 16)
    v = torch.randn(M, H, generator=g) * (0.5 + torch.rand(M, 1, generator=g)) + 0.3 * torch.randn(M, 1, generator=g)
    st = row_stats(v, rows)
    mean, var = v.mean(-1, keepdim=True), v.var(-1, unbiased=False, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    gamma = 1.0 + 0.2 * torch.randn(H, generator=g)
    beta = 0.1 * torch.randn(H, generator=g)
    a = (torch.randn(M, I, generator=g) * 0.7).to(torch.bfloat16)
    W = (torch.randn(H, I, generator=g) * 0.03).to(torch.bfloat16)
    cb = 0.05 * torch.randn(H, generator=g) + beta
    want2 = a.float() @ W.float().t() + cb + gamma * ((v.to(torch.float16).float() - mean) * rstd)
    Wp = (torch.randn(I, H, generator=g) * 0.03)
    Wf = (Wp * gamma[None, :]).to(torch.bfloat16)
    gsum = Wf.float().sum(1)
    h = Wp @ beta + 0.05 * torch.randn(I, generator=g)
    x16 = v.to(torch.bfloat16)
    want1 = rstd * (x16.float() @ Wf.float().t() - mean * gsum) + h
    for variant in variants:
        ops.force_gemm_variant(variant)
        out16, out32, so = ops.linear_ln(a.to(dev), W.to(dev), cb.to(dev), gamma.to(dev), st.to(dev), eps, 2, rs=v.to(torch.float16).to(dev))
        torch.cuda.synchronize()
        pattern("variant %d mode 2 fp16 stream" % variant, out32, want2, 1e-2)
        pattern("variant %d mode 2 stats" % variant, so[:, :M].reshape(-1, 2 * M), row_stats(want2, rows)[:, :M].reshape(-1, 2 * M), 5e-2)
        got = ops.linear_ln(x16.to(dev), Wf.to(dev), h.to(dev), gsum.to(dev), st.to(dev), eps, 1)
        torch.cuda.synchronize()
        pattern("variant %d mode 1" % variant, got, want1, 5e-2)
    ops.force_gemm_variant(None)




def probe():
    """mode 1 with identity statistics, g = 0, h = 0: the output is the plain product; print what sits at the bad places."""
    M, H, I, eps = 192, 768, 256, 1e-12
    g = torch.Generator().manual_seed(3)
    rows = ops.round_up(M, 16)
    x = (torch.randn(M, H, generator=g)).to(torch.bfloat16)
    W = (torch.randn(I, H, generator=g) * 0.05).to(torch.bfloat16)
    st = torch.zeros(H // 128, rows, 2)
    st[0, :, 1] = float(H)
    want = x.float() @ W.float().t()
    for variant in (23, 19):
        ops.force_gemm_variant(variant)
        got = ops.linear_ln(x.to(dev), W.to(dev), torch.zeros(I).to(dev), torch.zeros(I).to(dev), st.to(dev), eps, 1).float().cpu()
        torch.cuda.synchronize()
        d = (got - want).abs()
        bad = (~(d <= 5e-2)).nonzero()
        print("variant", variant, "bad", len(bad))
        for r, c in bad[:24].tolist():
            print("   row %3d col %3d got %12.5e want %9.5f | got bits %08x | row+1 want %9.5f | col-2 want %9.5f" % (
                r, c, got[r, c], want[r, c], got[r, c].view(torch.int32).item() & 0xffffffff, want[min(r + 1, M - 1), c], want[r, c - 2]))
    ops.force_gemm_variant(None)


if __name__ == "__main__":
    probe() if (len(sys.argv) > 1 and sys.argv[1] == "probe") else main()
