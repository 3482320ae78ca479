"""Yardstick only (not on the product path): what the vendor BLAS (hipBLASLt / rocBLAS behind torch.matmul)
reaches on the encoder's GEMM shapes, beside this repo's hand-written kernels, on the same GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from visitron_amd import ops

dev = "cuda:0"
BF16 = torch.bfloat16


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


M = int(sys.argv[1]) if len(sys.argv) > 1 else 58368
print("NT  y[M,N] = x[M,K] w[N,K]^T   M=%d" % M)
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    x = torch.randn(M, K, device=dev).to(BF16)
    w = torch.randn(N, K, device=dev).to(BF16)
    y = torch.empty(M, N, device=dev, dtype=BF16)
    ops.autotune_linear(M, N, K, device=dev)
    t_blas = timeit(lambda: torch.matmul(x, w.t(), out=y))
    t_ours = timeit(lambda: ops.linear(x, w, out=y))
    best = None
    per = []
    for v in ops.GEMM_CANDIDATES:   # every autotuner candidate, one by one (the tuned pick above is one noisy measurement)
        ops.set_gemm_variant(v)
        try:
            t = timeit(lambda: ops.linear(x, w, out=y))
        except RuntimeError:
            continue
        per.append((v, t))
        if best is None or t < best[1]:
            best = (v, t)
    ops.set_gemm_variant(-1)
    fl = 2.0 * M * N * K
    print("  N=%5d K=%5d  blas %7.1f us %6.0f TF | ours(tuned) %7.1f us %6.0f TF | best variant %2d %7.1f us %6.0f TF  (%s)" % (
        N, K, t_blas * 1e6, fl / t_blas * 1e-12, t_ours * 1e6, fl / t_ours * 1e-12, best[0], best[1] * 1e6,
        fl / best[1] * 1e-12, " ".join("%d:%.0f" % (v, t * 1e6) for v, t in per)))
print("TN  dw[N,K] = dy[M,N]^T x[M,K]")
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    x = torch.randn(M, K, device=dev).to(BF16)
    dy = torch.randn(M, N, device=dev).to(BF16)
    dw32 = torch.zeros(N, K, device=dev)
    dwb = torch.empty(N, K, device=dev, dtype=BF16)
    t_blas = timeit(lambda: torch.matmul(dy.t(), x, out=dwb))
    t_ours = timeit(lambda: ops.wgrad([dict(dy=dy, x=x, dw=dw32)], M))
    fl = 2.0 * M * N * K
    print("  N=%5d K=%5d  blas %7.1f us %6.0f TF | ours %7.1f us %6.0f TF" % (N, K, t_blas * 1e6, fl / t_blas * 1e-12,
                                                                              t_ours * 1e6, fl / t_ours * 1e-12))
