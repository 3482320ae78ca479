"""LayerNorm forward / backward kernels alone, on the training step's shapes (run on the GPU box):
    python tools/ln_bench.py [M] [H] [sets]
M rows (default 50 820: the real rows of BASELINE configs[2]'s batch), fp16 pre-LayerNorm sums, bf16 gradients, dropout-masked
second copy of dx as in the step.  `sets` operand sets are rotated (default 6 = 1.9 GB: nothing is served from the 256 MB
infinity cache, as in the step where 40 other kernels run between two LayerNorms); sets = 1 is the warm-cache figure.
Prints microseconds per launch and the rate on the algorithmic bytes (forward 6 B, backward 8 B per element)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops  # noqa: E402


def timed(fn, n):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 50820
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    sets = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    gamma = (1.0 + 0.1 * torch.randn(H, generator=g)).to(dev)
    beta = (0.1 * torch.randn(H, generator=g)).to(dev)
    xs = [(torch.randn(M, H, generator=g) * 2.0).to(dev, torch.float16) for _ in range(sets)]
    dys = [torch.randn(M, H, generator=g).to(dev, torch.bfloat16) for _ in range(sets)]
    ys = [torch.empty(M, H, dtype=torch.bfloat16, device=dev) for _ in range(sets)]
    yhs = [torch.empty(M, H, dtype=torch.float16, device=dev) for _ in range(sets)]
    dxs = [torch.empty(M, H, dtype=torch.bfloat16, device=dev) for _ in range(sets)]
    dx2s = [torch.empty(M, H, dtype=torch.bfloat16, device=dev) for _ in range(sets)]
    dgam, dbet = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    ws = torch.empty(ops.LN_BWD_WS_ROWS * 2 * H, dtype=torch.float32, device=dev)
    n = 20 * sets

    def fwd2(i):
        k = i % sets
        ops.layernorm(xs[k], gamma, beta, 1e-12, out=ys[k], out_h=yhs[k])

    def fwd1(i):
        k = i % sets
        ops.layernorm(xs[k], gamma, beta, 1e-12, out=ys[k])

    def bwd(i):
        k = i % sets
        ops.layernorm_bwd(xs[k], dys[k], gamma, 1e-12, dgam, dbet, dx=dxs[k], ws=ws, dx_dropped=dx2s[k], drop=(0.1, 1234, 7))

    def bwd1(i):
        k = i % sets
        ops.layernorm_bwd(xs[k], dys[k], gamma, 1e-12, dgam, dbet, dx=dxs[k], ws=ws)

    for name, fn, bytes_per in (("layernorm fwd, bf16 + fp16 out", fwd2, 6), ("layernorm fwd, bf16 out", fwd1, 4),
                                ("layernorm bwd + reduce, dx + masked dx", bwd, 8), ("layernorm bwd + reduce, dx only", bwd1, 6)):
        us = timed(fn, n)
        print("%-42s M=%d H=%d sets=%d  %7.1f us  %5.2f TB/s" % (name, M, H, sets, us, bytes_per * M * H / us / 1e6))


if __name__ == "__main__":
    main()
