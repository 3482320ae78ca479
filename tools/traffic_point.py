"""One point of the traffic experiment (tools/traffic_experiment.sh): the persistent NT GEMM (variant 16) on one encoder shape at
M rows, `reps` launches back to back (operands stay where the last launch left them: L2 / Infinity Cache warm when they fit)
and, with --cold, each launch behind a 512 MiB fill of another buffer (nothing of the operands left in the 256 MiB Infinity
Cache).  Prints one JSON line with the HIP-event time per launch; under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace`
the per-dispatch counters of exactly these launches land in the profiler's csv (kernel name gemm_nt_bf16_v8*).
usage: python tools/traffic_point.py <M> <shape: attn_out | ffn_up> [--cold] [--reps 10]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("M", type=int)
ap.add_argument("shape", choices=["attn_out", "ffn_up"])
ap.add_argument("--cold", action="store_true")
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
M = a.M
N, K, act, res = (768, 768, 0, True) if a.shape == "attn_out" else (3072, 768, 1, False)
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
w = (torch.randn(N, K, generator=g) * 0.03).to(dev, torch.bfloat16)
b = torch.randn(N, generator=g).to(dev)
r = torch.randn(M, N, generator=g).to(dev, torch.bfloat16) if res else None
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev) if a.cold else None
ops.set_gemm_variant(16)
for _ in range(2):
    ops.linear(x, w, b, residual=r, act=act, out=out)
torch.cuda.synchronize()
times = []
for _ in range(a.reps):
    if flush is not None:
        flush.fill_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.linear(x, w, b, residual=r, act=act, out=out)
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) * 1e3)
times.sort()
alg_read = 2.0 * (M * K + N * K) + (2.0 * M * N if res else 0.0)
alg_write = 2.0 * M * N
print(json.dumps(dict(M=M, shape=a.shape, N=N, K=K, cold=a.cold, us_median=round(times[len(times) // 2], 2), us_min=round(times[0], 2),
                      alg_read_MB=round(alg_read / 1e6, 2), alg_write_MB=round(alg_write / 1e6, 2),
                      operand_set_MB=round((alg_read + alg_write) / 1e6, 1), tflops=round(2.0 * M * N * K / times[len(times) // 2] / 1e6, 1))))
