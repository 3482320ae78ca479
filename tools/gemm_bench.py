"""A/B the GEMM kernel variants on the encoder's layer shapes in ONE process (interleaved rounds,
HIP events on the launch stream, random data).  python tools/gemm_bench.py [--batch 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=228)
    ap.add_argument("--variants", default="1,14,9,10,11,15,16")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    M = a.batch * a.seq
    shapes = [("qkv", 2304, 768, 0, False), ("attn_out", 768, 768, 0, True),
              ("ffn_up", 3072, 768, 1, False), ("ffn_down", 768, 3072, 0, True)]
    variants = [int(v) for v in a.variants.split(",")]
    g = torch.Generator(device="cpu").manual_seed(0)
    total = {v: 0.0 for v in variants}
    for name, N, K, act, res in shapes:
        x = (torch.randn(M, K, generator=g)).to(dev, torch.bfloat16)
        w = (torch.randn(N, K, generator=g) * 0.03).to(dev, torch.bfloat16)
        b = torch.randn(N, generator=g).to(dev)
        r = torch.randn(M, N, generator=g).to(dev, torch.bfloat16) if res else None
        outs = {}
        for v in variants:
            lib.vt_debug_set_gemm_variant(v)
            outs[v] = ops.linear(x, w, b, residual=r, act=act)
        torch.cuda.synchronize()
        ref = outs[variants[0]]
        for v in variants[1:]:
            if not torch.equal(ref, outs[v]):
                d = float((ref.float() - outs[v].float()).abs().max())
                print("  !! variant %d differs from variant %d on %s: max abs %.4g" % (v, variants[0], name, d))
        out = torch.empty_like(ref)
        times = {v: [] for v in variants}
        for _ in range(a.rounds):
            for v in variants:
                lib.vt_debug_set_gemm_variant(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ops.linear(x, w, b, residual=r, act=act, out=out)
                e0.record()
                for _ in range(a.reps):
                    ops.linear(x, w, b, residual=r, act=act, out=out)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / a.reps * 1e3)
        flops = 2.0 * M * N * K
        line = "%-9s M=%d N=%d K=%d |" % (name, M, N, K)
        for v in variants:
            t = sorted(times[v])
            med = t[len(t) // 2]
            total[v] += med
            line += "  v%d %.1fus %.0fTF" % (v, med, flops / med / 1e6)
        print(line, flush=True)
    print("sum per layer (us): " + "  ".join("v%d %.1f" % (v, total[v]) for v in variants))
    lib.vt_debug_set_gemm_variant(-1)


if __name__ == "__main__":
    main()
