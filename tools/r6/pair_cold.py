"""FFN-up -> FFN-down over L distinct buffer sets (as the training layers have) against one reused set: python tools/r6/pair_cold.py [M] [L]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from visitron_amd import ops

dev = "cuda:0"
BF16, F16 = torch.bfloat16, torch.float16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 7091
L = int(sys.argv[2]) if len(sys.argv) > 2 else 12
H, I = 768, 3072
ops.ensure_gemm_workspace()
w1 = [(torch.randn(I, H, device=dev) * 0.03).to(BF16) for _ in range(L)]
w2 = [(torch.randn(H, I, device=dev) * 0.03).to(BF16) for _ in range(L)]
b1, b2 = torch.zeros(I, device=dev), torch.zeros(H, device=dev)
x = [torch.randn(M, H, device=dev).to(BF16) for _ in range(L)]
h = [torch.empty(M, I, device=dev, dtype=BF16) for _ in range(L)]
dg = [torch.empty(M, I, device=dev, dtype=BF16) for _ in range(L)]
rh = [torch.randn(M, H, device=dev).to(F16) for _ in range(L)]
y = [torch.empty(M, H, device=dev, dtype=F16) for _ in range(L)]
mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
drop = (0.1, 1234, ops.site_out(0))
VU, VD = [int(t) for t in os.environ.get("VUVD", "19,21" if M < 16384 else "18,18").split(",")]


def up(i):
    ops.set_gemm_variant(VU)
    ops.linear(x[i], w1[i], b1, act=ops.ACT_GELU, out=h[i], pre_act_out=dg[i])


def down(i):
    ops.set_gemm_variant(VD)
    ops.linear(h[i], w2[i], b2, residual=rh[i], out=y[i], residual_ln=(mean, rstd, gamma, beta), drop=drop)


def timed(seq, reps=6):
    for f, i in seq:
        f(i)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(seq) * reps + 1)]
    ev[0].record()
    k = 1
    for _ in range(reps):
        for f, i in seq:
            f(i)
            ev[k].record()
            k += 1
    torch.cuda.synchronize()
    t = {}
    for j in range(1, k):
        f, _ = seq[(j - 1) % len(seq)]
        t.setdefault(f.__name__, []).append(ev[j - 1].elapsed_time(ev[j]) * 1e3)
    return {n: sum(v) / len(v) for n, v in t.items()}


one = timed([(up, 0), (down, 0)] * L)
many = timed([p for i in range(L) for p in ((up, i), (down, i))])
print("M=%d: one buffer set reused: up %.1f us, down %.1f us | %d distinct sets in turn: up %.1f us, down %.1f us" % (
    M, one["up"], one["down"], L, many["up"], many["down"]))
# which operand's first touch it is: distinct activations with ONE weight pair, and distinct weights with ONE activation set
w1_all, w2_all = list(w1), list(w2)
w1[:] = [w1_all[0]] * L
w2[:] = [w2_all[0]] * L
act_only = timed([p for i in range(L) for p in ((up, i), (down, i))])
w1[:], w2[:] = w1_all, w2_all
xs, hs, dgs, rhs, ys = list(x), list(h), list(dg), list(rh), list(y)
x[:], h[:], dg[:], rh[:], y[:] = [xs[0]] * L, [hs[0]] * L, [dgs[0]] * L, [rhs[0]] * L, [ys[0]] * L
w_only = timed([p for i in range(L) for p in ((up, i), (down, i))])
x[:], h[:], dg[:], y[:] = xs, hs, dgs, ys                  # distinct everything except the residual
no_res = timed([p for i in range(L) for p in ((up, i), (down, i))])
print("   distinct activations, one weight pair: up %.1f, down %.1f | distinct weights, one activation set: up %.1f, down %.1f | "
      "all distinct but the residual: up %.1f, down %.1f" % (act_only["up"], act_only["down"], w_only["up"], w_only["down"],
                                                             no_res["up"], no_res["down"]))
ops.set_gemm_variant(-1)

# a stand-in prefetch: read the layer's W2 (torch reduction) BEFORE its FFN-up, all sets distinct
def pre(i):
    w2[i].view(torch.int32).sum()


pf = timed([p for i in range(L) for p in ((pre, i), (up, i), (down, i))])
print("   with W2 read once before FFN-up: prefetch %.1f, up %.1f, down %.1f" % (pf["pre"], pf["up"], pf["down"]))
