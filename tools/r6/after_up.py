"""Why FFN-down runs 40 us slower right behind FFN-up at M = 50 845: what has to lie between them for the penalty to go?
python tools/r6/after_up.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from visitron_amd import ops

dev = "cuda:0"
BF16, F16 = torch.bfloat16, torch.float16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 50845
H, I = 768, 3072
VU, VD = [int(t) for t in os.environ.get("VUVD", "19,18").split(",")]
ops.ensure_gemm_workspace()
x = torch.randn(M, H, device=dev).to(BF16)
w1 = (torch.randn(I, H, device=dev) * 0.03).to(BF16)
w2 = (torch.randn(H, I, device=dev) * 0.03).to(BF16)
b1, b2 = torch.zeros(I, device=dev), torch.zeros(H, device=dev)
h = torch.empty(M, I, device=dev, dtype=BF16)
h2 = torch.randn(M, I, device=dev).to(BF16)
dg = torch.empty(M, I, device=dev, dtype=BF16)
rh = torch.randn(M, H, device=dev).to(F16)
rb = torch.randn(M, H, device=dev).to(BF16)
y = torch.empty(M, H, device=dev, dtype=F16)
yb = torch.empty(M, H, device=dev, dtype=BF16)
mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
drop = (0.1, 1234, ops.site_out(0))
big = torch.empty(M, 2 * I, device=dev, dtype=BF16)       # 625 MB: what FFN-up writes
small = torch.empty(1024, device=dev)


def up():
    ops.set_gemm_variant(VU)
    ops.linear(x, w1, b1, act=ops.ACT_GELU, out=h, pre_act_out=dg)


def up_plain():
    ops.set_gemm_variant(VU)
    ops.linear(x, w1, b1, out=h)


def down(src=None):
    ops.set_gemm_variant(VD)
    ops.linear(h if src is None else src, w2, b2, residual=rh, out=y, residual_ln=(mean, rstd, gamma, beta), drop=drop)


def down_plainres():
    ops.set_gemm_variant(VD)
    ops.linear(h, w2, b2, residual=rb, out=yb)


def measure(pre, fn, reps=12):
    """time of fn alone when each call is preceded by pre() (events around fn only)"""
    for _ in range(3):
        pre(); fn()
    torch.cuda.synchronize()
    tot = 0.0
    evs = []
    for _ in range(reps):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / reps * 1e3


nothing = lambda: None
cases = [
    ("down behind down (a loop of itself)", lambda: down(), down),
    ("down behind FFN-up (GELU, two outputs)", up, down),
    ("down behind FFN-up without GELU / second output", up_plain, down),
    ("down behind a 625 MB fill", lambda: big.fill_(1.0), down),
    ("down behind FFN-up and a 4 KB fill", lambda: (up(), small.fill_(0.0)), down),
    ("down behind FFN-up and a 625 MB fill", lambda: (up(), big.fill_(1.0)), down),
    ("down (bf16 residual, no LayerNorm rebuild / dropout) behind FFN-up", up, down_plainres),
    ("down reading another H behind FFN-up", up, lambda: down(h2)),
    ("FFN-up behind down", lambda: down(), up),
    ("FFN-up behind FFN-up", up, up),
]
for name, pre, fn in cases:
    print("%-70s %7.1f us" % (name, measure(pre, fn)))
ops.set_gemm_variant(-1)
