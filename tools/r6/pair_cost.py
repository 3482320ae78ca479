"""Does FFN-down run slower right behind FFN-up than alone?  python tools/r6/pair_cost.py [M]
Loops of: up alone | down alone | up -> down on the H just written | up -> down on another (clean) H buffer."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from visitron_amd import ops

dev = "cuda:0"
BF16, F16 = torch.bfloat16, torch.float16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 7091
H, I = 768, 3072


def timeit(fn, iters=40, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


ops.ensure_gemm_workspace()
x = torch.randn(M, H, device=dev).to(BF16)
w1 = (torch.randn(I, H, device=dev) * 0.03).to(BF16)
w2 = (torch.randn(H, I, device=dev) * 0.03).to(BF16)
b1, b2 = torch.zeros(I, device=dev), torch.zeros(H, device=dev)
h = torch.empty(M, I, device=dev, dtype=BF16)
h_clean = torch.randn(M, I, device=dev).to(BF16)
dg = torch.empty(M, I, device=dev, dtype=BF16)
rh = torch.randn(M, H, device=dev).to(F16)
mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
y = torch.empty(M, H, device=dev, dtype=F16)
drop = (0.1, 1234, ops.site_out(0))
up = lambda: ops.linear(x, w1, b1, act=ops.ACT_GELU, out=h, pre_act_out=dg)
down = lambda src: ops.linear(src, w2, b2, residual=rh, out=y, residual_ln=(mean, rstd, gamma, beta), drop=drop)
for vu, vd in ((-1, -1), (19, 21), (23, 21), (19, 1), (19, 14)):
    def setv(v):
        ops.set_gemm_variant(v)
    def f_up():
        setv(vu); up()
    def f_down():
        setv(vd); down(h)
    def f_pair():
        setv(vu); up(); setv(vd); down(h)
    def f_pair_clean():
        setv(vu); up(); setv(vd); down(h_clean)
    f_up()
    t_up, t_down, t_pair, t_pc = timeit(f_up), timeit(f_down), timeit(f_pair), timeit(f_pair_clean)
    print("M=%d up v%d %.1f us | down v%d %.1f us | sum %.1f | up->down(H written) %.1f (%+.1f) | up->down(clean H) %.1f (%+.1f)" % (
        M, vu, t_up, vd, t_down, t_up + t_down, t_pair, t_pair - t_up - t_down, t_pc, t_pc - t_up - t_down))
ops.set_gemm_variant(-1)
