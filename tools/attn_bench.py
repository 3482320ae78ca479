"""Stand-alone timing of the fused attention kernels at the bench shape (B x 12 heads x S, head size 64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visitron_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 228
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
nh, dev = 12, "cuda:0"
H = nh * 64
qkv = (torch.randn(B * S, 3 * H, device=dev) * 0.8).to(torch.bfloat16)
dctx = torch.randn(B * S, H, device=dev).to(torch.bfloat16)
mask = torch.ones(B, S, device=dev)
lse = torch.zeros(B, nh, S, device=dev)
drop = (p, 1234, 0) if p > 0 else ops.NO_DROP
delta = torch.empty(B, nh, S, device=dev)
out = torch.empty(B * S, 3 * H, device=dev, dtype=torch.bfloat16)
ctx = ops.attention_fwd(qkv, B, S, nh, mask=mask, lse=lse, drop=drop)
ops.set_attn_bwd_waves(int(os.environ.get("ATTN_BWD_WAVES", "17")))   # 17 default (persistent), 16 one pair per workgroup, 8 eight waves, 10 separate delta pass, 4 four waves


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


tf = timeit(lambda: ops.attention_fwd(qkv, B, S, nh, mask=mask, lse=lse, drop=drop))
tb = timeit(lambda: ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh, mask=mask, out=out, delta_ws=delta, drop=drop))
fl = 4.0 * B * nh * S * S * 64
print("%s B=%d S=%d p=%.2f: fwd %.1f us (%.0f TF/s)  bwd %.1f us (%.0f TF/s)" % (
    os.environ.get("VT_HIP_LIB", "default lib"), B, S, p, tf, fl / tf * 1e-6, tb, 2.5 * fl / tb * 1e-6))
if p > 0 and hasattr(ops, "keep_words"):   # the forward writing its keep words, the backward reading them
    kb = torch.zeros(ops.keep_words(B, nh, S), dtype=torch.int32, device=dev)
    tfk = timeit(lambda: ops.attention_fwd(qkv, B, S, nh, mask=mask, lse=lse, drop=drop, keep_bits=kb))
    tbk = timeit(lambda: ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh, mask=mask, out=out, delta_ws=delta, drop=drop, keep_bits=kb))
    print("    with keep words: fwd %.1f us  bwd %.1f us" % (tfk, tbk))
if p > 0 and S == 228:   # the training step's layout: real rows only, per-sequence lengths as bench.py's synthetic batch draws them
    g = torch.Generator().manual_seed(0)
    lens = (128 - torch.randint(0, 33, (B,), generator=g)) + torch.randint(75, 101, (B,), generator=g)
    keepm = torch.arange(S)[None, :] < lens[:, None]
    seq = ops.SeqLayout(keepm.to(dev))
    qc, dc = qkv[:seq.rows].contiguous(), dctx[:seq.rows].contiguous()
    kb = torch.zeros(ops.keep_words(B, nh, S), dtype=torch.int32, device=dev)
    ctxc = ops.attention_fwd(qc, B, S, nh, lse=lse, drop=drop, seq=seq, keep_bits=kb)
    outc = torch.empty(seq.rows, 3 * H, device=dev, dtype=torch.bfloat16)
    tfc = timeit(lambda: ops.attention_fwd(qc, B, S, nh, lse=lse, drop=drop, seq=seq, keep_bits=kb))
    tbc = timeit(lambda: ops.attention_bwd(qc, dc, ctxc, lse, B, S, nh, out=outc, delta_ws=delta, drop=drop, seq=seq, keep_bits=kb))
    print("    compacted rows (%d of %d, %d odd lengths), keep words: fwd %.1f us  bwd %.1f us" % (
        seq.rows, B * S, int((lens % 2).sum()), tfc, tbc))
