"""What the row-range split of the persistent weight-gradient kernel costs: M rows with one range against 2 M rows with two
(the same K-steps per workgroup).  python tools/r6/wgrad_split_cost.py"""
import os
import subprocess
import sys

if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import torch
    from visitron_amd import ops

    dev = "cuda:0"
    H, I = 768, 3072
    M = int(sys.argv[1])
    shapes = [(3 * H, H), (H, H), (I, H), (H, I)]
    mk = lambda n: (torch.randn(M, n, device=dev) * 0.1).to(torch.bfloat16)
    probs = [dict(dy=mk(N), x=mk(K), dw=torch.zeros(N, K, device=dev), db=torch.zeros(N, device=dev)) for N, K in shapes]
    for _ in range(5):
        ops.wgrad(probs, M)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ops.wgrad(probs, M)
    e1.record()
    torch.cuda.synchronize()
    print("M=%d VT_WGRAD_SPLIT=%s: %.1f us per launch" % (M, os.environ.get("VT_WGRAD_SPLIT", "auto"), e0.elapsed_time(e1) / 30 * 1e3))
else:
    for M, split in ((3520, "1"), (7040, "2"), (7040, "1"), (4864, "1"), (9728, "2"), (1792, "1"), (3584, "2")):
        env = dict(os.environ, VT_WGRAD_SPLIT=split)
        subprocess.run([sys.executable, os.path.abspath(__file__), str(M)], env=env)
