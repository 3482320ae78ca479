"""Is the rollout LSTM's one-launch-per-position loop cheaper as a captured HIP graph?  Times 511 dependent lstm_step
launches (forward with training saves, then the backward loop) issued directly against a torch.cuda.CUDAGraph replay of
the same launches.  Usage: python tools/lstm_graph_bench.py [B] [S] [hs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 511
    hs = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ops.LSTM_PERSISTENT = False
    xproj = torch.randn(B, S, 4 * hs, device=dev) * 0.5
    w_hh = (torch.randn(4 * hs, hs, device=dev) * 0.05).to(torch.bfloat16)
    w_hh_t = w_hh.t().contiguous()
    lens = torch.full((B,), S, dtype=torch.int32, device=dev)
    d_out = torch.randn(B, S, hs, device=dev)
    state = {}

    def fwd():
        state["f"] = ops.lstm_sequence_train(xproj, w_hh, S, lens, False)

    def bwd():
        state["g"] = ops.lstm_sequence_bwd(d_out, None, None, state["f"][3], w_hh_t, S, lens, False)

    print("direct launches: forward %.3f ms, backward %.3f ms" % (timed(fwd), timed(bwd)))
    # capture (allocations inside the captured region come from the graph's private pool and stay valid for replays)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd()
        bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gf):
        fwd()
    ref_out = state["f"][0]
    with torch.cuda.graph(gb):
        bwd()
    torch.cuda.synchronize()
    print("graph replay:    forward %.3f ms, backward %.3f ms" % (timed(gf.replay), timed(gb.replay)))
    # same numbers?
    gf.replay()
    torch.cuda.synchronize()
    a = ref_out.clone()
    ops_out = ops.lstm_sequence_train(xproj, w_hh, S, lens, False)[0]
    print("max |graph - direct| = %.3e" % float((a - ops_out).abs().max()))


if __name__ == "__main__":
    main()
