"""The rollout decoder step (AttnDecoderLSTM.forward, agent_models.py:384-428: ten launches of 5-20 us each) issued
directly against the same launches replayed from a captured HIP graph, per (batch, context length), at the reference's
sizes (angle 4, embedding 64, hidden 512, features 2052, 36 views).  Also a chain of STEPS decoder steps feeding each
other (h_tilde -> prev_h1, c_1 -> c_0, agent.py:383) as one graph: what the rollout loop would replay between two
simulator calls.  Usage: python tools/decoder_graph_bench.py [B] [L] [candidates] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd.rollout import AttnDecoderLSTM  # noqa: E402


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    C = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ang, emb, hs, feat = 4, 64, 512, 2048 + 4
    dec = AttnDecoderLSTM(ang, emb, hs, 0.5, feature_size=feat).eval().to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    action = torch.randn(B, ang, generator=g, device=dev)
    feature = torch.randn(B, 36, feat, generator=g, device=dev).abs() * 0.3
    cand = torch.randn(B, C, feat, generator=g, device=dev).abs() * 0.3
    h1 = torch.randn(B, hs, generator=g, device=dev) * 0.3
    c0 = torch.randn(B, hs, generator=g, device=dev) * 0.3
    ctx = torch.randn(B, L, hs, generator=g, device=dev) * 0.5
    mask = torch.zeros(B, L, dtype=torch.bool, device=dev)
    mask[:, L - L // 4:] = True

    def one_step(h, c):
        with torch.no_grad():
            return dec(action, feature, cand, None, h, c, ctx, mask)

    def chain():
        h, c = h1, c0
        for _ in range(steps):
            h_1, c_1, logit, h_tilde = one_step(h, c)
            h, c = h_tilde, c_1
        return logit

    one_step(h1, c0)          # packed weights built outside any capture
    t_direct = timed(lambda: one_step(h1, c0))
    t_chain = timed(chain, reps=20)

    # one step as a graph
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            one_step(h1, c0)
    torch.cuda.current_stream().wait_stream(s)
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        out1 = one_step(h1, c0)
    t_graph = timed(g1.replay)
    # the chain as a graph
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        out2 = chain()
    t_graph_chain = timed(g2.replay, reps=20)
    want = chain()
    g2.replay()
    torch.cuda.synchronize()
    err = float((out2 - want).abs().max())
    print("decoder step B=%d context %d candidates %d: direct %.1f us, graph replay %.1f us" % (B, L, C, t_direct, t_graph))
    print("chain of %d steps: direct %.1f us (%.1f per step), one graph %.1f us (%.1f per step); replay == direct to %.1e" % (
        steps, t_chain, t_chain / steps, t_graph_chain, t_graph_chain / steps, err))


if __name__ == "__main__":
    main()
