"""The inference trunk forward captured into a HIP graph (torch.cuda.CUDAGraph) and replayed, beside the direct call:
python tools/fwd_graph_bench.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd.config import BertConfig  # noqa: E402
from visitron_amd.modeling import PreTrainOscar  # noqa: E402
from visitron_amd.synth import make_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
trunk = PreTrainOscar(BertConfig()).eval().to(dev).bert
batch = make_batch(BertConfig(), B, 128, 100, seed=1234, device=dev, with_labels=False)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    want = trunk(**batch)
    t_direct = timed(lambda: trunk(**batch))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            trunk(**batch)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = trunk(**batch)
    g.replay()
    torch.cuda.synchronize()
    print("max |graph - direct| sequence_output %.3e pooled %.3e" % (float((out[0] - want[0]).abs().max()),
                                                                      float((out[1] - want[1]).abs().max())))
    t_graph = timed(g.replay)
print("B=%d: direct %.3f ms, graph replay %.3f ms" % (B, t_direct, t_graph))
