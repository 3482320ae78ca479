"""Host (enqueue) time of a training step against its device time: python tools/host_time.py [B ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visitron_amd.config import BertConfig
from visitron_amd.modeling import PreTrainOscar
from visitron_amd.synth import make_batch
from visitron_amd.training import PretrainEngine

dev = "cuda:0"
cfg = BertConfig(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
torch.manual_seed(0)
model = PreTrainOscar(cfg).to(dev).train()
eng = PretrainEngine(model)
for B in [int(x) for x in sys.argv[1:]] or [8, 36]:
    batch = {k: v.to(dev) for k, v in make_batch(cfg, B, seed=1).items()}
    for _ in range(6):
        eng.train_step(batch)
    torch.cuda.synchronize()
    n = 20
    # (a) back to back: device time per step
    t0 = time.perf_counter()
    for _ in range(n):
        eng.train_step(batch)
    t_enq = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    # (b) a synchronisation before every step: the host's own time to get a step enqueued, by phase
    fb = st = 0.0
    for _ in range(n):
        torch.cuda.synchronize()
        a = time.perf_counter()
        eng.forward_backward(batch)
        b = time.perf_counter()
        eng.optimizer_step()
        c = time.perf_counter()
        fb += b - a
        st += c - b
    torch.cuda.synchronize()
    print("B=%d: %.2f ms per step back to back (host returned after %.2f ms per step); from an idle device the host needs %.2f ms "
          "for forward_backward (incl. its wait for the row counts) + %.2f ms for the optimizer" % (B, t_all * 1e3, t_enq * 1e3, fb / n * 1e3, st / n * 1e3))
