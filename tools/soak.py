"""A few hundred optimizer steps of the pretrain engine on a small fixed pool of synthetic batches (BASELINE configs[2]
shape, dropout 0.1, the reference's AdamW + warm-up schedule): the loss must stay finite and come down, the persistent
weight-gradient kernel must report no timed-out turn, and the eval-mode inference path must see the trained weights.
Usage: python tools/soak.py [steps] [batch] [text] [regions]      (prints one line per 25 steps; exit code 1 on a failed check)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops  # noqa: E402
from visitron_amd.config import BertConfig  # noqa: E402
from visitron_amd.modeling import PreTrainOscar  # noqa: E402
from visitron_amd.synth import make_batch  # noqa: E402
from visitron_amd.training import PretrainEngine  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    R = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = BertConfig(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    model = PreTrainOscar(cfg).to(dev).train()
    eng = PretrainEngine(model, lr=1e-4, warmup_steps=20, t_total=steps + 50)
    pool = [{k: v.to(dev) for k, v in make_batch(cfg, B, T, R, seed=100 + i).items()} for i in range(4)]
    # as the reference's loader leaves them (data_loader_pretrain.py:549-613): 80 % of the supervised positions carry [MASK]
    # (id 103), thousands of times per batch -- the long runs of the embedding-gradient kernel
    g = torch.Generator(device=dev).manual_seed(5)
    for b in pool:
        sup = b["labels"][:, :T] != -1
        hit = sup & (torch.rand(sup.shape, generator=g, device=dev) < 0.8)
        b["input_ids"][hit] = 103
    print("[MASK] positions per batch: %s" % [int((b["input_ids"] == 103).sum()) for b in pool], flush=True)
    model.eval()
    with torch.no_grad():
        before = float(model(**pool[0])[0])
    model.train()
    t0 = time.time()
    first = last = None
    for s in range(steps):
        out = eng.train_step(pool[s % len(pool)])
        if s % 25 == 0 or s == steps - 1:
            vals = [float(v) for v in out]
            if not all(v == v and abs(v) < 1e6 for v in vals[:4]):
                print("step %d: non-finite loss %s" % (s, vals))
                return 1
            first = vals[0] if first is None else first
            last = vals[0]
            print("step %4d  loss %.4f  mlm %.4f  action %.4f  token %.4f  acc %.3f/%.3f/%.3f  (%.1f s)"
                  % (s, *vals, time.time() - t0), flush=True)
    late = ops.wgrad_turn_timeouts()
    model.eval()
    with torch.no_grad():
        after = float(model(**pool[0])[0])
    print("eval loss on batch 0: %.4f before, %.4f after %d steps; wgrad turn time-outs: %d" % (before, after, steps, late))
    ok = last < first and after < before and late == 0
    print("soak %s" % ("ok" if ok else "FAILED"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
