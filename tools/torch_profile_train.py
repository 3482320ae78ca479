"""torch.profiler view of one engine.train_step at the bench shape: which host-side torch ops launch kernels around the
HIP library calls (the 'glue' of DESIGN §5).  python tools/torch_profile_train.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd.config import BertConfig  # noqa: E402
from visitron_amd.modeling import PreTrainOscar  # noqa: E402
from visitron_amd.synth import make_batch  # noqa: E402
from visitron_amd.training import PretrainEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
cfg = BertConfig(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
torch.manual_seed(0)
full = PreTrainOscar(cfg).to(dev).train()
engine = PretrainEngine(full, lr=5e-5, weight_decay=0.05, eps=1e-8, schedule="linear", warmup_steps=0, t_total=20000)
batch = make_batch(cfg, B, 128, 100, seed=1234, device=dev, with_labels=True)
for _ in range(3):
    engine.train_step(batch)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    engine.train_step(batch)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = [e for e in ka if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
tot = 0.0
for e in rows[:70]:
    tot += e.device_time_total
    print("%-28s %3d calls %8.1f us   %s" % (e.key, e.count, e.device_time_total, str(e.input_shapes)[:110]))
print("sum of the listed aten ops: %.1f us" % tot)
