"""torch.profiler view of one trunk forward (inference path, bench --mode fwd shape): the torch ops that launch kernels
around the HIP library calls.  python tools/torch_profile_fwd.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd.config import BertConfig  # noqa: E402
from visitron_amd.modeling import PreTrainOscar  # noqa: E402
from visitron_amd.synth import make_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
cfg = BertConfig(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
torch.manual_seed(0)
trunk = PreTrainOscar(cfg).eval().to(dev).bert
batch = make_batch(cfg, B, 128, 100, seed=1234, device=dev, with_labels=False)
with torch.no_grad():
    for _ in range(3):
        trunk(**batch)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    with torch.no_grad():
        trunk(**batch)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = [e for e in ka if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.device_time_total)
tot = 0.0
for e in rows[:50]:
    tot += e.device_time_total
    print("%-28s %3d calls %8.1f us   %s" % (e.key, e.count, e.device_time_total, str(e.input_shapes)[:120]))
print("sum of the listed aten ops: %.1f us" % tot)
print("---- device kernels")
ks = [e for e in prof.key_averages() if e.device_type is not None and e.device_time_total > 0 and not e.key.startswith("aten::")]
ks.sort(key=lambda e: -e.device_time_total)
for e in ks[:30]:
    print("%-90s %3d %8.1f us" % (e.key[:90], e.count, e.device_time_total))
