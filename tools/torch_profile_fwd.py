"""torch.profiler view of one trunk forward (which host-side torch ops launch copies / elementwise kernels around the HIP library calls)."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visitron_amd.config import BertConfig
from visitron_amd.modeling import PreTrainOscar
from visitron_amd.synth import make_batch
dev = torch.device("cuda:0")
cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
torch.manual_seed(0)
full = PreTrainOscar(cfg).eval().to(dev)
trunk = full.bert
batch = make_batch(cfg, 64, 128, 100, seed=1234, device=dev, with_labels=False)
with torch.no_grad():
    for _ in range(3): trunk(**batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    with torch.no_grad():
        trunk(**batch)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
