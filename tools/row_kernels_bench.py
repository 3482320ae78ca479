"""Times the small row kernels of the training step's bookkeeping at the bench shape (B=256, S=228)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops
from visitron_amd.config import BertConfig
from visitron_amd.synth import make_batch
dev = torch.device("cuda:0")
cfg = BertConfig()
b = make_batch(cfg, 256, 128, 100, seed=1234, device=dev, with_labels=True)
B, S = 256, 228
lab, tl = b["labels"].reshape(-1).contiguous(), b["token_labels"].reshape(-1).contiguous()
mask = b["attention_mask"].float().contiguous()
err = torch.zeros(1, dtype=torch.int32, device=dev)
def timed(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
vals, tiles = ops.batch_row_counts(lab, tl, mask, err, B, S)
print("counts", vals)
print("batch_row_counts (+ zero fill, + read-back): %.1f us" % timed(lambda: ops.batch_row_counts(lab, tl, mask, err, B, S)))
print("batch_row_lists: %.1f us" % timed(lambda: ops.batch_row_lists(lab, tl, mask, B, S, vals[1], vals[2], vals[3], tiles)))
ids = b["input_ids"].reshape(-1)
de = torch.randn(ids.numel(), 768, device=dev)
grad = torch.zeros(cfg.vocab_size, 768, device=dev)
print("embed_table_grad (sort + runs): %.1f us" % timed(lambda: ops.embed_table_grad(ids, de, grad, skip_id=0)))
print("  of which torch.sort(int32, stable): %.1f us" % timed(lambda: torch.sort(ids.to(torch.int32), stable=True)))
ids_m = ids.clone(); ids_m[torch.randperm(ids.numel(), device=dev)[:4000]] = 103   # [MASK] 4000 times
print("embed_table_grad, 4000 x [MASK]: %.1f us" % timed(lambda: ops.embed_table_grad(ids_m, de, grad, skip_id=0)))
pos = (torch.arange(ids.numel(), device=dev) % 128)
g2 = torch.zeros(512, 768, device=dev)
print("embed_table_grad, position table (128 runs of %d): %.1f us" % (ids.numel() // 128, timed(lambda: ops.embed_table_grad(pos, de, g2))))
