import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from visitron_amd import ops
from visitron_amd.config import mini_config
from visitron_amd.modeling import PreTrainOscar
from visitron_amd.synth import deterministic_state_dict, make_batch
from visitron_amd.training import PretrainEngine
dev = torch.device("cuda:0")
cfg = mini_config()
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob = p, p
b = make_batch(cfg, 5, text_len=20, region_len=9, seed=11)
bd = {k: v.to(dev) for k, v in b.items()}
def run(unrolled, compact):
    m = PreTrainOscar(cfg)
    m.load_state_dict(deterministic_state_dict(m, seed=5, weight_std=0.05))
    m.tie_weights()
    m = m.to(dev).train()
    eng = PretrainEngine(m)
    eng.compact_rows = compact
    if unrolled:
        ops.profiling_was = ops.profiling
        ops.profiling = lambda: True
    out = eng.forward_backward(bd)
    if unrolled:
        ops.profiling = ops.profiling_was
    torch.cuda.synchronize()
    bufs = eng._buffers(5, 29)
    res = {"loss": torch.tensor([float(x) for x in out[:4]]), "g": eng.flat.g.clone()}
    n = eng.last_rows
    for l, d in enumerate(bufs.layers):
        for k in ("qkv", "ctx", "attn_pre", "attn_out", "mid", "out_pre", "out"):
            res["L%d.%s" % (l, k)] = d[k][:n].float().clone()
    return res
for compact in (False, True):
    a = run(False, compact)
    if len(sys.argv) > 2:
        ops.LN_RESIDUAL = False
        run(False, compact)
        ops.LN_RESIDUAL = True
    c = run(True, compact)
    print("compact", compact)
    for k in a:
        d = float((a[k] - c[k]).abs().max())
        if d > 0:
            print("   %-14s max diff %.3e (scale %.3e)" % (k, d, float(a[k].abs().max())))
