import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops
dev = torch.device("cuda:0")
B, S, nh = 1, 64, 1
H = 64
g = torch.Generator().manual_seed(1)
drop = (0.2, 1234, ops.site_attn(2))
qkv = (torch.randn(B * S, 3 * H, generator=g) * 0.9).to(dev, torch.bfloat16)
words = torch.full((ops.keep_words(B, nh, S),), -1, dtype=torch.int32, device=dev)
lse = torch.zeros((B, nh, S), dtype=torch.float32, device=dev)
ops.attention_fwd(qkv, B, S, nh, lse=lse, drop=drop, keep_bits=words)
torch.cuda.synchronize()
nqb = 2
w = words.view(nqb, nqb * 32).cpu().to(torch.int64) & 0xFFFFFFFF
bits = ((w[:, None, :] >> torch.arange(32)[None, :, None]) & 1).reshape(64, 64).bool()   # [query][key]
want = ops.attn_dropout_mask(S, drop, 0, device=dev).bool().cpu()
print("equal frac", (bits == want).float().mean().item(), "transposed", (bits.t() == want).float().mean().item())
print("keep rate got", bits.float().mean().item(), "want", want.float().mean().item())
# per key column agreement
col = (bits == want).float().mean(0)
print("per-key agreement", [round(float(x), 2) for x in col])
row = (bits == want).float().mean(1)
print("per-query agreement", [round(float(x), 2) for x in row])
# try key permutations: for each got key column find the want column it equals
for k in range(32):
    m = [(k2, float((bits[:, k] == want[:, k2]).float().mean())) for k2 in range(64)]
    best = max(m, key=lambda t: t[1])
    print(k, best)
