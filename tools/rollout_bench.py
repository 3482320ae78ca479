"""Rollout caller at the reference's sizes (agent.py:110-125: trunk S=511 text-only, encoder LSTM 768->512, decoder
hidden 512, features 2048+4, 36 views): OscarEncoder.forward and one AttnDecoderLSTM step on the GPU, HIP-event timed,
with the per-kernel split from ops.profile_*, then one training iteration of the rollout's shape (agent.py:497-518: encoder,
`steps` teacher-forced decoder steps, cross-entropy, backward through all of it, Adam on both modules).
Usage: python tools/rollout_bench.py [B] [S] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops  # noqa: E402
from visitron_amd.config import BertConfig  # noqa: E402
from visitron_amd.modeling import BertImgModelwithLocationEmbeds  # noqa: E402
from visitron_amd.rollout import AttnDecoderLSTM, OscarEncoder  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, (time.perf_counter() - t0) * 1e3 / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 511
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    bert = BertImgModelwithLocationEmbeds(cfg).eval().to(dev)
    enc = OscarEncoder(None, bert, 512, 512, 0.5).eval().to(dev)
    dec = AttnDecoderLSTM(4, 64, 512, 0.5, feature_size=2048 + 4).eval().to(dev)
    g = torch.Generator().manual_seed(1)
    lengths = torch.sort(torch.randint(S // 4, S + 1, (B,), generator=g), descending=True).values
    lengths[0] = S
    ids = torch.randint(1000, cfg.vocab_size, (B, S), generator=g)
    pad = torch.arange(S)[None, :] >= lengths[:, None]
    ids[pad] = 0
    ids, mask = ids.to(dev), pad.byte().to(dev)
    out = {}

    def run_enc():
        with torch.no_grad():          # the reference evaluates under no_grad (agent.py:55)
            out["enc"] = enc(ids, lengths, mask)

    ms, wall = timed(run_enc, 5)
    print("OscarEncoder.forward  B=%d S=%d: %.2f ms GPU (%.2f ms wall) -> %.0f instructions/s  (rows below the lengths "
          "only: %d of %d)" % (B, S, ms, wall, B / ms * 1e3, int(lengths.sum()), B * S))
    enc.compact_rows = False
    ms2, _ = timed(run_enc, 5)
    enc.compact_rows = True
    print("   every padded row computed: %.2f ms" % ms2)
    ops.profile_begin()
    run_enc()
    for k, v in sorted(ops.profile_end().items(), key=lambda kv: -kv[1]["ms"]):
        print("   %-22s %8.3f ms  n=%d" % (k, v["ms"], v["n"]))
    ctx, h_t, c_t = out["enc"]
    action = torch.randn(B, 4, device=dev)
    feature = torch.randn(B, 36, 2052, device=dev).abs()
    cand = torch.randn(B, 12, 2052, device=dev).abs()
    h1 = h_t

    def run_dec():
        with torch.no_grad():
            out["dec"] = dec(action, feature, cand, h_t, h1, c_t, ctx, mask[:, : ctx.shape[1]])

    ms, wall = timed(run_dec, 20)
    print("AttnDecoderLSTM step  B=%d: %.3f ms GPU (%.3f ms wall)" % (B, ms, wall))
    ops.profile_begin()
    run_dec()
    for k, v in sorted(ops.profile_end().items(), key=lambda kv: -kv[1]["ms"]):
        print("   %-22s %8.3f ms  n=%d" % (k, v["ms"], v["n"]))

    # ---- one training iteration of the rollout (dropout 0.5 in the rollout modules as in the reference, 0.1 in the trunk)
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    del out["enc"], out["dec"]
    cfg_t = BertConfig(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    bert_t = BertImgModelwithLocationEmbeds(cfg_t).to(dev)
    enc_t = OscarEncoder(None, bert_t, 512, 512, 0.5).to(dev).train()
    dec_t = AttnDecoderLSTM(4, 64, 512, 0.5, feature_size=2048 + 4).to(dev).train()
    opt_e = torch.optim.Adam(enc_t.parameters(), lr=1e-4)
    opt_d = torch.optim.Adam(dec_t.parameters(), lr=1e-4)
    target = torch.randint(0, 12, (steps, B), device=dev)
    bmask = mask.bool()
    marks = {}

    def train_iter():
        opt_e.zero_grad()
        opt_d.zero_grad()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        ctx_t, h, c = enc_t(ids, lengths, bmask)
        ev[1].record()
        h1_, loss = h, 0.0
        for i in range(steps):
            h, c, logit, h1_ = dec_t(action, feature, cand, h1_, h, c, ctx_t, bmask[:, : ctx_t.shape[1]])
            loss = loss + torch.nn.functional.cross_entropy(logit, target[i])
        ev[2].record()
        loss.backward()
        ev[3].record()
        torch.nn.utils.clip_grad_norm_(enc_t.parameters(), 40.0)
        torch.nn.utils.clip_grad_norm_(dec_t.parameters(), 40.0)
        opt_e.step()
        opt_d.step()
        ev[4].record()
        marks["ev"], marks["loss"] = ev, loss

    ms, wall = timed(train_iter, 3)
    ev = marks["ev"]
    parts = [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
    print("rollout training iteration  B=%d S=%d, %d decoder steps: %.1f ms GPU (%.1f ms wall) = %.0f instructions/s; "
          "encoder forward %.1f, decoder steps forward %.1f, backward %.1f, clip + Adam %.1f ms; loss %.3f"
          % (B, S, steps, ms, wall, B / ms * 1e3, parts[0], parts[1], parts[2], parts[3], float(marks["loss"].detach())))
    print("   peak device memory %.1f GB" % (torch.cuda.max_memory_allocated() / 2 ** 30))
    ops.profile_begin()
    train_iter()
    for k, v in sorted(ops.profile_end().items(), key=lambda kv: -kv[1]["ms"]):
        print("   %-24s %8.3f ms  n=%d" % (k, v["ms"], v["n"]))


if __name__ == "__main__":
    main()
