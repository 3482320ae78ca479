"""One rank of tests/test_gpu_round2.py::test_engine_under_two_ranks_matches_single_rank_accumulation (launched by
torch.distributed.run, two ranks sharing cuda:0, gloo collectives): PretrainEngine.train_step with world_size 2 --
the overlapped chunked backward + bucketed all-reduce of the flat gradient slab + fused AdamW -- on this rank's
shard; dumps the all-reduced gradients, the updated weights, the 7-tuple and the rank-averaged metrics."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, comm_dtype = sys.argv[1], sys.argv[2]
    mode = sys.argv[3] if len(sys.argv) > 3 else "mini"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)             # both ranks on the one GPU of the test box
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="gloo")
    from visitron_amd import ops
    from visitron_amd.config import mini_config
    from visitron_amd.distributed import all_reduce_metrics
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import deterministic_state_dict, make_batch
    from visitron_amd.training import PretrainEngine

    if mode == "base2":
        # tests/test_gpu_round3.py::test_engine_under_two_ranks_matches_the_oracle_gradients: the base LAYER shape on two
        # layers, the autotuner's own kernel choices, one layer per chunk (two chunk boundaries + the tail); compared
        # against the CPU oracle's autograd, not against another run of these kernels
        from visitron_amd.config import BertConfig

        cfg = BertConfig(num_hidden_layers=2, vocab_size=2048, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                         max_position_embeddings=64)
        m = PreTrainOscar(cfg)
        m.load_state_dict(deterministic_state_dict(m, seed=5, weight_std=0.03))
        bsz, text_len, region_len, per_chunk, bucket_mb, calls = 6, 40, 24, 1, 4.0, 4
    elif mode == "b36":
        # tests/test_gpu_round6.py: BASELINE configs[3]'s per-GPU shape -- the base config (12 layers), 36 x (128 + 100) per
        # rank, three layers per chunk as the bench runs it
        from visitron_amd.config import BertConfig

        cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        m = PreTrainOscar(cfg)
        m.load_state_dict(deterministic_state_dict(m, seed=5, weight_std=0.03))
        bsz, text_len, region_len, per_chunk, bucket_mb, calls = 36, 128, 100, 3, 25.0, 6
    else:
        ops.force_gemm_variant(1)             # one kernel variant everywhere: the comparison is then order-exact
        ops.set_wgrad_kernel(-8)
        cfg = mini_config(num_hidden_layers=4)
        m = PreTrainOscar(cfg)
        m.load_state_dict(deterministic_state_dict(m, seed=5))
        bsz, text_len, region_len, per_chunk, bucket_mb, calls = 3, 20, 10, 2, 0.05, 4
    m.tie_weights()
    m = m.to(dev).eval()
    eng = PretrainEngine(m, lr=1e-3, weight_decay=0.05, schedule="constant", warmup_steps=0, bucket_mb=bucket_mb,
                         grad_comm_dtype=comm_dtype)
    assert (eng.g16 is not None) == (comm_dtype == "bf16")
    eng.compact_min_rows = 0
    assert eng.world == world == 2
    shard = {k: v.to(dev) for k, v in make_batch(cfg, bsz, text_len=text_len, region_len=region_len, seed=100 + rank).items()}
    # the step, with the gradients captured between the all-reduce and AdamW -- which runs per arrived range (the layer
    # chunks in launch order, then the embeddings / heads tail): every element exactly once
    total = eng.flat.total
    captured = {"g": torch.zeros(total), "cover": torch.zeros(total, dtype=torch.bool), "calls": 0, "scale": None}
    adam_ranges = eng._adam_ranges

    def spy(consts, ranges, grad_scale, grads=None):
        torch.cuda.synchronize()
        assert (grads is not None) == (comm_dtype == "bf16")      # AdamW is handed the reduced bf16 copy
        src = eng.flat.g if grads is None else grads
        for s_, e_ in ranges:
            assert not bool(captured["cover"][s_:e_].any())
            captured["g"][s_:e_] = src[s_:e_].detach().float().cpu()
            captured["cover"][s_:e_] = True
        captured["scale"] = grad_scale
        captured["calls"] += 1
        return adam_ranges(consts, ranges, grad_scale, grads)

    eng._adam_ranges = spy
    out = eng.train_step(shard, overlap=True, layers_per_chunk=per_chunk)
    torch.cuda.synchronize()
    assert abs(captured["scale"] - 1.0 / world) < 1e-12
    assert bool(captured["cover"].all()) and captured["calls"] == calls   # the heads (first), the layer chunks, the tail
    assert eng.step_count == 1 and eng.sched_step == 1
    metrics = all_reduce_metrics([v if torch.is_tensor(v) else torch.tensor(float(v), device=dev) for v in out])
    torch.save({"g": captured["g"], "p": eng.flat.p.detach().cpu(), "out": [float(v) for v in out],
                "metrics": [float(v) for v in metrics],
                "names": [n for n, _, _, _, _ in eng.flat.entries],
                "ranges": [(o, o + cnt) for _, _, o, cnt, _ in eng.flat.entries]},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
