"""One rank of tests/test_gpu_round4.py::test_two_rank_loss_curve_bf16_exchange_follows_fp32_exchange (torch.distributed.run, two
ranks sharing cuda:0, gloo collectives): six PretrainEngine.train_step calls under world_size 2 with the gradient exchange in
the given dtype; rank 0 dumps the rank-averaged 7-tuples of every step and the final weights' checksum."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, comm_dtype = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="gloo")
    from visitron_amd import ops
    from visitron_amd.config import mini_config
    from visitron_amd.distributed import all_reduce_metrics
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import deterministic_state_dict, make_batch
    from visitron_amd.training import PretrainEngine

    ops.force_gemm_variant(1)                 # the same kernels (and summation order) in both runs: only the exchange differs
    ops.set_wgrad_kernel(-8)
    cfg = mini_config(num_hidden_layers=4)
    m = PreTrainOscar(cfg)
    m.load_state_dict(deterministic_state_dict(m, seed=9))
    m.tie_weights()
    m = m.to(dev).eval()                      # eval: no dropout, the two runs see the same function
    eng = PretrainEngine(m, lr=2e-3, weight_decay=0.05, schedule="constant", warmup_steps=0, bucket_mb=0.05,
                         grad_comm_dtype=comm_dtype)
    eng.compact_min_rows = 0
    assert eng.world == world == 2
    pool = [{k: v.to(dev) for k, v in make_batch(cfg, 4, text_len=24, region_len=12, seed=300 + 2 * i + rank).items()}
            for i in range(2)]
    curve = []
    for step in range(6):
        out = eng.train_step(pool[step % 2], overlap=True, layers_per_chunk=2)
        met = all_reduce_metrics([v if torch.is_tensor(v) else torch.tensor(float(v), device=dev) for v in out])
        curve.append([float(v) for v in met])
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"curve": curve, "p": eng.flat.p.detach().cpu()}, os.path.join(out_dir, "curve_%s.pt" % comm_dtype))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
