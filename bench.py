#!/usr/bin/env python
"""Headline bench of the encoder hot path (BASELINE.json metric / SURVEY.md section 8d).

  python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in
the environment, one rank per GPU), or started plainly -- then this process starts that launcher itself as a CHILD
process before anything here touches the GPU, relays its output and exits with its code.

A step = one pass of the hot path over one synthetic batch that is already resident in HBM:
  --mode train  (default) the pretrain step of tasks/viewpoint_select/pretrain.py:150-193 on
                PreTrainOscar: forward (trunk + MLM / region-token / action heads + losses), backward,
                gradient all-reduce over the data-parallel group (RCCL), fused AdamW, LR schedule;
                bf16 compute, B=256 x (128 text + 100 region) per GPU [BASELINE configs[2]], weak scaling
  --mode fwd    BertImgModelwithLocationEmbeds.forward (embeddings + region projection + 12-layer
                encoder + pooler), same shapes   [BASELINE configs[1]]
The train-mode line also carries the forward-only rate measured in the same process.
Prints ONE JSON line (rank 0) with whole-job samples/s, the live per-kernel roofline of the
dominant kernel (the MFMA GEMM), and -- at N=1 -- the CPU oracle timed on the host cores.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0


def enc_flops_per_seq(S, H=768, L=12, I=3072):
    """F_enc(S) = L * (24 S H^2 + 4 S^2 H) for I = 4H (multiply-add = 2)."""
    return L * (2 * S * H * (3 * H) + 2 * S * H * H + 2 * 2 * S * H * I + 4 * S * S * H)


def enc_flops_rows(lens, H=768, L=12, I=3072):
    """The same count over the rows actually computed: sequences of `lens` real positions each (padding rows dropped)."""
    rows = float(sum(lens))
    return L * ((2 * H * 3 * H + 2 * H * H + 4 * H * I) * rows + 4 * H * float(sum(n * n for n in lens)))


def _self_launch(a):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a child (nothing in this process
    has touched the GPU yet -- not even torch is imported), relay its output, exit with its code."""
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def _pct(xs, q):
    xs = sorted(xs)
    if not xs:
        return None
    k = (len(xs) - 1) * q
    lo, hi = int(k), min(int(k) + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (k - lo)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (SURVEY 8d protocol: 20 warm-up + 100 timed)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", default="train", choices=["train", "fwd"])
    ap.add_argument("--batch", type=int, default=None,
                    help="sequences per GPU (default: 256 in train mode = BASELINE configs[2], 64 in fwd mode = configs[1])")
    ap.add_argument("--text", type=int, default=128)
    ap.add_argument("--regions", type=int, default=100)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: all ranks use cuda:0")
    ap.add_argument("--dropout", type=float, default=0.1,
                    help="hidden / attention-probs dropout in the train step (reference defaults: --drop_out 0.1, "
                         "params.py:299, and BERT-base attention_probs_dropout_prob 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-fwd-rate", action="store_true",
                    help="skip the forward-only rate measured after the timed region (profiler runs: the trace then "
                         "holds the timed step's kernels only)")
    a = ap.parse_args()
    if a.batch is None:
        a.batch = 256 if a.mode == "train" else 64
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(a)
    global torch
    import torch

    t_start = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if a.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend=a.backend)
    assert a.gpus == world, "--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (a.gpus, world)

    from visitron_amd import ops
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds, PreTrainOscar
    from visitron_amd.synth import make_batch
    from visitron_amd.training import PretrainEngine

    cfg = BertConfig(hidden_dropout_prob=a.dropout, attention_probs_dropout_prob=a.dropout)
    torch.manual_seed(0)  # reference init N(0, 0.02) (BertPreTrainedModel.init_weights), identical on every rank
    S = a.text + a.regions
    train = a.mode == "train"
    if train:
        full = PreTrainOscar(cfg).to(dev).train()
        trunk = full.bert
        # pretrain.py defaults: lr 5e-5, weight_decay 0.05 (scripts), adam eps 1e-8, linear schedule
        engine = PretrainEngine(full, lr=5e-5, weight_decay=0.05, eps=1e-8, schedule="linear", warmup_steps=0,
                                t_total=20000)
        batch = make_batch(cfg, a.batch, a.text, a.regions, seed=1234 + rank, device=dev, with_labels=True)
        fwd_batch = {k: batch[k] for k in ("input_ids", "attention_mask", "img_feats", "img_location_embeddings") if k in batch}   # (--regions 0: text only)

        def step():
            return engine.train_step(batch)
    else:
        full = PreTrainOscar(cfg).eval().to(dev)   # the timed call is the trunk; the heads serve the logits diff below
        trunk = full.bert
        batch = make_batch(cfg, a.batch, a.text, a.regions, seed=1234 + rank, device=dev, with_labels=False)
        fwd_batch = batch

        def step():
            with torch.no_grad():
                return trunk(**batch)

    def fwd_step():
        was = trunk.training
        trunk.eval()
        with torch.no_grad():
            out = trunk(**fwd_batch)
        trunk.train(was)
        return out

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print("[bench] %s (t=%.1f s)" % (msg, time.perf_counter() - t_start), file=sys.stderr, flush=True)

    note("model built")
    for _ in range(a.warmup):
        step()
    sync_all()
    note("warm-up done")
    # HIP events around every step, on the stream the kernels are launched on (torch's current stream: the library is
    # handed torch.cuda.current_stream() with every call) -> median / p10 / p90 per step; `value` is the wall clock over
    # the whole region between the two barrier + synchronize brackets
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(a.steps):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps)]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    sync_all()

    ms_per_step = elapsed / a.steps * 1e3
    value = world * a.batch * a.steps / elapsed
    note("timed region done: %.3f ms/step" % ms_per_step)
    f_enc = enc_flops_per_seq(S, cfg.hidden_size, cfg.num_hidden_layers, cfg.intermediate_size)

    # the same K steps with every padded row computed (the reference's own amount of work per step): reported beside
    # `value`, not part of the timed region above
    value_all_rows = None
    # (the condition must not depend on a rank's own batch: every rank runs the same number of collective steps)
    if train and engine.compact_rows and a.batch * S >= engine.compact_min_rows and not a.no_fwd_rate:
        engine.compact_rows = False
        for _ in range(2):
            step()
        sync_all()
        t2 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t2
        if dist is not None:
            t = torch.tensor([el2], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        value_all_rows = world * a.batch * a.steps / el2
        engine.compact_rows = True
        step()                      # back on the default path (and its row count) for what follows
        sync_all()

    # forward-only rate in the same process (train mode): not part of the timed region above
    fwd_value = None
    if train and not a.no_fwd_rate:
        for _ in range(3):
            fwd_step()
        sync_all()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            fwd_step()
        torch.cuda.synchronize()
        fwd_elapsed = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([fwd_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fwd_elapsed = float(t.item())
        fwd_value = world * a.batch * a.steps / fwd_elapsed
        sync_all()

    # BASELINE configs[1] under the same clock (train mode, one GPU): the trunk forward at B = 64 -- the number north_star's
    # 40 % target is quoted on -- with its HIP-vs-oracle differences; outside the timed region above
    fwd_b64, fwd_b64_out = None, None
    if train and world == 1 and not a.no_fwd_rate:
        b64 = make_batch(cfg, 64, a.text, a.regions, seed=1234 + rank, device=dev, with_labels=False)
        was = trunk.training
        full.eval()
        with torch.no_grad():
            for _ in range(10):
                trunk(**b64)
            sync_all()
            e64 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
            t3 = time.perf_counter()
            e64[0].record()
            for i in range(a.steps):
                trunk(**b64)
                e64[i + 1].record()
            torch.cuda.synchronize()
            el64 = time.perf_counter() - t3
            ms64 = [e64[i].elapsed_time(e64[i + 1]) for i in range(a.steps)]
            if not a.no_cpu_baseline:
                fwd_b64_out = hip_outputs_for_diff(full, trunk, b64)
        full.train(was)
        rate64 = 64 * a.steps / el64
        fwd_b64 = {
            "workload": "trunk forward (embeddings + region projection + encoder + pooler), batch 64 x (%d text + %d region "
                        "tokens)%s; same process, after the timed region"
                        % (a.text, a.regions, " [BASELINE configs[1]]" if (a.text, a.regions) == (128, 100) else ""),
            "samples_per_sec": round(rate64, 2), "ms_per_forward": round(el64 / a.steps * 1e3, 4),
            "ms_per_forward_hip_events": {"median": round(_pct(ms64, 0.5), 4), "p10": round(_pct(ms64, 0.1), 4),
                                          "p90": round(_pct(ms64, 0.9), 4), "n": len(ms64)},
            "mfma_frac": round(f_enc * rate64 / (PEAK_BF16_TFLOPS * 1e12), 4), "steps": a.steps, "warmup": 10,
            "residual_stream": ("fp16, LayerNorms deferred into the GEMM epilogues" if trunk.encoder.serves_deferred_ln()
                                else "bf16 (seven-launch layer, LayerNorm passes)"),
        }
        sync_all()

    # BASELINE configs[3]'s per-GPU share under the same clock (train mode, one GPU, default shapes): the pretrain step at
    # B = 36 (8 x 36 viewpoint candidates over 8 GPUs) -- the number that bounds the strong scaling of a 288-sequence job;
    # same engine and weights, 10 warm-up (tunes the B = 36 shapes) + 20 timed steps, outside the timed region above
    b36 = None
    if train and world == 1 and not a.no_fwd_rate and a.batch != 36:
        bt36 = make_batch(cfg, 36, a.text, a.regions, seed=1234 + rank, device=dev, with_labels=True)
        for _ in range(10):
            engine.train_step(bt36)
        sync_all()
        e36 = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        t4 = time.perf_counter()
        e36[0].record()
        for i in range(20):
            engine.train_step(bt36)
            e36[i + 1].record()
        torch.cuda.synchronize()
        el36 = time.perf_counter() - t4
        ms36 = [e36[i].elapsed_time(e36[i + 1]) for i in range(20)]
        rate36 = 36 * 20 / el36
        b36 = {
            "workload": "pretrain step (fwd + bwd + fused AdamW), batch 36 x (%d text + %d region tokens) = BASELINE configs[3]'s "
                        "per-GPU share (8 x 36 candidates over 8 GPUs); same process and engine, after the timed region"
                        % (a.text, a.regions),
            "samples_per_sec": round(rate36, 2), "ms_per_step": round(el36 / 20 * 1e3, 4),
            "ms_per_step_hip_events": {"median": round(_pct(ms36, 0.5), 4), "p10": round(_pct(ms36, 0.1), 4),
                                       "p90": round(_pct(ms36, 0.9), 4), "n": 20},
            "mfma_frac_whole_step": round(3 * f_enc * rate36 / (PEAK_BF16_TFLOPS * 1e12), 4),
            "token_rows": {"padded": 36 * S, "computed": engine.last_rows},
            "steps": 20, "warmup": 10,
            # 8 GPUs x this rate over the one-GPU rate of the 288-sequence job is the ceiling of configs[3]'s strong scaling
            # before any communication; `value` above (B = 256) stands in for the 288-sequence rate within 1 %
            "strong_scaling_ceiling_8gpu": round(8 * rate36 / value, 2) if a.batch >= 256 else None,
        }
        for _ in range(2):
            step()                  # back on the timed batch's shapes for the kernel-timing replay below
        sync_all()

    # N > 1: what the collective actually saw, so that a SCALE record proves itself -- backend, ranks, DISTINCT devices,
    # library version, bytes per step -- and the exposed communication: the same step with the gradient exchange left out
    collective = None
    if dist is not None:
        pr = torch.cuda.get_device_properties(dev)
        ident = "%s|%s|%s:%s:%s" % (socket.gethostname(), getattr(pr, "uuid", None), getattr(pr, "pci_domain_id", None),
                                    getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None))
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        backend = dist.get_backend()
        try:
            ccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            ccl_version = None
        ms_nocomm = None
        if train:
            noop = lambda rng: None
            for _ in range(2):
                engine.train_step(batch, _force_comm=noop)
            sync_all()
            t5 = time.perf_counter()
            nn_ = max(5, min(a.steps, 20))
            for _ in range(nn_):
                engine.train_step(batch, _force_comm=noop)
            torch.cuda.synchronize()
            el5 = time.perf_counter() - t5
            t = torch.tensor([el5], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms_nocomm = float(t.item()) / nn_ * 1e3
            sync_all()
        collective = {
            "backend": backend, "library": ("RCCL" if backend == "nccl" else backend), "library_version": (ccl_version if backend == "nccl" else None),
            "world_size": dist.get_world_size(), "distinct_devices": len(set(idents)), "devices": idents,
            "all_ranks_on_one_gpu": bool(a.share_gpu),
            "bytes_reduced_per_step": (int(engine.flat.total) * (2 if engine.g16 is not None else 4) if train else 0),
            "grad_comm_dtype": (engine.grad_comm_dtype if train else None),
            "ms_per_step_without_gradient_exchange": None if ms_nocomm is None else round(ms_nocomm, 4),
            "exposed_communication_ms": None if ms_nocomm is None else round(ms_per_step - ms_nocomm, 4),
            "env": {k: os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "VT_GEMM_RESERVE_CUS", "VT_GRAD_COMM")},
        }

    note("side measurements done")
    # ---- live per-kernel timing (HIP events on the launch stream), outside the timed region ----
    roofline = None
    kernels = None
    hbm_step_bytes, traffic_src = None, None
    if rank == 0 and not a.no_kernel_timing:
        ops.profile_begin()
        for _ in range(3):
            if train:
                engine.forward_backward(batch)
                engine.optimizer_step()
            else:
                step()
        torch.cuda.synchronize()
        kernels = ops.profile_end()
        gemm = kernels.get("gemm_nt_bf16")
        if gemm and gemm["ms"] > 0:
            achieved = gemm["flops"] / (gemm["ms"] * 1e-3) / 1e12
            # HBM bytes per launch of the same kernels: not measurable from inside this process; taken from the
            # committed PMC passes of this command (profiles/README.md), null when that file is absent
            traffic, traffic_src = None, None
            here = os.path.dirname(os.path.abspath(__file__))
            # (the PMC passes of THIS configuration: train B=256 or forward B=64, default shapes; newest round first)
            default_shape = a.text == 128 and a.regions == 100 and a.batch == (256 if train else 64)
            rels = ()
            if default_shape:
                name = "train_b256_pmc_hbm_traffic.json" if train else "fwd_b64_pmc_hbm_traffic.json"
                rels = ("profiles/r06/" + name, "profiles/r05/" + name)
            elif train and a.text == 512 and a.regions == 144 and a.batch == 64:   # BASELINE configs[4]'s shape on one GPU
                rels = ("profiles/r06/cfg5_pmc_hbm_traffic.json", "profiles/r05/cfg5_pmc_hbm_traffic.json")
            for rel in rels:
                tp = os.path.join(here, rel)
                if os.path.exists(tp):
                    pmc = json.load(open(tp))
                    if pmc.get("kernel_tree") != kernel_tree_stamp():
                        traffic_src = rel + " (NOT reported: taken on another tree, kernel_tree %s)" % pmc.get("kernel_tree")
                        continue
                    traffic, traffic_src = pmc["hbm_bytes_per_launch"], rel
                    hbm_step_bytes = pmc.get("all_kernels_hbm_bytes_per_step")
                    break
            roofline = {
                "kernel": "gemm_nt_bf16 (NT GEMM family; per shape the autotuner picks among the persistent 256x256-tile "
                          "kernel with 128x128 wave tiles / AGPR accumulators and the older 128x128 .. 256x256 tiles)",
                "bound": "mfma", "achieved": round(achieved, 2),
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
                "launches": gemm["n"], "avg_us": round(gemm["ms"] * 1e3 / gemm["n"], 2),
                "flops_per_launch": gemm["flops"] / gemm["n"],
            }

    # FLOPs of the rows actually computed (training drops the padding rows): per step and GPU
    f_exec = None
    if train and engine.last_layout is not None:
        f_exec = 3 * enc_flops_rows(engine.last_layout.length.tolist(), cfg.hidden_size, cfg.num_hidden_layers,
                                    cfg.intermediate_size)
    note("kernel timing done")
    cpu_baseline = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        gpu_out = fwd_b64_out
        if not train:   # the first two sequences of the timed batch: HIP outputs for the logits diff against the oracle
            with torch.no_grad():
                gpu_out = hip_outputs_for_diff(full, trunk, batch)
        cpu_baseline = run_cpu_baseline(cfg, a.text, a.regions, train, gpu_out)
        if train and fwd_b64 is not None and cpu_baseline is not None:
            fwd_b64["max_abs_diff_hip_vs_oracle"] = cpu_baseline.pop("max_abs_diff_hip_vs_oracle", None)

    if rank == 0:
        out = {
            "metric": ("encoder fwd+bwd samples/sec (seq=%d, 12L base)" if train else "encoder fwd samples/sec (seq=%d, 12L base)") % S,
            "value": round(value, 2),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": ("oscar base (12L/768d/12h), %d text + %d region tokens, batch %d per GPU; " % (a.text, a.regions, a.batch))
                            + ("pretrain step = PreTrainOscar fwd (trunk + MLM/region-token/action heads, losses) + bwd + "
                               "gradient all-reduce + fused AdamW [BASELINE configs[2]: pretrain step, batch 256 per GPU]" if train else
                               "trunk forward (embeddings + region projection + encoder + pooler) [BASELINE configs[1]]"),
                "global_batch": world * a.batch, "seq_len": S,
                # names the library the collective really ran on (gloo = a rehearsal, never a scaling number)
                "parallelism": (("dp%d (flat-slab gradient all-reduce over " + (
                    "RCCL" if (dist is None or dist.get_backend() == "nccl") else dist.get_backend() + ": REHEARSAL, not RCCL")
                    + (", all ranks sharing ONE GPU" if a.share_gpu else "") + ")") if train else "dp%d (replicas, no collective)") % world,
                # what the NT GEMM does beside the collective's kernels (ops.multi_rank_gemm_policy: decided from the
                # environment RCCL reads, no hand-set knob)
                "gemm_beside_collective": (engine.gemm_policy if train else None),
                # the data-parallel exchange moves a bf16 copy of the gradient slab by default (VT_GRAD_COMM=fp32: the fp32 slab)
                "grad_comm_dtype": (engine.grad_comm_dtype if (train and world > 1) else None),
                "weights": "random init N(0,0.02), seed 0",
                "dropout": a.dropout if train else 0.0,
                # the attention-probability dropout runs p quantised to 1/256 (one hash word per four keys)
                "attention_dropout_effective": (ops.attn_drop_p(a.dropout) if train else 0.0),
                # bf16 operands on the matrix cores, fp32 accumulation everywhere.  Inference keeps the residual stream
                # between sub-layers in fp16 (+ a bf16 copy for the next GEMM) and defers every LayerNorm into the GEMM
                # epilogues that consume it; training keeps bf16 activations and LayerNorm passes (its backward reads them) with
                # the pre-LayerNorm sums and the residual operands as fp16 (VT_F16_STREAM=0: all bf16)
                "residual_stream": (((("fp16 pre-LayerNorm sums; each residual add rebuilds the previous LayerNorm from them and its saved "
                                       "row statistics (seven-launch layer, one-output LayerNorm passes)") if ops.LN_RESIDUAL else
                                      "fp16 copies beside the bf16 GEMM operands (seven-launch layer, LayerNorm passes)") if ops.F16_STREAM
                                     else "bf16 (seven-launch layer, LayerNorm passes)") if train else
                                    ("fp16, LayerNorms deferred into the GEMM epilogues" if trunk.encoder.serves_deferred_ln()
                                     else "bf16 (seven-launch layer, LayerNorm passes)")),
                # training computes the real rows only: positions with attention mask 0 (text tails, missing regions;
                # 12.5 % of this synthetic batch, as in the reference's data) are read by nothing in the step -- same
                # losses and gradients as the padded run (tests/test_gpu_train.py); VT_COMPACT_ROWS=0 computes them all
                "token_rows_per_gpu": {"padded": a.batch * S, "computed": (engine.last_rows if train else a.batch * S)},
            },
            "encoder_flops_per_seq_fwd": f_enc,
            "ms_per_step_hip_events": {"median": round(_pct(step_ms, 0.5), 4), "p10": round(_pct(step_ms, 0.1), 4),
                                       "p90": round(_pct(step_ms, 0.9), 4), "n": len(step_ms)},
            # algorithmic FLOPs of the PADDED batch (every position, as the reference computes it) over the measured time
            "mfma_frac_whole_step": round((3 if train else 1) * f_enc * value / world / (PEAK_BF16_TFLOPS * 1e12), 4),
            # FLOPs of the rows actually executed (training skips the padding rows): the like-for-like roofline fraction
            "mfma_frac_whole_step_executed": (None if f_exec is None else
                                              round(f_exec / (ms_per_step * 1e-3) / (PEAK_BF16_TFLOPS * 1e12), 4)),
            "value_all_padded_rows_computed": None if value_all_rows is None else round(value_all_rows, 2),
            "fwd_samples_per_sec": None if fwd_value is None else round(fwd_value, 2),
            "fwd_b64": fwd_b64,
            "b36": b36,
            "collective": collective,
            "mfma_frac_whole_forward": (None if (train and fwd_value is None) else
                                        round(f_enc * (fwd_value if train else value) / world / (PEAK_BF16_TFLOPS * 1e12), 4)),
            "roofline": roofline,
            # the whole step's HBM traffic (every kernel of the committed PMC passes of this command, FETCH_SIZE doubled per
            # the gfx950 correction, Infinity-Cache hits included in FETCH_SIZE: an upper bound on DRAM bytes) over the time
            # measured here, against the 8 TB/s peak; null without a committed profile of this configuration
            "hbm": None if hbm_step_bytes is None else {
                "bytes_per_step": hbm_step_bytes, "gbps": round(hbm_step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                "peak_gbps": 8000.0, "frac": round(hbm_step_bytes / (ms_per_step * 1e-3) / 8e12, 4), "source": traffic_src},
            # the autotuner's pick per (M, N, K, kind) -- kind = act | 16 residual | 32 second output | 64 fp32 out |
            # deferred-LayerNorm mode << 8; kernel variant numbers as in csrc/gemm_bf16.hip
            "gemm_variants": {"%d,%d,%d,%d" % k: v for k, v in sorted(ops._tuned.items())},
            "device": device_info(dev),
            "cpu_baseline": cpu_baseline,
            "kernels_ms_per_step": None if kernels is None else {
                k: round(v["ms"] / 3, 4) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])},
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()   # rank 0's kernel-timing replay and print are done before any rank tears the group down
        dist.destroy_process_group()


def kernel_tree_stamp():
    """sha256 (16 hex) over the kernel sources and the Python that picks kernels: the identity of the tree a PMC profile was
    taken on (there is no .git on the GPU box).  profiles/*/..._pmc_hbm_traffic.json carry it; a profile from another tree is
    not reported as this run's traffic."""
    import hashlib

    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(here, "visitron_amd", "csrc", "*.h*")) + glob.glob(os.path.join(here, "visitron_amd", "csrc", "*.inc")))
    files += [os.path.join(here, "visitron_amd", f) for f in ("ops.py", "training.py", "modeling.py", "gemm_defaults.json")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def hip_outputs_for_diff(full, trunk, batch):
    """HIP outputs of the first two sequences of `batch` (alone, and inside the full batch) with the weights that produced
    them, for the max-abs comparison against the CPU oracle in run_cpu_baseline.  Call under torch.no_grad() in eval()."""
    sub = {k: v[:2] for k, v in batch.items()}
    outs, pooled, _, B2, S2 = full.bert.run_trunk(
        sub["input_ids"], attention_mask=sub["attention_mask"], img_feats=sub["img_feats"],
        img_location_embeddings=sub["img_location_embeddings"])
    sc, _, act = full.head_outputs(outs[-1], pooled)
    seq_sub = trunk(**sub)[0]                    # what a caller of the trunk is handed (fp32)
    seq_full = trunk(**batch)[0][:2]             # the same two sequences inside the full timed batch must give the same rows
    return dict(batch={k: v.cpu() for k, v in sub.items()}, state={k: v.detach().cpu() for k, v in full.state_dict().items()},
                sequence_output=seq_sub.float().cpu().view(B2, S2, -1), prediction_scores=sc.float().cpu().view(B2, S2, -1),
                action_scores=act.float().cpu(), sequence_output_in_full_batch=seq_full.float().cpu())


def device_info(dev):
    """What the roofline constants refer to: the device as the runtime reports it (name, CU count, peak shader clock)
    beside the peak used (dense bf16 MFMA, MI355X_MICROARCH.md)."""
    pr = torch.cuda.get_device_properties(dev)
    info = {"name": pr.name, "compute_units": pr.multi_processor_count, "peak_bf16_tflops_used": PEAK_BF16_TFLOPS,
            "hbm_gib": round(pr.total_memory / 2 ** 30, 1)}
    clk = getattr(pr, "clock_rate", None)
    if clk:
        info["max_shader_clock_mhz"] = round(clk / 1e3)
    return info


def run_cpu_baseline(cfg, T, R, train, gpu_out=None):
    """The oracle (a port of the reference's CPU path) on this box's host cores: bounded samples of the same step
    (train: PreTrainOscar forward + backward + pytorch-transformers AdamW; fwd: trunk forward) at B=2 (BASELINE
    configs[0]) and B=16.  In fwd mode it also checks the HIP outputs of two sequences of the timed batch
    against the oracle on the same weights (max |diff| of sequence_output / prediction_scores / action_scores)."""
    from oracle.modeling import BertImgModelwithLocationEmbeds as OracleTrunk
    from oracle.modeling import PreTrainOscar as OraclePreTrain
    from oracle.optim import AdamW, grouped_parameters
    from visitron_amd.synth import make_batch

    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = logical
    # the container's CPU share (cgroup quota): a GPU box hands one GPU's share of the host cores to this process, and
    # more threads than that share only fight over it (a first attempt with one thread per visible core did not finish)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(round(float(q) / float(per))))
    except Exception:
        try:   # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, int(round(q / per)))
        except Exception:
            pass
    # no visible quota: torch's CPU GEMMs at these sizes stop scaling around 32 threads (round 1: 32 threads finished
    # 14 steps in 12 s; one thread per visible core of the GPU host did not finish a step in minutes)
    threads = min(usable, quota) if quota else min(usable, 32)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    points = []
    model = (OraclePreTrain(cfg).train() if train else OracleTrunk(cfg).eval())
    opt = AdamW(grouped_parameters(model, 0.05), lr=5e-5, eps=1e-8) if train else None
    for B, budget_s in ((2, 6.0), (16, 12.0)):
        b = make_batch(cfg, B, T, R, seed=1234, with_labels=train)

        def it():
            if train:
                model.zero_grad()
                model(**b)[0].backward()
                opt.step()
            else:
                with torch.no_grad():
                    model(**b)
        tw = time.perf_counter()
        it()  # warm-up
        tw = time.perf_counter() - tw
        print("[bench] cpu baseline B=%d: warm-up iteration %.1f s (%d threads)" % (B, tw, threads), file=sys.stderr, flush=True)
        n, t0 = 0, time.perf_counter()
        while n < 1 or (time.perf_counter() - t0 < budget_s and n < 50):
            it()
            n += 1
        dt = time.perf_counter() - t0
        print("[bench] cpu baseline B=%d: %d iterations in %.1f s" % (B, n, dt), file=sys.stderr, flush=True)
        points.append({"batch": B, "value": round(B * n / dt, 3), "iterations": n, "seconds": round(dt, 1)})
    what = ("oracle fp32 pretrain step (fwd + bwd + AdamW, torch CPU ops)" if train else
            "oracle fp32 trunk forward (torch CPU ops)")
    best = max(points, key=lambda p_: p_["value"])
    out = {
        "value": best["value"], "unit": "samples/s", "cores": threads, "kind": "port",
        "sample": "%s, S=%d: %s" % (what, T + R, "; ".join("B=%d: %d iterations in %.1f s = %.3f samples/s" % (
            p_["batch"], p_["iterations"], p_["seconds"], p_["value"]) for p_ in points)),
        "points": points, "host_logical_cpus": logical, "host_usable_cpus": usable, "cgroup_cpu_quota": quota,
    }
    if gpu_out is not None:
        ref = OraclePreTrain(cfg).eval()
        ref.load_state_dict(gpu_out["state"])
        with torch.no_grad():
            bb = gpu_out["batch"]
            seq, pooled = ref.bert(**bb)[:2]
            scores, _, act = ref.heads(seq, pooled)
        d = lambda x, y: round(float((x.reshape(y.shape) - y).abs().max()), 6)
        out["max_abs_diff_hip_vs_oracle"] = {
            "what": "first 2 sequences of the timed batch, same weights: HIP bf16 path vs CPU fp32 oracle",
            "sequence_output": d(gpu_out["sequence_output"], seq), "prediction_scores": d(gpu_out["prediction_scores"], scores),
            "action_scores": d(gpu_out["action_scores"], act),
            "sequence_output_rows_inside_the_full_batch": d(gpu_out["sequence_output_in_full_batch"], seq),
            "sequence_output_rms": round(float((gpu_out["sequence_output"].reshape(seq.shape) - seq).pow(2).mean().sqrt()), 6),
            "sequence_output_absmax_of_reference": round(float(seq.abs().max()), 4),
            "tolerance": 5e-2,
        }
    return out


if __name__ == "__main__":
    main()
