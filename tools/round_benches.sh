#!/bin/bash
# The round's bench lines and profile sets (run on the GPU box from the repo root):  bash tools/round_benches.sh r06
# Profiles first (the bench lines read the HBM figures of THIS tree's passes: bench.kernel_tree_stamp), then the lines.
set -o pipefail
R=${1:-r06}
O=gpurun_out/${R}_benches
mkdir -p $O profiles/$R
bash tools/profile_round.sh $R > $O/profile_train.log 2>&1 && echo "profile train done" >> $O/progress.txt
bash tools/profile_round.sh $R fwd_b64 --mode fwd > $O/profile_fwd.log 2>&1 && echo "profile fwd done" >> $O/progress.txt
bash tools/profile_round.sh $R cfg5 --batch 64 --text 512 --regions 144 > $O/profile_cfg5.log 2>&1 && echo "profile cfg5 done" >> $O/progress.txt
# (profiles/ on the GPU box is not merged back: after the call, copy gpurun_out/prof_$R/* and $O/bench_*.json into profiles/$R/ in the build container)
cp gpurun_out/prof_$R/*_pmc_hbm_traffic.json gpurun_out/prof_$R/*_pmc_hbm_traffic.csv gpurun_out/prof_$R/*_kernel_stats.csv gpurun_out/prof_$R/tune_*.json gpurun_out/prof_$R/bench_*_live.json profiles/$R/ 2>/dev/null
# kernel-trace summary of the B = 36 step (configs[3]'s per-GPU share): gpurun_out/prof_$R/b36_kernel_stats.csv
bash tools/trace_step.sh prof_$R b36 --batch 36 > $O/trace_b36.log 2>&1 && echo "trace b36 done" >> $O/progress.txt
cp gpurun_out/prof_$R/b36_kernel_stats.csv profiles/$R/ 2>/dev/null
b() { name=$1; shift; python bench.py "$@" --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; echo $name >> $O/progress.txt; }
b b36 --batch 36 --steps 50 --warmup 10
# configs[3]'s per-GPU share under each multi-rank GEMM policy, on one rank (the kernel mix an 8-GPU run executes)
VT_FORCE_MULTI_RANK_GEMM=1 b b36_policy_stood_down --batch 36 --steps 50 --warmup 10
VT_FORCE_MULTI_RANK_GEMM=1 VT_GEMM_RESERVE_CUS=16 b b36_policy_reserved16 --batch 36 --steps 50 --warmup 10
b b288 --batch 288 --steps 30 --warmup 8
b fwd256 --mode fwd --batch 256 --steps 50 --warmup 10
b cfg5 --batch 64 --text 512 --regions 144 --steps 30 --warmup 8
b cfg5f --mode fwd --batch 64 --text 512 --regions 144 --steps 50 --warmup 10
# the reference's shipped shapes: pretrain 511 + 256 at the shipped per-GPU batch x 4, rollout text-only T = 511
b shipped_s767_b8 --text 511 --regions 256 --batch 8 --steps 30 --warmup 8
b shipped_s767_b8_fwd --mode fwd --text 511 --regions 256 --batch 8 --steps 50 --warmup 10
b text511_b8_fwd --mode fwd --text 511 --regions 0 --batch 8 --steps 50 --warmup 10
b 2rank --gpus 2 --share-gpu --backend gloo --batch 36 --steps 20 --warmup 5
python bench.py --mode fwd > $O/bench_fwd_b64.json 2> $O/bench_fwd_b64.err; echo fwdb64 >> $O/progress.txt
python bench.py > $O/bench_train_b256.json 2> $O/bench_train_b256.err; echo train >> $O/progress.txt
cp $O/bench_*.json profiles/$R/ 2>/dev/null
tail -3 $O/progress.txt
