set -o pipefail
mkdir -p gpurun_out/r04i
bash tools/profile_round.sh r04 > gpurun_out/r04i/profile_train.log 2>&1 && echo "profile train done" >> gpurun_out/r04i/progress.txt
bash tools/profile_round.sh r04 fwd_b64 --mode fwd > gpurun_out/r04i/profile_fwd.log 2>&1 && echo "profile fwd done" >> gpurun_out/r04i/progress.txt
bash tools/profile_round.sh r04 cfg5 --batch 64 --text 512 --regions 144 > gpurun_out/r04i/profile_cfg5.log 2>&1 && echo "profile cfg5 done" >> gpurun_out/r04i/progress.txt
# the bench lines below read the HBM figures of THIS tree's passes
cp gpurun_out/prof_r04/*_pmc_hbm_traffic.json gpurun_out/prof_r04/*_pmc_hbm_traffic.csv profiles/r04/ 2>/dev/null
python bench.py --batch 36 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r04i/bench_b36.json 2> gpurun_out/r04i/bench_b36.err; echo b36 >> gpurun_out/r04i/progress.txt
python bench.py --batch 288 --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/r04i/bench_b288.json 2> gpurun_out/r04i/bench_b288.err; echo b288 >> gpurun_out/r04i/progress.txt
python bench.py --mode fwd --batch 256 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r04i/bench_fwd256.json 2> gpurun_out/r04i/bench_fwd256.err; echo fwd256 >> gpurun_out/r04i/progress.txt
python bench.py --batch 64 --text 512 --regions 144 --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/r04i/bench_cfg5.json 2> gpurun_out/r04i/bench_cfg5.err; echo cfg5 >> gpurun_out/r04i/progress.txt
python bench.py --mode fwd --batch 64 --text 512 --regions 144 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r04i/bench_cfg5f.json 2> gpurun_out/r04i/bench_cfg5f.err; echo cfg5f >> gpurun_out/r04i/progress.txt
python bench.py --gpus 2 --share-gpu --backend gloo --batch 36 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04i/bench_2rank.json 2> gpurun_out/r04i/bench_2rank.err; echo 2rank >> gpurun_out/r04i/progress.txt
python bench.py --mode fwd > gpurun_out/r04i/bench_fwd_b64.json 2> gpurun_out/r04i/bench_fwd_b64.err; echo fwdb64 >> gpurun_out/r04i/progress.txt
python bench.py > gpurun_out/r04i/bench_train_b256.json 2> gpurun_out/r04i/bench_train_b256.err; echo train >> gpurun_out/r04i/progress.txt
tail -3 gpurun_out/r04i/progress.txt
