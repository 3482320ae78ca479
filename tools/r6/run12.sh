cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
python tools/sk_sweep.py 4730 6136 7150 | grep "^M=.*K= 3072\|^M=.*K= 2304" | cut -c1-150
for b in 4 8 16 24 36; do
python bench.py --batch $b --steps 60 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b${b}_v33b.log 2>&1
echo "b$b $(tail -1 gpurun_out/r6/b${b}_v33b.log | cut -c64-150)"
done
python bench.py --batch 2 --text 511 --regions 256 --steps 60 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/s767b2_v33b.log 2>&1
echo "s767 b2 $(tail -1 gpurun_out/r6/s767b2_v33b.log | cut -c64-150)"
python bench.py --batch 8 --text 511 --regions 256 --steps 40 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/s767b8_v33b.log 2>&1
echo "s767 b8 $(tail -1 gpurun_out/r6/s767b8_v33b.log | cut -c64-150)"
