cd $GRAFT_REPO_ROOT
for b in 8 16 24 28; do for s in 0 1; do
VT_WGRAD_SPLIT=$s python bench.py --batch $b --steps 60 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b${b}_wsplit${s}.log 2>&1
echo "b$b wgrad split $s $(tail -1 gpurun_out/r6/b${b}_wsplit${s}.log | cut -c64-150)"
done; done
for s in 0 1; do
VT_WGRAD_SPLIT=$s python bench.py --batch 8 --text 511 --regions 256 --steps 40 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/s767_wsplit${s}.log 2>&1
echo "s767b8 wgrad split $s $(tail -1 gpurun_out/r6/s767_wsplit${s}.log | cut -c64-150)"
done
