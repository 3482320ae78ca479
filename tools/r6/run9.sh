cd $GRAFT_REPO_ROOT
for s in 0 1 2; do echo "split $s"; VT_WGRAD_SPLIT=$s python tools/wgrad_bench.py 3584 7168 14336 2>&1 | grep "^M="; done
for rep in 1 2; do for s in 0 1; do
VT_WGRAD_SPLIT=$s python bench.py --batch 36 --steps 60 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b36_wsplit${s}_$rep.log 2>&1
echo "b36 wgrad split $s rep $rep $(tail -1 gpurun_out/r6/b36_wsplit${s}_$rep.log | cut -c64-150)"
done; done
