set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
export VT_COMPACT_MIN_ROWS=0 VT_GEMM_WS_REGIONS=0
for rep in 1 2; do
  for tree in old new; do
    d=.; [ $tree = old ] && d=ab_old
    (cd $d && python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-fwd-rate > $GRAFT_REPO_ROOT/gpurun_out/r6/ab_${tree}_b256_$rep.log 2>&1)
    echo "b256 $tree $rep $(tail -1 gpurun_out/r6/ab_${tree}_b256_$rep.log | cut -c64-150)"
    (cd $d && python bench.py --batch 36 --steps 60 --warmup 10 --no-cpu-baseline --no-fwd-rate > $GRAFT_REPO_ROOT/gpurun_out/r6/ab_${tree}_b36_$rep.log 2>&1)
    echo "b36 $tree $rep $(tail -1 gpurun_out/r6/ab_${tree}_b36_$rep.log | cut -c64-150)"
  done
done
