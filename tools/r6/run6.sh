set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
export VT_TUNE_VERBOSE=1
for rep in 1 2; do
python bench.py --batch 36 --steps 60 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b36_fine_$rep.log 2>&1
echo "b36 fine $rep $(tail -1 gpurun_out/r6/b36_fine_$rep.log | cut -c64-150)"
done
grep chosen gpurun_out/r6/b36_fine_1.log | grep "M=7168" | cut -c1-100
