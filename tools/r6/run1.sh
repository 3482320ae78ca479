set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
export VT_TUNE_VERBOSE=1
python bench.py --batch 36 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r6/b36_base.log 2>&1
VT_COMPACT_MIN_ROWS=0 python bench.py --batch 36 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r6/b36_compact.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6/b256_base.log 2>&1
tail -1 gpurun_out/r6/b36_base.log | cut -c1-300
tail -1 gpurun_out/r6/b36_compact.log | cut -c1-300
tail -1 gpurun_out/r6/b256_base.log | cut -c1-300
