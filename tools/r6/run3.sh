set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
export VT_TUNE_VERBOSE=1
for cfg in "base:" "cmp:VT_COMPACT_MIN_ROWS=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --batch 36 --steps 50 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b36b_$name.log 2>&1
  echo "$name $(tail -1 gpurun_out/r6/b36b_$name.log | cut -c1-160)"
done
