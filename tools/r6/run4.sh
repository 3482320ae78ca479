set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for b in 4 8 16 24; do
for cfg in "base:VT_COMPACT_MIN_ROWS=16384" "cmp:VT_COMPACT_MIN_ROWS=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b${b}_$name.log 2>&1
  echo "b$b $name $(tail -1 gpurun_out/r6/b${b}_$name.log | cut -c64-160)"
done
done
for cfg in "base:VT_COMPACT_MIN_ROWS=16384" "cmp:VT_COMPACT_MIN_ROWS=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --batch 8 --text 511 --regions 256 --steps 30 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/s767_$name.log 2>&1
  echo "s767 b8 $name $(tail -1 gpurun_out/r6/s767_$name.log | cut -c64-160)"
done
