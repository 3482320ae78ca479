set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
export VT_TUNE_FILE=gpurun_out/r6/tune_b36_tmp.json
for cfg in "base:" "ovw:VT_OVERLAP_WGRAD=1" "ova:VT_OVERLAP_ADAMW=1" "cmp:VT_COMPACT_MIN_ROWS=0" "cmp_ovw:VT_COMPACT_MIN_ROWS=0 VT_OVERLAP_WGRAD=1" "cmp_ovw_ova:VT_COMPACT_MIN_ROWS=0 VT_OVERLAP_WGRAD=1 VT_OVERLAP_ADAMW=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --batch 36 --steps 50 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b36_$name.log 2>&1
  echo "$name $(tail -1 gpurun_out/r6/b36_$name.log | cut -c1-160)"
done
