cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
bash tools/run_gpu_tests.sh r6_t9 tests/test_gpu_round6.py -m gpu -x -q -k "three_stage or splitk" || exit 1
for b in 8 16 24 36; do
python bench.py --batch $b --steps 60 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b${b}_v35.log 2>&1
echo "b$b $(tail -1 gpurun_out/r6/b${b}_v35.log | cut -c64-150) $(python - <<PY
import json
d=json.loads(open("gpurun_out/r6/b${b}_v35.log").read().strip().splitlines()[-1])
print({k:v for k,v in d["gemm_variants"].items() if v in (33,35)})
PY
)"
done
python bench.py --batch 8 --text 511 --regions 0 --steps 60 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/t511b8_v35.log 2>&1
echo "text511 b8 train $(tail -1 gpurun_out/r6/t511b8_v35.log | cut -c64-150)"
