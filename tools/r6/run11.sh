cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for b in 4 8 16; do
python bench.py --batch $b --steps 60 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b${b}_v33.log 2>&1
echo "b$b $(tail -1 gpurun_out/r6/b${b}_v33.log | cut -c64-150) $(python - <<PY
import json
d=json.loads(open("gpurun_out/r6/b${b}_v33.log").read().strip().splitlines()[-1])
print({k:v for k,v in d["gemm_variants"].items() if v==33})
PY
)"
done
python bench.py --batch 2 --text 511 --regions 256 --steps 60 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/s767b2_v33.log 2>&1
echo "s767 b2 $(tail -1 gpurun_out/r6/s767b2_v33.log | cut -c64-150)"
