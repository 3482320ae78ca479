set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for rep in 1 2; do
for g in 2048 256; do
VT_TUNE_BUCKET_LARGE=$g python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b256_bucket${g}_$rep.log 2>&1
echo "b256 bucket $g rep $rep $(tail -1 gpurun_out/r6/b256_bucket${g}_$rep.log | cut -c64-150)"
done
done
python - <<PY
import json
for g in (2048, 256):
    d=json.loads(open("gpurun_out/r6/b256_bucket%d_1.log"%g).read().strip().splitlines()[-1])
    print(g, {k:v for k,v in d["gemm_variants"].items() if k.startswith("50") or k.startswith("51")})
PY
