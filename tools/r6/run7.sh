set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for rep in 1 2 3; do
for g in 2048 512; do
VT_TUNE_BUCKET_SMALL=$g python bench.py --batch 36 --steps 60 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b36_bucket${g}_$rep.log 2>&1
echo "b36 bucket $g rep $rep $(tail -1 gpurun_out/r6/b36_bucket${g}_$rep.log | cut -c64-150)"
done
done
