set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
export VT_TUNE_VERBOSE=1
python bench.py --batch 36 --steps 50 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b36_sk.log 2>&1
echo "b36 $(tail -1 gpurun_out/r6/b36_sk.log | cut -c64-160)"
python bench.py --mode fwd --batch 64 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r6/f64_sk.log 2>&1
echo "f64 $(tail -1 gpurun_out/r6/f64_sk.log | cut -c60-160)"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b256_sk.log 2>&1
echo "b256 $(tail -1 gpurun_out/r6/b256_sk.log | cut -c64-160)"
VT_GEMM_WS_REGIONS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b256_nosk.log 2>&1
echo "b256 nosk $(tail -1 gpurun_out/r6/b256_nosk.log | cut -c64-160)"
VT_GEMM_WS_REGIONS=0 python bench.py --batch 36 --steps 50 --warmup 10 --no-cpu-baseline --no-fwd-rate > gpurun_out/r6/b36_nosk.log 2>&1
echo "b36 nosk $(tail -1 gpurun_out/r6/b36_nosk.log | cut -c64-160)"
VT_GEMM_WS_REGIONS=0 python bench.py --mode fwd --batch 64 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r6/f64_nosk.log 2>&1
echo "f64 nosk $(tail -1 gpurun_out/r6/f64_nosk.log | cut -c60-160)"
