#!/bin/bash
# The profile set of one round (run on the GPU box from the repo root): live bench line with the autotuner's choices saved,
# rocprofv3 kernel-trace summary of the same command with those choices read back (no tuning launches in the trace), and
# the two PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) that tools/summarize_pmc.py turns into HBM traffic.
#   usage: bash tools/profile_round.sh <tag>          -> gpurun_out/prof_<tag>/...
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export VT_TUNE_FILE=$PWD/$OUT/tune_b256.json
rm -f $VT_TUNE_FILE
python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_train_b256_live.json 2> $OUT/live.err || exit 1
echo "live done" >> $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fwd-rate --no-kernel-timing > $OUT/trace.out 2> $OUT/trace.err || exit 2
echo "trace done" >> $OUT/progress.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fwd-rate > $OUT/pmc_fetch.out 2> $OUT/pmc_fetch.err || exit 3
echo "fetch done" >> $OUT/progress.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fwd-rate > $OUT/pmc_write.out 2> $OUT/pmc_write.err || exit 4
echo "write done" >> $OUT/progress.txt
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/summarize_pmc.py $F $W $OUT/train_b256_pmc_hbm_traffic
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $S $OUT/train_b256_kernel_stats.csv
# the raw per-dispatch csv files are large: keep the summaries only
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/trace
ls -la $OUT
