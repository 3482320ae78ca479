#!/bin/bash
# Kernel-trace summary of one bench configuration (no PMC passes): tools/trace_step.sh <outdir-tag> <name> [bench args ...]
#   -> gpurun_out/<tag>/<name>_kernel_stats.csv, bench_<name>_live.json, tune_<name>.json
set -euo pipefail
TAG=$1; NAME=$2; shift 2
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
export VT_TUNE_FILE="$PWD/$OUT/tune_$NAME.json"
rm -f "$VT_TUNE_FILE"
python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 --no-fwd-rate "$@" > "$OUT/bench_${NAME}_live.json" 2> "$OUT/${NAME}_live.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$NAME" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fwd-rate --no-kernel-timing "$@" > "$OUT/${NAME}_trace.out" 2> "$OUT/${NAME}_trace.err"
S=$(find "$OUT/trace_$NAME" -name "*kernel_stats.csv" | head -1)
cp "$S" "$OUT/${NAME}_kernel_stats.csv"
rm -rf "$OUT/trace_$NAME"
tail -1 "$OUT/bench_${NAME}_live.json" | cut -c1-200
