#!/bin/bash
# The profile set of one bench configuration (run on the GPU box from the repo root): live bench line with the autotuner's
# choices saved, rocprofv3 kernel-trace summary of the same command with those choices read back (no tuning launches in the
# trace), and the two PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) that tools/summarize_pmc.py turns into HBM
# traffic.
#   usage: bash tools/profile_round.sh <tag> [name] [bench.py arguments ...]
#          bash tools/profile_round.sh r03                              -> gpurun_out/prof_r03/train_b256_*   (default bench)
#          bash tools/profile_round.sh r03 fwd_b64 --mode fwd           -> gpurun_out/prof_r03/fwd_b64_*
# Under rocprofv3 the program itself follows `--` (python3 bench.py ...): bench.py self-launches only for --gpus > 1.
set -euo pipefail
TAG=${1:-r03}
NAME=${2:-train_b256}
shift $(( $# > 2 ? 2 : $# ))
ARGS=("$@")
OUT="gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
export VT_TUNE_FILE="$PWD/$OUT/tune_$NAME.json"
rm -f "$VT_TUNE_FILE"
python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 "${ARGS[@]}" > "$OUT/bench_${NAME}_live.json" 2> "$OUT/${NAME}_live.err"
echo "$NAME live done" >> "$OUT/progress.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$NAME" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fwd-rate --no-kernel-timing "${ARGS[@]}" > "$OUT/${NAME}_trace.out" 2> "$OUT/${NAME}_trace.err"
echo "$NAME trace done" >> "$OUT/progress.txt"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_$NAME" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fwd-rate "${ARGS[@]}" > "$OUT/${NAME}_pmc_fetch.out" 2> "$OUT/${NAME}_pmc_fetch.err"
echo "$NAME fetch done" >> "$OUT/progress.txt"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_$NAME" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fwd-rate "${ARGS[@]}" > "$OUT/${NAME}_pmc_write.out" 2> "$OUT/${NAME}_pmc_write.err"
echo "$NAME write done" >> "$OUT/progress.txt"
F=$(find "$OUT/pmc_fetch_$NAME" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/pmc_write_$NAME" -name "*counter_collection.csv" | head -1)
S=$(find "$OUT/trace_$NAME" -name "*kernel_stats.csv" | head -1)
if [ -z "$F" ] || [ -z "$W" ] || [ -z "$S" ]; then
  echo "profile_round.sh: a profiler pass left no csv (fetch='$F' write='$W' stats='$S'); raw directories kept under $OUT" >&2
  exit 5
fi
python3 tools/summarize_pmc.py "$F" "$W" "$OUT/${NAME}_pmc_hbm_traffic" 3 "$(python3 -c 'import bench; print(bench.kernel_tree_stamp())')"
cp "$S" "$OUT/${NAME}_kernel_stats.csv"
# the raw per-dispatch csv files are large: keep the summaries only (reached only when every summary exists)
rm -rf "$OUT/pmc_fetch_$NAME" "$OUT/pmc_write_$NAME" "$OUT/trace_$NAME"
ls -la "$OUT"
