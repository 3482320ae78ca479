#!/bin/bash
# Is the NT GEMM's FETCH_SIZE excess over its algorithmic bytes (1.45 x at M = 50 845) HBM traffic or on-die (Infinity Cache)
# traffic?  For operand sets from 30 MB to 1.4 GB -- well under and well over the 256 MiB Infinity Cache -- this collects, per
# launch of the persistent kernel: FETCH_SIZE (doubled: gfx950), WRITE_SIZE, the DRAM-destined request count, and the launch
# time with warm caches and behind a 512 MiB flush.  Run on the GPU box from the repo root; summary: gpurun_out/traffic/summary.txt
set -euo pipefail
OUT=gpurun_out/traffic
mkdir -p "$OUT"
export TMPDIR=/tmp
: > "$OUT/points.jsonl"
for SHAPE in attn_out ffn_up; do
  for M in 8208 14592 29184 50845 101690 203380; do
    python3 tools/traffic_point.py $M $SHAPE >> "$OUT/points.jsonl"
    python3 tools/traffic_point.py $M $SHAPE --cold >> "$OUT/points.jsonl"
    for C in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/raw_${SHAPE}_${M}_$C" -- python3 tools/traffic_point.py $M $SHAPE --reps 4 > /dev/null 2> "$OUT/err_${SHAPE}_${M}_$C.txt"
    done
    echo "$SHAPE $M done" >> "$OUT/progress.txt"
  done
done
python3 tools/traffic_summary.py "$OUT" > "$OUT/summary.txt"
rm -rf "$OUT"/raw_* "$OUT"/err_*
cat "$OUT/summary.txt"
