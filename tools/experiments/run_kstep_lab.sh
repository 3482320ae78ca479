#!/bin/bash
# Run every K-step lab build on the QKV shape (K = 768: 12 steps per tile) and the FFN-down shape (K = 3072: 48 steps).
# usage: tools/experiments/run_kstep_lab.sh [out file]
cd "$(dirname "$0")"
out=${1:-/dev/stdout}
for shape in "58368 2304 768" "58368 768 3072"; do
  for b in bin/kstep_*; do
    timeout -k 5 60 $b $shape || echo "$b FAILED rc=$?"
  done
done > "$out" 2>&1
