#!/bin/bash
# Weight-gradient group of one layer at M = 50 845: time and FETCH_SIZE per launch for both workgroup -> (tile, row range)
# mappings (VT_WGRAD_ORDER=0 tile-major, 1 one row range per XCD pair).  FETCH_SIZE is doubled in the summary (gfx950 counts
# 64 B per 128-B request).  Run on the GPU box from the repo root.
set -uo pipefail
OUT=${1:-gpurun_out/wgrad_traffic}
mkdir -p "$OUT"
export TMPDIR=/tmp
for O in 0 1 0 1; do
  echo "VT_WGRAD_ORDER=$O" >> "$OUT/summary.txt"
  VT_WGRAD_ORDER=$O python3 tools/wgrad_bench.py 50845 2>&1 | grep -v amdgpu.ids >> "$OUT/summary.txt"
done
for O in 0 1; do
  rm -rf /tmp/wg_pmc_$O
  if ! VT_WGRAD_ORDER=$O rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/wg_pmc_$O -- python3 tools/wgrad_bench.py 50845 > /dev/null 2> "$OUT/pmc_err_$O.txt"; then
    echo "rocprofv3 FAILED for order $O (see pmc_err_$O.txt)" >> "$OUT/summary.txt"; continue
  fi
  echo "VT_WGRAD_ORDER=$O FETCH_SIZE per launch (raw counter in KB; x 2 for bytes on gfx950):" >> "$OUT/summary.txt"
  python3 tools/pmc_avg_by_kernel.py /tmp/wg_pmc_$O | grep -i wgrad >> "$OUT/summary.txt"
done
cat "$OUT/summary.txt"
