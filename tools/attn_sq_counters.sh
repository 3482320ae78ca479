#!/bin/bash
# SQ counters of the attention kernels at B = 256, S = 228, dropout 0.1 (two passes of four counters; SQ wait / active counters
# are in quad-cycles): bash tools/attn_sq_counters.sh > gpurun_out/attention_sq_counters.txt   (on the GPU box, repo root)
# ATTN_BWD_WAVES is set explicitly per pass (17 persistent 16-wave kernel, 16 one pair per workgroup, 8 eight waves); a profiler
# or kernel failure stops the script with the profiler's stderr shown -- no partial or stale summary.
set -uo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/attn_sq
rm -rf $OUT
mkdir -p $OUT
pass() {   # pass <dir> <waves> <counters...>
  local d=$1 w=$2; shift 2
  if ! ATTN_BWD_WAVES=$w rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -- python3 tools/attn_bench.py 256 228 0.1 3 > $d.out 2> $d.err; then
    echo "rocprofv3 FAILED (ATTN_BWD_WAVES=$w, counters $*):" >&2; tail -20 $d.err >&2; exit 1
  fi
}
for W in 17 16 8; do
  pass $OUT/a$W $W SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  pass $OUT/b$W $W SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  echo "==== ATTN_BWD_WAVES=$W (attn_bench.py 256 228 0.1 3: padded, then with keep words, then the compacted batch)"
  python3 tools/pmc_sq_summary.py $OUT/a$W attention || exit 1
  python3 tools/pmc_sq_summary.py $OUT/b$W attention || exit 1
done
rm -rf $OUT
