#!/bin/bash
# SQ counters of the attention kernels at B = 256, S = 228, dropout 0.1 (two passes of four counters; SQ wait / active counters
# are in quad-cycles): bash tools/attn_sq_counters.sh > gpurun_out/attention_sq_counters.txt   (on the GPU box, repo root)
set -euo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/attn_sq
mkdir -p $OUT
for W in 16 8; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/a$W -- python3 tools/attn_bench.py 256 228 0.1 3 > /dev/null 2>&1 || true
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/b$W -- python3 tools/attn_bench.py 256 228 0.1 3 > /dev/null 2>&1 || true
  echo "==== ATTN_BWD_WAVES=$W (attn_bench.py 256 228 0.1 3: padded, then with keep words, then the compacted batch)"
  python3 tools/pmc_sq_summary.py $OUT/a$W attention
  python3 tools/pmc_sq_summary.py $OUT/b$W attention
  export ATTN_BWD_WAVES=8
done
rm -rf $OUT
