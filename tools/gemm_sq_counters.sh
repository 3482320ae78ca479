#!/bin/bash
# SQ counters of the NT GEMM variants 16 (one wave per SIMD) and 25 (eight waves on shared stages) on the encoder's layer
# shapes at B = 256 (two passes of four counters; SQ wait / active counters are in quad-cycles):
#   bash tools/gemm_sq_counters.sh > gpurun_out/gemm_sq_counters.txt   (on the GPU box, repo root)
set -euo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/gemm_sq
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/a -- python3 tools/gemm_bench.py --batch 256 --variants 16,25 --rounds 1 --reps 2 > /dev/null 2>&1 || true
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/b -- python3 tools/gemm_bench.py --batch 256 --variants 16,25 --rounds 1 --reps 2 > /dev/null 2>&1 || true
python3 tools/pmc_sq_summary.py $OUT/a gemm_nt_bf16
python3 tools/pmc_sq_summary.py $OUT/b gemm_nt_bf16
rm -rf $OUT
