#!/bin/bash
# SQ counters of the NT GEMM variant 16 (one wave per SIMD) on the encoder's layer shapes at B = 256 (two passes of four
# counters; SQ wait / active counters are in quad-cycles).  VARIANTS=16,25 with VT_HIP_LIB pointing at the gemmlab build
# (tools/experiments) adds the eight-wave experiment.  A profiler or kernel failure stops the script with its stderr shown.
#   bash tools/gemm_sq_counters.sh > gpurun_out/gemm_sq_counters.txt   (on the GPU box, repo root)
set -uo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/gemm_sq
V=${VARIANTS:-16}
rm -rf $OUT
mkdir -p $OUT
pass() {
  local d=$1; shift
  if ! rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -- python3 tools/gemm_bench.py --batch 256 --variants $V --rounds 1 --reps 2 > $d.out 2> $d.err; then
    echo "rocprofv3 FAILED (counters $*):" >&2; tail -20 $d.err >&2; exit 1
  fi
}
pass $OUT/a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass $OUT/b SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python3 tools/pmc_sq_summary.py $OUT/a gemm_nt_bf16 || exit 1
python3 tools/pmc_sq_summary.py $OUT/b gemm_nt_bf16 || exit 1
rm -rf $OUT
