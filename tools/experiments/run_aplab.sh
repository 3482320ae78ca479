#!/bin/bash
cd "$(dirname "$0")/../.."
out=${1:-/dev/stdout}
{
  ATTN_BWD_WAVES=17 python tools/attn_bench.py 256 228 0.1 20
  for lib in tools/experiments/bin/libvisitron_hip_aplab*.so; do
    ATTN_BWD_WAVES=17 VT_HIP_LIB=$PWD/$lib python tools/attn_bench.py 256 228 0.1 20 || echo "$lib FAILED"
  done
} 2>&1 | grep -v amdgpu.ids > "$out"
