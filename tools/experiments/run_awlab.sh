#!/bin/bash
# attention backward (16-wave kernel) with parts removed: tools/attn_bench.py against each lab build of the library
cd "$(dirname "$0")/../.."
out=${1:-/dev/stdout}
{
  python tools/attn_bench.py 256 228 0.1 20
  for lib in tools/experiments/bin/libvisitron_hip_awlab*.so; do
    VT_HIP_LIB=$PWD/$lib python tools/attn_bench.py 256 228 0.1 20 || echo "$lib FAILED"
  done
} > "$out" 2>&1
