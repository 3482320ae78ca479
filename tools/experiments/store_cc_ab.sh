#!/bin/bash
# Cache-control hints on the persistent GEMM's C stores (twin builds with -DV7_STORE_CC=" nt" / " sc1" / " sc0 sc1 nt"),
# same box, training shapes (tools/epilogue_ab.py), then the step with each.
set -e
mkdir -p gpurun_out/epi
O=gpurun_out/epi/store_cc_ab.txt
: > $O
for shape in "50820 2304 768 0 0" "50820 768 768 0 0" "50820 3072 768 1 1" "50820 768 3072 0 0"; do
  python3 tools/epilogue_ab.py $shape 16 >> $O
  for tag in nt sc1 sc0sc1nt; do
    VT_HIP_LIB=$PWD/visitron_amd/lib_$tag/libvisitron_hip.so python3 tools/epilogue_ab.py $shape 16 >> $O
  done
done
for i in 1 2; do for tag in default nt sc1 sc0sc1nt; do
  if [ $tag = default ]; then unset VT_HIP_LIB; else export VT_HIP_LIB=$PWD/visitron_amd/lib_$tag/libvisitron_hip.so; fi
  python3 bench.py --no-cpu-baseline --no-fwd-rate --steps 20 --warmup 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step with $tag:', d['value'], d['ms_per_step'])" >> $O
done; done
cat $O
