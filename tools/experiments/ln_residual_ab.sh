#!/bin/bash
# VT_LN_RESIDUAL=0 (two-output LayerNorm, fp16 copy read by the residual add) against 1 (one output, LayerNorm rebuilt in the
# residual epilogue) on ONE box: bench lines alternating, then rocprofv3 kernel stats of each.
set -e
mkdir -p gpurun_out/lnres
export TMPDIR=/tmp
O=gpurun_out/lnres/ab.txt
: > $O
python3 bench.py --no-cpu-baseline --steps 5 --warmup 3 > /dev/null 2>&1
for i in 1 2; do for v in 0 1; do
  VT_LN_RESIDUAL=$v python3 bench.py --no-cpu-baseline --no-fwd-rate --steps 20 --warmup 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VT_LN_RESIDUAL=$v', d['value'], d['ms_per_step'])" >> $O
done; done
for v in 0 1; do
  rm -rf /tmp/tr_$v
  VT_LN_RESIDUAL=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fwd-rate --no-kernel-timing > /dev/null 2>&1
  S=$(find /tmp/tr_$v -name "*kernel_stats.csv" | head -1)
  echo "VT_LN_RESIDUAL=$v" >> $O
  python3 -c "
import csv
for r in list(csv.DictReader(open('$S')))[:14]:
    print('  %-74s %5s calls  %8.1f us' % (r['Name'][:74], r['Calls'], float(r['AverageNs']) / 1e3))
" >> $O
done
cat $O
