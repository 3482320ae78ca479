#!/bin/bash
# In-step A/B of the LayerNorm forward kernels on ONE box (rocprofv3 kernel stats of the default bench): VT_LN_FWD_ROWS = 0 / 1 / 2
set -e
mkdir -p gpurun_out/ln
export TMPDIR=/tmp
O=gpurun_out/ln/fwd_ab.txt
: > $O
python3 bench.py --no-cpu-baseline --steps 5 --warmup 3 > /dev/null 2>&1
for v in 0 1 2 0 1; do
  rm -rf /tmp/tr_$v
  VT_LN_FWD_ROWS=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fwd-rate --no-kernel-timing > /dev/null 2>&1
  S=$(find /tmp/tr_$v -name "*kernel_stats.csv" | head -1)
  echo "VT_LN_FWD_ROWS=$v" >> $O
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$S')):
    if 'layernorm' in r['Name'] or 'adamw' in r['Name'] or 'ln_bwd' in r['Name']:
        print('  %-70s %5s calls  %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
" >> $O
done
cat $O
