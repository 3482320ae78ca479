#!/bin/bash
O=gpurun_out/deferred_ab2; rm -rf $O; mkdir -p $O
for rep in 1 2; do
 for d in 1 0; do
  for B in 1 2 6 8 10 12; do
    VT_DEFERRED_LN=$d python bench.py --mode fwd --batch $B --no-cpu-baseline --steps 50 --warmup 10 > $O/fwd${B}_d${d}_$rep.json 2>> $O/err_$d_$rep.txt
  done
  VT_DEFERRED_LN=$d python bench.py --mode fwd --text 511 --regions 0 --batch 2 --no-cpu-baseline --steps 50 --warmup 10 > $O/t511b2_d${d}_$rep.json 2>> $O/err_$d_$rep.txt
  VT_DEFERRED_LN=$d python bench.py --mode fwd --text 511 --regions 0 --batch 4 --no-cpu-baseline --steps 50 --warmup 10 > $O/t511b4_d${d}_$rep.json 2>> $O/err_$d_$rep.txt
 done
done
python - <<'P'
import json, glob
for name in ('fwd1', 'fwd2', 'fwd6', 'fwd8', 'fwd10', 'fwd12', 't511b2', 't511b4'):
    row = []
    for d in (1, 0):
        v = []
        for f in sorted(glob.glob('gpurun_out/deferred_ab2/%s_d%d_*.json' % (name, d))):
            try:
                j = json.loads(open(f).read().strip().splitlines()[-1]); v.append(j['ms_per_step'])
            except Exception as e: print(f, 'ERR', e)
        row.append(v)
    print('%-7s deferred %s | seven-launch %s' % (name, ' '.join('%.3f' % x for x in row[0]), ' '.join('%.3f' % x for x in row[1])))
P
