#!/bin/bash
# inference layer loop: deferred LayerNorms (5 launches per layer, LN-mode GEMMs) against the seven-launch layer, by batch
O=gpurun_out/deferred_ab; rm -rf $O; mkdir -p $O
for rep in 1 2; do
 for d in 1 0; do
  VT_DEFERRED_LN=$d python bench.py --mode fwd --text 511 --regions 0 --batch 8 --no-cpu-baseline --steps 50 --warmup 10 > $O/t511b8_d${d}_$rep.json 2> $O/err_$d_$rep.txt
  for B in 4 16 32 64; do
    VT_DEFERRED_LN=$d python bench.py --mode fwd --batch $B --no-cpu-baseline --steps 50 --warmup 10 > $O/fwd${B}_d${d}_$rep.json 2>> $O/err_$d_$rep.txt
  done
 done
done
python - <<'P'
import json, glob
for name in ('t511b8', 'fwd4', 'fwd16', 'fwd32', 'fwd64'):
    row = []
    for d in (1, 0):
        v = []
        for f in sorted(glob.glob('gpurun_out/deferred_ab/%s_d%d_*.json' % (name, d))):
            try: v.append(json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step'])
            except Exception as e: print(f, 'ERR', e)
        row.append(v)
    print('%-7s deferred %s | seven-launch %s' % (name, ' '.join('%.3f' % x for x in row[0]), ' '.join('%.3f' % x for x in row[1])))
P
