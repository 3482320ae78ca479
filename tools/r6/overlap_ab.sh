#!/bin/bash
# weight gradients on a side stream (VT_OVERLAP_WGRAD=1) now that small batches take the one-tile wgrad kernel: one box, ms per step
O=gpurun_out/overlap_ab; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  for B in 8 24 36 48; do
    for m in 0 1; do
      VT_OVERLAP_WGRAD=$m python bench.py --batch $B --no-cpu-baseline --no-fwd-rate --steps 40 --warmup 10 > $O/b${B}_m${m}_$rep.json 2>> $O/err.txt
    done
  done
done
python - <<'P'
import json, glob
for B in (8, 24, 36, 48):
    row = []
    for m in (0, 1):
        v = []
        for f in sorted(glob.glob('gpurun_out/overlap_ab/b%d_m%d_*.json' % (B, m))):
            try: v.append(json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step'])
            except Exception: v.append(float('nan'))
        row.append(v)
    print('B=%-3d main stream %s | side stream %s' % (B, ' '.join('%.3f' % x for x in row[0]), ' '.join('%.3f' % x for x in row[1])))
P
