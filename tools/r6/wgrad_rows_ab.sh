#!/bin/bash
# the persistent weight-gradient kernel only from 12 288 rows (default) against always (VT_WGRAD_PERSISTENT_MIN_ROWS=0): one box
O=gpurun_out/wgrad_rows_ab; rm -rf $O; mkdir -p $O
for rep in 1 2; do
 for B in ${BATCHES:-4 8 16 24 36 48 64}; do
  for r in 0 12288; do
    VT_WGRAD_PERSISTENT_MIN_ROWS=$r python bench.py --batch $B --no-cpu-baseline --no-fwd-rate --steps 40 --warmup 10 > $O/b${B}_r${r}_$rep.json 2> $O/b${B}_r${r}_$rep.err
  done
 done
done
python - <<'P'
import json, glob
for B in (4, 8, 16, 24, 36, 48, 64):
    row = []
    for r in (0, 12288):
        v = []
        for f in sorted(glob.glob('gpurun_out/wgrad_rows_ab/b%d_r%d_*.json' % (B, r))):
            try: v.append(json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step'])
            except Exception as e: print(f, 'ERR', e)
        row.append((r, v))
    if row[0][1] and row[1][1]:
        a, b = sum(row[0][1]) / len(row[0][1]), sum(row[1][1]) / len(row[1][1])
        print('B=%-3d always persistent %s | one-tile below 12 288 rows %s | %+.1f %%' % (B, ' '.join('%.3f' % x for x in row[0][1]), ' '.join('%.3f' % x for x in row[1][1]), 100 * (b / a - 1)))
P
