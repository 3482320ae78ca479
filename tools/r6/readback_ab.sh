#!/bin/bash
# the step's read-back overlapped with the weight transposes (VT_STEP_OVERLAP_READBACK=1, default) against the old order
O=gpurun_out/readback_ab; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
 for B in 8 36 256; do
  for m in 0 1; do
    VT_STEP_OVERLAP_READBACK=$m python bench.py --batch $B --no-cpu-baseline --no-fwd-rate --steps 40 --warmup 10 > $O/b${B}_m${m}_$rep.json 2> $O/b${B}_m${m}_$rep.err
  done
 done
done
python - <<'P'
import json, glob
for B in (8, 36, 256):
    for m in (0, 1):
        v = []
        for f in sorted(glob.glob('gpurun_out/readback_ab/b%d_m%d_*.json' % (B, m))):
            try: v.append(json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step'])
            except Exception as e: print(f, 'ERR', e)
        print('B=%d mode=%d' % (B, m), ' '.join('%.3f' % x for x in v))
P
