#!/bin/bash
O=gpurun_out/reverse_ab2; rm -rf $O; mkdir -p $O
for rep in 1 2 3 4 5 6; do
  for k in 0 3072; do
    VT_GEMM_REVERSE_K=$k python bench.py --no-cpu-baseline --no-fwd-rate --steps 30 --warmup 8 > $O/b256_k${k}_$rep.json 2> $O/b256_k${k}_$rep.err
  done
done
python - <<'P'
import json, glob, statistics
for k in (0, 3072):
    v = []
    for f in sorted(glob.glob('gpurun_out/reverse_ab2/b256_k%d_*.json' % k)):
        try: v.append(json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step'])
        except Exception as e: print(f, 'ERR', e)
    print('k=%d' % k, ' '.join('%.3f' % x for x in v), 'median %.3f mean %.3f' % (statistics.median(v), sum(v) / len(v)))
P
