#!/bin/bash
# weight prefetch in the training layer loops (VT_PREFETCH_WEIGHTS = 0 off / 1 per layer / 2 per GEMM) on one box
O=gpurun_out/prefetch_ab; mkdir -p $O
for rep in 1 2; do
for B in ${BATCHES:-8 36 64}; do
  for m in ${MODES:-0 1 2 3}; do
    VT_PREFETCH_WEIGHTS=$m VT_PREFETCH_MAX_ROWS=1000000 python bench.py --batch $B --no-cpu-baseline --no-fwd-rate --steps 40 --warmup 10 > $O/b${B}_m${m}_$rep.json 2> $O/b${B}_m${m}_$rep.err
    echo "B=$B mode=$m rep=$rep done" >> $O/progress.txt
  done
done
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/prefetch_ab/b*_m*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'])
    except Exception as e:
        print(f, 'ERR', e)
P
