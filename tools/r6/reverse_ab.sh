#!/bin/bash
# the persistent GEMM walking its tiles backwards for K >= k (consume H / g_mid in the reverse of the order written): one box
O=gpurun_out/reverse_ab; mkdir -p $O
for rep in 1 2 3; do
  for k in 0 3072 2304 768; do
    VT_GEMM_REVERSE_K=$k python bench.py --no-cpu-baseline --no-fwd-rate --steps 20 --warmup 8 > $O/b256_k${k}_$rep.json 2> $O/b256_k${k}_$rep.err
    echo "k=$k rep=$rep" >> $O/progress.txt
  done
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/reverse_ab/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
P
