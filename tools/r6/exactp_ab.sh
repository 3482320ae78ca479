#!/bin/bash
# exact-p attention dropout (VT_ATTN_DROPOUT_BITS=16) against the default on one box: the default bench line and B = 36
O=gpurun_out/exactp_ab; mkdir -p $O
for i in 1 2; do
  for bits in 8 16; do
    VT_ATTN_DROPOUT_BITS=$bits python bench.py --no-cpu-baseline --no-fwd-rate --steps 20 --warmup 8 > $O/b256_${bits}_$i.json 2> $O/b256_${bits}_$i.err
    echo "b256 bits=$bits run $i done" >> $O/progress.txt
  done
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/exactp_ab/b256_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d.get('kernels_ms_per_step', {})
    att = {n: v for n, v in k.items() if 'attention' in n}
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('attention_dropout_effective'), att)
P
