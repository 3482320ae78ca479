#!/bin/bash
O=gpurun_out/prefetch_infer; mkdir -p $O
for rep in 1 2; do
 for m in ${MODES:-0 1 3}; do
  VT_PREFETCH_INFER=$m python bench.py --mode fwd --no-cpu-baseline --steps 50 --warmup 10 > $O/fwd64_m${m}_$rep.json 2> $O/fwd64_m${m}_$rep.err
  VT_PREFETCH_INFER=$m python bench.py --mode fwd --text 511 --regions 0 --batch 8 --no-cpu-baseline --steps 50 --warmup 10 > $O/t511b8_m${m}_$rep.json 2>> $O/fwd64_m${m}_$rep.err
  VT_PREFETCH_INFER=$m python bench.py --mode fwd --batch 16 --no-cpu-baseline --steps 50 --warmup 10 > $O/fwd16_m${m}_$rep.json 2>> $O/fwd64_m${m}_$rep.err
 done
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/prefetch_infer/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
P
