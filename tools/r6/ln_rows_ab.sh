#!/bin/bash
# LayerNorm rows-per-wave settings at small batches, interleaved: one box, ms per step
O=gpurun_out/ln_rows; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  for cfg in "default X=1" "fwd1 VT_LN_FWD_ROWS=1" "bwd2 VT_LN_BWD_ROWS=2" "both VT_LN_FWD_ROWS=1 VT_LN_BWD_ROWS=2" "default2 X=1"; do
    set -- $cfg; name=$1; shift
    for B in 8 36; do env "$@" python bench.py --batch $B --no-cpu-baseline --no-fwd-rate --steps 40 --warmup 10 > $O/${name}_b${B}_$rep.json 2>> $O/err.txt; done
  done
done
python - <<'P'
import json, glob
for name in ('default', 'fwd1', 'bwd2', 'both', 'default2'):
    out = []
    for B in (8, 36):
        v = []
        for f in sorted(glob.glob('gpurun_out/ln_rows/%s_b%d_*.json' % (name, B))):
            try: v.append(json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step'])
            except Exception as e: v.append(float('nan'))
        out.append('B=%d %s mean %.3f' % (B, ' '.join('%.3f' % x for x in v), sum(v) / max(1, len(v))))
    print('%-10s %s' % (name, ' | '.join(out)))
P
