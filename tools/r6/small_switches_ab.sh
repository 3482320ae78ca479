#!/bin/bash
# round-4/5 defaults re-measured at small batches: one box, ms per step
O=gpurun_out/small_switches; rm -rf $O; mkdir -p $O
run() { name=$1; shift; for B in 8 36; do for rep in 1 2; do env "$@" python bench.py --batch $B --no-cpu-baseline --no-fwd-rate --steps 40 --warmup 10 > $O/${name}_b${B}_$rep.json 2>> $O/err.txt; done; done; }
run default X=1
run lnres0 VT_LN_RESIDUAL=0
run keepbits0 VT_ATTN_KEEP_BITS=0
run lnfwd1 VT_LN_FWD_ROWS=1
run lnbwd2 VT_LN_BWD_ROWS=2
run drop8 VT_ATTN_DROPOUT_BITS=8
python - <<'P'
import json, glob
for name in ('default', 'lnres0', 'keepbits0', 'lnfwd1', 'lnbwd2', 'drop8'):
    out = []
    for B in (8, 36):
        v = []
        for f in sorted(glob.glob('gpurun_out/small_switches/%s_b%d_*.json' % (name, B))):
            try: v.append(json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step'])
            except Exception as e: v.append(float('nan'))
        out.append('B=%d %s' % (B, ' '.join('%.3f' % x for x in v)))
    print('%-10s %s' % (name, ' | '.join(out)))
P
