#!/bin/bash
# Where do the AdamW launches of VT_OVERLAP_ADAMW=1 land in time?  (rocprofv3 kernel trace of 4 steps; run on the GPU box)
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/adamw
rm -rf /tmp/tr_adam
VT_OVERLAP_ADAMW=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_adam -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fwd-rate --no-kernel-timing > /dev/null 2>&1
F=$(find /tmp/tr_adam -name "*kernel_trace.csv" | head -1)
python3 - "$F" > gpurun_out/adamw/overlap_trace.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")) for r in rows))
ad = [e for e in ev if "adamw" in e[2]]
last = ad[-12:]
t0 = last[0][0]
for s, e, n, q in last:
    others = [(max(s, s2) , min(e, e2), n2) for s2, e2, n2, q2 in ev if e2 > s and s2 < e and "adamw" not in n2]
    ov = sum(b - a for a, b, _ in others)
    print("adamw q=%s start %9.1f us dur %7.1f us; other kernels running during it: %7.1f us of overlap: %s" % (
        q, (s - t0) / 1e3, (e - s) / 1e3, ov / 1e3, sorted({n2[:40] for _, _, n2 in others})))
PY
cat gpurun_out/adamw/overlap_trace.txt
