#!/bin/bash
# launch-order view of one inference forward: gpurun_out/seq/fwd_b<B>.txt  (args: B [extra bench args])
B=${1:-64}; shift
O=gpurun_out/seq; mkdir -p $O
export TMPDIR=/tmp
rm -rf $O/trace_f$B
rocprofv3 --kernel-trace --output-format csv -d $O/trace_f$B -- python3 bench.py --mode fwd --batch $B --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing "$@" > $O/f$B.out 2> $O/f$B.err
T=$(find $O/trace_f$B -name "*kernel_trace.csv" | head -1)
python3 - "$T" $O/fwd_b$B.txt <<'P'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last forward: from the last embed_layernorm* kernel to the end
starts = [i for i, r in enumerate(rows) if "embed_layernorm" in r["Kernel_Name"]]
lo = starts[-1]
out = open(sys.argv[2], "w")
t0 = int(rows[lo]["Start_Timestamp"]); tp = t0
busy = 0
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:60]
    out.write("%9.1f us  %-60s %8.1f us  gap %6.1f us  grid %s\n" % ((s - t0) / 1e3, n, (e - s) / 1e3, (s - tp) / 1e3, r.get("Grid_Size_X", "?")))
    tp = max(tp, e); busy += e - s
out.write("span %.3f ms busy %.3f ms, %d dispatches\n" % ((tp - t0) / 1e6, busy / 1e6, len(rows) - lo))
P
rm -rf $O/trace_f$B
tail -3 $O/fwd_b$B.txt
