"""One step of a rocprofv3 --kernel-trace run in launch order: per dispatch the kernel's short name, its duration and the idle
gap since the previous dispatch's end, then the totals.  The step taken is the last complete window between two
`adamw_flat` pairs.   python tools/step_sequence.py <kernel_trace.csv> [out.txt]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"]
ad = [i for i, r in enumerate(rows) if "adamw_flat" in name(r)]
ends = [i for k, i in enumerate(ad) if k + 1 == len(ad) or ad[k + 1] != i + 1 and not any("adamw" in name(rows[j]) for j in range(i + 1, min(i + 3, len(rows))))]
if len(ends) < 2:
    sys.exit("fewer than two optimizer updates in the trace")
lo, hi = ends[-2] + 1, ends[-1] + 1
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:64]


t_prev = int(rows[lo - 1]["End_Timestamp"])
t0 = int(rows[lo]["Start_Timestamp"])
busy = gap_total = 0
by = {}
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - t_prev
    d = e - s
    busy += d
    gap_total += max(gap, 0)
    k = short(name(r))
    by.setdefault(k, [0, 0])
    by[k][0] += 1
    by[k][1] += d
    out.write("%9.1f us  %-64s %8.1f us  gap %6.1f us  grid %s wg %s\n" % ((s - t0) / 1e3, k, d / 1e3, gap / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")),
                                                                       r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
    t_prev = max(t_prev, e)
span = t_prev - t0
out.write("\nstep span %.3f ms, kernels busy %.3f ms, idle between dispatches %.3f ms, %d dispatches\n" % (span / 1e6, busy / 1e6, gap_total / 1e6, hi - lo))
for k, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    out.write("  %-64s %4d x %8.1f us = %7.3f ms\n" % (k, c, d / c / 1e3, d / 1e6))
