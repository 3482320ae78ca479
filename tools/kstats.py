"""Per-step view of a rocprofv3 kernel_stats.csv: tools/kstats.py <csv> <steps in the trace> [top N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 13.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total %.3f ms per step over %g steps" % (tot / steps * 1e-6, steps))
for r in rows[:top]:
    print("%-86s %7.1f/step %8.1f us  %6.3f ms/step %5.1f%%" % (r["Name"][:86], float(r["Calls"]) / steps, float(r["AverageNs"]) * 1e-3,
          float(r["TotalDurationNs"]) / steps * 1e-6, 100 * float(r["TotalDurationNs"]) / tot))
