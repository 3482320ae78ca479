"""Summary table of tools/traffic_experiment.sh (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
pts = collections.defaultdict(dict)
for line in open(os.path.join(out, "points.jsonl")):
    d = json.loads(line)
    pts[(d["shape"], d["M"])]["cold" if d["cold"] else "warm"] = d


def counter(shape, M, name):
    vals = []
    for f in glob.glob(os.path.join(out, "raw_%s_%d_%s" % (shape, M, name), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "gemm_nt_bf16" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    vals = vals[2:] if len(vals) > 2 else vals     # the two warm-up launches
    return sum(vals) / len(vals) * 1024 / 1e6 if vals else float("nan")


print("# NT GEMM (persistent 256x256-tile kernel), per launch; FETCH_SIZE x 2 (gfx950), MB = 1e6 bytes")
print("%-9s %7s %9s | %9s %9s %6s | %9s %9s %6s | %8s %8s %7s" % (
    "shape", "M", "set MB", "alg rd", "FETCH", "ratio", "alg wr", "WRITE", "ratio", "us warm", "us cold", "TF warm"))
for (shape, M), d in sorted(pts.items()):
    w, c = d["warm"], d.get("cold", d["warm"])
    fe, wr = 2.0 * counter(shape, M, "FETCH_SIZE"), counter(shape, M, "WRITE_SIZE")
    print("%-9s %7d %9.1f | %9.1f %9.1f %6.2f | %9.1f %9.1f %6.2f | %8.1f %8.1f %7.0f" % (
        shape, M, w["operand_set_MB"], w["alg_read_MB"], fe, fe / w["alg_read_MB"], w["alg_write_MB"], wr, wr / w["alg_write_MB"],
        w["us_median"], c["us_median"], w["tflops"]))
