"""profiles/: turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, as the guide prescribes) into a
per-kernel HBM traffic table and the per-launch figure bench.py reports as roofline.traffic for the NT GEMM family.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, so it is doubled
(MI355X_MICROARCH.md, HBM / rocprofv3 section).
The JSON also carries the sum over EVERY kernel of the passes divided by the steps they ran (`steps`, default 3 = bench.py
--steps 3 --warmup 1 with the side measurements switched off): the whole step's HBM bytes, which bench.py turns into
HBM GB/s against the 8 TB/s peak (BASELINE configs[4] asks for that figure).
The whole-step figure counts ONLY the dispatches between the end of the first optimizer update and the end of the last one
(adamw_flat launches delimit the steps in dispatch order), divided by the steps in between: initialisation, weight packing and
the warm-up step's tuning launches stay out.  Without an optimizer in the trace (forward benches) it falls back to every
dispatch / `steps` and says so (`window`).  `kernel_tree` = bench.kernel_tree_stamp() of the tree the passes ran on: bench.py
reports the figure only while it matches the running tree.
usage: python tools/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out_prefix> [steps] [kernel_tree]"""
import collections
import csv
import json
import sys


def agg(path, ctr):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == ctr:
            d[r["Kernel_Name"]][0] += 1
            d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return d


def step_window_bytes(path, ctr, scale):
    """bytes of the dispatches inside whole optimizer steps, and the number of such steps (None, 0: no optimizer in the trace)"""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == ctr]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ends = [i for i, r in enumerate(rows) if "adamw_flat" in r["Kernel_Name"] and
            (i + 1 == len(rows) or "adamw_flat" not in rows[i + 1]["Kernel_Name"])]
    if len(ends) < 2:
        return None, 0
    inside = rows[ends[0] + 1: ends[-1] + 1]
    return sum(float(r["Counter_Value"]) for r in inside) * 1024 * scale, len(ends) - 1


fetch, write, prefix = sys.argv[1:4]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
tree = sys.argv[5] if len(sys.argv) > 5 else None
f, w = agg(fetch, "FETCH_SIZE"), agg(write, "WRITE_SIZE")
rows = []
for name in set(f) | set(w):
    nf, fv = f.get(name, [0, 0.0])
    nw, wv = w.get(name, [0, 0.0])
    n = max(nf, nw)
    rows.append(dict(kernel=name, launches=n, read_MB_per_launch=round(2.0 * fv * 1024 / max(nf, 1) / 1e6, 2),
                     write_MB_per_launch=round(wv * 1024 / max(nw, 1) / 1e6, 2)))
rows.sort(key=lambda r: -(r["read_MB_per_launch"] + r["write_MB_per_launch"]) * r["launches"])
with open(prefix + ".csv", "w", newline="") as fh:
    wr = csv.DictWriter(fh, fieldnames=["kernel", "launches", "read_MB_per_launch", "write_MB_per_launch"])
    wr.writeheader()
    wr.writerows(rows)
nt = [r for r in rows if "gemm_nt_bf16" in r["kernel"]]
n = sum(r["launches"] for r in nt)
tot = sum((r["read_MB_per_launch"] + r["write_MB_per_launch"]) * r["launches"] for r in nt) * 1e6
every = sum((r["read_MB_per_launch"] + r["write_MB_per_launch"]) * r["launches"] for r in rows) * 1e6
fb, fs = step_window_bytes(fetch, "FETCH_SIZE", 2.0)
wb, ws = step_window_bytes(write, "WRITE_SIZE", 1.0)
if fs and ws:
    per_step, window = fb / fs + wb / ws, "dispatches between optimizer updates (%d steps)" % min(fs, ws)
else:
    per_step, window = every / steps, "every dispatch of the passes / %d (no optimizer in the trace: includes initialisation)" % steps
json.dump({"kernel_family": "gemm_nt_bf16*", "launches": n, "hbm_bytes_per_launch": round(tot / n),
           "all_kernels_hbm_bytes_per_step": round(per_step), "window": window, "steps_in_the_passes": steps, "kernel_tree": tree,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --steps 3 --warmup 1, "
                     "FETCH_SIZE doubled (gfx950)"}, open(prefix + ".json", "w"), indent=1)
print("NT GEMM family: %d launches, %.1f MB per launch; whole step: %.1f MB (%s)" % (n, tot / n / 1e6, per_step / 1e6, window))
