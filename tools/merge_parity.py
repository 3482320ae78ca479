"""Merge a run's gpurun_out/parity_measured.txt (written by tests/conftest.py at the end of a GPU session) into the
committed record: profiles/rNN/parity_measured.txt (one line per check, the larger measured error of the runs merged so
far) and tests/golden/parity_measured.json (the lookup tests/helpers.effective_bound tightens the stated bounds with).

    python tools/merge_parity.py [--round r02] [gpurun_out/parity_measured.txt ...]
"""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(path):
    rows = {}
    if not os.path.exists(path):
        return rows
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        f = line.rstrip("\n").split()
        # name (may contain spaces) | kind | measured | bound | ratio
        name, kind, meas, bound = " ".join(f[:-4]), f[-4], float(f[-3]), float(f[-2])
        if name not in rows or meas > rows[name][1]:
            rows[name] = (kind, meas, bound)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r02")
    ap.add_argument("runs", nargs="*", default=[os.path.join(ROOT, "gpurun_out", "parity_measured.txt")])
    a = ap.parse_args()
    txt = os.path.join(ROOT, "profiles", a.round, "parity_measured.txt")
    js = os.path.join(ROOT, "tests", "golden", "parity_measured.json")
    rows = read(txt)
    for r in a.runs:
        for name, (kind, meas, bound) in read(r).items():
            if name not in rows or meas > rows[name][1]:
                rows[name] = (kind, meas, bound)
            else:
                rows[name] = (rows[name][0], rows[name][1], bound)
    with open(txt, "w") as fh:
        fh.write("# name | kind | measured | bound | measured/bound\n")
        for name, (kind, meas, bound) in rows.items():
            fh.write("%-60s %-7s %.4e %.4e %.2f\n" % (name, kind, meas, bound, meas / bound if bound else 0.0))
    rec = json.load(open(js)) if os.path.exists(js) else {}
    for name, (_, meas, _) in rows.items():
        rec[name] = max(float(rec.get(name, 0.0)), float("%.5g" % meas))
    json.dump(rec, open(js, "w"), indent=0, sort_keys=True)
    print("%d checks in %s, %d in %s" % (len(rows), os.path.relpath(txt, ROOT), len(rec), os.path.relpath(js, ROOT)))


if __name__ == "__main__":
    main()
