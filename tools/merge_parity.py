"""Turn ONE full GPU run made with VT_PARITY_RECORD=1 (gpurun_out/parity_measured.txt, written by tests/conftest.py: every
check against its STATED bound) into the round's committed record:

  profiles/rNN/parity_measured.txt   one line per check (name, kind, measured, stated bound)
  tests/golden/parity_measured.json  the lookup tests/helpers.effective_bound tightens the stated bounds with:
                                     min(stated, max(2 x recorded, stated / 10))

The record is REPLACED by the round's measurement -- it is not the maximum over history, so a value can go down as well as
up -- and every check whose error rose by more than 1.5 x against the previous record (or is new) is listed on stdout: that
list, with its reason, belongs in the commit message.  Checks that no longer exist leave the record.

    python tools/merge_parity.py --round r05 [gpurun_out/parity_measured.txt]
"""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(path):
    rows = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        f = line.rstrip("\n").split()
        # name (may contain spaces) | kind | measured | bound | ratio
        name, kind, meas, bound = " ".join(f[:-4]), f[-4], float(f[-3]), float(f[-2])
        if name not in rows or meas > rows[name][1]:      # a name checked several times in the run: its largest error
            rows[name] = (kind, meas, bound)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", required=True)
    ap.add_argument("run", nargs="?", default=os.path.join(ROOT, "gpurun_out", "parity_measured.txt"))
    a = ap.parse_args()
    rows = read(a.run)
    os.makedirs(os.path.join(ROOT, "profiles", a.round), exist_ok=True)
    txt = os.path.join(ROOT, "profiles", a.round, "parity_measured.txt")
    js = os.path.join(ROOT, "tests", "golden", "parity_measured.json")
    old = json.load(open(js)) if os.path.exists(js) else {}
    with open(txt, "w") as fh:
        fh.write("# name | kind | measured | stated bound | measured/bound   (one full run with VT_PARITY_RECORD=1)\n")
        for name, (kind, meas, bound) in rows.items():
            fh.write("%-60s %-7s %.4e %.4e %.2f\n" % (name, kind, meas, bound, meas / bound if bound else 0.0))
    rec = {name: float("%.5g" % meas) for name, (_, meas, _) in rows.items()}
    rose = sorted((n for n in rec if n in old and rec[n] > 1.5 * old[n] and rec[n] > 0.05 * rows[n][2]),
                  key=lambda n: -rec[n] / max(old[n], 1e-30))
    new = [n for n in rec if n not in old]
    gone = [n for n in old if n not in rec]
    json.dump(rec, open(js, "w"), indent=0, sort_keys=True)
    print("%d checks in %s and %s (%d new, %d gone)" % (len(rows), os.path.relpath(txt, ROOT), os.path.relpath(js, ROOT), len(new), len(gone)))
    for n in rose:
        print("ROSE  %-70s %.3e -> %.3e (stated bound %.1e)" % (n, old[n], rec[n], rows[n][2]))
    if not rose:
        print("no check rose by more than 1.5 x (among those above 5 % of their stated bound)")


if __name__ == "__main__":
    main()
