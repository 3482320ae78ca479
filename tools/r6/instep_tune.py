"""In-step tuning at B = 256: the step's time with ONE GEMM kind forced to another variant (a VT_TUNE_FILE made from the committed
choices), everything else as tuned -- do the tuner's warm-loop choices hold inside the step?  python tools/r6/instep_tune.py"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
base = json.load(open(os.path.join(root, "profiles/r06/tune_train_b256.json")))
out = os.path.join(root, "gpurun_out/instep_tune")
os.makedirs(out, exist_ok=True)
keys = ["50944,3072,768,33", "50944,3072,768,19", "50944,768,3072,16", "50944,2304,768,0", "50944,768,2304,16", "50944,768,768,16"]
cands = [16, 18, 19, 20]


def run(tag, table):
    path = os.path.join(out, "tune_%s.json" % tag)
    json.dump(table, open(path, "w"))
    env = dict(os.environ, VT_TUNE_FILE=path)
    ms = []
    for rep in range(2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-fwd-rate", "--steps", "30", "--warmup", "8"],
                           env=env, capture_output=True, text=True)
        try:
            ms.append(json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"])
        except Exception:
            ms.append(float("nan"))
    json.dump(table, open(path, "w"))   # (the run may have appended other shapes: restore)
    return ms


res = {"base": run("base", dict(base))}
print("base", res["base"], flush=True)
for k in keys:
    for v in cands:
        if base.get(k) == v:
            continue
        t = dict(base)
        t[k] = v
        tag = "%s_v%d" % (k.replace(",", "_"), v)
        res[tag] = run(tag, t)
        print("%-28s tuned v%-2d -> v%-2d  %s" % (k, base.get(k), v, " ".join("%.3f" % x for x in res[tag])), flush=True)
res["base_again"] = run("base2", dict(base))
print("base again", res["base_again"])
json.dump(res, open(os.path.join(out, "result.json"), "w"), indent=1)
