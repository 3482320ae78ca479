"""Measure the GEMM kernel choices for the path's own shapes on this MI355X and write visitron_amd/gemm_defaults.json
("M,N,K,kind" -> variant): what the autotuner falls back on (VT_AUTOTUNE=0) and keeps unless a candidate beats it by more
than ops.TUNE_KEEP_DEFAULT on the box at hand.  Usage: python tools/make_gemm_defaults.py [out.json] [M ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from visitron_amd import ops  # noqa: E402

H, I = 768, 3072
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                         "visitron_amd", "gemm_defaults.json")
Ms = [int(x) for x in sys.argv[2:]] or [456, 8208, 14592, 29184, 51200, 58368]
ops._defaults = {}          # measure from scratch: no hysteresis towards an older table
ops.TUNE_ROUNDS = 5
for M in Ms:
    ops.autotune_encoder_shapes_ln(M, H, I)
    ops.autotune_encoder_shapes(M, H, I, training=False)
    ops.autotune_encoder_shapes(M, H, I, training=True)
    print("M = %d done" % M, flush=True)
table = {"%d,%d,%d,%d" % k: int(v) for k, v in sorted(ops._tuned.items())}
table["_source"] = "tools/make_gemm_defaults.py on %s, median of %d interleaved rounds" % (
    torch.cuda.get_device_properties(0).name, ops.TUNE_ROUNDS)
with open(out, "w") as fh:
    json.dump(table, fh, indent=0, sort_keys=True)
print("wrote %d entries to %s" % (len(table) - 1, out))
