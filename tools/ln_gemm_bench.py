"""Per-shape times of the deferred-LayerNorm GEMMs beside the plain ones (every candidate variant, the autotuner's own
timing loop): python tools/ln_gemm_bench.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VT_TUNE_VERBOSE"] = "1"
import torch
from visitron_amd import ops

H, I = 768, 3072
for M in [int(x) for x in sys.argv[1:]] or [14592]:
    print("==== M = %d" % M)
    print("-- qkv plain / deferred-LN (mode 1)")
    ops.autotune_linear(M, 3 * H, H)
    ops.autotune_linear(M, 3 * H, H, ln_mode=1)
    print("-- out-proj (+residual) plain / deferred-LN (mode 2)")
    ops.autotune_linear(M, H, H, residual=True)
    ops.autotune_linear(M, H, H, ln_mode=2)
    print("-- ffn-up GELU plain / deferred-LN (mode 1)")
    ops.autotune_linear(M, I, H, act=ops.ACT_GELU)
    ops.autotune_linear(M, I, H, act=ops.ACT_GELU, ln_mode=1)
    print("-- ffn-down (+residual) plain / deferred-LN (mode 2)")
    ops.autotune_linear(M, H, I, residual=True)
    ops.autotune_linear(M, H, I, ln_mode=2)
