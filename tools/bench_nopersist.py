"""bench.py with the persistent GEMM excluded from the autotuner (what a multi-rank run uses): python tools/bench_nopersist.py [bench args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import visitron_amd.ops as ops  # noqa: E402

ops.PERSISTENT_GEMM_OK = False
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
