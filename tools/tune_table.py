"""Per-variant timing of the encoder's GEMM shapes with their real epilogues (bias / GELU + saved derivative /
residual / multiply), as the autotuner sees them."""
import sys, os
os.environ["VT_TUNE_VERBOSE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 58368
print(ops.autotune_encoder_shapes(M, 768, 3072, training=True, device="cuda:0"))
