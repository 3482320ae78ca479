"""Per-variant timing of the encoder's forward GEMM shapes (inference epilogues) at M token rows."""
import sys, os
os.environ["VT_TUNE_VERBOSE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visitron_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 14592
print(ops.autotune_encoder_shapes(M, 768, 3072, training=False, device="cuda:0"))
