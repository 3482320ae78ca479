import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The tests' forwards are small (a few hundred token rows).  In the product an inference forward below 2 800 rows takes the
# seven-launch layer (round 6: faster there); the tests keep sending theirs through the deferred-LayerNorm loop -- the path
# large forwards run and most inference tests were written against -- unless a test sets the encoder's
# `deferred_ln_min_rows` itself (tests/test_gpu_round6.py does, for the rule).  The seven-launch layer has its own tests
# (`deferred_ln = False`, the compacted / training-mode forwards).
os.environ.setdefault("VT_DEFERRED_LN_MIN_ROWS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def pytest_sessionfinish(session, exitstatus):
    """Every parity check made through helpers.check_close is listed (name, measured error, bound) in
    gpurun_out/parity_measured.txt, so the bounds in the tests can be audited against what the kernels deliver."""
    try:
        import helpers
    except Exception:
        return
    rows = helpers.measured()
    if not rows:
        return
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_measured.txt"), "w") as fh:
        fh.write("# name | kind | measured | bound | measured/bound\n")
        for name, kind, err, bound in rows:
            fh.write("%-60s %-7s %.4e %.4e %.2f\n" % (name, kind, err, bound, err / bound if bound else float("nan")))
