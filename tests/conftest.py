import os
import sys

# single-threaded BLAS: the oracle's matrices are small, OpenBLAS thread spin-up dominates otherwise
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a Jacobi SVD that hits its sweep limit (mpbp_stats.jacobi_not_converged) is surfaced by the host mirror as a
    # RuntimeWarning: in the test suite it is a failure
    config.addinivalue_line("filterwarnings", "error:libmpbp_hip:RuntimeWarning")
