"""Batched R-only QR (mpbp_selftest_qr_batched): the communication-avoiding form against the launch-per-panel form
(MPBP_DEBUG_NO_CAQR=1 in the environment selects the latter) over shapes x problem counts.  usage: qrbench3.py [rowsxcolsxnprob ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpbp_amd  # noqa: E402

L = mpbp_amd._lib.lib()
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(23400, 900, 1), (23400, 900, 4), (21600, 900, 16), (6400, 1600, 1), (6400, 1600, 4), (6400, 1600, 16), (16384, 4096, 1), (7200, 900, 32), (7200, 900, 128)]
tag = "launch-per-panel" if os.environ.get("MPBP_DEBUG_NO_CAQR") else "communication-avoiding"
for rows, cols, nprob in shapes:
    rng = np.random.default_rng(0)
    A = np.tile(rng.standard_normal(rows * cols), nprob)
    R = np.zeros(nprob * min(rows, cols) * cols)
    ms = C.c_double(0)
    best = 1e30
    for rep in range(3):
        rc = L.mpbp_selftest_qr_batched(0, rows, cols, nprob, 0, dp(A), dp(R), C.byref(ms))
        assert rc == 0
        best = min(best, ms.value)
    fl = (2.0 * rows * cols * cols - 2.0 / 3.0 * cols ** 3) * nprob
    print(f"{tag:24s} {rows:6d} x {cols:5d} x {nprob:4d}: {best:8.2f} ms  {fl / best * 1e-9:6.2f} TFLOP/s", flush=True)
