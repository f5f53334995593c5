"""Times the grid-level batched QR (mpbp_selftest_qr_batched) against the workgroup-per-problem QR (mpbp_selftest_qr_bench)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpbp_amd  # noqa: E402

L = mpbp_amd._lib.lib()
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
for rows, cols, nprob, tall in [(1600, 400, 512, 0), (1600, 400, 512, 1), (6400, 1600, 16, 0), (16384, 4096, 1, 0), (21600, 900, 16, 0), (7200, 900, 128, 0), (5400, 900, 256, 0)]:
    if len(sys.argv) > 1 and str(cols) not in sys.argv[1:] and f"{rows}x{cols}x{nprob}" not in sys.argv[1:]:
        continue
    rng = np.random.default_rng(0)
    A1 = rng.standard_normal(rows * cols)
    A = np.tile(A1, nprob)
    k = min(rows, cols)
    R = np.zeros(nprob * k * cols)
    ms = C.c_double(0)
    for rep in range(2):
        rc = L.mpbp_selftest_qr_batched(0, rows, cols, nprob, tall, dp(A), dp(R), C.byref(ms))
        assert rc == 0
    fl = (2.0 * rows * cols * cols - 2.0 / 3.0 * cols ** 3) * nprob
    line = f"batched rows={rows} cols={cols} nprob={nprob} tall={tall}: {ms.value:.2f} ms  {fl / ms.value * 1e-9:.2f} TFLOP/s"
    if rows <= 2048:
        ms1 = C.c_double(0)
        L.mpbp_selftest_qr_bench(0, rows, cols, nprob, 2, C.byref(ms1))
        line += f"   | workgroup-per-problem: {ms1.value:.2f} ms  {fl / ms1.value * 1e-9:.2f} TFLOP/s"
    print(line, flush=True)
