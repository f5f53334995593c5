"""Workgroup-per-problem QR (mpbp_selftest_qr_bench) at different numbers of concurrently resident problems: separates the
intrinsic per-workgroup speed (few problems, everything cache resident, no contention) from memory-system effects."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpbp_amd  # noqa: E402

L = mpbp_amd._lib.lib()
rows, cols = 1600, 400
fl = 2.0 * rows * cols * cols - 2.0 / 3.0 * cols ** 3
for nprob in [1, 8, 32, 64, 128, 256, 512]:
    ms = C.c_double(0)
    L.mpbp_selftest_qr_bench(0, rows, cols, nprob, 2, C.byref(ms))
    L.mpbp_selftest_qr_bench(0, rows, cols, nprob, 2, C.byref(ms))
    print(f"nprob={nprob:4d}: {ms.value:8.3f} ms  {fl * nprob / ms.value * 1e-9:7.2f} TFLOP/s  per-workgroup {fl / ms.value * 1e-6 * (nprob / max(1, -(-nprob // 256))) / min(nprob, 256):7.2f} GFLOP/s", flush=True)
