"""Phase split of the workgroup-form QR micro-benchmark (run with MPBP_QR_PROF=1)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpbp_amd
L = mpbp_amd._lib.lib()
ms = C.c_double(0)
for nb in (256, 64):
    L.mpbp_selftest_qr_bench(0, 1600, 400, nb, 3, C.byref(ms))
    print("nblocks", nb, "ms", ms.value, flush=True)
