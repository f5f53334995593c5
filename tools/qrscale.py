"""QR building block at different numbers of concurrent workgroups (contention vs latency floor)."""
import ctypes as C, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import mpbp_amd
L = mpbp_amd._lib.lib()
ms = C.c_double()
for nb in (16, 64, 128, 256):
    L.mpbp_selftest_qr_bench(0, 1600, 400, nb, 2, C.byref(ms))
    print(f"QR 1600x400 x{nb} blocks: {ms.value:.2f} ms/launch", flush=True)
