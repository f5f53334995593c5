import ctypes as C, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import mpbp_amd
L = mpbp_amd._lib.lib()
ms = C.c_double()
shapes = [(1600, 400, 256), (1600, 400, 512), (400, 80, 256)] if len(sys.argv) < 2 else [(1600, 400, 256)]
reps = 3 if len(sys.argv) < 2 else 1
for (r, c, nb) in shapes:
    L.mpbp_selftest_qr_bench(0, r, c, nb, reps, C.byref(ms))
    fl = 2.0 * c * c * (r - c / 3.0) * nb
    print(f"QR {r}x{c} x{nb} blocks: {ms.value:.2f} ms/launch  {fl/ms.value/1e9:.2f} TFLOP/s  ({ms.value*256/nb:.2f} ms per problem-wave)")
