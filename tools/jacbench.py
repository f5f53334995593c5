"""Jacobi SVD building block, LDS resident: 512-thread and single-wave variants."""
import ctypes as C, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import mpbp_amd
L = mpbp_amd._lib.lib()
L.mpbp_selftest_jacobi_bench.argtypes = [C.c_int32] * 6 + [C.POINTER(C.c_double)] * 2
ms, sw = C.c_double(), C.c_double()
for (m, n, nb, var) in [(80, 80, 256, 0), (80, 40, 256, 0), (80, 40, 1024, 1), (80, 20, 1024, 1), (80, 80, 1024, 1)]:
    rc = L.mpbp_selftest_jacobi_bench(0, m, n, nb, var, 3, C.byref(ms), C.byref(sw))
    print(f"jacobi {m}x{n} x{nb} variant {'v64' if var else 'v512'}: rc={rc} {ms.value:.3f} ms/launch, {sw.value:.1f} sweeps, {ms.value*1e3/sw.value:.1f} us/sweep", flush=True)
