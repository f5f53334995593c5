"""Per-phase split (MPBP_QR_PROF=1) of the workgroup-per-problem 1600x400 QR with 1 and 256 workgroups resident."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpbp_amd
L = mpbp_amd._lib.lib()
for nprob in [1, 256]:
    ms = C.c_double(0)
    L.mpbp_selftest_qr_bench(0, 1600, 400, nprob, 2, C.byref(ms))
    print(nprob, ms.value, flush=True)
