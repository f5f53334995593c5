"""Cost of moving the whole message state across the boundary at BASELINE configs[1] (the boundary takes host buffers:
mpbp_set_messages / mpbp_get_messages; the bench times sweeps on device-resident state).  Prints the C-call times of one
full download and one full upload of all E messages, the per-sweep read-out the reference's callback needs (beliefs), and
the PCIe-inclusive rate a caller would see who moved everything every sweep."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import networkx as nx  # noqa: E402
import numpy as np  # noqa: E402
import mpbp_amd as M  # noqa: E402

N, T, Mb = 1024, 50, 20
G = nx.random_regular_graph(3, N, seed=0)
A = nx.to_numpy_array(G, nodelist=range(N))
phi = [[np.array([0.9, 0.1]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(0.1, 0.05)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
allnodes = np.arange(N, dtype=np.int32)
for s in range(4):
    t0 = time.time(); M.onebpiter(bp, allnodes, M.TruncBond(Mb)); ts = time.time() - t0
    print(f"sweep {s}: {ts:.2f} s", flush=True)
E = bp.g.ne()
b = bp.bonds()
sizes = (b[:, :-1].astype(np.int64) * b[:, 1:] * 4).sum(axis=1)
offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
data = np.zeros(int(sizes.sum()))
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
t0 = time.time(); rc = bp._L.mpbp_get_messages(bp._h, offs.ctypes.data_as(C.POINTER(C.c_int64)), dp(data)); tg = time.time() - t0
assert rc == 0
t0 = time.time(); rc = bp._L.mpbp_set_messages(bp._h, b.ctypes.data_as(C.POINTER(C.c_int32)), offs.ctypes.data_as(C.POINTER(C.c_int64)), dp(data)); tu = time.time() - t0
assert rc == 0
t0 = time.time(); bel = M.beliefs(bp); tb = time.time() - t0
gb = data.nbytes / 1e9
print(f"messages: {gb:.2f} GB; download {tg:.3f} s ({gb / tg:.1f} GB/s), upload {tu:.3f} s ({gb / tu:.1f} GB/s), beliefs read-out {tb * 1e3:.1f} ms")
print(f"sweep {ts:.2f} s -> {E / ts:.0f} edge-updates/s resident; {E / (ts + tg + tu):.0f} if every message crossed PCIe both ways every sweep; "
      f"{E / (ts + tb):.0f} with the per-sweep belief read-out the reference's callback needs")
