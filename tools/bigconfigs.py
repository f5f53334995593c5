"""BASELINE configs[3] (karate, T=200, d=40) and configs[4] (infinite graph k=3, T=200, d=64) at their stated sizes:
sweep times and sanity flags.  Usage: python tools/bigconfigs.py [inf|karate] [T] [bond] [sweeps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import mpbp_amd as M  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "inf"
if which == "inf":
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    Mb = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    ns = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    phi = [np.array([0.9, 0.1]) if t == 0 else np.ones(2) for t in range(T + 1)]
    bp = M.mpbp_infinite_graph(3, [M.SISFactor(0.1, 0.2)] * (T + 1), 2, phi, max_bond=Mb)
    for s in range(ns):
        t0 = time.time(); M.onebpiter(bp, [0], M.TruncBond(Mb)); st = bp.last_stats
        b = np.array(M.beliefs(bp)[0])
        print(f"infinite k=3 T={T} d={Mb} iter {s}: {time.time()-t0:.2f}s flags nan={st.nan_flag} cap={st.capacity_flag} "
              f"jac={st.jacobi_not_converged} jacobi sweeps/call={st.jacobi_sweeps / max(1, st.jacobi_calls):.1f} maxbond={bp.bonds().max()} b[T]={b[-1]} sum-err={np.abs(b.sum(axis=1)-1).max():.1e}", flush=True)
else:
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    Mb = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    ns = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    A = np.loadtxt(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "karate.txt"))
    N = 34
    phi = [[np.array([0.0, 1.0]) if (t == 0 and i == 0) else (np.array([1.0, 0.0]) if t == 0 else np.ones(2)) for t in range(T + 1)] for i in range(N)]
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(0.1, 0.05)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
    for s in range(ns):
        t0 = time.time(); M.onebpiter(bp, np.arange(N, dtype=np.int32), M.TruncBond(Mb)); st = bp.last_stats
        b = np.array(M.beliefs(bp))
        print(f"karate T={T} d={Mb} sweep {s}: {time.time()-t0:.2f}s n_compress={st.n_compress} flags nan={st.nan_flag} cap={st.capacity_flag} "
              f"jac={st.jacobi_not_converged} jacobi sweeps/call={st.jacobi_sweeps / max(1, st.jacobi_calls):.1f} maxbond={bp.bonds().max()} belief sum err={np.abs(b.sum(axis=2)-1).max():.2e} min={b.min():.2e}", flush=True)
