import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
import mpbp_amd as M
A = np.loadtxt("/root/repo/tests/golden/karate.txt")
N, T, Mb = 34, 30, 24
phi = [[np.array([0.0, 1.0]) if (t == 0 and i == 0) else (np.array([1.0, 0.0]) if t == 0 else np.ones(2)) for t in range(T + 1)] for i in range(N)]
bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(0.1, 0.05)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
for s in range(4):
    t0 = time.time(); M.onebpiter(bp, np.arange(N, dtype=np.int32), M.TruncBond(Mb)); st = bp.last_stats
    b = np.array(M.beliefs(bp))
    print(f"karate T={T} d={Mb} sweep {s}: {time.time()-t0:.2f}s n_compress={st.n_compress} flags nan={st.nan_flag} cap={st.capacity_flag} jac={st.jacobi_not_converged} maxbond={bp.bonds().max()} belief sum err={np.abs(b.sum(axis=2)-1).max():.2e} min={b.min():.2e}", flush=True)
T, k, Mb = 40, 3, 32
phi = [np.array([0.9, 0.1]) if t == 0 else np.ones(2) for t in range(T + 1)]
bp = M.mpbp_infinite_graph(k, [M.SISFactor(0.1, 0.2)] * (T + 1), 2, phi, max_bond=Mb)
for s in range(5):
    t0 = time.time(); M.onebpiter(bp, [0], M.TruncBond(Mb)); st = bp.last_stats
    b = np.array(M.beliefs(bp)[0])
    print(f"infinite k=3 T={T} d={Mb} iter {s}: {time.time()-t0:.2f}s flags nan={st.nan_flag} cap={st.capacity_flag} jac={st.jacobi_not_converged} maxbond={bp.bonds().max()} b[T]={b[-1]}", flush=True)
