"""One-off parity check at the production dimensions of BASELINE configs[1] (T = 50, TruncBond(20): product bond
400, 1600 x 400 QRs, 4-panel updates) on a small 3-regular graph, against the numpy oracle, sweep by sweep.
Test infrastructure (imports oracle/); the oracle needs ~12 s per node update, so this is a tool, not a test.
usage: python tools/parity_d20.py [n_nodes=8] [sweeps=3]"""
import os
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import networkx as nx
import numpy as np
import mpbp_amd as M
import oracle.factors as OF
import oracle.mpbp as O
import oracle.tensor_trains as OT

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T, Mb, lam, rho, gam = 50, 20, 0.1, 0.05, 0.1
A = nx.to_numpy_array(nx.random_regular_graph(3, N, seed=0), nodelist=range(N))
phi = [[np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
for s in range(sweeps):
    t0 = time.time()
    M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(Mb), tol=0.0)
    t1 = time.time()
    O.iterate(obp, maxiter=1, svd_trunc=OT.TruncBond(Mb), tol=0.0, shuffle_nodes=False, jacobi=True)
    t2 = time.time()
    b, ob = np.array(M.beliefs(bp)), np.array(O.beliefs(obp))
    f, of = M.bethe_free_energy(bp), O.bethe_free_energy(obp)
    pb, opb = M.pair_beliefs(bp)[0], O.pair_beliefs(obp)[0]
    e3 = max(np.abs(np.array(x) - np.array(y)).max() for x, y in zip(pb, opb))
    print(f"sweep {s}: max bond {int(bp.bonds().max())}  beliefs {np.abs(b - ob).max():.2e}  pair beliefs {e3:.2e}  "
          f"free energy {abs(f - of) / max(1, abs(of)):.2e}   (device {t1 - t0:.1f} s, oracle {t2 - t1:.1f} s)", flush=True)
