"""Device path vs the restated periodic oracle (oracle/periodic.py) on the reference's periodic loopy case
(test/periodic.jl:70-110: homogeneous Glauber on the complete graph of 4 nodes, T = 2, damp 0.2), Jacobi sweeps,
with a binding cap (TruncBond(10), the reference's) and a non-binding one.  Prints max |belief difference| per sweep."""
import os, sys
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")      # the GPU box shows 256 hardware threads behind a 16-CPU quota
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpbp_amd as M
from oracle import factors as OF, mpbp as O, periodic as OP, tensor_trains as OT

T, k, m0 = 2, 3, 0.5
N = k + 1
A = np.ones((N, N)) - np.eye(N)
phi_i = [np.array([(1 + m0) / 2, (1 - m0) / 2]) if t == 0 else np.ones(2) for t in range(T + 1)]
phi_i[1] = np.array([0.4, 0.6]); phi_i[T] = np.array([0.95, 0.05])
for cap in (10, 256):
    for damp in (0.0, 0.2):
        bp = M.periodic_mpbp(M.IndexedBiDiGraph(A), [[M.HomogeneousGlauberFactor(1.0, 0.0, 1.0)] * (T + 1)] * N, 2, T, phi=[phi_i] * N, max_bond=min(cap, 64))
        obp = OP.periodic_mpbp(O.IndexedBiDiGraph(A), [[OF.HomogeneousGlauberFactor(1.0, 0.0, 1.0)] * (T + 1)] * N, [2] * N, T, phi=[phi_i] * N)
        dmax = []
        for s in range(6):
            M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(min(cap, 64)), tol=0.0, damp=damp)
            OP.iterate(obp, 1, OT.TruncBond(cap), damp=damp, jacobi=True)
            d = max(np.abs(np.array(M.beliefs(bp)[i]) - np.array(OP.beliefs(obp)[i])).max() for i in range(N))
            dmax.append(d)
            print(f"  cap {cap} damp {damp} sweep {s}: {d:.2e}", file=sys.stderr, flush=True)
        print(f"cap {cap:3d} damp {damp}: oracle max bond {max(max(m.bonds) for m in obp.mu)}, device max bond {bp.bonds().max()}, "
              f"max |belief diff| per sweep: " + " ".join(f"{d:.1e}" for d in dmax), flush=True)
