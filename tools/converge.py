"""Convergence run of BASELINE configs[1] (SIS, 3-regular N=1024, T=50, bond 20): 25 Jacobi sweeps through `iterate`
with the reference callback; prints the belief changes per sweep.  usage: python tools/converge.py"""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import networkx as nx
import mpbp_amd as M
N, T, Mb = 1024, 50, 20
lam, rho, gam = 0.1, 0.05, 0.1
g = M.IndexedBiDiGraph(nx.to_numpy_array(nx.random_regular_graph(3, N, seed=0), nodelist=range(N)))
phi = [[np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
bp = M.mpbp(g, [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
cb = M.CB_BP(bp)
t0 = time.time()
it, cb = M.iterate(bp, maxiter=25, svd_trunc=M.TruncBond(Mb), tol=1e-9, cb=cb)
print("iterations", it, "time %.1f s" % (time.time() - t0))
print("deltas", ["%.2e" % d for d in cb.deltas] if hasattr(cb, "deltas") else getattr(cb, "Deltas", None))
b = np.array(M.beliefs(bp)); print("belief sums ok", np.abs(b.sum(axis=2) - 1).max(), "min", b.min(), "F", M.bethe_free_energy(bp))
st = bp.last_stats; print("flags nan", st.nan_flag, "cap", st.capacity_flag, "jac", st.jacobi_not_converged)
