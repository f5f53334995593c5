"""Timing of BASELINE configs[2] phases on one GPU (the full-size test's steps), progress to stdout."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, networkx as nx
import mpbp_amd as M
N, T, Mb = int(os.environ.get('CFG3_N', '2048')), int(sys.argv[1]) if len(sys.argv) > 1 else 100, 30
G = nx.gnp_random_graph(N, 4 / (N - 1), seed=0)
A = nx.to_numpy_array(G, nodelist=range(N))
m0 = -0.6
phi = [[np.array([(1 + m0) / 2, (1 - m0) / 2]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
t0 = time.time()
bp = M.Glauber(M.Ising(0.5 * A, np.zeros(N), 1.0), T, phi=phi).mpbp(max_bond=Mb)
print("setup", time.time() - t0, flush=True)
deg = A.sum(axis=0).astype(int)
allnodes = np.arange(N, dtype=np.int32)
if os.environ.get("CFG3_RANDOM"):
    rng = np.random.default_rng(5)
    ptr, ine, oute = bp.g.nbr_arrays()
    tgt = [int(i) for i in np.nonzero((deg >= 2) & (deg <= 5))[0][:48]] + ([int(np.argmax(deg))] if os.environ.get('CFG3_HUB') else [])
    msgs = [None] * bp.g.ne()
    for i in tgt:
        for p_ in range(ptr[i], ptr[i + 1]):
            msgs[int(ine[p_])] = M.random_message(T, 2, Mb, rng)
    bp.set_messages(msgs)
else:
    for s in range(2):
        t0 = time.time(); M.onebpiter(bp, allnodes, M.TruncBond(Mb)); print("full sweep", s, time.time() - t0, "maxbond", bp.bonds().max(), flush=True)
hub = int(np.argmax(deg))
subs = [] if (os.environ.get('CFG3_ONLY_FULL') or os.environ.get('CFG3_HUB')) else [[int(np.nonzero(deg == 4)[0][0])], [int(i) for i in np.nonzero((deg >= 2) & (deg <= 5))[0][:48]]][:(0 if os.environ.get('CFG3_ONLY_FULL') else 2)]
if len(sys.argv) > 2: subs.append([hub])
for sub in subs:
    t0 = time.time(); M.onebpiter(bp, np.array(sub, dtype=np.int32), M.TruncBond(Mb)); st = bp.last_stats
    print("nodes", sub[:4], len(sub), "deg", deg[sub][:8], time.time() - t0, "s", "n_compress", st.n_compress, "flags", st.nan_flag, st.capacity_flag, st.jacobi_not_converged, flush=True)
