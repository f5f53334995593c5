"""Generates the golden fixtures of tests/golden/ from the CPU oracle (and stores the reference's own
hard-coded known answer verbatim).  Run from the repo root:  python tests/golden/make_golden.py
The reference itself is Julia and cannot run in this image, so fixtures are (a) the numbers the reference's
tests hold (test/sis_infinite_graph.jl:21-29) and (b) oracle outputs on seeded inputs, the oracle being pinned
to (a) and to brute-force enumeration by tests/test_oracle.py."""
import json
import os
import sys

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import networkx as nx  # noqa: E402
import numpy as np  # noqa: E402

from oracle import factors as OF  # noqa: E402
from oracle import mpbp as O  # noqa: E402
from oracle import tensor_trains as OT  # noqa: E402
from oracle.exact import exact_marginals, exact_pair_marginals, exact_prob  # noqa: E402


def known_answer():
    ref = [[0.9000000001671186, 0.0999999998328814], [0.8932690998131098, 0.10673090018689023],
           [0.8899420329322244, 0.11005796706777556], [0.8884643888492034, 0.11153561115079656],
           [0.8880305235706524, 0.1119694764293476], [0.8882121515614524, 0.11178784843854758],
           [0.8887717202217936, 0.1112282797782064]]
    json.dump({"source": "reference test/sis_infinite_graph.jl:1-30 (verbatim)",
               "params": {"T": 6, "k": 3, "gamma": 0.1, "lambda": 0.1, "rho": 0.2, "svd_trunc": "TruncBond(10)",
                          "maxiter": 200, "tol": 1e-14},
               "beliefs": ref}, open(os.path.join(HERE, "sis_infinite_graph_reference.json"), "w"), indent=1)


def rrg_sweeps(N=16, T=10, Mb=8, sweeps=3, name="sis_rrg16_T10_M8_jacobi.npz"):
    lam, rho, gam = 0.1, 0.05, 0.1
    A = nx.to_numpy_array(nx.random_regular_graph(3, N, seed=0), nodelist=range(N))
    phi = [[np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    bp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
    out = {"A": A, "params": np.array([N, T, Mb, sweeps, lam, rho, gam])}
    for s in range(sweeps):
        O.iterate(bp, maxiter=1, svd_trunc=OT.TruncBond(Mb), tol=0.0, shuffle_nodes=False, jacobi=True)
        out[f"beliefs_{s}"] = np.array(O.beliefs(bp))
        out[f"f_{s}"] = bp.f.copy()
    pb, lz = O.pair_beliefs(bp)
    out["pair_beliefs"] = np.array(pb)
    out["pair_logz"] = lz
    out["bonds"] = np.array([m.bonds for m in bp.mu])
    np.savez_compressed(os.path.join(HERE, name), **out)


def star_exact():
    T = 3
    A = np.array([[0, 1, 1, 1], [1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0]])
    lam, rho, gam, alpha = 0.5, 0.4, 0.5, 0.1
    phi = [[np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(4)]
    rng = np.random.default_rng(111)
    for i in range(4):
        phi[i][T] = np.array([1.0, 0.0]) if rng.random() < 0.5 else np.array([0.0, 1.0])
    bp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho, alpha)] * (T + 1)] * 4, [2] * 4, T, phi=phi)
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(bp)
    np.savez_compressed(os.path.join(HERE, "sis_star4_T3_exact.npz"), A=A, phi=np.array(phi),
                        params=np.array([T, lam, rho, gam, alpha]), marginals=np.array(exact_marginals(bp, p)),
                        pair_marginals=np.array(exact_pair_marginals(bp, p)), Z=Z)


def _save_sweeps(bp, sweeps, trunc, name, extra, jacobi=True):
    out = dict(extra)
    import time
    for s in range(sweeps):
        t0 = time.time()
        O.iterate(bp, maxiter=1, svd_trunc=trunc, tol=0.0, shuffle_nodes=False, jacobi=jacobi)
        out[f"beliefs_{s}"] = np.array(O.beliefs(bp))
        out[f"f_{s}"] = bp.f.copy()
        print(name, "sweep", s, f"{time.time() - t0:.1f} s", "max bond", max(max(m.bonds) for m in bp.mu), flush=True)
    out["bonds"] = np.array([m.bonds for m in bp.mu])
    if jacobi:
        pb, lz = O.pair_beliefs(bp)
        out["pair_beliefs"] = np.array(pb)
        out["pair_logz"] = lz
    np.savez_compressed(os.path.join(HERE, name), **out)


def bondcap_infinite(T=8, Mb=64, iters=5, name="sis_inf_k3_T8_M64.npz"):
    """BASELINE configs[4] at its bond cap (TruncBond(64): product bond 4096), chain shortened to T = 8 so that the
    oracle finishes offline (reference test/sis_infinite_graph.jl:3-12 scaled; sequential in-place iterations)."""
    k, gam, lam, rho = 3, 0.1, 0.1, 0.2
    phi = [np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)]
    bp = O.mpbp_infinite_graph(k, [OF.SISFactor(lam, rho) for _ in range(T + 1)], 2, phi)
    _save_sweeps(bp, iters, OT.TruncBond(Mb), name, {"params": np.array([k, T, Mb, iters, lam, rho, gam])}, jacobi=False)


def bondcap_karate(T=6, Mb=40, sweeps=5, name="sis_karate8_T6_M40.npz"):
    """BASELINE configs[3] at its bond cap (TruncBond(40): product bond 1600) on the induced subgraph of the karate
    club on node 0 and its first 7 neighbours (node 0 keeps degree 7), T = 6."""
    Afull = np.loadtxt(os.path.join(HERE, "karate.txt"))
    nb = [0] + list(np.nonzero(Afull[:, 0])[0][:7])
    A = Afull[np.ix_(nb, nb)]
    N = len(nb)
    lam, rho = 0.1, 0.05
    phi = [[np.array([0.0, 1.0]) if (t == 0 and i == 0) else (np.array([1.0, 0.0]) if t == 0 else np.ones(2))
            for t in range(T + 1)] for i in range(N)]
    bp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
    _save_sweeps(bp, sweeps, OT.TruncBond(Mb), name, {"A": A, "params": np.array([N, T, Mb, sweeps, lam, rho])})


def bondcap_glauber(T=6, Mb=30, sweeps=5, name="glauber_er8_T6_M30.npz"):
    """BASELINE configs[2] at its bond cap (TruncBond(30): product bond 900, nstates = l+1 growing to degree 5) on an
    8-node Erdos-Renyi graph that has a node of degree >= 5, T = 6 (glauber_bp.jl:22-44,128-131)."""
    N = 8
    for seed in range(100):
        G = nx.gnp_random_graph(N, 4 / 7, seed=seed)
        deg = [d for _, d in G.degree()]
        if max(deg) >= 5 and nx.is_connected(G) and sum(deg) <= 28:
            break
    A = nx.to_numpy_array(G, nodelist=range(N))
    J = 0.5 * A
    h = np.zeros(N)
    m0 = -0.6
    phi = [[np.array([(1 + m0) / 2, (1 - m0) / 2]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    bp = O.mpbp(O.IndexedBiDiGraph(A), OF.glauber_factors(A != 0, J, h, 1.0, T), [2] * N, T, phi=phi)
    _save_sweeps(bp, sweeps, OT.TruncBond(Mb), name, {"A": A, "params": np.array([N, T, Mb, sweeps, 0.5, 0.0, 1.0, m0, seed])})


def hub_update(kind):
    """ONE update (reference onebpiter!, src/recursive_bp_factor.jl:146-165) of a HIGH-DEGREE node from seeded random
    messages at the saturated bond profile on all its in-edges - the shape class that BASELINE configs[2] / [3] create at
    their hubs (cavity chains of 3z-2 products at the bond cap) and that whole-graph fixtures cannot reach offline:
      glauber9  homogeneous Glauber (glauber_bp.jl:22-44), degree 9 (nstates grows to 10: sweep-2 factors of 600 columns,
                Y_t up to 900 x 20 rows per rank index), T = 6, TruncBond(30)
      karate17  SIS (sis_bp.jl), degree 17 = the larger hub of notebooks/karate.txt (49 products in CavityTools order),
                T = 6, TruncBond(40)
    The graph is the star hub + z leaves: with given in-messages the update of the hub does not see anything else.
    Inputs are NOT stored: they are `random_message(T, q, Mb, rng)` of the product package drawn edge by edge from
    numpy.random.default_rng(seed) (the test re-creates them); outputs: hub belief, f[hub], pair beliefs and bonds of the
    hub's edges."""
    import time
    from mpbp_amd import random_message
    z, T, Mb, seed = {"glauber9": (9, 6, 30, 11), "karate17": (17, 6, 40, 12)}[kind]
    N = z + 1
    A = np.zeros((N, N))
    A[0, 1:] = A[1:, 0] = 1
    g = O.IndexedBiDiGraph(A)
    if kind == "glauber9":
        m0 = -0.6
        phi = [[np.array([(1 + m0) / 2, (1 - m0) / 2]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
        w = OF.glauber_factors(A != 0, 0.5 * A, np.zeros(N), 1.0, T)
        params = np.array([z, T, Mb, seed, 0.5, 0.0, 1.0, m0])
    else:
        lam, rho = 0.1, 0.05
        phi = [[np.array([0.0, 1.0]) if (t == 0 and i == 1) else (np.array([1.0, 0.0]) if t == 0 else np.ones(2))
                for t in range(T + 1)] for i in range(N)]
        w = [[OF.SISFactor(lam, rho)] * (T + 1)] * N
        params = np.array([z, T, Mb, seed, lam, rho])
    bp = O.mpbp(g, w, [2] * N, T, phi=phi)
    rng = np.random.default_rng(seed)
    for (_, _, e) in g.inedges(0):
        bp.mu[e] = OT.TensorTrain(random_message(T, 2, Mb, rng))
    t0 = time.time()
    O.onebpiter(bp, 0, OT.TruncBond(Mb))
    print(kind, f"hub update {time.time() - t0:.1f} s", flush=True)
    pb, lz = O.pair_beliefs(bp)
    hub_edges = [e for (_, _, e) in g.inedges(0)] + [e for (_, _, e) in g.outedges(0)]
    np.savez_compressed(os.path.join(HERE, f"hub_{kind}.npz"), params=params, belief=np.array(O.beliefs(bp)[0]), f=bp.f[0],
                        hub_edges=np.array(hub_edges), pair_beliefs=np.array([pb[e] for e in hub_edges]),
                        out_bonds=np.array([bp.mu[e].bonds for (_, _, e) in g.outedges(0)]))


if __name__ == "__main__":
    if "--hub" in sys.argv:
        # OPENBLAS_NUM_THREADS=6 python tests/golden/make_golden.py --hub glauber9|karate17      (minutes each)
        hub_update(sys.argv[sys.argv.index("--hub") + 1])
        sys.exit(0)
    if "--bondcap" in sys.argv:
        # fixtures at the bond caps of BASELINE configs[2..4] (reduced N / T); minutes to tens of minutes each:
        #   OPENBLAS_NUM_THREADS=4 python tests/golden/make_golden.py --bondcap glauber|karate|infinite
        {"glauber": bondcap_glauber, "karate": bondcap_karate, "infinite": bondcap_infinite}[sys.argv[sys.argv.index("--bondcap") + 1]]()
        sys.exit(0)
    known_answer()
    rrg_sweeps()
    star_exact()
    if "--production-dims" in sys.argv:
        # BASELINE configs[1] dimensions (T = 50, TruncBond(20): product bond 400) on 8 nodes; the oracle needs about
        # 100 s per saturated sweep, so this one is only regenerated on request
        rrg_sweeps(8, 50, 20, 5, "sis_rrg8_T50_M20_jacobi.npz")
    print(sorted(os.listdir(HERE)))
