"""Round-4 review item 1(a): can the gauge sweep of `op` (sweep 1 of the device algorithm) be an
all-GEMM Cholesky form instead of the blocked Householder QR?  Zero GPU minutes: the numpy restatement of the device
algorithm (oracle/device_algorithm.py) is run with four forms of the triangular factor

    householder (shipped) | gram (one Cholesky of Y^T Y) | cholqr2 (shifted CholeskyQR2) | cholqr3

inside full BP runs, each against the REFERENCE-algorithm oracle (two SVD sweeps, oracle/mpbp.py::op_kron_compress =
recursive_bp_factor.jl:118-131) on

  (1) the random-train cases of tests/test_oracle.py::test_device_algorithm_equals_reference_compress_with_binding_truncation,
  (2) the reference's known answer (test/sis_infinite_graph.jl:1-30: TruncBond(10), 200 iterations),
  (3) 10 Jacobi sweeps of the configs[1]-sized fixture case sis_rrg8 (T = 50, TruncBond(20): 1600 x 400 factors).

Test infrastructure (imports oracle/).  usage: python tools/gram_gauge_experiment.py [sweeps=10] [workers=8] > profiles/r04_gram_gauge_errors.txt
"""
import json
import multiprocessing as mp
import os
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import networkx as nx  # noqa: E402
import numpy as np  # noqa: E402

import oracle.device_algorithm as DA  # noqa: E402
import oracle.factors as OF  # noqa: E402
import oracle.mpbp as O  # noqa: E402
import oracle.tensor_trains as OT  # noqa: E402

REF_OP = O.op_kron_compress
GAUGES = ["householder", "gram", "cholqr2", "cholqr3"]


def use(gauge):
    """route every `op` of the oracle's BP through the device algorithm with the given gauge form (None = reference)"""
    if gauge is None:
        O.op_kron_compress = REF_OP
    else:
        O.op_kron_compress = lambda wi, a, b, T, tr: DA.op_kron_compress_qr(wi, a, b, T, tr, gauge=gauge)


def relerr(x, ref):
    x, ref = np.asarray(x), np.asarray(ref)
    return float((np.abs(x - ref) / np.maximum(np.abs(ref), 1e-300)).max())


# ---------------------------------------------------------------- (1) random trains, binding truncation
def case_random_trains():
    print("(1) random trains of tests/test_oracle.py (T = 5, bonds 1-3-5-5-4-2-1, 40 evaluations each): max relative error of the"
          "\n    truncated function against the reference's two SVD sweeps")
    for gauge in GAUGES:
        rng = np.random.default_rng(7)
        T, worst = 5, 0.0
        for (w, d1, d2) in [(OF.SISFactor(0.2, 0.1), 1, 1), (OF.HomogeneousGlauberFactor(0.4, 0.1, 1.0), 2, 1)]:
            wi = [w] * (T + 1)
            prof = [1, 3, 5, 5, 4, 2, 1]

            def rand_train(d):
                ny = w.nstates(d)
                return OT.TensorTrain([rng.random((prof[t], prof[t + 1], ny, 2)) + 0.05 for t in range(T + 1)], logz=0.3 * d)
            for trunc_f in (lambda: OT.TruncBond(4), lambda: OT.TruncBondMax(3), lambda: OT.TruncThresh(1e-3)):
                A, B = rand_train(d1), rand_train(d2)
                ref, _ = REF_OP(wi, (A.copy(), d1), (B.copy(), d2), T, trunc_f())
                dev, _ = DA.op_kron_compress_qr(wi, (A.copy(), d1), (B.copy(), d2), T, trunc_f(), gauge=gauge)
                assert ref.bonds == dev.bonds
                ny = w.nstates(d1 + d2)
                for _ in range(40):
                    x = [(int(rng.integers(ny)), int(rng.integers(2))) for _ in range(T + 1)]
                    ra, rb = OT.evaluate(ref, x), OT.evaluate(dev, x)
                    worst = max(worst, abs(ra - rb) / max(abs(ra), 1e-300))
        print(f"    {gauge:12s} {worst:.2e}")


# ---------------------------------------------------------------- (2) the reference's known answer
def case_known_answer():
    ref = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "sis_infinite_graph_reference.json")))["beliefs"])
    T, k, gam, lam, rho = 6, 3, 0.1, 0.1, 0.2
    phi = [np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)]
    print("(2) known answer of test/sis_infinite_graph.jl (T = 6, k = 3, TruncBond(10), 200 iterations, tol 1e-14): max relative"
          "\n    belief error against the reference's 14 numbers / against the reference-algorithm oracle; free energy")
    res = {}
    for gauge in [None] + GAUGES:
        use(gauge)
        bp = O.mpbp_infinite_graph(k, [OF.SISFactor(lam, rho) for _ in range(T + 1)], 2, phi)
        it, _ = O.iterate(bp, maxiter=200, svd_trunc=OT.TruncBond(10), tol=1e-14)
        res[gauge] = (np.array(O.beliefs(bp)[0]), O.bethe_free_energy(bp), it)
    use(None)
    b0, f0, _ = res[None]
    for gauge in [None] + GAUGES:
        b, f, it = res[gauge]
        print(f"    {str(gauge or 'reference alg.'):15s} vs vector {relerr(b, ref):.2e}   vs oracle {relerr(b, b0):.2e}   "
              f"free energy {abs(f - f0) / max(1.0, abs(f0)):.2e}   ({it} iterations)")


# ---------------------------------------------------------------- (3) configs[1] dimensions
_BP = None


def _node_update(args):
    gauge, i = args
    use(gauge)
    bp = _BP
    O.onebpiter(bp, i, OT.TruncBond(20), 0.0)
    return i, [(e[2], bp.mu[e[2]]) for e in bp.g.outedges(i)], bp.b[i], bp.f[i]


def jacobi_sweep(bp, gauge, pool):
    global _BP
    _BP = bp
    N = len(list(bp.g.vertices()))
    out = pool.map(_node_update, [(gauge, i) for i in range(N)])
    for i, mus, b, f in out:
        for e, m in mus:
            bp.mu[e] = m
        bp.b[i] = b
        bp.f[i] = f


def case_rrg8(sweeps, workers, observed):
    N, T, Mb, lam, rho, gam = 8, 50, 20, 0.1, 0.05, 0.1
    A = nx.to_numpy_array(nx.random_regular_graph(3, N, seed=0), nodelist=range(N))
    phi = [[np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    if observed:
        # noisy observations of a planted trajectory (the inference setting of the reference's tests, test/sis_small_tree.jl:
        # observations enter through phi): small marginals and a free energy away from zero
        rng = np.random.default_rng(3)
        for i in range(N):
            for t in rng.choice(np.arange(1, T + 1), size=6, replace=False):
                phi[i][t] = np.array([0.02, 0.98]) if rng.random() < 0.4 else np.array([0.98, 0.02])
        print(f"(4) the same graph and dimensions with 6 noisy observations per node (2 % flip rate) - small marginals, F != 0; {sweeps} sweeps,"
              "\n    same columns (free-energy terms: max |df_i| / max(1, max |f_i|))")
    else:
        print(f"(3) sis_rrg8_T50_M20 (the configs[1]-sized fixture: 3-regular, N = 8, T = 50, TruncBond(20), Y_t 1600 x 400), {sweeps} Jacobi"
              "\n    sweeps, every form on its own trajectory against the reference-algorithm oracle on its own:"
              "\n    max relative error of beliefs / pair beliefs (entries > 1e-12) / per-node free-energy terms (max |df_i| / max(1, max |f_i|)),"
              "\n    and the smallest belief entry")
    runs = {}
    for gauge in [None] + GAUGES:
        runs[gauge] = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
    fix = np.load(os.path.join(ROOT, "tests", "golden", "sis_rrg8_T50_M20_jacobi.npz"))
    worst = {g: [0.0, 0.0, 0.0] for g in GAUGES}
    for s in range(sweeps):
        t0 = time.time()
        obs = {}
        for gauge in [None] + GAUGES:
            global _BP
            _BP = runs[gauge]
            with mp.get_context("fork").Pool(workers) as pool:          # fork AFTER _BP is set: workers see the snapshot
                jacobi_sweep(runs[gauge], gauge, pool)
            use(None)
            pb, _ = O.pair_beliefs(runs[gauge])
            obs[gauge] = (np.array(O.beliefs(runs[gauge])), np.concatenate([np.ravel(p) for p in pb]), np.array(runs[gauge].f))
        b0, p0, f0 = obs[None]
        if not observed and f"beliefs_{s}" in fix:
            assert relerr(b0, fix[f"beliefs_{s}"]) < 1e-9                  # the committed fixture is this run
        line = f"    sweep {s + 1:2d}  max bond {max(max(m.bonds) for m in runs[None].mu):3d}  min belief {b0.min():.1e}"
        for gauge in GAUGES:
            b, p, f = obs[gauge]
            big = p0 > 1e-12
            e = (relerr(b, b0), relerr(p[big], p0[big]), float(np.abs(f - f0).max() / max(1.0, np.abs(f0).max())))
            worst[gauge] = [max(a, c) for a, c in zip(worst[gauge], e)]
            line += f"   {gauge}: {e[0]:.1e} / {e[1]:.1e} / {e[2]:.1e}"
        print(line + f"   [{time.time() - t0:.0f} s]", flush=True)
    print("    worst over the run:")
    for gauge in GAUGES:
        print(f"      {gauge:12s} beliefs {worst[gauge][0]:.2e}   pair beliefs {worst[gauge][1]:.2e}   free-energy terms {worst[gauge][2]:.2e}")


if __name__ == "__main__":
    sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    case_random_trains()
    case_known_answer()
    case_rrg8(sweeps, workers, False)
    case_rrg8(sweeps, workers, True)
