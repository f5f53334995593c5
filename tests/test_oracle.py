"""Pins the CPU oracle (oracle/) against the reference's own known answers and exactness
properties.  CPU only."""
import itertools

import numpy as np
import pytest

from oracle.exact import exact_autocorrelations, exact_marginals, exact_pair_marginals, exact_prob
from oracle.factors import (DampedFactor, GenericFactor, HomogeneousGlauberFactor, IntegerGlauberFactor,
                            PMJGlauberFactor, RecursiveTraceFactor, RestrictedRecursiveBPFactor, SIRSFactor,
                            SISFactor, SISHeterogeneousFactor, glauber_factors)
from oracle.mpbp import (MPEM3, IndexedBiDiGraph, autocorrelations, beliefs, bethe_free_energy, evaluate_mpem3,
                         iterate, logprob, mpbp, mpbp_infinite_bipartite_graph, mpbp_infinite_graph, mpem2,
                         pair_beliefs)
from oracle.tensor_trains import (TruncBond, TruncBondMax, TruncBondThresh, TruncThresh, evaluate,
                                  normalization_log, orthogonalize_left, orthogonalize_right, rand_tt)

RTOL = 1.5e-8     # Julia's isapprox default (sqrt(eps)), the tolerance of every `≈` in the reference tests


def _maxerr(a, b):
    return max(np.abs(np.asarray(x) - np.asarray(y)).max() for x, y in zip(a, b))


def _flat(bb):
    return [p for b in bb for p in b]


def test_known_answer_sis_infinite_graph():
    """reference test/sis_infinite_graph.jl:1-30 - the only hard-coded numbers of the reference."""
    T, k, gam, lam, rho = 6, 3, 0.1, 0.1, 0.2
    wi = [SISFactor(lam, rho) for _ in range(T + 1)]
    phi = [np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)]
    bp = mpbp_infinite_graph(k, wi, 2, phi)
    iterate(bp, maxiter=200, svd_trunc=TruncBond(10), tol=1e-14)
    ref = [[0.9000000001671186, 0.0999999998328814],
           [0.8932690998131098, 0.10673090018689023],
           [0.8899420329322244, 0.11005796706777556],
           [0.8884643888492034, 0.11153561115079656],
           [0.8880305235706524, 0.1119694764293476],
           [0.8882121515614524, 0.11178784843854758],
           [0.8887717202217936, 0.1112282797782064]]
    b = beliefs(bp)[0]
    np.testing.assert_allclose(np.array(b), np.array(ref), rtol=RTOL, atol=0)


def _sis_star(alpha=0.1, seed=111, T=3):
    A = np.array([[0, 1, 1, 1], [1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0]])
    g = IndexedBiDiGraph(A)
    N = 4
    lam, rho, gam = 0.5, 0.4, 0.5
    w = [[SISFactor(lam, rho, alpha) for _ in range(T + 1)] for _ in range(N)]
    phi = [[np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    rng = np.random.default_rng(seed)
    for i in range(N):     # hard observation at the last time (draw_node_observations!(…, last_time=true))
        phi[i][T] = np.array([1.0, 0.0]) if rng.random() < 0.5 else np.array([0.0, 1.0])
    return g, w, phi, N, T


def test_sis_small_tree_exact():
    """reference test/sis_small_tree.jl:4-50 (binding cap TruncBondMax(4))."""
    g, w, phi, N, T = _sis_star()
    bp = mpbp(g, w, [2] * N, T, phi=phi)
    iterate(bp, maxiter=10, svd_trunc=TruncBondMax(4))
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(bp)
    assert _maxerr(_flat(beliefs(bp)), _flat(exact_marginals(bp, p))) < RTOL
    assert abs(np.exp(-bethe_free_energy(bp)) - Z) / Z < RTOL
    assert _maxerr(_flat(pair_beliefs(bp)[0]), _flat(exact_pair_marginals(bp, p))) < RTOL
    f = lambda x, i: x - 1
    assert _maxerr(autocorrelations(f, bp), exact_autocorrelations(f, bp, p)) < RTOL
    for m in bp.mu:   # test/normalizations.jl:46-51
        assert abs(normalization_log(m)) < 1e-12


def test_sis_small_tree_cross_paths():
    """reference test/sis_small_tree.jl:53-98: Restricted / Generic / RecursiveTrace wrappers."""
    g, w, phi, N, T = _sis_star()
    bp = mpbp(g, w, [2] * N, T, phi=phi)
    iterate(bp, maxiter=10, svd_trunc=TruncBondMax(4))
    b = _flat(beliefs(bp))
    pb = _flat(pair_beliefs(bp)[0])
    for wrap, tr in ((RestrictedRecursiveBPFactor, TruncBondMax(4)), (GenericFactor, TruncBondMax(4)),
                     (lambda x: RecursiveTraceFactor(x, 2), TruncBond(10))):
        bp2 = mpbp(g, [[wrap(x) for x in wi] for wi in w], [2] * N, T, phi=phi)
        rng = np.random.default_rng(0)
        for _ in range(5):
            X = rng.integers(0, 2, size=(N, T + 1))
            with np.errstate(divide="ignore"):
                assert np.isclose(logprob(bp, X), logprob(bp2, X), rtol=1e-12) or \
                    (np.isinf(logprob(bp, X)) and np.isinf(logprob(bp2, X)))
        iterate(bp2, maxiter=10, svd_trunc=tr)
        assert _maxerr(_flat(beliefs(bp2)), b) < RTOL
        assert _maxerr(_flat(pair_beliefs(bp2)[0]), pb) < RTOL


def test_sis_observe_everything_free_energy():
    """reference test/sis_small_tree.jl:100-111: -F_bethe == logprob(X) when every (i,t) is observed."""
    g, w, _, N, T = _sis_star(alpha=0.1)
    X = np.array([[0, 1, 1, 0], [1, 1, 0, 0], [0, 0, 1, 1], [0, 0, 0, 1]])
    phi = [[np.eye(2)[X[i, t]] * (0.5 if t == 0 else 1.0) for t in range(T + 1)] for i in range(N)]
    bp = mpbp(g, w, [2] * N, T, phi=phi)
    iterate(bp, maxiter=10, svd_trunc=TruncBond(4), tol=0)
    with np.errstate(divide="ignore"):
        lp = logprob(bp, X)
    assert np.isfinite(lp)
    assert abs(-bethe_free_energy(bp) - lp) < 1e-10 * abs(lp)


def test_glauber_small_tree_exact():
    """reference test/glauber_small_tree.jl:3-72: 5 nodes (star of 4 + isolated), T=2, TruncBondThresh(10)."""
    T = 2
    J = np.array([[0, 1, 0, 0, 0], [1, 0, 1, 1, 0], [0, 1, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 0, 0, 0]], float)
    N = 5
    rng = np.random.default_rng(111)
    h = rng.standard_normal(N)
    g = IndexedBiDiGraph(J != 0)
    w = glauber_factors(J != 0, J, h, 1.0, T)
    assert all(isinstance(wi[0], HomogeneousGlauberFactor) for wi in w)
    phi = [[np.array([0.75, 0.25]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    phi[1][2] = np.array([0.0, 1.0])
    phi[3][1] = np.array([1.0, 0.0])
    bp = mpbp(g, w, [2] * N, T, phi=phi)
    iterate(bp, maxiter=20, svd_trunc=TruncBondThresh(10))
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(bp)
    assert _maxerr(_flat(beliefs(bp)), _flat(exact_marginals(bp, p))) < RTOL
    assert abs(np.exp(-bethe_free_energy(bp)) - Z) / Z < RTOL
    assert _maxerr(_flat(pair_beliefs(bp)[0]), _flat(exact_pair_marginals(bp, p))) < RTOL
    # DampedFactor variant (test/glauber_small_tree.jl:88-131)
    wd = [[DampedFactor(x, 0.3) for x in wi] for wi in w]
    bpd = mpbp(g, wd, [2] * N, T, phi=phi)
    iterate(bpd, maxiter=20, svd_trunc=TruncBondThresh(10))
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(bpd)
    assert _maxerr(_flat(beliefs(bpd)), _flat(exact_marginals(bpd, p))) < RTOL
    assert abs(np.exp(-bethe_free_energy(bpd)) - Z) / Z < RTOL


def test_glauber_config1_path_exact():
    """BASELINE config 1 as literally written: Glauber on the 3-node path, T=3, bond 4
    (graph of reference test/glauber_small_tree.jl:323-325)."""
    T = 3
    J = np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]], float)
    rng = np.random.default_rng(0)
    h = rng.standard_normal(3)
    g = IndexedBiDiGraph(J != 0)
    w = glauber_factors(J != 0, J, h, 1.0, T)
    phi = [[np.array([0.6, 0.4]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(3)]
    bp = mpbp(g, w, [2] * 3, T, phi=phi)
    iterate(bp, maxiter=10, svd_trunc=TruncBond(16))
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(bp)
    assert _maxerr(_flat(beliefs(bp)), _flat(exact_marginals(bp, p))) < RTOL
    assert abs(np.exp(-bethe_free_energy(bp)) - Z) / Z < RTOL


def test_integer_and_pmj_glauber_exact():
    """reference test/glauber_small_tree.jl:174-318 (IntegerGlauber, J=[0 -1 2;…]) and
    test/glauber_pmJ_small_tree.jl:3-62 (±J 4-node tree, T=3, β=2, TruncThresh(0))."""
    T = 2
    J = np.array([[0, -1, 2], [-1, 0, 0], [2, 0, 0]], float)
    rng = np.random.default_rng(5)
    h = rng.standard_normal(3)
    g = IndexedBiDiGraph(J != 0)
    w = glauber_factors(J != 0, J, h, 1.0, T)
    assert isinstance(w[0][0], IntegerGlauberFactor)
    phi = [[np.array([0.75, 0.25]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(3)]
    bp = mpbp(g, w, [2] * 3, T, phi=phi)
    iterate(bp, maxiter=10, svd_trunc=TruncBondThresh(15))
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(bp)
    assert _maxerr(_flat(beliefs(bp)), _flat(exact_marginals(bp, p))) < RTOL
    assert abs(np.exp(-bethe_free_energy(bp)) - Z) / Z < RTOL

    T = 3
    J = np.array([[0, 1, 0, 0], [1, 0, -1, 1], [0, -1, 0, 0], [0, 1, 0, 0]], float)
    h = rng.standard_normal(4)
    g = IndexedBiDiGraph(J != 0)
    w = glauber_factors(J != 0, J, h, 2.0, T)
    assert isinstance(w[1][0], PMJGlauberFactor)
    phi = [[np.array([0.75, 0.25]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(4)]
    bp = mpbp(g, w, [2] * 4, T, phi=phi)
    iterate(bp, maxiter=10, svd_trunc=TruncThresh(0.0))
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(bp)
    assert _maxerr(_flat(beliefs(bp)), _flat(exact_marginals(bp, p))) < RTOL
    assert abs(np.exp(-bethe_free_energy(bp)) - Z) / Z < RTOL


def test_sirs_and_heterogeneous_sis_exact():
    """reference test/sirs_small_tree.jl:1-49 (q=3) and test/sis_heterogeneous.jl:1-47."""
    T = 2
    A = np.array([[0, 1, 1], [1, 0, 0], [1, 0, 0]])
    g = IndexedBiDiGraph(A)
    w = [[SIRSFactor(0.4, 0.4, 0.3, 0.05) for _ in range(T + 1)] for _ in range(3)]
    phi = [[np.array([0.5, 0.5, 0.0]) if t == 0 else np.ones(3) for t in range(T + 1)] for _ in range(3)]
    phi[2][2] = np.array([0.0, 0.0, 1.0])
    bp = mpbp(g, w, [3] * 3, T, phi=phi)
    iterate(bp, maxiter=10, svd_trunc=TruncBond(27))
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(bp)
    assert _maxerr(_flat(beliefs(bp)), _flat(exact_marginals(bp, p))) < RTOL
    assert abs(np.exp(-bethe_free_energy(bp)) - Z) / Z < RTOL

    T = 3
    A = np.array([[0, 1, 1, 1], [1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0]])
    g = IndexedBiDiGraph(A)
    rng = np.random.default_rng(0)
    w = []
    for i in range(4):
        lam = rng.random(len(g.neighbors(i)))
        w.append([SISHeterogeneousFactor(lam, 0.4, 0.1) for _ in range(T + 1)])
    phi = [[np.array([0.5, 0.5]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(4)]
    phi[1][T] = np.array([0.0, 1.0])
    bp = mpbp(g, w, [2] * 4, T, phi=phi)
    iterate(bp, maxiter=10, svd_trunc=TruncBond(16))
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(bp)
    assert _maxerr(_flat(beliefs(bp)), _flat(exact_marginals(bp, p))) < RTOL
    assert abs(np.exp(-bethe_free_energy(bp)) - Z) / Z < RTOL


def test_pair_observations_exact():
    """reference test/pair_observations.jl:3-58: non-trivial ψ on the edges."""
    T = 2
    A = np.array([[0, 1, 1], [1, 0, 0], [1, 0, 0]])
    g = IndexedBiDiGraph(A)
    w = [[SISFactor(0.5, 0.4, 0.1) for _ in range(T + 1)] for _ in range(3)]
    phi = [[np.array([0.5, 0.5]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(3)]
    rng = np.random.default_rng(3)
    psi = [None] * g.E
    for (i, j, ij) in g.edges():
        if i < j:
            m = [rng.random((2, 2)) + 0.1 for _ in range(T + 1)]
            psi[ij] = m
            psi[g.rev[ij]] = [x.T.copy() for x in m]
    bp = mpbp(g, w, [2] * 3, T, phi=phi, psi=psi)
    iterate(bp, maxiter=10, svd_trunc=TruncBond(16))
    p, Z = exact_prob(bp)
    assert _maxerr(_flat(beliefs(bp)), _flat(exact_marginals(bp, p))) < RTOL
    assert abs(np.exp(-bethe_free_energy(bp)) - Z) / Z < RTOL
    assert _maxerr(_flat(pair_beliefs(bp)[0]), _flat(exact_pair_marginals(bp, p))) < RTOL


def test_mpem2_and_orthogonalize_preserve_function():
    """reference test/mpems.jl:3-40: evaluate invariant under orthogonalize_*! and mpem2."""
    rng = np.random.default_rng(0)
    T, q = 3, 2
    A = rand_tt([1, 3, 4, 3, 1], q, q, rng=rng)
    xs = [tuple((int(a), int(b)) for a, b in zip(x[::2], x[1::2])) for x in itertools.product(range(q), repeat=2 * (T + 1))]
    e0 = np.array([evaluate(A, x) for x in xs])
    for fn in (orthogonalize_left, orthogonalize_right):
        B = fn(A.copy(), TruncThresh(0.0))
        np.testing.assert_allclose(np.array([evaluate(B, x) for x in xs]), e0, rtol=1e-10, atol=1e-13)
    bonds = [1, 3, 4, 3, 1]
    cores = [rng.random((bonds[t], bonds[t + 1], q, q, q)) for t in range(T + 1)]
    cores[-1][:] = cores[-1][:, :, :, :, :1]
    B3 = MPEM3(cores)
    C = mpem2(B3)
    np.testing.assert_allclose(np.array([evaluate(C, x) for x in xs]), np.array([evaluate_mpem3(B3, x) for x in xs]),
                               rtol=1e-10, atol=1e-13)


def test_infinite_graph_equals_complete_graph():
    """reference test/glauber_infinite_graph.jl:6-46 idea: BP on the infinite k-regular graph equals
    BP on any finite k-regular graph with identical nodes; here k=2 -> the 3-cycle."""
    T, k = 2, 2
    wi = [HomogeneousGlauberFactor(0.4, 0.2, 1.0) for _ in range(T + 1)]
    phi = [np.array([0.7, 0.3]) if t == 0 else np.ones(2) for t in range(T + 1)]
    bpi = mpbp_infinite_graph(k, wi, 2, phi)
    iterate(bpi, maxiter=30, svd_trunc=TruncBond(16), tol=1e-15)
    A = np.ones((3, 3)) - np.eye(3)
    g = IndexedBiDiGraph(A)
    bp = mpbp(g, [wi] * 3, [2] * 3, T, phi=[phi] * 3)
    iterate(bp, maxiter=30, svd_trunc=TruncBond(16), tol=1e-15)
    assert _maxerr(beliefs(bpi)[0], beliefs(bp)[0]) < 1e-9
    assert abs(bethe_free_energy(bpi) - bethe_free_energy(bp) / 3) < 1e-9


def test_baseline_config0_glauber_3node_path_exact():
    """BASELINE.json configs[0] / SURVEY 8(d) config 1: Glauber on the 3-node path, J = 1, beta = 1, seeded fields,
    T = 3, TruncBond(4): must equal brute-force enumeration to 1e-8."""
    T = 3
    J = np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]], float)
    h = np.random.default_rng(0).standard_normal(3)
    phi = [[np.array([0.75, 0.25]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(3)]
    import oracle.factors as OF
    import oracle.mpbp as O
    bp = O.mpbp(O.IndexedBiDiGraph(J != 0), OF.glauber_factors(J != 0, J, h, 1.0, T), [2] * 3, T, phi=phi)
    O.iterate(bp, maxiter=10, svd_trunc=TruncBond(4), tol=0.0, shuffle_nodes=False)
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(bp)
    assert np.abs(np.array(O.beliefs(bp)) - np.array(exact_marginals(bp, p))).max() < 1e-8
    assert abs(np.exp(-O.bethe_free_energy(bp)) - Z) / Z < 1e-8


def test_device_algorithm_equals_reference_compress_with_binding_truncation():
    """The R-only-QR gauge sweep + truncating SVD of M_t = N_t Lf_{t+1} (what the HIP engine computes,
    oracle/device_algorithm.py) gives the same truncated FUNCTION as the reference's two SVD sweeps (`compress!` inside
    `op`, recursive_bp_factor.jl:127) when the cap binds: the two output trains differ by a gauge only."""
    from oracle import mpbp as O
    from oracle.device_algorithm import op_kron_compress_qr
    from oracle.factors import SISFactor, HomogeneousGlauberFactor
    from oracle.tensor_trains import TensorTrain, TruncBond, TruncBondMax, TruncThresh, evaluate
    rng = np.random.default_rng(7)
    T = 5
    for (w, d1, d2) in [(SISFactor(0.2, 0.1), 1, 1), (HomogeneousGlauberFactor(0.4, 0.1, 1.0), 2, 1)]:
        wi = [w] * (T + 1)
        prof = [1, 3, 5, 5, 4, 2, 1]

        def rand_train(d):
            ny = w.nstates(d)
            return TensorTrain([rng.random((prof[t], prof[t + 1], ny, 2)) + 0.05 for t in range(T + 1)], logz=0.3 * d)
        for trunc_f in (lambda: TruncBond(4), lambda: TruncBondMax(3), lambda: TruncThresh(1e-3)):
            A, B = rand_train(d1), rand_train(d2)
            ta, tb = trunc_f(), trunc_f()
            ref, _ = O.op_kron_compress(wi, (A.copy(), d1), (B.copy(), d2), T, ta)
            dev, _ = op_kron_compress_qr(wi, (A.copy(), d1), (B.copy(), d2), T, tb)
            assert ref.bonds == dev.bonds
            if isinstance(ta, TruncBondMax):
                assert abs(ta.maxerr - tb.maxerr) < 1e-12 and ta.maxerr > 1e-6       # the cap really binds
            ny = w.nstates(d1 + d2)
            for _ in range(40):
                x = [(int(rng.integers(ny)), int(rng.integers(2))) for _ in range(T + 1)]
                ra, rb = evaluate(ref, x), evaluate(dev, x)
                assert abs(ra - rb) <= 1e-11 * max(abs(ra), 1e-300) + 1e-14


# ------------------------------------------------------------------------------------------ chains periodic in time
def test_periodic_oracle_mpem2_preserves_the_function():
    """reference test/mpems.jl:55-65: evaluate(mpem2(B)) == evaluate(B) for a random PeriodicMPEM3 (src/mpems.jl:96-155)."""
    import itertools
    from oracle import periodic as OP
    rng = np.random.default_rng(5)
    L, q, d = 3, 2, 3
    B = OP.PeriodicMPEM3([rng.random((d, d, q, q, q)) for _ in range(L)], logz=0.3)
    C = OP.mpem2(B)
    for x in itertools.product(itertools.product(range(q), range(q)), repeat=L):
        assert abs(OP.evaluate(C, [tuple(v) for v in x]) - OP.evaluate_mpem3(B, [tuple(v) for v in x])) < 1e-12


def _periodic_tree_model():
    from oracle import factors as OF, mpbp as O
    T = 2
    J = np.array([[0, 1, 0, 0, 0], [1, 0, 1, 1, 0], [0, 1, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 0, 0, 0]], float)
    N = 5
    rng = np.random.default_rng(111)
    h = rng.standard_normal(N)
    g = O.IndexedBiDiGraph(J != 0)
    psi = [[np.ones((2, 2)) for _ in range(T + 1)] for _ in range(g.E)]
    obs = [(0, 1, 0, np.array([[0.1, 0.9], [0.3, 0.4]])), (1, 3, 1, np.array([[0.4, 0.6], [0.5, 0.9]])),
           (1, 2, T, rng.random((2, 2)) + 0.05)]
    for (i, j, t, m) in obs:
        for (a, b, e) in g.edges():
            if (a, b) == (i, j):
                psi[e][t] = m
            if (a, b) == (j, i):
                psi[e][t] = m.T.copy()
    phi = [[np.array([0.75, 0.25]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    phi[2][1] = np.array([0.0, 1.0])
    phi[4][2] = np.array([1.0, 0.0])
    return g, OF.glauber_factors(J != 0, J, h, 1.0, T), phi, psi, N, T


def test_periodic_oracle_is_exact_on_the_reference_tree():
    """reference test/periodic.jl:1-68 (5-node tree, pair + node observations, T = 2): the restated periodic path
    (oracle/periodic.py: PeriodicMPEM3, periodic mpem2, periodic _f_bp_partial; TensorTrains' periodic sweeps as recalled)
    with a non-binding cap equals brute-force enumeration (src/exact.jl:24-26): beliefs, Z, pair beliefs."""
    from oracle import periodic as OP
    from oracle import tensor_trains as OT
    from oracle.exact import exact_marginals, exact_pair_marginals, exact_prob
    g, w, phi, psi, N, T = _periodic_tree_model()
    bp = OP.periodic_mpbp(g, w, [2] * N, T, phi=phi, psi=psi)
    OP.iterate(bp, 6, OT.TruncBondThresh(64, 1e-14))
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(bp, periodic=True)
    b = OP.beliefs(bp)
    ex = exact_marginals(bp, p)
    assert max(np.abs(np.array(b[i][t]) - ex[i][t]).max() for i in range(N) for t in range(T + 1)) < 1e-9
    assert abs(np.exp(-float(np.sum(bp.f))) - Z) / Z < 1e-9
    pb, _ = OP.pair_beliefs(bp)
    pex = exact_pair_marginals(bp, p)
    assert max(np.abs(np.array(pb[e][t]) - pex[e][t]).max() for e in range(g.E) for t in range(T + 1)) < 1e-9


def test_generic_factor_tables_of_the_host_mirror_match_the_functor():
    """`BPFactor.generic_table` (the dense table `mpbp_set_generic_factor` takes: w[x', x, x_1..x_deg], first index fastest)
    against direct evaluation of the oracle's functor, for GenericGlauberFactor and a GenericFactor-wrapped SIS factor; and
    `glauber_factors` picks the generic type exactly when the reference does (src/Models/glauber/glauber_bp.jl:121-142)."""
    import itertools
    import mpbp_amd as M
    from oracle import factors as OF
    cases = [(M.GenericGlauberFactor([0.3, -0.7, 1.1], 0.2, 0.9), OF.GenericGlauberFactor([0.3, -0.7, 1.1], 0.2, 0.9), 3),
             (M.GenericFactor(M.SISFactor(0.3, 0.2, 0.05)), OF.GenericFactor(OF.SISFactor(0.3, 0.2, 0.05)), 2)]
    for w, ow, deg in cases:
        tab = w.generic_table(deg, 2).reshape((2, 2) + (2,) * deg, order="F")
        for xs in itertools.product(range(2), repeat=deg):
            for x in range(2):
                for xn in range(2):
                    assert abs(tab[(xn, x) + xs] - ow(xn + 1, [v + 1 for v in xs], x + 1)) < 1e-15
        assert np.allclose(tab.sum(axis=0), 1.0)              # a transition probability
    J = np.array([[0, 0.3, 0], [0.3, 0, -0.7], [0, -0.7, 0]], float)
    assert all(isinstance(wi[0], M.GenericGlauberFactor) for wi in M.glauber_factors(J != 0, J, np.zeros(3), 1.0, 2))
    J2 = np.where(J != 0, 0.5, 0.0)
    assert all(isinstance(wi[0], M.HomogeneousGlauberFactor) for wi in M.glauber_factors(J2 != 0, J2, np.zeros(3), 1.0, 2))


def test_oracle_heterogeneous_nstates_tree_is_exact():
    """`nstates(bp, i)` differing from node to node (reference src/mpbp.jl:22-26; messages of edge i->j are MPEM2s over q_i x q_j,
    src/recursive_bp_factor.jl:155): SIS nodes (q = 2) and SIRS nodes (q = 3) on one tree with random node observations - BP
    without truncation against brute-force enumeration.  Pins the oracle for the device's padded-state implementation
    (mpbp_set_node_states)."""
    from oracle import factors as OF, mpbp as O
    from oracle.exact import exact_marginals, exact_prob
    from oracle.tensor_trains import TruncThresh
    T = 2                                # 36^3 trajectories to enumerate
    A = np.array([[0, 1, 1, 0], [1, 0, 0, 1], [1, 0, 0, 0], [0, 1, 0, 0]])
    qs = [2, 3, 2, 3]
    w = [[OF.SISFactor(0.3, 0.2)] * (T + 1) if q == 2 else [OF.SIRSFactor(0.3, 0.2, 0.1)] * (T + 1) for q in qs]
    rng = np.random.default_rng(0)
    phi = [[rng.random(q) + 0.1 for _ in range(T + 1)] for q in qs]
    bp = O.mpbp(O.IndexedBiDiGraph(A), w, qs, T, phi=phi)
    O.iterate(bp, maxiter=6, svd_trunc=TruncThresh(0.0), tol=1e-14, shuffle_nodes=False)
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(bp)
    m = exact_marginals(bp, p)
    b = O.beliefs(bp)
    assert [np.array(x).shape for x in b] == [(T + 1, q) for q in qs]
    assert max(np.abs(np.array(b[i]) - np.array(m[i])).max() for i in range(4)) < 1e-12
    assert abs(np.exp(-O.bethe_free_energy(bp)) - Z) / Z < 1e-12
