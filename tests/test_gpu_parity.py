"""Parity of the HIP path (through the C ABI) with the CPU oracle on the same inputs, and with the
reference's own known answers.  Tolerance: 1e-6 relative (BASELINE.json north_star); most cases are
checked far tighter because the device algorithm is backward stable."""
import os
import sys

import networkx as nx
import numpy as np
import pytest

import mpbp_amd as M
from oracle import factors as OF
from oracle import mpbp as O
from oracle import tensor_trains as OT
from oracle.exact import exact_marginals, exact_pair_marginals, exact_prob

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def _flat(bb):
    return np.array([p for b in bb for p in b])


def _rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_known_answer_sis_infinite_graph_gpu():
    """reference test/sis_infinite_graph.jl:1-30 on the device path."""
    T, k, gam, lam, rho = 6, 3, 0.1, 0.1, 0.2
    phi = [np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)]
    bp = M.mpbp_infinite_graph(k, [M.SISFactor(lam, rho)] * (T + 1), 2, phi, max_bond=10)
    it, cb = M.iterate(bp, maxiter=200, svd_trunc=M.TruncBond(10), tol=1e-14)
    ref = [[0.9000000001671186, 0.0999999998328814],
           [0.8932690998131098, 0.10673090018689023],
           [0.8899420329322244, 0.11005796706777556],
           [0.8884643888492034, 0.11153561115079656],
           [0.8880305235706524, 0.1119694764293476],
           [0.8882121515614524, 0.11178784843854758],
           [0.8887717202217936, 0.1112282797782064]]
    np.testing.assert_allclose(np.array(M.beliefs(bp)[0]), np.array(ref), rtol=1.5e-8, atol=0)


def _sis_star_inputs(T=3):
    A = np.array([[0, 1, 1, 1], [1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0]])
    lam, rho, gam, alpha = 0.5, 0.4, 0.5, 0.1
    phi = [[np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(4)]
    rng = np.random.default_rng(111)
    for i in range(4):
        phi[i][T] = np.array([1.0, 0.0]) if rng.random() < 0.5 else np.array([0.0, 1.0])
    return A, lam, rho, alpha, phi, T


@pytest.mark.parametrize("schedule", ["sequential", "colored", "jacobi"])
def test_sis_small_tree_exact_gpu(schedule):
    """reference test/sis_small_tree.jl:4-50 against brute-force enumeration (oracle/exact.py)."""
    A, lam, rho, alpha, phi, T = _sis_star_inputs()
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho, alpha)] * (T + 1)] * 4, 2, T, phi=phi, max_bond=16)
    tr = M.TruncBondMax(4)
    M.iterate(bp, maxiter=10, svd_trunc=tr, schedule=schedule)
    og = O.IndexedBiDiGraph(A)
    obp = O.mpbp(og, [[OF.SISFactor(lam, rho, alpha)] * (T + 1)] * 4, [2] * 4, T, phi=phi)
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(obp)
    assert _rel(_flat(M.beliefs(bp)), _flat(exact_marginals(obp, p))) < 1e-9
    assert abs(np.exp(-M.bethe_free_energy(bp)) - Z) / Z < 1e-9
    pb, _ = M.pair_beliefs(bp)
    assert _rel(_flat(pb), _flat(exact_pair_marginals(obp, p))) < 1e-9
    assert tr.maxerr < 1e-7


def _loopy(N, T, lam, rho, gam, seed=0):
    G = nx.random_regular_graph(3, N, seed=seed)
    A = nx.to_numpy_array(G)
    phi = [[np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    return A, phi


@pytest.mark.parametrize("N,T,Mb,sweeps", [(8, 5, 4, 3), (16, 10, 8, 3)])
def test_sis_loopy_jacobi_sweeps_match_oracle(N, T, Mb, sweeps):
    """Binding truncation on a loopy 3-regular graph: every Jacobi sweep must reproduce the oracle's
    beliefs, pair beliefs and free energy (SURVEY 8d parity read-outs)."""
    lam, rho, gam = 0.1, 0.05, 0.1
    A, phi = _loopy(N, T, lam, rho, gam)
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
    for s in range(sweeps):
        M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(Mb), tol=0.0)
        O.iterate(obp, maxiter=1, svd_trunc=OT.TruncBond(Mb), tol=0.0, shuffle_nodes=False, jacobi=True)
        assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < RTOL, f"sweep {s}"
        # per-node free-energy terms (the sum cancels to ~0 without observations)
        import ctypes as C
        f = np.zeros(N)
        bp._L.mpbp_free_energy(bp._h, f.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.abs(f - obp.f).max() < RTOL * max(1.0, np.abs(obp.f).max()), f"sweep {s}"
    pb, lz = M.pair_beliefs(bp)
    opb, olz = O.pair_beliefs(obp)
    assert _rel(_flat(pb), _flat(opb)) < RTOL
    assert np.abs(lz - olz).max() < RTOL * max(1.0, np.abs(olz).max())
    assert (bp.bonds() <= Mb).all()
    ob = np.array([m.bonds for m in obp.mu])
    assert np.array_equal(bp.bonds(), ob)


@pytest.mark.parametrize("mode", ["grid", "grid_colstep"])
def test_batched_gauge_sweep_matches_oracle(mode, monkeypatch):
    """The same loopy parity case with sweep 1 of every cavity product forced through the batched, grid-level gauge sweep
    (MPBP_GAUGE=grid; register panels, or column-step panels with MPBP_DEBUG_FORCE_TALL=1) instead of the workgroup
    engine: beliefs, free energy and bonds after every sweep against the oracle."""
    monkeypatch.setenv("MPBP_GAUGE", "grid")
    monkeypatch.setenv("MPBP_DEBUG_NO_SMALL", "1")
    if mode == "grid_colstep":
        monkeypatch.setenv("MPBP_DEBUG_FORCE_TALL", "1")
    N, T, Mb = 8, 6, 6
    lam, rho, gam = 0.15, 0.1, 0.2
    A, phi = _loopy(N, T, lam, rho, gam)
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
    for s in range(3):
        M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(Mb), tol=0.0)
        O.iterate(obp, maxiter=1, svd_trunc=OT.TruncBond(Mb), tol=0.0, shuffle_nodes=False, jacobi=True)
        assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < RTOL, f"sweep {s}"
    assert abs(M.bethe_free_energy(bp) - O.bethe_free_energy(obp)) < RTOL * max(1.0, np.abs(obp.f).sum())
    assert np.array_equal(bp.bonds(), np.array([m.bonds for m in obp.mu]))


def _fnodes(bp):
    import ctypes as C
    f = np.zeros(bp.g.nv())
    bp._L.mpbp_free_energy(bp._h, f.ctypes.data_as(C.POINTER(C.c_double)))
    return f


def test_split_sweep_equals_one_pass_and_oracle(monkeypatch):
    """mpbp_sweep splits a node list whose work trains do not fit the device and lets the halves read a snapshot of the
    in-edges (include/mpbp_hip.h: "results are those of one pass").  MPBP_DEBUG_SPLIT_NODES=2 forces the split (recursively,
    down to pairs of nodes) on a loopy graph with damping (which reads the LIVE slab while prep reads the snapshot),
    TruncBondMax (maxerr merged over the parts) and the n_compress count: identical to the unsplit call, equal to the oracle."""
    N, T, Mb, damp = 10, 6, 6, 0.3
    lam, rho, gam = 0.15, 0.1, 0.2
    A, phi = _loopy(N, T, lam, rho, gam)
    res = {}
    for mode in ("one", "split"):
        if mode == "split":
            monkeypatch.setenv("MPBP_DEBUG_SPLIT_NODES", "2")
        bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
        tr = M.TruncBondMax(Mb)
        out = []
        for s in range(3):
            M.iterate(bp, maxiter=1, svd_trunc=tr, tol=0.0, damp=damp)
            st = bp.last_stats
            out.append((_flat(M.beliefs(bp)), _fnodes(bp), int(st.n_compress), float(st.maxerr)))
        res[mode] = (out, bp.bonds().copy(), tr.maxerr)
    monkeypatch.delenv("MPBP_DEBUG_SPLIT_NODES")
    for (b1, f1, n1, e1), (b2, f2, n2, e2) in zip(res["one"][0], res["split"][0]):
        assert _rel(b1, b2) < 1e-12 and np.allclose(f1, f2, rtol=1e-11, atol=1e-12)
        assert n1 == n2 and abs(e1 - e2) <= 1e-12 * max(e1, 1e-300)
    assert np.array_equal(res["one"][1], res["split"][1])
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
    otr = OT.TruncBondMax(Mb)
    for s in range(3):
        O.iterate(obp, maxiter=1, svd_trunc=otr, tol=0.0, shuffle_nodes=False, jacobi=True, damp=damp)
        assert _rel(res["split"][0][s][0], _flat(O.beliefs(obp))) < RTOL, f"sweep {s}"
    assert abs(res["split"][2] - otr.maxerr) < 1e-6 * max(otr.maxerr, 1e-12)


def test_cooperative_panel_timeout_is_recovered(monkeypatch):
    """The cooperative column-step kernel of the batched gauge sweep waits for its sibling workgroups inside one launch; if
    they are not co-resident an arrival counter times out, the kernel leaves Y untouched and raises a flag.  launch_engine
    then repeats the batch with one launch per column step - nothing has been committed to the message slab yet - and the
    context stays in that mode.  MPBP_DEBUG_COOP_FAIL_ONCE injects the time-out on the first attempt: same results and
    counters as a clean run, no error."""
    monkeypatch.setenv("MPBP_GAUGE", "grid")
    monkeypatch.setenv("MPBP_DEBUG_NO_SMALL", "1")
    monkeypatch.setenv("MPBP_DEBUG_FORCE_TALL", "1")
    N, T, Mb = 8, 6, 6
    lam, rho, gam = 0.15, 0.1, 0.2
    A, phi = _loopy(N, T, lam, rho, gam)
    res = []
    for inject in (False, True):
        if inject:
            monkeypatch.setenv("MPBP_DEBUG_COOP_FAIL_ONCE", "1")
        bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
        out = []
        for s in range(2):
            M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(Mb), tol=0.0)
            out.append((_flat(M.beliefs(bp)), int(bp.last_stats.n_compress)))
        res.append(out)
    monkeypatch.delenv("MPBP_DEBUG_COOP_FAIL_ONCE")
    for (b1, n1), (b2, n2) in zip(*res):
        assert _rel(b1, b2) < 1e-12 and n1 == n2


def test_glauber_small_tree_gpu():
    """reference test/glauber_small_tree.jl:3-72 structure (star of 4 + isolated node, T=2,
    TruncBondThresh(10)); HomogeneousGlauberFactor with growing nstates = l+1."""
    T = 2
    J = np.array([[0, 1, 0, 0, 0], [1, 0, 1, 1, 0], [0, 1, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 0, 0, 0]], float)
    rng = np.random.default_rng(111)
    h = rng.standard_normal(5)
    phi = [[np.array([0.75, 0.25]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(5)]
    phi[1][2] = np.array([0.0, 1.0])
    phi[3][1] = np.array([1.0, 0.0])
    gl = M.Glauber(M.Ising(J, h, 1.0), T, phi=phi)
    bp = gl.mpbp(max_bond=16)
    M.iterate(bp, maxiter=20, svd_trunc=M.TruncBondThresh(10), schedule="colored")
    og = O.IndexedBiDiGraph(J != 0)
    obp = O.mpbp(og, OF.glauber_factors(J != 0, J, h, 1.0, T), [2] * 5, T, phi=phi)
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(obp)
    assert _rel(_flat(M.beliefs(bp)), _flat(exact_marginals(obp, p))) < 1e-9
    assert abs(np.exp(-M.bethe_free_energy(bp)) - Z) / Z < 1e-9


def _exact_logZ_periodic(obp):
    with np.errstate(divide="ignore"):
        return exact_prob(obp, periodic=True)


def test_periodic_glauber_small_tree_exact_gpu():
    """reference test/periodic.jl:1-68: chains periodic in time (`periodic_mpbp`, src/mpbp.jl:399-409; periodic
    `_f_bp_partial`, src/recursive_bp_factor.jl:89-101) on the 5-node tree with pair observations, node observations and a
    biased phi at every first time; beliefs, Z, pair beliefs and autocorrelations against brute-force enumeration
    (src/exact.jl:24-26) with non-binding truncation."""
    from oracle.exact import exact_autocorrelations
    T = 2
    J = np.array([[0, 1, 0, 0, 0], [1, 0, 1, 1, 0], [0, 1, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 0, 0, 0]], float)
    N = 5
    rng = np.random.default_rng(111)
    h = rng.standard_normal(N)
    g = M.IndexedBiDiGraph(J != 0)
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
    phi[2][1] = np.array([0.0, 1.0])                       # hard node observations, as draw_node_observations! leaves them
    phi[4][2] = np.array([1.0, 0.0])
    w = M.glauber_factors(J != 0, J, h, 1.0, T)
    bp = M.periodic_mpbp(g, w, 2, T, phi=phi, psi=psi, max_bond=16)
    assert M.is_periodic(bp)
    M.iterate(bp, maxiter=20, svd_trunc=M.TruncBondThresh(16), schedule="colored")
    ow = OF.glauber_factors(J != 0, J, h, 1.0, T)
    obp = O.mpbp(O.IndexedBiDiGraph(J != 0), ow, [2] * N, T, phi=phi, psi=psi)
    p, Z = _exact_logZ_periodic(obp)
    assert _rel(_flat(M.beliefs(bp)), _flat(exact_marginals(obp, p))) < 1e-9
    assert abs(np.exp(-M.bethe_free_energy(bp)) - Z) / Z < 1e-9
    pb, _ = M.pair_beliefs(bp)
    assert _rel(_flat(pb), _flat(exact_pair_marginals(obp, p))) < 1e-9
    f = lambda x, i: 2 * x - 3
    r = M.autocorrelations(f, bp)
    rex = exact_autocorrelations(f, obp, p)
    assert max(np.abs(a - b).max() for a, b in zip(r, rex)) < 1e-9
    # the open-chain model on the same inputs is a different distribution: the switch is not a no-op
    bp0 = M.mpbp(g, w, 2, T, phi=phi, psi=psi, max_bond=16)
    M.iterate(bp0, maxiter=20, svd_trunc=M.TruncBondThresh(16), schedule="colored")
    assert _rel(_flat(M.beliefs(bp0)), _flat(M.beliefs(bp))) > 1e-4


def test_periodic_sis_chain_exact_gpu():
    """Periodic chains with a longer ring in time (T = 4) and the SIS factor on a 3-node path: enumeration."""
    T = 4
    A = np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]])
    phi = [[np.array([0.6, 0.4]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(3)]
    phi[0][3] = np.array([0.2, 0.8])
    bp = M.periodic_mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(0.4, 0.3, 0.05)] * (T + 1)] * 3, 2, T, phi=phi, max_bond=48)
    M.iterate(bp, maxiter=10, svd_trunc=M.TruncBond(48), schedule="colored")
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(0.4, 0.3, 0.05)] * (T + 1)] * 3, [2] * 3, T, phi=phi)
    p, Z = _exact_logZ_periodic(obp)
    assert _rel(_flat(M.beliefs(bp)), _flat(exact_marginals(obp, p))) < 1e-8
    assert abs(np.exp(-M.bethe_free_energy(bp)) - Z) / Z < 1e-8


def test_periodic_infinite_graph_matches_complete_graph_gpu():
    """reference test/periodic.jl:70-110: `periodic_mpbp_infinite_graph` (k = 3 aliases of one message, damping 0.2,
    TruncBond(10)) against `periodic_mpbp` on the complete graph of k + 1 nodes - the same local fixed point."""
    T, k, m0 = 2, 3, 0.5
    wi = [M.HomogeneousGlauberFactor(1.0, 0.0, 1.0)] * (T + 1)
    phi_i = [np.array([(1 + m0) / 2, (1 - m0) / 2]) if t == 0 else np.ones(2) for t in range(T + 1)]
    phi_i[1] = np.array([0.4, 0.6])
    phi_i[T] = np.array([0.95, 0.05])
    bp = M.periodic_mpbp_infinite_graph(k, wi, 2, phi_i, max_bond=10)
    M.iterate(bp, maxiter=150, svd_trunc=M.TruncBond(10), tol=1e-12, damp=0.2)
    N = k + 1
    A = np.ones((N, N)) - np.eye(N)
    bpc = M.periodic_mpbp(M.IndexedBiDiGraph(A), [wi] * N, 2, T, phi=[phi_i] * N, max_bond=10)
    M.iterate(bpc, maxiter=150, svd_trunc=M.TruncBond(10), tol=1e-12, damp=0.2)
    b_inf, b_c = np.array(M.beliefs(bp)[0]), np.array(M.beliefs(bpc)[0])
    assert np.abs(b_inf - b_c).max() < 1e-7
    pb_inf, pb_c = np.array(M.pair_beliefs(bp)[0][0]), np.array(M.pair_beliefs(bpc)[0][0])
    assert np.abs(pb_inf - pb_c).max() < 1e-7


def test_periodic_loopy_damped_sweeps_match_periodic_oracle_gpu():
    """Chains periodic in time on a LOOPY graph with damping, sweep by sweep, against the restated periodic path of the
    reference (oracle/periodic.py: PeriodicMPEM3, periodic mpem2 / _f_bp_partial of src/mpems.jl:96-155 and
    src/recursive_bp_factor.jl:89-101, damping through the ring-closed _compose) while NO truncation binds on either side:
    the case of reference test/periodic.jl:70-110 (homogeneous Glauber, complete graph of 4 nodes, T = 2, damp 0.2).
    The device keeps a message as an open train that carries x_i^1 (exact bond <= 4 at T = 2), the oracle as a ring with
    a boundary bond (21 after the second damped sweep, cap 32): the same functions, so beliefs and pair beliefs agree to
    rounding.  Under a BINDING cap the two representations truncate different matrices (TruncBond(10) as in
    test/periodic.jl:87: 7e-3 on the beliefs after the second sweep, tools/periodic_compare.py,
    profiles/r03_periodic_compare.txt) - that regime is unpinned, DESIGN.md section 7."""
    from oracle import periodic as OP
    T, k, m0, damp = 2, 3, 0.5, 0.2
    N = k + 1
    A = np.ones((N, N)) - np.eye(N)
    phi_i = [np.array([(1 + m0) / 2, (1 - m0) / 2]) if t == 0 else np.ones(2) for t in range(T + 1)]
    phi_i[1] = np.array([0.4, 0.6])
    phi_i[T] = np.array([0.95, 0.05])
    bp = M.periodic_mpbp(M.IndexedBiDiGraph(A), [[M.HomogeneousGlauberFactor(1.0, 0.0, 1.0)] * (T + 1)] * N, 2, T,
                         phi=[phi_i] * N, max_bond=16)
    obp = OP.periodic_mpbp(O.IndexedBiDiGraph(A), [[OF.HomogeneousGlauberFactor(1.0, 0.0, 1.0)] * (T + 1)] * N, [2] * N, T,
                           phi=[phi_i] * N)
    for s in range(2):
        M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(16), tol=0.0, damp=damp)
        OP.iterate(obp, 1, OT.TruncBond(32), damp=damp, jacobi=True)
        assert max(max(m.bonds) for m in obp.mu) < 32                       # nothing was truncated in the ring form
        assert _rel(_flat(M.beliefs(bp)), _flat(OP.beliefs(obp))) < 1e-10, f"sweep {s}"
    pb, _ = M.pair_beliefs(bp)
    opb, _ = OP.pair_beliefs(obp)
    assert _rel(_flat(pb), _flat(opb)) < 1e-10
    assert np.allclose(_fnodes(bp), obp.f, rtol=1e-9, atol=1e-11)


def test_initial_bond_size_d_gpu():
    """`mpbp(...; d)` with d > 1 (src/mpbp.jl:60-70: `flat_mpem2(q, q, T; d)`): the same uniform function in a redundant
    bond, so the first sweeps must give what d = 1 gives."""
    N, T, Mb = 8, 5, 6
    lam, rho, gam = 0.2, 0.1, 0.15
    A, phi = _loopy(N, T, lam, rho, gam)
    g = M.IndexedBiDiGraph(A)
    bp1 = M.mpbp(g, [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
    bp3 = M.mpbp(g, [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, d=3, phi=phi, max_bond=Mb)
    assert bp3.bonds().max() == 3
    for _ in range(2):
        M.iterate(bp1, maxiter=1, svd_trunc=M.TruncBond(Mb), tol=0.0)
        M.iterate(bp3, maxiter=1, svd_trunc=M.TruncBond(Mb), tol=0.0)
        assert _rel(_flat(M.beliefs(bp3)), _flat(M.beliefs(bp1))) < 1e-9
    assert abs(M.bethe_free_energy(bp3) - M.bethe_free_energy(bp1)) < 1e-8


def test_sirs_q3_tree_gpu():
    """reference test/sirs_small_tree.jl (q = 3)."""
    T = 2
    A = np.array([[0, 1, 1], [1, 0, 0], [1, 0, 0]])
    phi = [[np.array([0.5, 0.5, 0.0]) if t == 0 else np.ones(3) for t in range(T + 1)] for _ in range(3)]
    phi[2][2] = np.array([0.0, 0.0, 1.0])
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SIRSFactor(0.4, 0.4, 0.3, 0.05)] * (T + 1)] * 3, 3, T, phi=phi, max_bond=27)
    M.iterate(bp, maxiter=10, svd_trunc=M.TruncBond(27), schedule="colored")
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SIRSFactor(0.4, 0.4, 0.3, 0.05)] * (T + 1)] * 3, [3] * 3, T, phi=phi)
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(obp)
    assert _rel(_flat(M.beliefs(bp)), _flat(exact_marginals(obp, p))) < 1e-9
    assert abs(np.exp(-M.bethe_free_energy(bp)) - Z) / Z < 1e-9


def test_integer_glauber_heterogeneous_sis_and_damped_factor_gpu():
    """The factor families of SURVEY 8(a) a18 that had only oracle-side tests: `IntegerGlauberFactor` (reference
    test/glauber_small_tree.jl:174-318, J = [0 -1 2; ...]), `SIS_heterogeneousFactor` (test/sis_heterogeneous.jl:1-47: one
    rate per incoming neighbour), `DampedFactor` (test/glauber_small_tree.jl:88-131, src/recursive_bp_factor.jl:183-206) -
    the device path against brute-force enumeration and, sweep by sweep, against the oracle."""
    # IntegerGlauber: couplings -1 / 2 on a 3-node star
    T = 2
    J = np.array([[0, -1, 2], [-1, 0, 0], [2, 0, 0]], float)
    rng = np.random.default_rng(5)
    h = rng.standard_normal(3)
    phi = [[np.array([0.75, 0.25]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(3)]
    w = M.glauber_factors(J != 0, J, h, 1.0, T)
    assert isinstance(w[0][0], M.IntegerGlauberFactor)
    bp = M.mpbp(M.IndexedBiDiGraph(J != 0), w, 2, T, phi=phi, max_bond=16)
    M.iterate(bp, maxiter=10, svd_trunc=M.TruncBondThresh(15), schedule="colored")
    obp = O.mpbp(O.IndexedBiDiGraph(J != 0), OF.glauber_factors(J != 0, J, h, 1.0, T), [2] * 3, T, phi=phi)
    with np.errstate(divide="ignore"):
        pex, Z = exact_prob(obp)
    assert _rel(_flat(M.beliefs(bp)), _flat(exact_marginals(obp, pex))) < 1e-9
    assert abs(np.exp(-M.bethe_free_energy(bp)) - Z) / Z < 1e-9
    assert _rel(_flat(M.pair_beliefs(bp)[0]), _flat(exact_pair_marginals(obp, pex))) < 1e-9

    # heterogeneous SIS: a star of 4, rates drawn per (node, neighbour), an observation at the last time
    T = 3
    A = np.array([[0, 1, 1, 1], [1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0]])
    g, og = M.IndexedBiDiGraph(A), O.IndexedBiDiGraph(A)
    rng = np.random.default_rng(0)
    lams = [rng.random(len(og.neighbors(i))) for i in range(4)]
    phi = [[np.array([0.5, 0.5]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(4)]
    phi[1][T] = np.array([0.0, 1.0])
    bp = M.mpbp(g, [[M.SIS_heterogeneousFactor(l, 0.4, 0.1)] * (T + 1) for l in lams], 2, T, phi=phi, max_bond=16)
    M.iterate(bp, maxiter=10, svd_trunc=M.TruncBond(16), schedule="colored")
    obp = O.mpbp(og, [[OF.SISHeterogeneousFactor(l, 0.4, 0.1) for _ in range(T + 1)] for l in lams], [2] * 4, T, phi=phi)
    with np.errstate(divide="ignore"):
        pex, Z = exact_prob(obp)
    assert _rel(_flat(M.beliefs(bp)), _flat(exact_marginals(obp, pex))) < 1e-9
    assert abs(np.exp(-M.bethe_free_energy(bp)) - Z) / Z < 1e-9
    # the same model with a BINDING cap, Jacobi sweeps against the oracle's (rates differ per neighbour position: the cavity
    # order matters)
    bp = M.mpbp(g, [[M.SIS_heterogeneousFactor(l, 0.4, 0.1)] * (T + 1) for l in lams], 2, T, phi=phi, max_bond=3)
    obp = O.mpbp(og, [[OF.SISHeterogeneousFactor(l, 0.4, 0.1) for _ in range(T + 1)] for l in lams], [2] * 4, T, phi=phi)
    for _ in range(3):
        M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(3), tol=0.0)
        O.iterate(obp, maxiter=1, svd_trunc=OT.TruncBond(3), tol=0.0, shuffle_nodes=False, jacobi=True)
        assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < RTOL
    assert np.array_equal(bp.bonds(), np.array([m.bonds for m in obp.mu])) and bp.bonds().max() == 3

    # DampedFactor(p = 0.3) around the homogeneous Glauber factors of the 5-node tree
    T = 2
    J = np.array([[0, 1, 0, 0, 0], [1, 0, 1, 1, 0], [0, 1, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 0, 0, 0]], float)
    rng = np.random.default_rng(111)
    h = rng.standard_normal(5)
    phi = [[np.array([0.75, 0.25]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(5)]
    phi[1][2] = np.array([0.0, 1.0])
    phi[3][1] = np.array([1.0, 0.0])
    w = [[M.DampedFactor(x, 0.3) for x in wi] for wi in M.glauber_factors(J != 0, J, h, 1.0, T)]
    bp = M.mpbp(M.IndexedBiDiGraph(J != 0), w, 2, T, phi=phi, max_bond=16)
    M.iterate(bp, maxiter=20, svd_trunc=M.TruncBondThresh(10), schedule="colored")
    ow = [[OF.DampedFactor(x, 0.3) for x in wi] for wi in OF.glauber_factors(J != 0, J, h, 1.0, T)]
    obp = O.mpbp(O.IndexedBiDiGraph(J != 0), ow, [2] * 5, T, phi=phi)
    with np.errstate(divide="ignore"):
        pex, Z = exact_prob(obp)
    assert _rel(_flat(M.beliefs(bp)), _flat(exact_marginals(obp, pex))) < 1e-9
    assert abs(np.exp(-M.bethe_free_energy(bp)) - Z) / Z < 1e-9


def test_observe_everything_free_energy_is_logprob_gpu():
    """reference test/sis_small_tree.jl:100-111 and test/glauber_small_tree.jl:75-86: with every (i, t) observed the
    Bethe free energy is minus the log-probability of the observed trajectory (TruncBond(4) is not binding: every message
    is a product state)."""
    A, lam, rho, alpha, _, T = _sis_star_inputs()
    X = np.array([[0, 1, 1, 0], [1, 1, 0, 0], [0, 0, 1, 1], [0, 0, 0, 1]])
    phi = [[np.eye(2)[X[i, t]] * (0.5 if t == 0 else 1.0) for t in range(T + 1)] for i in range(4)]
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho, alpha)] * (T + 1)] * 4, 2, T, phi=phi, max_bond=4)
    M.iterate(bp, maxiter=10, svd_trunc=M.TruncBond(4), tol=0.0, schedule="colored")
    with np.errstate(divide="ignore"):
        lp = M.logprob(bp, X + 1)
    assert np.isfinite(lp) and abs(-M.bethe_free_energy(bp) - lp) < 1e-10 * abs(lp)
    # Glauber on the 5-node tree
    T = 2
    J = np.array([[0, 1, 0, 0, 0], [1, 0, 1, 1, 0], [0, 1, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 0, 0, 0]], float)
    h = np.random.default_rng(111).standard_normal(5)
    X = np.array([[0, 1, 1], [1, 1, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0]])
    phi = [[np.eye(2)[X[i, t]] * (0.75 if (t == 0 and X[i, t] == 0) else (0.25 if t == 0 else 1.0)) for t in range(T + 1)]
           for i in range(5)]
    gl = M.Glauber(M.Ising(J, h, 1.0), T, phi=phi)
    bp = gl.mpbp(max_bond=4)
    M.iterate(bp, maxiter=10, svd_trunc=M.TruncBond(4), tol=0.0, schedule="colored")
    with np.errstate(divide="ignore"):
        lp = M.logprob(bp, X + 1)
    assert np.isfinite(lp) and abs(-M.bethe_free_energy(bp) - lp) < 1e-10 * abs(lp)


def test_sis_heterogeneous_equals_homogeneous_with_binding_cap_gpu():
    """reference test/sis_heterogeneous_compare_homogeneous.jl: `SIS_heterogeneous` with one rate everywhere against `SIS`
    on a loopy 5-node graph, TruncBond(3) (binding), iterated to a fixed point."""
    T, N = 3, 5
    A = np.array([[0, 1, 1, 0, 0], [1, 0, 1, 0, 0], [1, 1, 0, 1, 0], [0, 0, 1, 0, 1], [0, 0, 0, 1, 0]])
    lam, rho, gam = 0.15, 0.12, 0.13
    g = M.IndexedBiDiGraph(A)
    phi = [[np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    deg = A.sum(axis=0)
    bu = M.mpbp(g, [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=3)
    bh = M.mpbp(g, [[M.SIS_heterogeneousFactor([lam] * int(deg[i]), rho)] * (T + 1) for i in range(N)], 2, T, phi=phi, max_bond=3)
    for b in (bu, bh):
        M.iterate(b, maxiter=200, svd_trunc=M.TruncBond(3), tol=1e-12, schedule="sequential", shuffle_nodes=False)
    pu, ph = _flat(M.beliefs(bu)), _flat(M.beliefs(bh))
    assert bu.bonds().max() == 3 and _rel(ph, pu) < 1e-10
    assert abs(M.bethe_free_energy(bu) - M.bethe_free_energy(bh)) < 1e-10


def test_glauber_infinite_graph_equals_complete_graph_gpu():
    """reference test/glauber_infinite_graph.jl: marginals and free energy of the infinite k-regular graph (k aliases of one
    message, observations at t = 1 and t = T, damping 0.1, TruncThresh(0)) against BP on the complete graph of k + 1 nodes,
    and the bipartite variant (k = (3, 2)) against the complete bipartite graph.  TruncThresh(0) keeps every non-zero
    singular value: the cap must cover the structural bound (n_y q)^2 = 64 of the cavity trains (n_y = 4 at degree 3)."""
    T, k, m0 = 3, 3, 0.5
    wi = [M.HomogeneousGlauberFactor(1.0, 0.0, 1.0)] * (T + 1)
    phi_i = [np.array([(1 + m0) / 2, (1 - m0) / 2]) if t == 0 else np.ones(2) for t in range(T + 1)]
    phi_i[1] = np.array([0.4, 0.6])
    phi_i[T] = np.array([0.95, 0.05])
    bp = M.mpbp_infinite_graph(k, wi, 2, phi_i, max_bond=64)
    M.iterate(bp, maxiter=150, svd_trunc=M.TruncThresh(0.0), tol=1e-15, damp=0.1)
    Nc = k + 1
    Ac = np.ones((Nc, Nc)) - np.eye(Nc)
    bc = M.mpbp(M.IndexedBiDiGraph(Ac), [wi] * Nc, 2, T, phi=[phi_i] * Nc, max_bond=64)
    M.iterate(bc, maxiter=150, svd_trunc=M.TruncThresh(0.0), tol=1e-15, schedule="sequential", shuffle_nodes=False)
    assert _rel(np.array(M.beliefs(bp)[0]), np.array(M.beliefs(bc)[0])) < 1e-8
    assert abs(M.bethe_free_energy(bp) - M.bethe_free_energy(bc) / Nc) < 1e-8

    kk = (3, 2)
    JA, JB, h = 1.0, -0.2, -0.1
    w = [[M.HomogeneousGlauberFactor(JA, h, 1.0)] * (T + 1), [M.HomogeneousGlauberFactor(JB, h, 1.0)] * (T + 1)]
    phi = [[np.array([(1 + m0) / 2, (1 - m0) / 2]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(2)]
    phi[0][1] = np.array([0.4, 0.6])
    phi[1][T] = np.array([0.95, 0.05])
    bp = M.mpbp_infinite_bipartite_graph(kk, w, (2, 2), phi, max_bond=64)
    M.iterate(bp, maxiter=150, svd_trunc=M.TruncThresh(0.0), tol=1e-15, damp=0.1)
    Nc = sum(kk)
    Ac = np.zeros((Nc, Nc))
    Ac[:kk[1], kk[1]:] = 1                       # complete_bipartite_graph(2, 3): two nodes of degree 3, three of degree 2
    Ac = Ac + Ac.T
    wc = [w[0] if i < kk[1] else w[1] for i in range(Nc)]
    pc = [phi[0] if i < kk[1] else phi[1] for i in range(Nc)]
    bc = M.mpbp(M.IndexedBiDiGraph(Ac), wc, 2, T, phi=pc, max_bond=64)
    M.iterate(bc, maxiter=150, svd_trunc=M.TruncThresh(0.0), tol=1e-15, schedule="sequential", shuffle_nodes=False)
    b_inf, b_c = M.beliefs(bp), M.beliefs(bc)
    assert _rel(np.array(b_inf[0]), np.array(b_c[0])) < 1e-8 and _rel(np.array(b_inf[1]), np.array(b_c[kk[0]])) < 1e-8
    assert abs(M.bethe_free_energy(bp) - M.bethe_free_energy(bc) / Nc) < 1e-8


def test_pair_observations_gpu():
    """reference test/pair_observations.jl: time-dependent psi on the edges."""
    T = 2
    A = np.array([[0, 1, 1], [1, 0, 0], [1, 0, 0]])
    g = M.IndexedBiDiGraph(A)
    rng = np.random.default_rng(3)
    psi = [None] * g.E
    for (i, j, ij) in g.edges():
        if i < j:
            m = [rng.random((2, 2)) + 0.1 for _ in range(T + 1)]
            psi[ij] = m
            psi[int(g.rev[ij])] = [x.T.copy() for x in m]
    phi = [[np.array([0.5, 0.5]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(3)]
    bp = M.mpbp(g, [[M.SISFactor(0.5, 0.4, 0.1)] * (T + 1)] * 3, 2, T, phi=phi, psi=psi, max_bond=16)
    M.iterate(bp, maxiter=10, svd_trunc=M.TruncBond(16), schedule="colored")
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(0.5, 0.4, 0.1)] * (T + 1)] * 3, [2] * 3, T, phi=phi, psi=psi)
    p, Z = exact_prob(obp)
    assert _rel(_flat(M.beliefs(bp)), _flat(exact_marginals(obp, p))) < 1e-9
    assert abs(np.exp(-M.bethe_free_energy(bp)) - Z) / Z < 1e-9
    assert _rel(_flat(M.pair_beliefs(bp)[0]), _flat(exact_pair_marginals(obp, p))) < 1e-9
    # reference test/normalizations.jl:46-51: every stored message is normalised (z = 1, the represented function sums to 1)
    for cores in bp.get_messages():
        assert abs(OT.normalization_log(OT.TensorTrain(cores))) < 1e-12


def test_error_paths_do_not_abort():
    A = np.array([[0, 1], [1, 0]])
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(0.5, 0.4)] * 3] * 2, 2, 2, max_bond=4)
    with pytest.raises(M.MPBPError):
        M.onebpiter(bp, [0], M.TruncBond(4), damp=1.5)      # reference asserts 0 <= damp < 1
    with pytest.raises(M.MPBPError):
        M.onebpiter(bp, [0, 0], M.TruncBond(4))
    with pytest.raises(M.MPBPError):
        M.onebpiter(bp, [7], M.TruncBond(4))


def test_full_size_config2_properties():
    """BASELINE configs[1] at full size (N=1024, T=50, bond 20): size-independent properties after 3 Jacobi
    sweeps (the oracle would need hours here): normalised beliefs, transposed pair beliefs on reverse edges,
    bounded bonds, finite free energy, no NaN / capacity / Jacobi flags."""
    N, T, Mb = 1024, 50, 20
    A = nx.to_numpy_array(nx.random_regular_graph(3, N, seed=0), nodelist=range(N))
    gam = 0.1
    phi = [[np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    g = M.IndexedBiDiGraph(A)
    bp = M.mpbp(g, [[M.SISFactor(0.1, 0.05)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
    for _ in range(3):
        M.onebpiter(bp, np.arange(N, dtype=np.int32), M.TruncBond(Mb))
        st = bp.last_stats
        assert st.nan_flag == 0 and st.capacity_flag == 0 and st.jacobi_not_converged == 0
    assert st.n_compress == N * (7 + 3)
    b = np.array(M.beliefs(bp))
    assert np.isfinite(b).all() and (b >= -1e-12).all() and np.abs(b.sum(axis=2) - 1).max() < 1e-12
    assert np.abs(b[:, 0, 1] - gam).max() < 1e-6          # time-0 marginal = prior up to the truncation error
    bonds = bp.bonds()
    assert bonds.max() == Mb and (bonds[:, 0] == 1).all() and (bonds[:, -1] == 1).all()
    pb, lz = M.pair_beliefs(bp)
    pb = np.array(pb)
    assert np.abs(pb.sum(axis=(2, 3)) - 1).max() < 1e-12
    rev = g.rev
    assert np.abs(pb - np.transpose(pb[rev], (0, 1, 3, 2))).max() < 1e-10
    assert np.isfinite(M.bethe_free_energy(bp))
    # without observations the SIS prior is normalised: Z = 1  =>  F -> 0 as BP converges; already small
    assert abs(M.bethe_free_energy(bp)) < 1e-2 * N


def test_full_size_config1_sampled_node_updates_match_oracle():
    """BASELINE configs[1] at FULL size (SIS on random_regular_graph(3, 1024, seed=0), T=50, TruncBond(20)) against the oracle
    where the oracle can follow: after four sweeps on the device (bonds at the cap everywhere) the in-messages of two nodes are
    read back, the device does its fifth sweep, and the oracle (reference algorithm: two SVD sweeps per `op`,
    recursive_bp_factor.jl:146-165) updates those two nodes from the very same messages - belief and free-energy term of each
    at the contract's 1e-6 (observed ~1e-12).  The property test above cannot see a wrong-but-normalised result; this can."""
    import ctypes as C
    N, T, Mb = 1024, 50, 20
    A = nx.to_numpy_array(nx.random_regular_graph(3, N, seed=0), nodelist=range(N))
    gam, lam, rho = 0.1, 0.1, 0.05
    phi = [[np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    g = M.IndexedBiDiGraph(A)
    bp = M.mpbp(g, [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
    allnodes = np.arange(N, dtype=np.int32)
    for _ in range(4):
        M.onebpiter(bp, allnodes, M.TruncBond(Mb))
    assert bp.bonds().max() == Mb
    ptr, ine, oute = g.nbr_arrays()
    sample = [5, 777]
    snap = {}
    for i in sample:
        e_in = [int(ine[p_]) for p_ in range(ptr[i], ptr[i + 1])]
        msgs = bp.get_messages(edges=e_in)
        snap[i] = [(e, msgs[e]) for e in e_in]
        assert max(c.shape[0] for _, m in snap[i] for c in m) == Mb            # saturated inputs
    M.onebpiter(bp, allnodes, M.TruncBond(Mb))
    b = np.array(M.beliefs(bp))
    f = np.zeros(N)
    bp._L.mpbp_free_energy(bp._h, f.ctypes.data_as(C.POINTER(C.c_double)))
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
    for i in sample:
        for e, m in snap[i]:
            obp.mu[e] = OT.TensorTrain([np.array(c) for c in m])
        O.onebpiter(obp, i, OT.TruncBond(Mb))
        ob = np.array(O.beliefs(obp)[i])
        assert np.abs(b[i] - ob).max() <= 1e-6 * max(1.0, np.abs(ob).max()), (i, np.abs(b[i] - ob).max())
        assert (np.abs(b[i] - ob) / np.maximum(ob, 1e-300)).max() < 1e-6
        assert abs(f[i] - obp.f[i]) <= 1e-6 * max(1.0, abs(obp.f[i])), (i, f[i], obp.f[i])


def _mixed_sis_sirs(A, qs, T, seed):
    rng = np.random.default_rng(seed)
    phi = [[rng.random(q) + 0.1 for _ in range(T + 1)] for q in qs]
    w = [[M.SISFactor(0.3, 0.2)] * (T + 1) if q == 2 else [M.SIRSFactor(0.3, 0.2, 0.1)] * (T + 1) for q in qs]
    ow = [[OF.SISFactor(0.3, 0.2)] * (T + 1) if q == 2 else [OF.SIRSFactor(0.3, 0.2, 0.1)] * (T + 1) for q in qs]
    return phi, w, ow


def test_heterogeneous_nstates_tree_exact_gpu():
    """Nodes with different numbers of states (`nstates(bp, i)`, reference src/mpbp.jl:22-26) on the device: SIS nodes (q = 2)
    and SIRS nodes (q = 3) on one tree, random node observations, no truncation - beliefs, pair beliefs (q_i x q_j blocks)
    and the partition function against brute-force enumeration at 1e-9, and against the oracle.  The device pads every node to
    q = 3 states; the padding must carry exactly zero weight (mpbp_set_node_states)."""
    T = 2                                # 36^3 trajectories to enumerate
    A = np.array([[0, 1, 1, 0], [1, 0, 0, 1], [1, 0, 0, 0], [0, 1, 0, 0]])
    qs = [2, 3, 2, 3]
    phi, w, ow = _mixed_sis_sirs(A, qs, T, 0)
    bp = M.mpbp(M.IndexedBiDiGraph(A), w, qs, T, phi=phi, max_bond=16)
    M.iterate(bp, maxiter=6, svd_trunc=M.TruncThresh(0.0), tol=1e-14, shuffle_nodes=False)
    obp = O.mpbp(O.IndexedBiDiGraph(A), ow, qs, T, phi=phi)
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(obp)
    m = exact_marginals(obp, p)
    pm = exact_pair_marginals(obp, p)
    b = M.beliefs(bp)
    assert [np.array(x).shape for x in b] == [(T + 1, q) for q in qs]
    assert max(np.abs(np.array(b[i]) - np.array(m[i])).max() for i in range(4)) < 1e-9
    assert abs(np.exp(-M.bethe_free_energy(bp)) - Z) / Z < 1e-9
    pb, _ = M.pair_beliefs(bp)
    for e, (i, j, _) in enumerate(bp.g.edges()):
        assert np.array(pb[e]).shape == (T + 1, qs[i], qs[j])
        assert np.abs(np.array(pb[e]) - np.array(pm[e])).max() < 1e-9


def test_heterogeneous_nstates_loopy_binding_truncation_matches_oracle_gpu():
    """The same mixed SIS / SIRS model on a loopy graph with a binding TruncBond(4) and damping: every Jacobi sweep against the
    oracle (beliefs, per-node free-energy terms, pair beliefs) at the contract's 1e-6.  Bond tables may exceed the oracle's
    by directions of zero weight where q_i q_j (not the cap) limits a bond, so they are compared as upper bounds."""
    import ctypes as C
    T = 5
    A = nx.to_numpy_array(nx.random_regular_graph(3, 6, seed=3), nodelist=range(6))
    qs = [2, 3, 3, 2, 3, 2]
    phi, w, ow = _mixed_sis_sirs(A, qs, T, 1)
    bp = M.mpbp(M.IndexedBiDiGraph(A), w, qs, T, phi=phi, max_bond=4)
    obp = O.mpbp(O.IndexedBiDiGraph(A), ow, qs, T, phi=phi)
    for s in range(4):
        M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(4), tol=0.0, damp=0.2)
        O.iterate(obp, maxiter=1, svd_trunc=OT.TruncBond(4), tol=0.0, shuffle_nodes=False, jacobi=True, damp=0.2)
        b, ob = M.beliefs(bp), O.beliefs(obp)
        for i in range(6):
            np.testing.assert_allclose(np.array(b[i]), np.array(ob[i]), rtol=1e-6, atol=1e-12)
        f = np.zeros(6)
        bp._L.mpbp_free_energy(bp._h, f.ctypes.data_as(C.POINTER(C.c_double)))
        np.testing.assert_allclose(f, np.array(obp.f), rtol=1e-6, atol=1e-9)
    pb, lz = M.pair_beliefs(bp)
    opb, olz = O.pair_beliefs(obp)
    for e in range(bp.g.ne()):
        np.testing.assert_allclose(np.array(pb[e]), np.array(opb[e]), rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(lz, olz, rtol=1e-6, atol=1e-9)
    assert bp.bonds().max() == 4
    assert (bp.bonds() >= np.array([m.bonds for m in obp.mu])).all()


def _check_node_properties(bp, nodes, Mb, st):
    assert st.nan_flag == 0 and st.capacity_flag == 0 and st.jacobi_not_converged == 0
    b = np.array(M.beliefs(bp))[nodes]
    assert np.isfinite(b).all() and (b >= -1e-12).all() and np.abs(b.sum(axis=2) - 1).max() < 1e-12
    bonds = bp.bonds()
    assert bonds.max() == Mb and (bonds[:, 0] == 1).all() and (bonds[:, -1] == 1).all()
    return b


def _saturate_inputs(bp, g, nodes, Mb, seed):
    """Random normalised messages at the saturated bond profile on every in-edge of `nodes` (M.random_message): the
    dimensions of a converged state without the sweeps that lead there."""
    rng = np.random.default_rng(seed)
    ptr, ine, oute = g.nbr_arrays()
    msgs = [None] * g.ne()
    for i in nodes:
        for p in range(ptr[i], ptr[i + 1]):
            msgs[int(ine[p])] = M.random_message(bp.T, bp.q, Mb, rng)
    bp.set_messages(msgs)


def test_full_size_baseline_configs2_glauber_er_properties():
    """BASELINE configs[2] at its stated dimensions (homogeneous Glauber J=0.5, h=0, beta=1 on
    networkx.gnp_random_graph(2048, 4/2047, seed=0), T=100, TruncBond(30); glauber_bp.jl:22-44,128-131): one update, at
    saturated incoming bonds, of nodes of degree 0 .. 6 (product bond 900, nstates = l+1 growing along the cavity:
    Y_t up to 900 x 7 x 2 = 12600 rows), through the batched gauge sweep.  Size-independent properties.
    (The two sweeps that bring the whole graph to the cap take ~90 s on one GPU and are not repeated here; a saturated sweep
    of one of the eight node blocks takes 211 - 228 s, profiles/r04_config2_all_shards.txt; the degree-9 hub is pinned by the
    fixture tests/golden/hub_glauber9.npz.)"""
    N, T, Mb = 2048, 100, 30
    G = nx.gnp_random_graph(N, 4 / (N - 1), seed=0)
    A = nx.to_numpy_array(G, nodelist=range(N))
    m0 = -0.6
    phi = [[np.array([(1 + m0) / 2, (1 - m0) / 2]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    gl = M.Glauber(M.Ising(0.5 * A, np.zeros(N), 1.0), T, phi=phi)
    bp = gl.mpbp(max_bond=Mb)
    deg = A.sum(axis=0).astype(int)
    assert (deg == 0).any() and deg.max() >= 9            # isolated nodes and a long cavity chain are part of the config
    sub = np.array(sorted(int(np.nonzero(deg == z)[0][0]) for z in range(0, 7)), dtype=np.int32)
    _saturate_inputs(bp, bp.g, sub, Mb, 5)
    M.onebpiter(bp, sub, M.TruncBond(Mb))
    st = bp.last_stats
    assert st.n_compress == sum(max(3 * int(deg[i]) - 2, 0) + int(deg[i]) for i in sub)
    b = _check_node_properties(bp, sub, Mb, st)
    iso = int(np.nonzero(sub == np.nonzero(deg == 0)[0][0])[0][0])
    assert abs(b[iso, 0, 0] - (1 + m0) / 2) < 1e-12       # isolated node: the prior itself (random inputs elsewhere)
    assert np.isfinite(M.bethe_free_energy(bp))


def test_full_size_baseline_configs3_karate_properties():
    """BASELINE configs[3] at its stated dimensions (SIS lambda=0.1 rho=0.05 on notebooks/karate.txt, node 0 infected at
    t=0, T=200, TruncBond(40)): one update at saturated incoming bonds of nodes of degree 1 .. 5 (products of bond
    40 x 40 = 1600, Y_t = 6400 x 1600).  Full sweeps over the whole graph including the two hubs (cavity chains of 46 /
    49 products; 91 / 122 s for the third / fourth whole-graph sweep) are recorded in
    profiles/r04_config3_karate_T200_d40_full_sweeps.log; the degree-17 hub is pinned by tests/golden/hub_karate17.npz."""
    A = _karate()
    N, T, Mb = 34, 200, 40
    phi = [[np.array([0.0, 1.0]) if (t == 0 and i == 0) else (np.array([1.0, 0.0]) if t == 0 else np.ones(2))
            for t in range(T + 1)] for i in range(N)]
    g = M.IndexedBiDiGraph(A)
    bp = M.mpbp(g, [[M.SISFactor(0.1, 0.05)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
    deg = A.sum(axis=0).astype(int)
    sub = np.array(sorted(int(np.nonzero(deg == z)[0][0]) for z in (1, 2, 3, 4, 5)), dtype=np.int32)
    _saturate_inputs(bp, g, sub, Mb, 6)
    M.onebpiter(bp, sub, M.TruncBond(Mb))
    st = bp.last_stats
    b = _check_node_properties(bp, sub, Mb, st)
    assert np.abs(b[:, 0, 1]).max() < 1e-9               # none of these is node 0: observed susceptible at t = 0
    assert np.isfinite(M.bethe_free_energy(bp))


def test_full_size_baseline_configs4_infinite_graph_properties():
    """BASELINE configs[4] at its stated dimensions (src/infinite_graph.jl:8-43: k=3 copies of one message, SIS
    lambda=0.1 rho=0.2, gamma=0.1, T=200, TruncBond(64)): four in-place iterations - bonds 2, 8, 64 and then one
    iteration with every product at bond 64 x 64 = 4096 (Y_t = 16384 x 4096, 537 MB per time step)."""
    T, Mb, gam = 200, 64, 0.1
    phi = [np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)]
    bp = M.mpbp_infinite_graph(3, [M.SISFactor(0.1, 0.2)] * (T + 1), 2, phi, max_bond=Mb)
    prev = None
    for it in range(4):
        M.onebpiter(bp, [0], M.TruncBond(Mb))
        st = bp.last_stats
        assert st.nan_flag == 0 and st.capacity_flag == 0 and st.jacobi_not_converged == 0
        b = np.array(M.beliefs(bp)[0])
        assert np.isfinite(b).all() and np.abs(b.sum(axis=1) - 1).max() < 1e-12
        assert abs(b[0, 1] - gam) < 1e-6
        if prev is not None:
            assert np.abs(b - prev).max() < 0.2           # iterations move towards the fixed point, no blow-up
        prev = b
    assert bp.bonds().max() == Mb
    assert np.isfinite(M.bethe_free_energy(bp))


def _karate():
    import os
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "karate.txt")
    return np.loadtxt(p)


def test_sis_karate_heterogeneous_degree_matches_oracle():
    """BASELINE configs[3] structure (karate club, degrees 1..17 -> cavity DAGs up to 49 ops, 17 levels) at a
    size the oracle finishes in seconds: T=4, bond 5, 2 Jacobi sweeps, node 0 infected at t=0."""
    A = _karate()
    N, T, Mb = A.shape[0], 4, 5
    assert N == 34 and int(A.sum()) == 156 and int(A.sum(axis=0).max()) == 17
    phi = [[np.array([0.0, 1.0]) if (t == 0 and i == 0) else (np.array([1.0, 0.0]) if t == 0 else np.ones(2))
            for t in range(T + 1)] for i in range(N)]
    lam, rho = 0.1, 0.05
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
    for s in range(2):
        M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(Mb), tol=0.0)
        O.iterate(obp, maxiter=1, svd_trunc=OT.TruncBond(Mb), tol=0.0, shuffle_nodes=False, jacobi=True)
        assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < RTOL, f"sweep {s}"
    import ctypes as C
    f = np.zeros(N)
    bp._L.mpbp_free_energy(bp._h, f.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.abs(f - obp.f).max() < RTOL * max(1.0, np.abs(obp.f).max())
    assert np.array_equal(bp.bonds(), np.array([m.bonds for m in obp.mu]))


def test_glauber_erdos_renyi_loopy_matches_oracle():
    """BASELINE configs[2] structure (homogeneous Glauber on an ER graph: nstates = l+1 grows along the cavity,
    isolated and degree-1 nodes present) at oracle-feasible size."""
    G = nx.gnp_random_graph(14, 4 / 13, seed=1)
    A = nx.to_numpy_array(G, nodelist=range(14))
    N, T, Mb = 14, 4, 6
    J = 0.5 * A
    h = np.zeros(N)
    m0 = -0.6
    phi = [[np.array([(1 + m0) / 2, (1 - m0) / 2]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    gl = M.Glauber(M.Ising(J, h, 1.0), T, phi=phi)
    bp = gl.mpbp(max_bond=Mb)
    obp = O.mpbp(O.IndexedBiDiGraph(A), OF.glauber_factors(A != 0, J, h, 1.0, T), [2] * N, T, phi=phi)
    for s in range(2):
        M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(Mb), tol=0.0)
        O.iterate(obp, maxiter=1, svd_trunc=OT.TruncBond(Mb), tol=0.0, shuffle_nodes=False, jacobi=True)
        assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < RTOL, f"sweep {s}"
    pb, lz = M.pair_beliefs(bp)
    opb, olz = O.pair_beliefs(obp)
    assert _rel(_flat(pb), _flat(opb)) < RTOL
    assert abs(M.bethe_free_energy(bp) - O.bethe_free_energy(obp)) < RTOL * max(1.0, np.abs(obp.f).sum())


def test_trunc_thresh_loopy_matches_oracle():
    """default_truncator = TruncThresh(1e-6) (reference src/mpems.jl:161) on a loopy graph: data-dependent bonds."""
    N, T = 8, 6
    lam, rho, gam = 0.2, 0.1, 0.15
    A, phi = _loopy(N, T, lam, rho, gam)
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=16)
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
    for s in range(3):
        M.iterate(bp, maxiter=1, svd_trunc=M.TruncThresh(1e-6), tol=0.0)
        O.iterate(obp, maxiter=1, svd_trunc=OT.TruncThresh(1e-6), tol=0.0, shuffle_nodes=False, jacobi=True)
        assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < RTOL, f"sweep {s}"
    assert np.array_equal(bp.bonds(), np.array([m.bonds for m in obp.mu]))


def test_infinite_bipartite_graph_matches_oracle():
    """reference src/infinite_graph.jl:68-122 (two node types, aliased edges)."""
    T = 4
    k = (2, 3)
    w = [[M.HomogeneousGlauberFactor(0.3, 0.1, 1.0)] * (T + 1), [M.HomogeneousGlauberFactor(0.3, -0.2, 1.0)] * (T + 1)]
    ow = [[OF.HomogeneousGlauberFactor(0.3, 0.1, 1.0)] * (T + 1), [OF.HomogeneousGlauberFactor(0.3, -0.2, 1.0)] * (T + 1)]
    phi = [[np.array([0.7, 0.3]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(2)]
    bp = M.mpbp_infinite_bipartite_graph(k, w, (2, 2), phi, max_bond=8)
    obp = O.mpbp_infinite_bipartite_graph(k, ow, (2, 2), phi)
    M.iterate(bp, maxiter=15, svd_trunc=M.TruncBond(8), tol=0.0, schedule="sequential", shuffle_nodes=False)
    O.iterate(obp, maxiter=15, svd_trunc=OT.TruncBond(8), tol=0.0, shuffle_nodes=False)
    assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < RTOL
    assert abs(M.bethe_free_energy(bp) - O.bethe_free_energy(obp)) < RTOL * max(1.0, abs(O.bethe_free_energy(obp)))


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_rank_sharded_sweeps_equal_single_rank(tmp_path, backend):
    """Multi-GPU path end to end: two ranks each update their node block and all-gather the message slots
    after every sweep; beliefs and f must equal the single-process run up to rounding.  `gloo`: both ranks share the one
    GPU of the test box (host-staged exchange); `nccl`: the production exchange - RCCL in-place all_gather_into_tensor of
    the slab, one GPU per rank - needs two GPUs and is skipped otherwise."""
    import os
    import subprocess
    import sys
    import torch
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("the RCCL exchange needs one GPU per rank (2 GPUs)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--nodes", "32", "--T", "8", "--bond", "6", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = str(tmp_path / "one.npy")
    two = str(tmp_path / "two.npy")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common + ["--dump-beliefs", one],
                        capture_output=True, text=True, env=env, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                         "--gpus", "2", "--backend", backend] + common + ["--dump-beliefs", two],
                        capture_output=True, text=True, env=env, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    a = np.load(one, allow_pickle=True).item()
    b = np.load(two, allow_pickle=True).item()
    assert np.abs(a["beliefs"] - b["beliefs"]).max() < 1e-12
    assert np.abs(a["f"] - b["f"]).max() < 1e-10
    import json
    j2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j2["scaling"] == "strong"


def test_damping_loopy_matches_oracle():
    """set_msg! with damp > 0 (reference src/recursive_bp_factor.jl:168-179): _compose + compress! + normalize!."""
    N, T, Mb = 8, 6, 6
    lam, rho, gam = 0.2, 0.1, 0.15
    A, phi = _loopy(N, T, lam, rho, gam)
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
    for s in range(3):
        M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(Mb), tol=0.0, damp=0.3)
        O.iterate(obp, maxiter=1, svd_trunc=OT.TruncBond(Mb), tol=0.0, shuffle_nodes=False, jacobi=True, damp=0.3)
        assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < RTOL, f"sweep {s}"
        import ctypes as C
        f = np.zeros(N)
        bp._L.mpbp_free_energy(bp._h, f.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.abs(f - obp.f).max() < RTOL * max(1.0, np.abs(obp.f).max()), f"sweep {s}"
    pb, _ = M.pair_beliefs(bp)
    opb, _ = O.pair_beliefs(obp)
    assert _rel(_flat(pb), _flat(opb)) < RTOL


def test_damping_infinite_graph_compounds_like_reference():
    """Infinite regular graph with damping: the reference's loop re-reads bp.mu[1] for each of the k aliased
    out-edges, so the damping compounds k times per iteration (test/glauber_infinite_graph.jl:22)."""
    T, k = 5, 3
    w = [M.HomogeneousGlauberFactor(0.3, 0.1, 1.0)] * (T + 1)
    ow = [OF.HomogeneousGlauberFactor(0.3, 0.1, 1.0)] * (T + 1)
    phi = [np.array([0.7, 0.3]) if t == 0 else np.ones(2) for t in range(T + 1)]
    bp = M.mpbp_infinite_graph(k, w, 2, phi, max_bond=8)
    obp = O.mpbp_infinite_graph(k, ow, 2, phi)
    for s in range(6):
        M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(8), tol=0.0, damp=0.4)
        O.iterate(obp, maxiter=1, svd_trunc=OT.TruncBond(8), tol=0.0, damp=0.4)
        assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < RTOL, f"iteration {s}"
    assert abs(M.bethe_free_energy(bp) - O.bethe_free_energy(obp)) < RTOL * max(1.0, abs(O.bethe_free_energy(obp)))


def test_large_problem_code_paths_forced(monkeypatch):
    """Configs 3-5 (bond 30-64) use code paths that small tests never reach: operands and Jacobi buffers in
    global memory instead of LDS, the global-memory QR panel (> 2048 rows).  MPBP_DEBUG_FORCE_GENERIC=1 forces
    them on a small loopy case; results must still match the oracle."""
    monkeypatch.setenv("MPBP_DEBUG_FORCE_GENERIC", "1")
    N, T, Mb = 8, 6, 6
    lam, rho, gam = 0.2, 0.1, 0.15
    A, phi = _loopy(N, T, lam, rho, gam)
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
    for s in range(2):
        M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(Mb), tol=0.0)
        O.iterate(obp, maxiter=1, svd_trunc=OT.TruncBond(Mb), tol=0.0, shuffle_nodes=False, jacobi=True)
        assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < RTOL, f"sweep {s}"
    assert np.array_equal(bp.bonds(), np.array([m.bonds for m in obp.mu]))


def test_single_wave_engine_variant_forced(monkeypatch):
    """Small problems run on single-wave workgroups (v64::eng_kernel) only when there are many of them; with few
    (every small test) the launcher prefers the 512-thread engine.  MPBP_DEBUG_FORCE_SMALL=1 keeps them on the
    single-wave engine: finalisation, bond-1 products and - with max_bond 4 - the cavity products themselves."""
    monkeypatch.setenv("MPBP_DEBUG_FORCE_SMALL", "1")
    for Mb in (4, 8):
        N, T = 8, 6
        lam, rho, gam = 0.2, 0.1, 0.15
        A, phi = _loopy(N, T, lam, rho, gam)
        bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
        obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
        for s in range(2):
            M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(Mb), tol=0.0)
            O.iterate(obp, maxiter=1, svd_trunc=OT.TruncBond(Mb), tol=0.0, shuffle_nodes=False, jacobi=True)
            assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < RTOL, f"max_bond {Mb} sweep {s}"
        assert abs(M.bethe_free_energy(bp) - O.bethe_free_energy(obp)) < RTOL * max(1.0, abs(O.bethe_free_energy(obp)))
        assert np.array_equal(bp.bonds(), np.array([m.bonds for m in obp.mu]))


def test_autocorrelations_match_enumeration():
    """reference test/sis_small_tree.jl:36-49: two-time observables from the belief trains (`bp.b[i]`)."""
    from oracle.exact import exact_autocorrelations
    A, lam, rho, alpha, phi, T = _sis_star_inputs()
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho, alpha)] * (T + 1)] * 4, 2, T, phi=phi, max_bond=16)
    M.iterate(bp, maxiter=10, svd_trunc=M.TruncBondMax(4), schedule="colored")
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho, alpha)] * (T + 1)] * 4, [2] * 4, T, phi=phi)
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(obp)
    f = lambda x, i: x - 1
    r = M.autocorrelations(f, bp)
    rex = exact_autocorrelations(f, obp, p)
    assert max(np.abs(a - b).max() for a, b in zip(r, rex)) < 1e-9
    c = M.autocovariances(f, bp)
    assert np.isfinite(np.array(c)).all()
    # the belief train is normalised and reproduces the marginals
    tr = M.belief_train(bp, 0)
    assert tr[0].shape[0] == 1 and tr[-1].shape[1] == 1


def test_twovar_marginals_on_device_match_oracle_scan():
    """mpbp_twovar_marginals (tv_env_kernel / tv_kernel: TensorTrains `twovar_marginals` of the belief trains on the
    device, reference src/mpbp.jl:239-255) against the oracle's scan of the same trains, with and without maxdist, on a
    loopy graph with binding truncation (belief bonds up to q * max_bond)."""
    N, T, Mb = 8, 7, 5
    lam, rho, gam = 0.2, 0.1, 0.15
    A, phi = _loopy(N, T, lam, rho, gam)
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
    M.iterate(bp, maxiter=3, svd_trunc=M.TruncBond(Mb), tol=0.0)
    for maxdist in (None, 3):
        tu = M.beliefs_tu(bp, sites=[0, 3, 7], maxdist=maxdist)
        for k, i in enumerate([0, 3, 7]):
            ref = OT.twovar_marginals(OT.TensorTrain(M.belief_train(bp, i)), maxdist=maxdist)
            for t in range(T + 1):
                for u in range(T + 1):
                    if ref[t][u] is None:
                        assert tu[k][t][u] is None
                    else:
                        assert np.abs(tu[k][t][u] - ref[t][u]).max() < 1e-12, (i, t, u)
                        assert abs(tu[k][t][u].sum() - 1) < 1e-12
    # consistency with the one-time marginals
    b = M.beliefs(bp)
    tu = M.beliefs_tu(bp, sites=[2])[0]
    assert np.abs(tu[1][5].sum(axis=1) - b[2][1]).max() < 1e-10 and np.abs(tu[1][5].sum(axis=0) - b[2][5]).max() < 1e-10


@pytest.mark.parametrize("case", ["path3_plus_isolated_T1", "path3_plus_isolated_T5", "single_edge", "hub6"])
def test_edge_case_graphs(case):
    """Ragged inputs: an isolated node (degree 0: no cavity at all), leaves, the shortest possible chain (T = 1),
    a two-node graph, and a degree-6 hub (cavity chain of depth 6, packed launches)."""
    if case.startswith("path3"):
        A = np.zeros((4, 4)); A[0, 1] = A[1, 0] = A[1, 2] = A[2, 1] = 1
        T, Mb = (1 if case.endswith("T1") else 5), 4
    elif case == "single_edge":
        A = np.zeros((2, 2)); A[0, 1] = A[1, 0] = 1
        T, Mb = 3, 4
    else:
        A = np.zeros((7, 7)); A[0, 1:] = 1; A[1:, 0] = 1
        T, Mb = 4, 6
    N = A.shape[0]
    lam, rho, gam = 0.3, 0.2, 0.2
    phi = [[np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho)] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb)
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho)] * (T + 1)] * N, [2] * N, T, phi=phi)
    for s in range(2):
        M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(Mb), tol=0.0)
        O.iterate(obp, maxiter=1, svd_trunc=OT.TruncBond(Mb), tol=0.0, shuffle_nodes=False, jacobi=True)
    assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < RTOL
    assert abs(M.bethe_free_energy(bp) - O.bethe_free_energy(obp)) < RTOL * max(1.0, abs(O.bethe_free_energy(obp)))


def test_more_observables_match_enumeration_and_oracle():
    """pair_correlations, alternate_marginals / alternate_correlations (src/mpbp.jl:264-286), logprob
    (src/mpbp.jl:301-324), the factor call `w(x', x_nbrs, x)`, reset_observations / is_free_dynamics
    (src/mpbp.jl:89-111) on the reference's 4-star (test/sis_small_tree.jl)."""
    A, lam, rho, alpha, phi, T = _sis_star_inputs()
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho, alpha)] * (T + 1)] * 4, 2, T, phi=phi, max_bond=16)
    M.iterate(bp, maxiter=10, svd_trunc=M.TruncBondMax(4), schedule="colored")
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho, alpha)] * (T + 1)] * 4, [2] * 4, T, phi=phi)
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(obp)
    # p(x_i^t, x_j^{t+1}) per directed edge from the enumeration
    am = M.alternate_marginals(bp)
    for (i, j, e) in obp.g.edges():
        for t in range(T):
            a, b = i * (T + 1) + t, j * (T + 1) + t + 1
            m = p.sum(axis=tuple(c for c in range(p.ndim) if c not in (a, b)))
            m = m if a < b else m.T
            assert np.abs(am[e][t] - m).max() < 1e-9, (e, t)
    f2 = lambda x, y: (x - 1) * (y - 1)
    ac = M.alternate_correlations(f2, bp)
    assert abs(ac[0][0] - sum(f2(x + 1, y + 1) * am[0][0][x, y] for x in range(2) for y in range(2))) < 1e-14
    pc = M.pair_correlations(f2, bp)
    pb = M.pair_beliefs(bp)[0]
    assert abs(pc[2][1] - pb[2][1][1, 1]) < 1e-14
    # the train form of the pair beliefs reproduces the pair beliefs
    tr = M.pair_beliefs_as_mpem(bp, edges=[0])[0]
    summed = [c.sum(axis=(2, 3)) for c in tr]
    lv, rv = [np.ones(1)], [np.ones(1)]
    for c in summed:
        lv.append(lv[-1] @ c)
    for c in reversed(summed):
        rv.append(c @ rv[-1])
    rv = rv[::-1]
    for t in range(T + 1):
        m = np.einsum("m,mnxy,n->xy", lv[t], tr[t], rv[t + 1])
        assert np.abs(m / m.sum() - pb[0][t]).max() < 1e-12
    # logprob of a trajectory vs the oracle (and, observing nothing, the factor call sums to one)
    rng = np.random.default_rng(5)
    for _ in range(5):
        X = rng.integers(0, 2, size=(4, T + 1))
        with np.errstate(divide="ignore"):
            a, b = M.logprob(bp, X + 1), O.logprob(obp, X)
        assert (a == b) or abs(a - b) < 1e-12 * max(1.0, abs(b))
    w = M.SISFactor(lam, rho, alpha)
    for xn in ([1, 1, 2], [2, 2, 2], []):
        for x in (1, 2):
            assert abs(sum(w(xp, xn, x) for xp in (1, 2)) - 1.0) < 1e-14
            assert abs(w(2, xn, x) - OF.SISFactor(lam, rho, alpha)(2, xn, x)) < 1e-15
    assert not M.is_free_dynamics(bp)          # the last time is observed
    M.reset_observations(bp)
    assert M.is_free_dynamics(bp)
    M.reset(bp)
    M.iterate(bp, maxiter=10, svd_trunc=M.TruncBondMax(4), schedule="colored")
    b = M.beliefs(bp)
    assert abs(b[0][0][1] - 0.5) < 1e-9        # free dynamics: the time-0 marginal is uniform after reset_observations


def test_randomised_models_graphs_truncations():
    """tools/fuzz.py: 40 random cases (graph, SIS / SIRS / Glauber variants, T, bond cap, SVDTrunc rule, damping,
    number of sweeps) against the oracle; beliefs, pair beliefs and free energy within 1e-6 (observed <= 2e-12)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for args in (["40", "123"], ["10", "5", "big"]):       # "big": bond caps 10..18 (2- and 4-panel QR paths)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz.py")] + args, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_baseline_config0_glauber_3node_path_exact_gpu():
    """BASELINE.json configs[0]: Glauber on the 3-node path, T = 3, TruncBond(4), against enumeration (1e-8)."""
    T = 3
    J = np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]], float)
    h = np.random.default_rng(0).standard_normal(3)
    phi = [[np.array([0.75, 0.25]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(3)]
    gl = M.Glauber(M.Ising(J, h, 1.0), T, phi=phi)
    bp = gl.mpbp(max_bond=4)
    M.iterate(bp, maxiter=10, svd_trunc=M.TruncBond(4), schedule="sequential")
    obp = O.mpbp(O.IndexedBiDiGraph(J != 0), OF.glauber_factors(J != 0, J, h, 1.0, T), [2] * 3, T, phi=phi)
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(obp)
    assert _rel(_flat(M.beliefs(bp)), _flat(exact_marginals(obp, p))) < 1e-8
    assert abs(np.exp(-M.bethe_free_energy(bp)) - Z) / Z < 1e-8


# ------------------------------------------------------------------------------------------------------------------
# generic (exhaustive-trace) factors on the device: f_bp / f_bp_dummy_neighbor (reference src/bp_core.jl:18-93) and the
# generic onebpiter! (src/mpbp.jl:117-154) through mpbp_set_generic_factor
# ------------------------------------------------------------------------------------------------------------------
def test_generic_factor_equals_recursive_and_enumeration_sis_tree_gpu():
    """reference test/sis_small_tree.jl:68-83: `GenericFactor.(w)` forces the exhaustive-trace update; beliefs and pair
    beliefs must equal the recursive path's - and, on a tree, brute-force enumeration."""
    A, lam, rho, alpha, phi, T = _sis_star_inputs()
    tr = M.TruncBondMax(4)
    bp = M.mpbp(M.IndexedBiDiGraph(A), [[M.SISFactor(lam, rho, alpha)] * (T + 1)] * 4, 2, T, phi=phi, max_bond=4)
    M.iterate(bp, maxiter=10, svd_trunc=tr, schedule="sequential")
    wg = [[M.GenericFactor(M.SISFactor(lam, rho, alpha)) for _ in range(T + 1)] for _ in range(4)]
    bpg = M.mpbp(M.IndexedBiDiGraph(A), wg, 2, T, phi=phi, max_bond=4)
    M.iterate(bpg, maxiter=10, svd_trunc=M.TruncBondMax(4), schedule="sequential")
    assert _rel(_flat(M.beliefs(bpg)), _flat(M.beliefs(bp))) < 1e-9
    pbg, lzg = M.pair_beliefs(bpg)
    pb, lz = M.pair_beliefs(bp)
    assert _rel(_flat(pbg), _flat(pb)) < 1e-9
    assert abs(M.bethe_free_energy(bpg) - M.bethe_free_energy(bp)) < 1e-9
    obp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(lam, rho, alpha)] * (T + 1)] * 4, [2] * 4, T, phi=phi)
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(obp)
    assert _rel(_flat(M.beliefs(bpg)), _flat(exact_marginals(obp, p))) < 1e-9
    assert abs(np.exp(-M.bethe_free_energy(bpg)) - Z) / Z < 1e-9


def test_generic_glauber_factor_real_couplings_tree_exact_gpu():
    """reference test/glauber_pmJ_small_tree.jl:64-86 with couplings that no recursive factor type covers
    (`glauber_factors` then picks `GenericGlauberFactor`, src/Models/glauber/glauber_bp.jl:121-142): exact on a tree,
    Z, marginals and autocorrelations against enumeration; time-dependent fields through phi."""
    T = 3
    J = np.array([[0, 0.3, 0, 0], [0.3, 0, -0.7, 1.1], [0, -0.7, 0, 0], [0, 1.1, 0, 0]], float)
    rng = np.random.default_rng(5)
    h = rng.standard_normal(4)
    phi = [[np.array([0.6, 0.4]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(4)]
    phi[2][T] = np.array([0.0, 1.0])
    w = M.glauber_factors(J != 0, J, h, 1.0, T)
    assert all(isinstance(wi[0], M.GenericGlauberFactor) for wi in w)
    bp = M.mpbp(M.IndexedBiDiGraph(J != 0), w, 2, T, phi=phi, max_bond=8)
    M.iterate(bp, maxiter=10, svd_trunc=M.TruncThresh(0.0), schedule="colored")
    ow = OF.glauber_factors(J != 0, J, h, 1.0, T)
    assert all(isinstance(wi[0], OF.GenericGlauberFactor) for wi in ow)
    obp = O.mpbp(O.IndexedBiDiGraph(J != 0), ow, [2] * 4, T, phi=phi)
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(obp)
    assert _rel(_flat(M.beliefs(bp)), _flat(exact_marginals(obp, p))) < 1e-9
    assert abs(np.exp(-M.bethe_free_energy(bp)) - Z) / Z < 1e-9
    pb, _ = M.pair_beliefs(bp)
    assert _rel(_flat(pb), _flat(exact_pair_marginals(obp, p))) < 1e-9


def test_generic_factor_loopy_binding_truncation_matches_oracle_gpu():
    """The generic update under a BINDING cap on a loopy graph, sweep by sweep against the oracle's `onebpiter_generic`
    (src/mpbp.jl:117-154: the belief is compressed before it is marginalised, messages are stored without damping), with
    generic and recursive nodes mixed in one graph (dispatch per node, src/mpbp.jl:191)."""
    N, T, Mb = 8, 4, 4
    lam, rho, gam = 0.2, 0.1, 0.15
    A, phi = _loopy(N, T, lam, rho, gam, seed=3)
    gen = [i % 2 == 0 for i in range(N)]
    w = [[(M.GenericFactor(M.SISFactor(lam, rho)) if gen[i] else M.SISFactor(lam, rho)) for _ in range(T + 1)] for i in range(N)]
    ow = [[(OF.GenericFactor(OF.SISFactor(lam, rho)) if gen[i] else OF.SISFactor(lam, rho)) for _ in range(T + 1)] for i in range(N)]
    bp = M.mpbp(M.IndexedBiDiGraph(A), w, 2, T, phi=phi, max_bond=Mb)
    obp = O.mpbp(O.IndexedBiDiGraph(A), ow, [2] * N, T, phi=phi)
    for s in range(3):
        M.iterate(bp, maxiter=1, svd_trunc=M.TruncBond(Mb), tol=0.0, damp=0.2)
        O.iterate(obp, maxiter=1, svd_trunc=OT.TruncBond(Mb), tol=0.0, shuffle_nodes=False, jacobi=True, damp=0.2)
        assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < RTOL, f"sweep {s}"
        f = _fnodes(bp)
        assert np.abs(f - obp.f).max() < RTOL * max(1.0, np.abs(obp.f).max()), f"sweep {s}"
    pb, lz = M.pair_beliefs(bp)
    opb, olz = O.pair_beliefs(obp)
    assert _rel(_flat(pb), _flat(opb)) < RTOL
    assert (bp.bonds() <= Mb).all()
    assert np.array_equal(bp.bonds(), np.array([m.bonds for m in obp.mu]))
    assert bp.last_stats.nan_flag == 0 and bp.last_stats.capacity_flag == 0


def test_generic_factor_limits_are_reported_not_aborted():
    """The exhaustive update is exponential in the degree: a product bond beyond what one workgroup's panel holds is refused
    with MPBP_EUNSUPPORTED where the factor is set - at CONSTRUCTION of the host mirror, with the limit in the message
    (round-3 advisor: it used to surface at the first sweep) - and so is a generic factor on a chain periodic in time."""
    A, lam, rho, alpha, phi, T = _sis_star_inputs()
    wg = [[M.GenericFactor(M.SISFactor(lam, rho, alpha)) for _ in range(T + 1)] for _ in range(4)]
    with pytest.raises(M.MPBPError) as ei:
        M.mpbp(M.IndexedBiDiGraph(A), wg, 2, T, phi=phi, max_bond=16)      # hub: 2 x 16^3 x 2 rows
    assert "generic factor" in str(ei.value) and "2048" in str(ei.value)
    with pytest.raises(M.MPBPError):
        M.periodic_mpbp(M.IndexedBiDiGraph(A), wg, 2, T, phi=phi, max_bond=4)


def test_generic_factor_degree4_isolated_node_and_time_dependent_tables_gpu():
    """Generic factors at the edges of the exhaustive update: a degree-4 hub (product bond 4^3 before the compress), a
    degree-0 node (belief from phi and the factor alone, no neighbours to sum over) and a factor that changes with time
    (nt = T + 1 blocks of the table) - all exact on this forest, against enumeration (2^18 trajectories: T = 2) and the
    oracle's generic update."""
    T = 2
    A = np.zeros((6, 6))
    for k in (1, 2, 3, 4):
        A[0, k] = A[k, 0] = 1                     # node 0: degree 4; node 5: isolated
    rng = np.random.default_rng(9)
    Jt = [0.4 + 0.3 * t for t in range(T + 1)]    # time-dependent coupling
    h = rng.standard_normal(6) * 0.3
    phi = [[np.array([0.7, 0.3]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(6)]
    phi[2][T] = np.array([1.0, 0.0])
    deg = A.sum(1).astype(int)
    w = [[M.GenericGlauberFactor([Jt[t] * (1 + 0.1 * k) for k in range(deg[i])], h[i], 1.0) for t in range(T + 1)] for i in range(6)]
    ow = [[OF.GenericGlauberFactor([Jt[t] * (1 + 0.1 * k) for k in range(deg[i])], h[i], 1.0) for t in range(T + 1)] for i in range(6)]
    bp = M.mpbp(M.IndexedBiDiGraph(A), w, 2, T, phi=phi, max_bond=4)
    M.iterate(bp, maxiter=6, svd_trunc=M.TruncThresh(0.0), schedule="colored")
    obp = O.mpbp(O.IndexedBiDiGraph(A), ow, [2] * 6, T, phi=phi)
    with np.errstate(divide="ignore"):
        p, Z = exact_prob(obp)
    assert _rel(_flat(M.beliefs(bp)), _flat(exact_marginals(obp, p))) < 1e-9
    assert abs(np.exp(-M.bethe_free_energy(bp)) - Z) / Z < 1e-9
    O.iterate(obp, maxiter=6, svd_trunc=OT.TruncThresh(0.0), tol=0.0, shuffle_nodes=False)
    assert _rel(_flat(M.beliefs(bp)), _flat(O.beliefs(obp))) < 1e-9
