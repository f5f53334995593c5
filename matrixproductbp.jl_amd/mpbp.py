"""Host orchestration mirroring the reference's driver (src/mpbp.jl, src/infinite_graph.jl): the
`MPBP` container, `mpbp(...)` constructors, `iterate!`, `onebpiter!`, `CB_BP`, `beliefs`,
`pair_beliefs`, `bethe_free_energy`.  Everything proportional to edges x T x bond^2 runs in
libmpbp_hip.so (include/mpbp_hip.h); this module only keeps the graph, evaluates factor tables and owns
the sweep loop and the convergence callback, as the reference's `iterate!` does (src/mpbp.jl:185-198).

Names follow the reference with `!` dropped (`iterate!` -> `iterate`, `onebpiter!` -> `onebpiter`).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import MPBPError, Stats, Trunc
from .factors import BPFactor, RecursiveBPFactor

__all__ = ["IndexedBiDiGraph", "InfiniteRegularGraph", "InfiniteBipartiteRegularGraph", "MPBP", "mpbp",
           "mpbp_infinite_graph", "mpbp_infinite_bipartite_graph", "iterate", "onebpiter", "CB_BP", "beliefs",
           "means", "belief_train", "twovar_marginals", "autocorrelations", "autocovariances", "pair_beliefs",
           "pair_beliefs_as_mpem", "pair_correlations", "alternate_marginals", "alternate_correlations", "expectation",
           "logprob", "reset", "reset_observations", "is_free_dynamics", "bethe_free_energy", "reset_messages", "TruncThresh", "TruncBond",
           "TruncBondMax", "TruncBondThresh", "default_truncator", "color_classes"]


# ----------------------------------------------------------------------------- SVDTrunc functors
class SVDTrunc:
    def _abi(self) -> Trunc:
        raise NotImplementedError


class TruncThresh(SVDTrunc):
    def __init__(self, eps):
        self.eps = float(eps)

    def _abi(self):
        return Trunc(_lib.MPBP_TRUNC_THRESH, 0, self.eps)

    def bond_capacity(self, q, T):
        return None

    def __repr__(self):
        return f"TruncThresh({self.eps})"


class TruncBond(SVDTrunc):
    def __init__(self, mprime):
        self.mprime = int(mprime)

    def _abi(self):
        return Trunc(_lib.MPBP_TRUNC_BOND, self.mprime, 0.0)

    def bond_capacity(self, q, T):
        return self.mprime

    def __repr__(self):
        return f"TruncBond({self.mprime})"


class TruncBondMax(TruncBond):
    def __init__(self, mprime):
        super().__init__(mprime)
        self.maxerr = 0.0

    def _abi(self):
        return Trunc(_lib.MPBP_TRUNC_BOND_MAX, self.mprime, 0.0)

    def __repr__(self):
        return f"TruncBondMax({self.mprime})"


class TruncBondThresh(TruncBond):
    def __init__(self, mprime, eps=0.0):
        super().__init__(mprime)
        self.eps = float(eps)

    def _abi(self):
        return Trunc(_lib.MPBP_TRUNC_BOND_THRESH, self.mprime, self.eps)

    def __repr__(self):
        return f"TruncBondThresh({self.mprime}, {self.eps})"


def default_truncator():
    """src/mpems.jl:161"""
    return TruncThresh(1e-6)


# ----------------------------------------------------------------------------------------- graphs
class IndexedBiDiGraph:
    """Symmetric adjacency -> directed edge ids = CSC positions (reference src/mpbp.jl:41-57,220-228)."""

    def __init__(self, A):
        A = (np.asarray(A) != 0)
        if A.shape[0] != A.shape[1] or not np.array_equal(A, A.T):
            raise ValueError("adjacency must be square and symmetric")
        self.N = A.shape[0]
        rows, cols = np.nonzero(A.T)          # column-major order of the nonzeros
        self.colptr = np.zeros(self.N + 1, dtype=np.int32)
        np.add.at(self.colptr, rows + 1, 1)
        self.colptr = np.cumsum(self.colptr).astype(np.int32)
        self.rowval = cols.astype(np.int32)   # for column j=rows[k]: source node cols[k]
        self.E = int(self.rowval.size)
        self.dst = rows.astype(np.int32)
        pos = {(int(s), int(d)): k for k, (s, d) in enumerate(zip(self.rowval, self.dst))}
        self.rev = np.array([pos[(int(d), int(s))] for s, d in zip(self.rowval, self.dst)], dtype=np.int32)

    def nv(self):
        return self.N

    def ne(self):
        return self.E

    def neighbors(self, i):
        return self.rowval[self.colptr[i]:self.colptr[i + 1]]

    def edges(self):
        return [(int(s), int(d), k) for k, (s, d) in enumerate(zip(self.rowval, self.dst))]

    def nbr_arrays(self):
        """(nbr_ptr, in_edge, out_edge) of the C ABI: in-edge ids of column i are the CSC positions,
        the out-edge to the same neighbour is the reverse edge."""
        in_edge = np.arange(self.E, dtype=np.int32)
        return self.colptr.copy(), in_edge, self.rev.copy()

    def degree(self, i):
        return int(self.colptr[i + 1] - self.colptr[i])


class InfiniteRegularGraph:
    """src/infinite_graph.jl:8-20: one node, one message, k aliases of the edge."""

    def __init__(self, k):
        self.k, self.N, self.E = int(k), 1, 1

    def nv(self):
        return 1

    def ne(self):
        return 1

    def edges(self):
        return [(0, 0, 0)]

    def nbr_arrays(self):
        z = np.zeros(self.k, dtype=np.int32)
        return np.array([0, self.k], dtype=np.int32), z, z.copy()

    def degree(self, i):
        return self.k


class InfiniteBipartiteRegularGraph:
    """src/infinite_graph.jl:68-91"""

    def __init__(self, k):
        self.k, self.N, self.E = (int(k[0]), int(k[1])), 2, 2

    def nv(self):
        return 2

    def ne(self):
        return 2

    def edges(self):
        return [(1, 0, 0), (0, 1, 1)]

    def nbr_arrays(self):
        k0, k1 = self.k
        ptr = np.array([0, k0, k0 + k1], dtype=np.int32)
        in_edge = np.array([0] * k0 + [1] * k1, dtype=np.int32)    # inedges(i) = edge id i
        out_edge = np.array([1] * k0 + [0] * k1, dtype=np.int32)   # outedges(i) = edge id 3-i
        return ptr, in_edge, out_edge

    def degree(self, i):
        return self.k[i]


def color_classes(g):
    """Greedy colouring: nodes of one class share no edge, so updating a class as one batch is
    identical to the reference's sequential in-place sweep over that class (src/mpbp.jl:190-192)."""
    N = g.nv()
    color = -np.ones(N, dtype=int)
    for i in range(N):
        used = {int(color[j]) for j in g.neighbors(i) if color[j] >= 0}
        c = 0
        while c in used:
            c += 1
        color[i] = c
    return [np.nonzero(color == c)[0].astype(np.int32) for c in range(int(color.max()) + 1)]


# ------------------------------------------------------------------------------------------- MPBP
def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class MPBP:
    """Mirror of the `MPBP` struct (src/mpbp.jl:1-33): graph `g`, factors `w[i][t]`, `ϕ[i][t][x]`,
    `ψ[e][t][x_i,x_j]`; messages `μ`, beliefs `b` and free-energy terms `f` live on the device."""

    def __init__(self, g, w, phi, psi, q, T, max_bond=None, device=0, slot_of_edge=None, n_slots=0,
                 ext_cores=None, ext_bonds=None, stream=None, periodic=False):
        N, E = g.nv(), g.ne()
        assert len(w) == len(phi) == N, f"{len(w)}, {len(phi)}, {N}"          # src/mpbp.jl:20
        assert len(psi) == E
        assert all(len(wi) == T + 1 for wi in w)
        assert all(len(p) == T + 1 for p in phi) and all(len(p) == T + 1 for p in psi)
        # nstates(bp, i) may differ from node to node (src/mpbp.jl:22-26): the device works with q = max_i q_i and treats the
        # states beyond q_i as padding of exactly zero weight (mpbp_set_node_states, include/mpbp_hip.h)
        qn = np.atleast_1d(np.asarray(q, dtype=np.int64))
        qn = np.full(N, int(qn[0])) if qn.size == 1 else qn
        if qn.size != N or qn.min() < 1:
            raise MPBPError(-1, f"q must be one positive integer or one per node ({N}), got {q}")
        self.qnode = np.ascontiguousarray(qn, dtype=np.int32)
        self.g, self.w, self.q, self.T = g, w, int(qn.max()), int(T)
        for i in range(N):
            # dispatch on the factor type as src/mpbp.jl:191 does (eltype(bp.w[i])): recursive or generic per node
            if not all(isinstance(wt, BPFactor) for wt in w[i]):
                raise MPBPError(-4, "factors must be BPFactor instances (callable as w(x_next, x_neighbours, x))")
            if len({isinstance(wt, RecursiveBPFactor) for wt in w[i]}) != 1:
                raise MPBPError(-4, f"node {i}: recursive and generic factors cannot be mixed along one node's chain")
            if periodic and not isinstance(w[i][0], RecursiveBPFactor):
                raise MPBPError(-4, "generic factors on chains periodic in time are not supported on the device path")
        self.max_bond = int(max_bond) if max_bond is not None else 16
        self.phi = np.zeros((self.q, T + 1, N))
        self.psi = np.zeros((self.q, self.q, T + 1, E))
        ends = self._edge_ends(g)
        for i in range(N):
            for t in range(T + 1):
                v = np.asarray(phi[i][t], dtype=float)
                if v.shape != (self.qnode[i],):
                    raise MPBPError(-1, f"phi[{i}][{t}] must have nstates({i}) = {self.qnode[i]} entries")
                self.phi[:v.size, t, i] = v
        for e in range(E):
            for t in range(T + 1):
                m = np.asarray(psi[e][t], dtype=float)
                if ends is not None and m.shape != (self.qnode[ends[e][0]], self.qnode[ends[e][1]]):
                    raise MPBPError(-1, f"psi[{e}][{t}] must be nstates(src) x nstates(dst) = {self.qnode[ends[e][0]]} x {self.qnode[ends[e][1]]}")
                self.psi[:m.shape[0], :m.shape[1], t, e] = m
        self._ends = ends
        self._check_psis()
        L = _lib.lib()
        self._L = L
        ptr, ine, oute = g.nbr_arrays()
        self._ptr, self._in, self._out = (np.ascontiguousarray(a, dtype=np.int32) for a in (ptr, ine, oute))
        d = _lib.Desc()
        d.n_nodes, d.n_edges, d.T, d.q = N, E, T, self.q
        d.nbr_ptr, d.in_edge, d.out_edge = _ip(self._ptr), _ip(self._in), _ip(self._out)
        d.max_bond, d.device = self.max_bond, device
        self._slot = None
        if slot_of_edge is not None:
            self._slot = np.ascontiguousarray(slot_of_edge, dtype=np.int32)
            d.slot_of_edge, d.n_slots = _ip(self._slot), int(n_slots)
        d.ext_cores = ext_cores
        d.ext_bonds = ext_bonds
        d.stream = stream
        d.periodic = 1 if periodic else 0
        self.periodic = bool(periodic)
        h = C.c_void_p()
        _lib.check(L.mpbp_create(C.byref(h), C.byref(d)))
        self._h = h
        if (self.qnode != self.q).any():
            _lib.check(L.mpbp_set_node_states(h, _ip(self.qnode)), h)
        self._set_factors()
        _lib.check(L.mpbp_set_phi(h, _dp(np.asfortranarray(self.phi).ravel(order="F"))), h)
        _lib.check(L.mpbp_set_psi(h, _dp(np.asfortranarray(self.psi).ravel(order="F"))), h)
        self.last_stats = None

    @staticmethod
    def _edge_ends(g):
        """(src, dst) node of every edge id, or None where the edge set is implicit (InfiniteRegularGraph: one node)."""
        try:
            return {int(e): (int(i), int(j)) for (i, j, e) in g.edges()}
        except Exception:
            return None

    def _check_psis(self):
        """src/mpbp.jl:40-58: ψ on i->j must be the transpose of ψ on j->i."""
        g = self.g
        if not isinstance(g, IndexedBiDiGraph):
            return
        for e in range(g.E):
            r = g.rev[e]
            if not np.array_equal(self.psi[:, :, :, e], np.transpose(self.psi[:, :, :, r], (1, 0, 2))):
                raise AssertionError("check_ψs failed: ψ[i→j] != ψ[j→i]'")

    def _set_factors(self):
        L, h, T, q = self._L, self._h, self.T, self.q
        cache = {}
        for i in range(self.g.nv()):
            deg = self.g.degree(i)
            wi = self.w[i]
            const = all(wt is wi[0] or wt.key() == wi[0].key() for wt in wi)
            ws = [wi[0]] if const else list(wi)
            k = (tuple(wt.key() for wt in ws), deg)
            if not isinstance(wi[0], RecursiveBPFactor):
                # generic BPFactor: exhaustive-trace update (src/bp_core.jl:18-93) from the dense transition table
                if k not in cache:
                    cache[k] = np.ascontiguousarray(np.concatenate([wt.generic_table(deg, q) for wt in ws]))
                _lib.check(L.mpbp_set_generic_factor(h, i, deg, len(ws), _dp(cache[k])), h)
                continue
            if k not in cache:
                tabs = [wt.tables(deg, q) for wt in ws]
                ny = tabs[0][0]
                cache[k] = (ny, *(np.ascontiguousarray(np.concatenate([tb[j] for tb in tabs])) for j in (1, 2, 3, 4)))
            ny, py, pxy, pyy, py0 = cache[k]
            if pxy.size == 0:
                pxy = np.zeros(1)
            _lib.check(L.mpbp_set_factor(h, i, deg, _ip(ny), len(ws), _dp(py), _dp(pxy), _dp(pyy), _dp(py0)), h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.mpbp_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # --- message access (packed Julia layout) ------------------------------------------------
    def bonds(self):
        b = np.zeros((self.g.ne(), self.T + 2), dtype=np.int32)
        _lib.check(self._L.mpbp_get_bonds(self._h, _ip(b)), self._h)
        return b

    def get_messages(self, edges=None):
        """list over edges of lists over t of arrays [b_t, b_{t+1}, q, q] (`bp.μ`); `edges` restricts the
        download (other entries are None)."""
        b = self.bonds()
        q, E = self.q, self.g.ne()
        sel = np.ones(E, dtype=bool) if edges is None else np.isin(np.arange(E), np.asarray(edges))
        sizes = (b[:, :-1].astype(np.int64) * b[:, 1:] * q * q)
        tot = np.where(sel, sizes.sum(axis=1), 0)
        offs = np.concatenate([[0], np.cumsum(tot)[:-1]]).astype(np.int64)
        offs[~sel] = -1
        data = np.zeros(int(tot.sum()))
        _lib.check(self._L.mpbp_get_messages(self._h, offs.ctypes.data_as(C.POINTER(C.c_int64)), _dp(data)), self._h)
        out = []
        for e in range(E):
            if not sel[e]:
                out.append(None)
                continue
            o, cores = int(offs[e]), []
            for t in range(self.T + 1):
                n = int(sizes[e, t])
                cores.append(data[o:o + n].reshape((b[e, t], b[e, t + 1], q, q), order="F").copy())
                o += n
            out.append(cores)
        return out

    def set_messages(self, msgs):
        """`bp.μ[e] = msgs[e]` (lists over t of arrays [b_t, b_{t+1}, q, q], normalised); entries that are None keep
        the message the device holds."""
        q, E, T = self.q, self.g.ne(), self.T
        b = np.ones((E, T + 2), dtype=np.int32)
        chunks, offs, o = [], np.zeros(E, dtype=np.int64), 0
        for e in range(E):
            if msgs[e] is None:
                offs[e] = -1
                continue
            offs[e] = o
            for t in range(T + 1):
                a = np.asarray(msgs[e][t], dtype=np.float64)
                b[e, t], b[e, t + 1] = a.shape[0], a.shape[1]
                chunks.append(a.ravel(order="F"))
                o += a.size
        data = np.concatenate(chunks) if chunks else np.zeros(1)
        _lib.check(self._L.mpbp_set_messages(self._h, _ip(b), offs.ctypes.data_as(C.POINTER(C.c_int64)), _dp(data)),
                   self._h)


def random_message(T, q, bond, rng):
    """A normalised random MPEM2 at the saturated bond profile min(bond, q^2t, q^2(T+1-t)) with positive cores: a
    stand-in for a converged message when only the DIMENSIONS of the update matter (full-size tests, benchmarks of
    one saturated sweep without the sweeps that lead there)."""
    L = T + 1
    prof = [int(min(bond, float(q * q) ** min(t, L - t, 40))) for t in range(L + 1)]
    cores = [rng.uniform(0.5, 1.5, size=(prof[t], prof[t + 1], q, q)) / (prof[t + 1] * q * q) for t in range(L)]
    v, logz = np.ones((1, 1)), 0.0
    for a in cores:
        v = v @ a.sum(axis=(2, 3))
        s = np.abs(v).max()
        v, logz = v / s, logz + np.log(s)
    logz += np.log(v[0, 0])
    f = np.exp(-logz / L)
    return [a * f for a in cores]


def mpbp(g, w, q, T, d=1, phi=None, psi=None, max_bond=None, **kw):
    """src/mpbp.jl:60-70.  Messages start as `flat_mpem2(q_i, q_j, T; d)` (src/mpbp.jl:66): constant cores of bond
    `d` (1 at the two ends), normalised - for `d = 1` the library's initial state, for `d > 1` uploaded once.
    `max_bond` is the device capacity of every stored bond (>= the truncation cap used later)."""
    N = g.nv()
    q = np.atleast_1d(q)
    q = np.full(N, int(q[0])) if q.size == 1 else q
    phi = [[np.ones(q[i]) for _ in range(T + 1)] for i in range(N)] if phi is None else phi
    psi = [[np.ones((q[i], q[j])) for _ in range(T + 1)] for (i, j, _) in g.edges()] if psi is None else psi
    bp = MPBP(g, w, phi, psi, q, T, max_bond=max_bond, **kw)
    if d != 1:
        if d > bp.max_bond:
            raise MPBPError(-5, f"initial bond size d = {d} exceeds max_bond = {bp.max_bond}")
        L, qq = T + 1, bp.q
        prof = [1] + [d] * T + [1]
        # constant cores whose product sums to one: every entry of core t equals c_t with prod_t (c_t b_{t+1} q^2) = 1
        msg = [np.full((prof[t], prof[t + 1], qq, qq), 1.0 / (prof[t + 1] * qq * qq)) for t in range(L)]
        bp.set_messages([msg] * g.ne())
    return bp


def periodic_mpbp(g, w, q, T, phi=None, psi=None, max_bond=None, **kw):
    """`periodic_mpbp` (src/mpbp.jl:399-409): chains periodic in time - `w[i][T+1]` couples x^{T+1} back to x^1
    (src/exact.jl:24-26).  On the device the messages stay open trains (mpbp_desc::periodic, include/mpbp_hip.h)."""
    return mpbp(g, w, q, T, phi=phi, psi=psi, max_bond=max_bond, periodic=True, **kw)


def periodic_mpbp_infinite_graph(k, wi, qi, phi_i=None, psi=None, max_bond=None, **kw):
    """`periodic_mpbp_infinite_graph` (src/infinite_graph.jl:37-43)"""
    return mpbp_infinite_graph(k, wi, qi, phi_i, psi, max_bond=max_bond, periodic=True, **kw)


def is_periodic(bp):
    """src/mpbp.jl:113-114"""
    return bool(bp.periodic)


def mpbp_infinite_graph(k, wi, qi, phi_i=None, psi=None, max_bond=None, **kw):
    """src/infinite_graph.jl:22-35"""
    T = len(wi) - 1
    phi_i = [np.ones(qi) for _ in range(T + 1)] if phi_i is None else phi_i
    psi = [np.ones((qi, qi)) for _ in range(T + 1)] if psi is None else psi
    return MPBP(InfiniteRegularGraph(k), [wi], [phi_i], [psi], [qi], T, max_bond=max_bond, **kw)


def mpbp_infinite_bipartite_graph(k, w, qi, phi=None, psi=None, max_bond=None, **kw):
    """src/infinite_graph.jl:93-108"""
    T = len(w[0]) - 1
    phi = [[np.ones(qi[i]) for _ in range(T + 1)] for i in range(2)] if phi is None else phi
    psi = [[np.ones((qi[i], qi[1 - i])) for _ in range(T + 1)] for i in range(2)] if psi is None else psi
    # psi is indexed by edge id as in the reference (edge id i = message into node i); the reference
    # requires psi[0] == psi[1] (src/infinite_graph.jl:111)
    return MPBP(InfiniteBipartiteRegularGraph(k), w, phi, psi, [qi[0], qi[1]], T, max_bond=max_bond, **kw)


def reset_messages(bp: MPBP):
    """src/mpbp.jl:72-80"""
    _lib.check(bp._L.mpbp_reset_messages(bp._h), bp._h)


def onebpiter(bp: MPBP, i, svd_trunc=None, damp=0.0):
    """`onebpiter!(bp, i, U; svd_trunc, damp)` (src/recursive_bp_factor.jl:146-165); `i` may be a list of
    nodes, which are then updated together from the current messages (see mpbp_sweep in mpbp_hip.h)."""
    svd_trunc = default_truncator() if svd_trunc is None else svd_trunc
    nodes = np.ascontiguousarray(np.atleast_1d(i), dtype=np.int32)
    st = Stats()
    rc = bp._L.mpbp_sweep(bp._h, _ip(nodes), int(nodes.size), svd_trunc._abi(), float(damp), C.byref(st))
    _lib.check(rc, bp._h)
    bp.last_stats = st
    if isinstance(svd_trunc, TruncBondMax):
        svd_trunc.maxerr = max(svd_trunc.maxerr, st.maxerr)
    if st.nan_flag:
        print("Error: NaN in tensor train")      # reference: @error, then continues
    if st.jacobi_not_converged:
        # the one-sided Jacobi behind an SVDTrunc decision hit its sweep limit: singular vectors of that step are
        # less accurate than LAPACK's would be - never silently (the messages are still stored, as after a NaN)
        import warnings
        warnings.warn("libmpbp_hip: a Jacobi SVD did not converge within its sweep limit in this update "
                      "(mpbp_stats.jacobi_not_converged); results of this sweep may deviate from the reference",
                      RuntimeWarning, stacklevel=2)
    return None


def beliefs(bp: MPBP):
    """src/mpbp.jl:237: `b[i][t][x]`"""
    out = np.zeros((bp.q, bp.T + 1, bp.g.nv()))
    buf = np.zeros(out.size)
    _lib.check(bp._L.mpbp_beliefs(bp._h, _dp(buf)), bp._h)
    out = buf.reshape(out.shape, order="F")
    return [[out[:bp.qnode[i], t, i].copy() for t in range(bp.T + 1)] for i in range(bp.g.nv())]


def belief_train(bp: MPBP, i: int):
    """`bp.b[i]`: the belief as a normalised MPEM1 (list over t of arrays [b_t, b_{t+1}, q])."""
    bonds = np.zeros(bp.T + 2, dtype=np.int32)
    _lib.check(bp._L.mpbp_get_belief_train(bp._h, int(i), _ip(bonds), None, 0), bp._h)
    sizes = bonds[:-1].astype(np.int64) * bonds[1:] * bp.q
    data = np.zeros(int(sizes.sum()))
    _lib.check(bp._L.mpbp_get_belief_train(bp._h, int(i), _ip(bonds), _dp(data), data.size), bp._h)
    out, o = [], 0
    for t in range(bp.T + 1):
        out.append(data[o:o + int(sizes[t])].reshape((bonds[t], bonds[t + 1], bp.q), order="F").copy())
        o += int(sizes[t])
    return out


def twovar_marginals(cores, maxdist=None):
    """TensorTrains `twovar_marginals` for an MPEM1: `out[t][u][x_t, x_u]`, t < u <= t + maxdist, each
    normalised (used by beliefs_tu / autocorrelations, reference src/mpbp.jl:239-255)."""
    L = len(cores)
    maxdist = L if maxdist is None else maxdist
    summed = [c.sum(axis=2) for c in cores]
    r = [None] * (L + 1)
    r[L] = np.ones(1)
    for t in range(L - 1, -1, -1):
        v = summed[t] @ r[t + 1]
        r[t] = v / np.abs(v).max()
    out = [[None] * L for _ in range(L)]
    lv = np.ones(1)
    for t in range(L):
        mid = np.einsum("m,mnx->xn", lv, cores[t])
        for u in range(t + 1, min(L, t + maxdist + 1)):
            p = np.einsum("xm,mny,n->xy", mid, cores[u], r[u + 1])
            out[t][u] = p / p.sum()
            mid = mid @ summed[u]
            mid = mid / np.abs(mid).max()
        lv = lv @ summed[t]
        lv = lv / np.abs(lv).max()
    return out


def _abi_maxdist(maxdist):
    """mpbp_twovar_marginals takes maxdist <= 0 as "all distances"; here None means all, and 0 or less means NO pair
    (TensorTrains' twovar_marginals(; maxdist) loops u in t+1 : min(L, t + maxdist)): returns -1 for "skip the call"."""
    if maxdist is None:
        return 0
    if int(maxdist) <= 0:
        return -1
    return int(maxdist)


def beliefs_tu(bp: MPBP, sites=None, maxdist=None):
    """`beliefs_tu` (src/mpbp.jl:239-243): two-time marginals `out[i][t][u][x_t, x_u]` (t < u <= t + maxdist, else
    None) of the listed nodes, computed on the device from the belief trains (mpbp_twovar_marginals)."""
    sites = list(range(bp.g.nv())) if sites is None else list(sites)
    L, q = bp.T + 1, bp.q
    nodes = np.ascontiguousarray(sites, dtype=np.int32)
    buf = np.zeros(len(sites) * L * L * q * q)
    md = _abi_maxdist(maxdist)
    if md >= 0:           # maxdist = 0: no pair at all - nothing to compute (the ABI's 0 means "all distances")
        _lib.check(bp._L.mpbp_twovar_marginals(bp._h, _ip(nodes), int(nodes.size), md, _dp(buf)), bp._h)
    arr = buf.reshape((len(sites), L, L, q, q))           # [k][t][u][y][x] in memory order x fastest
    md = L if maxdist is None else int(maxdist)
    return [[[arr[k, t, u].T.copy() if t < u <= t + md else None for u in range(L)] for t in range(L)]
            for k in range(len(sites))]


def autocorrelations(f, bp: MPBP, sites=None, maxdist=None):
    """src/mpbp.jl:245-255: `r[i][t, u] = <f(x_i^t) f(x_i^u)>` for t < u (0 elsewhere); the O(T^2) two-time
    marginals come from the device (mpbp_twovar_marginals), only the q x q contraction with f runs here."""
    sites = list(range(bp.g.nv())) if sites is None else list(sites)
    L, q = bp.T + 1, bp.q
    nodes = np.ascontiguousarray(sites, dtype=np.int32)
    buf = np.zeros(len(sites) * L * L * q * q)
    md = _abi_maxdist(maxdist)
    if md >= 0:           # maxdist = 0: no pair at all (zeros), the same meaning as in beliefs_tu
        _lib.check(bp._L.mpbp_twovar_marginals(bp._h, _ip(nodes), int(nodes.size), md, _dp(buf)), bp._h)
    arr = buf.reshape((len(sites), L, L, q, q))
    out = []
    for k, i in enumerate(sites):
        fx = np.array([f(x + 1, i) for x in range(q)])
        out.append(np.einsum("tuyx,x,y->tu", arr[k], fx, fx))
    return out


def autocovariances(f, bp: MPBP, sites=None, maxdist=None):
    """src/mpbp.jl:288-294: `r - mu mu'`."""
    sites = list(range(bp.g.nv())) if sites is None else list(sites)
    mu = means(f, bp)
    r = autocorrelations(f, bp, sites, maxdist)
    return [ri - np.outer(mu[i], mu[i]) for ri, i in zip(r, sites)]


def means(f, bp: MPBP):
    """src/mpbp.jl:257-261 (`f(x, i)` with the 1-based state x)."""
    b = beliefs(bp)
    return [[sum(f(x + 1, i) * p[x] for x in range(len(p))) for p in bi] for i, bi in enumerate(b)]


def pair_beliefs(bp: MPBP):
    """src/mpbp.jl:202-235 (+ src/infinite_graph.jl:37-43,110-116): `(b[e][t][x_i,x_j], logz[i])`."""
    E, q, T = bp.g.ne(), bp.q, bp.T
    buf = np.zeros(q * q * (T + 1) * E)
    lz = np.zeros(E)
    _lib.check(bp._L.mpbp_pair_beliefs(bp._h, _dp(buf), _dp(lz)), bp._h)
    pb = buf.reshape((q, q, T + 1, E), order="F")
    ends = bp._ends
    qe = (lambda e: (bp.qnode[ends[e][0]], bp.qnode[ends[e][1]])) if ends is not None else (lambda e: (q, q))
    b = [[pb[:qe(e)[0], :qe(e)[1], t, e].copy() for t in range(T + 1)] for e in range(E)]
    g = bp.g
    if isinstance(g, InfiniteRegularGraph):
        logz = np.array([(1 / (g.k - 1) - 0.5) * lz[0]])
    elif isinstance(g, InfiniteBipartiteRegularGraph):
        # reference indexes by node i: f(μ[i], μ[3-i], ψ[i]); μ[i] is edge id i here as well
        logz = np.array([(1 / (g.k[i] - 1) - 0.5) * lz[i] for i in range(2)])
    else:
        logz = np.zeros(g.nv())
        for j in range(g.N):
            dj = g.degree(j)
            for k in range(g.colptr[j], g.colptr[j + 1]):
                logz[j] += (1 / dj - 0.5) * lz[k]
    return b, logz


def _rev_edge(g, e):
    if isinstance(g, InfiniteRegularGraph):
        return 0
    if isinstance(g, InfiniteBipartiteRegularGraph):
        return 1 - e
    return int(g.rev[e])


def expectation(f, p):
    """`expectation(f, p)` of the reference for a vector or a matrix of probabilities (1-based states)."""
    p = np.asarray(p)
    if p.ndim == 1:
        return float(sum(f(x + 1) * p[x] for x in range(p.shape[0])))
    return float(sum(f(x + 1, y + 1) * p[x, y] for x in range(p.shape[0]) for y in range(p.shape[1])))


def pair_beliefs_as_mpem(bp: MPBP, edges=None):
    """src/mpbp.jl:208-216, src/bp_core.jl:95-101: for the directed edge e = (i -> j) the (unnormalised) train
    `C[t][(a,b),(a',b'),x_i,x_j] = mu_ij[t][a,a',x_i,x_j] mu_ji[t][b,b',x_j,x_i] psi_ij[t][x_i,x_j]`, built on the
    host from the downloaded messages.  Returns a dict edge -> list of cores (bond = product of the two bonds)."""
    g, q, T = bp.g, bp.q, bp.T
    sel = list(range(g.ne())) if edges is None else [int(e) for e in edges]
    msgs = bp.get_messages(edges=sorted(set(sel) | {_rev_edge(g, e) for e in sel}))
    out = {}
    for e in sel:
        A, B = msgs[e], msgs[_rev_edge(g, e)]
        cores = []
        for t in range(T + 1):
            a, b = A[t], B[t]
            c = np.einsum("acxy,bdyx,xy->abcdxy", a, b, bp.psi[:, :, t, e])
            cores.append(c.reshape(a.shape[0] * b.shape[0], a.shape[1] * b.shape[1], q, q))
        out[e] = cores
    return out


def pair_correlations(f, bp: MPBP):
    """src/mpbp.jl:264-267: `<f(x_i^t, x_j^t)>` per directed edge."""
    return [[expectation(f, bt) for bt in be] for be in pair_beliefs(bp)[0]]


def alternate_marginals(bp: MPBP, edges=None):
    """src/mpbp.jl:270-280: `p(x_i^t, x_j^{t+1})`, t = 0..T-1, per directed edge (i -> j): the (t, t+1) two-time
    marginal of the pair belief train, summed over x_j^t and x_i^{t+1}.  Returns a dict edge -> list over t."""
    out = {}
    for e, cores in pair_beliefs_as_mpem(bp, edges).items():
        L = len(cores)
        summed = [c.sum(axis=(2, 3)) for c in cores]
        r = [None] * (L + 1)
        r[L] = np.ones(1)
        for t in range(L - 1, -1, -1):
            v = summed[t] @ r[t + 1]
            r[t] = v / np.abs(v).max()
        lv = np.ones(1)
        res = []
        for t in range(L - 1):
            left = np.einsum("m,mnx->xn", lv, cores[t].sum(axis=3))          # keep x_i^t
            right = np.einsum("mny,n->my", cores[t + 1].sum(axis=2), r[t + 2])   # keep x_j^{t+1}
            p = left @ right
            res.append(p / p.sum())
            lv = lv @ summed[t]
            lv = lv / np.abs(lv).max()
        out[e] = res
    return out


def alternate_correlations(f, bp: MPBP, edges=None):
    """src/mpbp.jl:283-286"""
    return {e: [expectation(f, p) for p in am] for e, am in alternate_marginals(bp, edges).items()}


def logprob(bp: MPBP, x):
    """src/mpbp.jl:301-324: log of the (unnormalised) posterior weight of the trajectory `x[i, t]` (1-based)."""
    g, T = bp.g, bp.T
    N = g.nv()
    x = np.asarray(x)
    assert x.shape == (N, T + 1)
    with np.errstate(divide="ignore"):
        lp = sum(np.log(bp.phi[x[i, 0] - 1, 0, i]) for i in range(N))
        for t in range(T):
            for i in range(N):
                nb = [int(v) for v in g.neighbors(i)]
                lp += np.log(bp.w[i][t](int(x[i, t + 1]), [int(x[j, t]) for j in nb], int(x[i, t])))
                lp += np.log(bp.phi[x[i, t + 1] - 1, t + 1, i])
        for t in range(T + 1):
            for (i, j, e) in g.edges():
                lp += 0.5 * np.log(bp.psi[x[i, t] - 1, x[j, t] - 1, t, e])
    return float(lp)


def reset_observations(bp: MPBP):
    """src/mpbp.jl:89-95: all `phi` := 1."""
    bp.phi[:] = 1.0
    _lib.check(bp._L.mpbp_set_phi(bp._h, _dp(np.asfortranarray(bp.phi).ravel(order="F"))), bp._h)


def reset(bp: MPBP, messages=True, beliefs=True, observations=False):
    """src/mpbp.jl:97-102 (the beliefs live on the device and are overwritten by the next update)."""
    if messages:
        reset_messages(bp)
    if observations:
        reset_observations(bp)


def is_free_dynamics(bp: MPBP):
    """src/mpbp.jl:105-111: no reweighting <-> every phi but the one at time zero is constant in x."""
    return bool(np.all(bp.phi[:, 1:, :] == bp.phi[:1, 1:, :]))


def bethe_free_energy(bp: MPBP):
    """src/mpbp.jl:298; bipartite reweighting src/infinite_graph.jl:118-122"""
    f = np.zeros(bp.g.nv())
    _lib.check(bp._L.mpbp_free_energy(bp._h, _dp(f)), bp._h)
    if isinstance(bp.g, InfiniteBipartiteRegularGraph):
        k = bp.g.k
        return float((f[0] * k[1] + f[1] * k[0]) / (k[0] + k[1]))
    return float(f.sum())


class CB_BP:
    """src/mpbp.jl:157-183 (without the progress bar)."""

    def __init__(self, bp, f=lambda x, i: x, showprogress=False):
        self.f = f
        self.m = [means(f, bp)]
        self.deltas = []

    def __call__(self, bp, it, svd_trunc):
        new = means(self.f, bp)
        old = self.m[-1]
        d = max(max(abs(a - b) for a, b in zip(mn, mo)) for mn, mo in zip(new, old)) if new else float("nan")
        self.deltas.append(d)
        self.m.append(new)
        return d


def iterate(bp: MPBP, maxiter=5, svd_trunc=None, showprogress=False, cb=None, tol=1e-10, nodes=None,
            shuffle_nodes=True, damp=0.0, schedule="jacobi", rng=None):
    """`iterate!` (src/mpbp.jl:185-198).  `schedule` picks how one sweep over `nodes` is issued:
      "jacobi"     all nodes from the messages at sweep start in one device call (fastest);
      "colored"    one device call per colour class = the reference's in-place sweep in class order;
      "sequential" one device call per node in the reference's order (exact reference semantics:
                   node order 1..N first, then a fresh random order per sweep when shuffle_nodes)."""
    svd_trunc = default_truncator() if svd_trunc is None else svd_trunc
    cb = CB_BP(bp) if cb is None else cb
    nodes = np.arange(bp.g.nv(), dtype=np.int32) if nodes is None else np.asarray(nodes, dtype=np.int32)
    rng = np.random.default_rng(0) if rng is None else rng
    classes = color_classes(bp.g) if schedule == "colored" else None
    for it in range(1, maxiter + 1):
        if schedule == "jacobi":
            onebpiter(bp, nodes, svd_trunc, damp)
        elif schedule == "colored":
            keep = set(int(v) for v in nodes)
            for cl in classes:
                sel = np.array([v for v in cl if int(v) in keep], dtype=np.int32)
                if sel.size:
                    onebpiter(bp, sel, svd_trunc, damp)
        elif schedule == "sequential":
            for i in nodes:
                onebpiter(bp, [int(i)], svd_trunc, damp)
        else:
            raise ValueError("schedule must be jacobi, colored or sequential")
        d = cb(bp, it, svd_trunc)
        if d < tol:
            return it, cb
        if shuffle_nodes and schedule == "sequential":
            nodes = rng.permutation(bp.g.nv()).astype(np.int32)
    return maxiter, cb
