"""MPBP update and observables - restatement of the reference hot path (TEST INFRASTRUCTURE).

Follows ``src/recursive_bp_factor.jl`` (whole), ``src/mpems.jl:27-94``, ``src/bp_core.jl``
(whole), ``src/mpbp.jl:1-70,117-154,185-261,298-324``, ``src/infinite_graph.jl:8-122`` and the
``cavity`` function of CavityTools.jl (not in the reference tree; call order restated from its
published source).
"""
from __future__ import annotations

import itertools

import numpy as np

from .factors import BPFactor, RecursiveBPFactor
from .tensor_trains import (TensorTrain, TruncThresh, accumulate_L, compose_sum, compress,
                            flat_tt, marginals, normalization_log, normalize,
                            normalize_eachmatrix, twovar_marginals)


# ----------------------------------------------------------------------------------- graph

class IndexedBiDiGraph:
    """IndexedGraphs.IndexedBiDiGraph of a symmetric adjacency: directed edge ``i->j`` has id =
    CSC position of (row i, column j) (reference mpbp.jl:41-57,220-228); ``inedges(j)`` and
    ``outedges(j)`` enumerate neighbours in ascending order."""

    def __init__(self, A):
        A = (np.asarray(A) != 0)
        assert A.shape[0] == A.shape[1] and np.array_equal(A, A.T), "adjacency must be symmetric"
        self.N = A.shape[0]
        self.colptr = [0]
        self.rowval = []
        for j in range(self.N):
            rows = np.nonzero(A[:, j])[0]
            self.rowval.extend(int(r) for r in rows)
            self.colptr.append(len(self.rowval))
        self.E = len(self.rowval)
        pos = {}
        for j in range(self.N):
            for k in range(self.colptr[j], self.colptr[j + 1]):
                pos[(self.rowval[k], j)] = k
        # rev[k] for edge k = (i->j) is the id of (j->i)  (``nonzeros(g.X)``)
        self.rev = [pos[(j, self.rowval[k])] for j in range(self.N) for k in range(self.colptr[j], self.colptr[j + 1])]

    def nv(self):
        return self.N

    def ne(self):
        return self.E

    def vertices(self):
        return range(self.N)

    def neighbors(self, i):
        return [self.rowval[k] for k in range(self.colptr[i], self.colptr[i + 1])]

    def inedges(self, i):
        """list of (src, dst, id) for edges src->i."""
        return [(self.rowval[k], i, k) for k in range(self.colptr[i], self.colptr[i + 1])]

    def outedges(self, i):
        return [(i, self.rowval[k], self.rev[k]) for k in range(self.colptr[i], self.colptr[i + 1])]

    def edges(self):
        out = []
        for j in range(self.N):
            for k in range(self.colptr[j], self.colptr[j + 1]):
                out.append((self.rowval[k], j, k))
        return out


class InfiniteRegularGraph:
    """infinite_graph.jl:8-20: one node, one stored message, k aliases of edge (1,1,1)."""

    def __init__(self, k):
        self.k = int(k)

    def nv(self):
        return 1

    def ne(self):
        return 1

    def vertices(self):
        return range(1)

    def edges(self):
        return [(0, 0, 0) for _ in range(self.k)]

    def inedges(self, i):
        assert i == 0
        return self.edges()

    def outedges(self, i):
        return self.inedges(i)

    def neighbors(self, i):
        return [0] * self.k


class InfiniteBipartiteRegularGraph:
    """infinite_graph.jl:68-91."""

    def __init__(self, k):
        self.k = (int(k[0]), int(k[1]))

    def nv(self):
        return 2

    def ne(self):
        return 2

    def vertices(self):
        return range(2)

    def inedges(self, i):
        return [(1 - i, i, i) for _ in range(self.k[i])]

    def outedges(self, i):
        return [(i, 1 - i, 1 - i) for _ in range(self.k[i])]

    def edges(self):
        return self.inedges(0) + self.inedges(1)


# ----------------------------------------------------------------------------------- MPEM3

class MPEM3:
    """mpems.jl:36-48: cores ``B[t][m,n,x_i^t,x_j^t,x_i^{t+1}]``."""

    def __init__(self, tensors, logz=0.0):
        self.tensors = tensors
        self.logz = float(logz)
        assert tensors[0].shape[0] == 1 and tensors[-1].shape[1] == 1

    def __len__(self):
        return len(self.tensors)

    def __getitem__(self, t):
        return self.tensors[t]


def evaluate_mpem3(B: MPEM3, x):
    """mpems.jl:56-64 (x[t] = (xi, xj), 0-based)."""
    M = np.ones((1, 1))
    L = len(B)
    for t in range(L - 1):
        M = M @ B[t][:, :, x[t][0], x[t][1], x[t + 1][0]]
    M = M @ B[L - 1][:, :, x[L - 1][0], x[L - 1][1], 0]
    return float(M[0, 0]) * np.exp(-B.logz)


def mpem2(B: MPEM3) -> TensorTrain:
    """mpems.jl:67-94: left->right sweep of un-truncated SVDs, ``M[(xi,xj,m),(n,xi')]``."""
    L = len(B)
    qi, qj, qi1 = B[0].shape[2], B[0].shape[3], B[0].shape[4]
    C = [None] * L
    logc = 0.0
    B0 = B[0]
    M = np.transpose(B0, (2, 3, 0, 1, 4)).reshape(qi * qj * B0.shape[0], B0.shape[1] * qi1, order="F")
    Bnew = np.ones((1, 1, 1, 1, 1))
    for t in range(L - 1):
        mt = np.max(np.abs(M))
        if np.isfinite(mt) and mt != 0:
            M = M / mt
            logc += np.log(mt)
        U, lam, Vt = np.linalg.svd(M, full_matrices=False)
        m = len(lam)
        Ct = np.transpose(U.reshape(qi, qj, -1, m, order="F"), (2, 3, 0, 1))
        C[t] = Ct
        Vtr = Vt.reshape(m, -1, qi1, order="F")                       # Vt[m, n, xi']
        Bn = B[t + 1]
        Bnew = np.einsum("m,mlx,lnxyz->mnxyz", lam, Vtr, Bn, optimize=True)
        M = np.transpose(Bnew, (2, 3, 0, 1, 4)).reshape(qi * qj * Bnew.shape[0], Bnew.shape[1] * qi1, order="F")
    C[L - 1] = Bnew[:, :, :, :, 0]
    return TensorTrain(C, B.logz - logc)


def marginalize(A: TensorTrain) -> TensorTrain:
    """mpems.jl:27-29"""
    return TensorTrain([a.sum(axis=3) for a in A], A.logz)


# ---------------------------------------------------------------------------------- cavity

def cavity(source, op, init):
    """CavityTools.cavity: ``dest[j]`` = ordered op-reduction of all sources but j closed with
    ``init``; returns (dest, full).  3z-2 calls of ``op`` in this exact order."""
    z = len(source)
    if z == 0:
        return [], init
    if z == 1:
        return [init], op(source[0], init)
    dest = [source[0]]
    for k in range(1, z):
        dest.append(op(dest[-1], source[k]))
    full = op(dest[-1], init)
    right = init
    for i in range(z - 1, 0, -1):
        dest[i] = op(dest[i - 1], right)
        right = op(source[i], right)
    dest[0] = right
    return dest, full


# ------------------------------------------------------------------------------------ MPBP

class MPBP:
    """mpbp.jl:1-33.  ``w[i][t]`` factors, ``phi[i][t][x]``, ``psi[e][t][xi,xj]``, ``mu[e]`` MPEM2
    (cores ``[m,n,x_src,x_dst]``), ``b[i]`` MPEM1, ``f[i]``."""

    def __init__(self, g, w, phi, psi, mu, b, f):
        self.g, self.w, self.phi, self.psi, self.mu, self.b, self.f = g, w, phi, psi, mu, b, f
        T = len(w[0]) - 1
        assert len(w) == len(phi) == len(b) == len(f) == g.nv()
        assert len(psi) == g.ne() and len(mu) == g.ne()
        assert all(len(wi) == T + 1 for wi in w)
        assert all(len(p) == T + 1 for p in phi) and all(len(p) == T + 1 for p in psi)

    @property
    def T(self):
        return len(self.b[0]) - 1

    def nstates(self, i):
        return self.b[i][0].shape[2]


def flat_mpem2(q1, q2, T, d=1, bondsizes=None):
    bondsizes = [1] + [d] * T + [1] if bondsizes is None else bondsizes
    return flat_tt(bondsizes, q1, q2)


def flat_mpem1(q, T, d=1, bondsizes=None):
    bondsizes = [1] + [d] * T + [1] if bondsizes is None else bondsizes
    return flat_tt(bondsizes, q)


def mpbp(g, w, q, T, d=1, phi=None, psi=None):
    """mpbp.jl:60-70"""
    N = g.nv()
    phi = [[np.ones(q[i]) for _ in range(T + 1)] for i in range(N)] if phi is None else phi
    psi = [[np.ones((q[i], q[j])) for _ in range(T + 1)] for (i, j, _) in g.edges()] if psi is None else psi
    mu = [flat_mpem2(q[i], q[j], T, d=d) for (i, j, _) in g.edges()]
    b = [flat_mpem1(q[i], T, d=d) for i in range(N)]
    return MPBP(g, w, phi, psi, mu, b, np.zeros(N))


def mpbp_infinite_graph(k, wi, qi, phi_i=None, psi=None, d=1):
    """infinite_graph.jl:22-35"""
    T = len(wi) - 1
    phi_i = [np.ones(qi) for _ in range(T + 1)] if phi_i is None else phi_i
    psi = [np.ones((qi, qi)) for _ in range(T + 1)] if psi is None else psi
    g = InfiniteRegularGraph(k)
    return MPBP(g, [wi], [phi_i], [psi], [flat_mpem2(qi, qi, T, d=d)], [flat_mpem1(qi, T, d=d)], np.zeros(1))


def mpbp_infinite_bipartite_graph(k, w, qi, phi=None, psi=None, d=(1, 1)):
    """infinite_graph.jl:93-108"""
    T = len(w[0]) - 1
    phi = [[np.ones(qi[i]) for _ in range(T + 1)] for i in range(2)] if phi is None else phi
    psi = [[np.ones((qi[i], qi[1 - i])) for _ in range(T + 1)] for i in range(2)] if psi is None else psi
    g = InfiniteBipartiteRegularGraph(k)
    mu = [flat_mpem2(qi[i], qi[1 - i], T, d=d[i]) for i in range(2)]
    b = [flat_mpem1(qi[i], T, d=d[i]) for i in range(2)]
    return MPBP(g, w, phi, psi, mu, b, np.zeros(2))


def reset_messages(bp):
    """mpbp.jl:72-80"""
    for A in bp.mu:
        for t in range(len(A)):
            A[t] = np.ones_like(A[t])
        normalize(A)


# ----------------------------------------------------------- recursive update (hot path)

def _f_bp_partial(A: TensorTrain, wi, phi_i, d, prob_name, qj, j) -> MPEM3:
    """recursive_bp_factor.jl:73-87.  ``A[t][m,n,y,x_i]`` -> ``B[t][m,n,x_i,x_j,x_i']``."""
    q = len(phi_i[0])
    L = len(A)
    B = []
    for t in range(L - 1):
        At = A[t]
        ny = At.shape[2]
        prob = getattr(wi[t], prob_name)
        W = np.zeros((q, q, qj, ny))
        for xn in range(q):
            for x in range(q):
                for xj in range(qj):
                    for y in range(ny):
                        W[xn, x, xj, y] = prob(xn + 1, x + 1, xj + 1, y + 1, d, j) * phi_i[t][x]
        B.append(np.einsum("pxjy,mnyx->mnxjp", W, At))
    AT = A[L - 1]
    BT = np.einsum("mnyx,x->mnx", AT, phi_i[L - 1])
    B.append(np.broadcast_to(BT[:, :, :, None, None], BT.shape + (qj, q)).copy())
    return MPEM3(B, A.logz)


def f_bp_partial_ij(A, wi, phi_i, d, qj, j):
    """recursive_bp_factor.jl:64-66"""
    return _f_bp_partial(A, wi, phi_i, d, "prob_y_partial", qj, j)


def f_bp_partial_i(A, wi, phi_i, d):
    """recursive_bp_factor.jl:69-71"""
    return _f_bp_partial(A, wi, phi_i, d, "prob_y_dummy", 1, 1)


def op_kron_compress(wi, a, b, T, svd_trunc):
    """The `op` closure of compute_prob_ys (recursive_bp_factor.jl:118-131): Kronecker product of two
    ỹ-messages through prob_yy, compress!, normalize_eachmatrix!.  `a`, `b` are (train, d) pairs."""
    B1, d1 = a
    B2, d2 = b
    cores = []
    for t in range(T + 1):
        w = wi[t]
        b1, b2 = B1[t], B2[t]
        ny = w.nstates(d1 + d2)
        Pyy = np.zeros((ny, b1.shape[2], b2.shape[2], b1.shape[3]))
        for y in range(ny):
            for y1 in range(b1.shape[2]):
                for y2 in range(b2.shape[2]):
                    for xi in range(b1.shape[3]):
                        Pyy[y, y1, y2, xi] = w.prob_yy(y + 1, y1 + 1, y2 + 1, xi + 1, d1, d2)
        B3 = np.einsum("yabx,ipax,jqbx->ijpqyx", Pyy, b1, b2, optimize=True)       # [m1,m2,n1,n2,y,x]
        s = B3.shape
        cores.append(B3.reshape(s[0] * s[1], s[2] * s[3], s[4], s[5], order="F"))
    Bout = TensorTrain(cores, B1.logz + B2.logz)
    compress(Bout, svd_trunc)
    normalize_eachmatrix(Bout)
    return Bout, d1 + d2


def prob_xy_apply(wi, qi, mu_k, psi_k, k, T):
    """One ỹ-message B_k of compute_prob_ys (recursive_bp_factor.jl:108-115), k 0-based."""
    cores = []
    for t in range(T + 1):
        w = wi[t]
        mk = mu_k[t]
        ny1 = w.nstates(1)
        qk = mk.shape[2]
        Pxy = np.zeros((ny1, qk, qi))
        for y in range(ny1):
            for xk in range(qk):
                for xi in range(qi):
                    Pxy[y, xk, xi] = w.prob_xy(y + 1, xk + 1, xi + 1, k + 1) * psi_k[t][xi, xk]
        cores.append(np.einsum("ykx,mnkx->mnyx", Pxy, mk))
    return TensorTrain(cores, mu_k.logz), 1


def init_train(wi, qi, T):
    """recursive_bp_factor.jl:133-138"""
    Minit = [np.array([[float(wi[t].prob_y0(y + 1, xi + 1)) for xi in range(qi)]
                       for y in range(wi[t].nstates(0))]).reshape(1, 1, wi[t].nstates(0), qi)
             for t in range(T + 1)]
    return TensorTrain(Minit), 0


def compute_prob_ys(wi, qi, mu_in, psi_out, T, svd_trunc):
    """recursive_bp_factor.jl:104-143.  Returns (C, full): cavity ỹ-messages and the full one."""
    B = []
    for k in range(len(psi_out)):
        cores = []
        for t in range(T + 1):
            w = wi[t]
            mk = mu_in[k][t]
            ny1 = w.nstates(1)
            qk = mk.shape[2]
            Pxy = np.zeros((ny1, qk, qi))
            for y in range(ny1):
                for xk in range(qk):
                    for xi in range(qi):
                        Pxy[y, xk, xi] = w.prob_xy(y + 1, xk + 1, xi + 1, k + 1) * psi_out[k][t][xi, xk]
            cores.append(np.einsum("ykx,mnkx->mnyx", Pxy, mk))
        B.append((TensorTrain(cores, mu_in[k].logz), 1))

    def op(a, b):
        return op_kron_compress(wi, a, b, T, svd_trunc)

    Minit = [np.array([[float(wi[t].prob_y0(y + 1, xi + 1)) for xi in range(qi)]
                       for y in range(wi[t].nstates(0))]).reshape(1, 1, wi[t].nstates(0), qi)
             for t in range(T + 1)]
    init = (TensorTrain(Minit), 0)
    dest, full = cavity(B, op, init)
    C = [dd[0] for dd in dest]
    return C, full[0]


def set_msg(bp, mu_j, edge_id, damp, svd_trunc):
    """recursive_bp_factor.jl:168-179"""
    assert 0 <= damp < 1
    mu_old = bp.mu[edge_id]
    logz = normalize(mu_j)
    if damp > 0:
        mu_j = compose_sum(mu_j, mu_old, damp / (1 - damp))
        compress(mu_j, svd_trunc)
        normalize(mu_j)
    bp.mu[edge_id] = mu_j
    return logz


def onebpiter_recursive(bp, i, svd_trunc, damp=0.0):
    """recursive_bp_factor.jl:146-165"""
    g = bp.g
    ein, eout = g.inedges(i), g.outedges(i)
    wi, phi_i, di = bp.w[i], bp.phi[i], len(ein)
    C, full = compute_prob_ys(wi, bp.nstates(i), [bp.mu[e[2]] for e in ein], [bp.psi[e[2]] for e in eout],
                              bp.T, svd_trunc)
    sumlogz = 0.0
    for j, e in enumerate(eout):
        B = f_bp_partial_ij(C[j], wi, phi_i, di - 1, bp.nstates(e[1]), j + 1)
        mu_j = compress(mpem2(B), svd_trunc, is_orthogonal="left")
        normalize_eachmatrix(mu_j)
        sumlogz += set_msg(bp, mu_j, e[2], damp, svd_trunc)
    B = f_bp_partial_i(full, wi, phi_i, di)
    bp.b[i] = marginalize(mpem2(B))
    logzi = normalize(bp.b[i])
    bp.f[i] = (di / 2 - 1) * logzi - 0.5 * sumlogz


# --------------------------------------------------- generic (exhaustive trace) update

def f_bp(A, wi, phi_i, psi_ni, j_index, periodic=False):
    """bp_core.jl:18-57 (j_index 0-based)."""
    T = len(A[0]) - 1
    q = len(phi_i[0])
    qj = psi_ni[j_index][0].shape[1]
    notj = [k for k in range(len(A)) if k != j_index]
    xin = list(itertools.product(*[range(psi_ni[k][0].shape[1]) for k in notj]))
    B = []
    for t in range(T + 1):
        rows = int(np.prod([A[k][t].shape[0] for k in notj])) if notj else 1
        cols = int(np.prod([A[k][t].shape[1] for k in notj])) if notj else 1
        Bt = np.zeros((rows, cols, q, qj, q))
        for xi in range(q):
            for xn in xin:
                At = np.ones((1, 1))
                for k, xk in zip(notj, xn):
                    At = np.kron(At, A[k][t][:, :, xk, xi] * psi_ni[k][t][xi, xk])
                for xj in range(qj):
                    for xnext in range(q):
                        w = phi_i[t][xi]
                        if t < T or periodic:
                            xs = list(xn[:j_index]) + [xj] + list(xn[j_index:])
                            w = w * wi[t](xnext + 1, [v + 1 for v in xs], xi + 1)
                        if w != 0:
                            Bt[:, :, xi, xj, xnext] += At * w
        B.append(Bt)
    return MPEM3(B, sum(A[k].logz for k in notj)), 0.0


def f_bp_dummy_neighbor(A, wi, phi_i, psi_ni, periodic=False):
    """bp_core.jl:60-93"""
    q = len(phi_i[0])
    T = len(phi_i) - 1
    ks = list(range(len(A)))
    xin = list(itertools.product(*[range(psi_ni[k][0].shape[1]) for k in ks]))
    B = []
    for t in range(T + 1):
        rows = int(np.prod([A[k][t].shape[0] for k in ks])) if ks else 1
        cols = int(np.prod([A[k][t].shape[1] for k in ks])) if ks else 1
        Bt = np.zeros((rows, cols, q, 1, q))
        for xi in range(q):
            for xn in xin:
                At = np.ones((1, 1))
                for k, xk in zip(ks, xn):
                    At = np.kron(At, A[k][t][:, :, xk, xi] * psi_ni[k][t][xi, xk])
                for xnext in range(q):
                    w = phi_i[t][xi]
                    if t < T or periodic:
                        w = w * wi[t](xnext + 1, [v + 1 for v in xn], xi + 1)
                    if w != 0:
                        Bt[:, :, xi, 0, xnext] += At * w
        B.append(Bt)
    return MPEM3(B, sum(A[k].logz for k in ks)), 0.0


def onebpiter_generic(bp, i, svd_trunc, damp=0.0):
    """mpbp.jl:117-138 (+ dummy neighbour :145-154)"""
    g = bp.g
    ein, eout = g.inedges(i), g.outedges(i)
    A = [bp.mu[e[2]] for e in ein]
    psi_out = [bp.psi[e[2]] for e in eout]
    sumlogz = 0.0
    new = []
    for j_ind, e in enumerate(eout):
        B, lz = f_bp(A, bp.w[i], bp.phi[i], psi_out, j_ind)
        sumlogz += lz
        mu_j = compress(mpem2(B), svd_trunc, is_orthogonal="left")
        sumlogz += normalize(mu_j)
        new.append((e[2], mu_j))
        bp.mu[e[2]] = mu_j
    di = len(ein)
    B, _ = f_bp_dummy_neighbor(A, bp.w[i], bp.phi[i], psi_out)
    bi = compress(mpem2(B), svd_trunc, is_orthogonal="left")
    bp.b[i] = marginalize(bi)
    logzi = normalization_log(bp.b[i])
    bp.f[i] = (di / 2 - 1) * logzi - 0.5 * sumlogz


def onebpiter(bp, i, svd_trunc=None, damp=0.0):
    """Dispatch on the factor type as mpbp.jl:191 does."""
    svd_trunc = TruncThresh(1e-6) if svd_trunc is None else svd_trunc
    if isinstance(bp.w[i][0], RecursiveBPFactor):
        onebpiter_recursive(bp, i, svd_trunc, damp)
    else:
        onebpiter_generic(bp, i, svd_trunc, damp)


# ------------------------------------------------------------------------ driver, observables

def beliefs(bp):
    """mpbp.jl:237"""
    return [marginals(b) for b in bp.b]


def means(f, bp):
    """mpbp.jl:257-261: expectation of ``f(x, i)`` (x is the 1-based state)."""
    out = []
    for i, b in enumerate(bp.b):
        out.append([sum(f(x + 1, i) * p[x] for x in range(len(p))) for p in marginals(b)])
    return out


class CB_BP:
    """mpbp.jl:157-183 without the progress bar."""

    def __init__(self, bp, f=lambda x, i: x):
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


def iterate(bp, maxiter=5, svd_trunc=None, cb=None, tol=1e-10, nodes=None, shuffle_nodes=True, damp=0.0,
            rng=None, jacobi=False):
    """mpbp.jl:185-198.  ``jacobi=True`` is NOT in the reference: it updates every node of the
    sweep from a snapshot of the messages (the schedule the GPU path uses), so that GPU sweeps can
    be compared with the oracle sweep by sweep."""
    svd_trunc = TruncThresh(1e-6) if svd_trunc is None else svd_trunc
    cb = CB_BP(bp) if cb is None else cb
    nodes = list(bp.g.vertices()) if nodes is None else list(nodes)
    rng = np.random.default_rng(0) if rng is None else rng
    for it in range(1, maxiter + 1):
        if jacobi:
            snap = [m.copy() for m in bp.mu]
            new = {}
            for i in nodes:
                saved = bp.mu
                bp.mu = [m.copy() for m in snap]
                onebpiter(bp, i, svd_trunc, damp)
                for e in bp.g.outedges(i):
                    new[e[2]] = bp.mu[e[2]]
                bp.mu = saved
            for e, m in new.items():
                bp.mu[e] = m
        else:
            for i in nodes:
                onebpiter(bp, i, svd_trunc, damp)
        d = cb(bp, it, svd_trunc)
        if d < tol:
            return it, cb
        if shuffle_nodes:
            nodes = list(rng.permutation(list(bp.g.vertices())))
    return maxiter, cb


def pair_belief_as_mpem(Aij, Aji, psi_ij):
    """bp_core.jl:95-101"""
    cores = []
    for a, b, p in zip(Aij, Aji, psi_ij):
        c = np.einsum("acxy,bdyx,xy->abcdxy", a, b, p)
        s = c.shape
        cores.append(c.reshape(s[0] * s[1], s[2] * s[3], s[4], s[5], order="F"))
    return TensorTrain(cores)


def pair_belief(Aij, Aji, psi_ij):
    """bp_core.jl:105-109: (marginals, z_ij)"""
    A = pair_belief_as_mpem(Aij, Aji, psi_ij)
    _, logz = accumulate_L(A)
    return marginals(A), np.exp(logz)


def pair_beliefs(bp):
    """mpbp.jl:202-205,218-235 and the infinite-graph overloads infinite_graph.jl:37-43,110-116."""
    g = bp.g
    if isinstance(g, InfiniteRegularGraph):
        b, z = pair_belief(bp.mu[0], bp.mu[0], bp.psi[0])
        return [b], np.array([(1 / (g.k - 1) - 0.5) * np.log(z)])
    if isinstance(g, InfiniteBipartiteRegularGraph):
        out, logz = [None, None], np.zeros(2)
        for i in range(2):
            out[i], z = pair_belief(bp.mu[i], bp.mu[1 - i], bp.psi[i])
            logz[i] = (1 / (g.k[i] - 1) - 0.5) * np.log(z)
        return out, logz
    b = [None] * g.ne()
    logz = np.zeros(g.nv())
    for j in range(g.N):
        dj = g.colptr[j + 1] - g.colptr[j]
        for k in range(g.colptr[j], g.colptr[j + 1]):
            ij, ji = k, g.rev[k]
            bij, zij = pair_belief(bp.mu[ij], bp.mu[ji], bp.psi[ij])
            logz[j] += (1 / dj - 0.5) * np.log(zij)
            b[ij] = bij
    return b, logz


def bethe_free_energy(bp):
    """mpbp.jl:298; bipartite reweighting infinite_graph.jl:118-122"""
    if isinstance(bp.g, InfiniteBipartiteRegularGraph):
        k = bp.g.k
        return (bp.f[0] * k[1] + bp.f[1] * k[0]) / (k[0] + k[1])
    return float(np.sum(bp.f))


def autocorrelations(f, bp, maxdist=None):
    """mpbp.jl:245-255"""
    out = []
    for i, b in enumerate(bp.b):
        tv = twovar_marginals(b, maxdist)
        L = len(b)
        r = np.zeros((L, L))
        for t in range(L):
            for u in range(t + 1, L):
                if tv[t][u] is not None:
                    p = tv[t][u]
                    r[t, u] = sum(f(x + 1, i) * f(y + 1, i) * p[x, y] for x in range(p.shape[0]) for y in range(p.shape[1]))
        out.append(r)
    return out


def logprob(bp, X):
    """mpbp.jl:301-324 (X[i,t] 0-based states)."""
    g = bp.g
    N, T = g.nv(), bp.T
    lp = 0.0
    for i in range(N):
        lp += np.log(bp.phi[i][0][X[i, 0]])
    for t in range(T):
        for i in range(N):
            nb = g.neighbors(i)
            lp += np.log(bp.w[i][t](X[i, t + 1] + 1, [X[k, t] + 1 for k in nb], X[i, t] + 1))
            lp += np.log(bp.phi[i][t + 1][X[i, t + 1]])
    for t in range(T + 1):
        for (i, j, ij) in g.edges():
            lp += 0.5 * np.log(bp.psi[ij][t][X[i, t], X[j, t]])
    return lp
