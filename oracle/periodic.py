"""TEST INFRASTRUCTURE (CPU oracle): chains periodic in time, restated from the reference.

In tree (restated line by line):
  PeriodicMPEM3 + evaluate            src/mpems.jl:96-122
  mpem2(::PeriodicMPEM3)              src/mpems.jl:124-155   (the last step folds lambda V^T into the FIRST core)
  _f_bp_partial(::PeriodicMPEM2, ...) src/recursive_bp_factor.jl:89-101   (every site, the last one included, carries W)
  periodic_mpbp                       src/mpbp.jl:399-409
  onebpiter! / set_msg! / pair_belief are the generic ones (src/recursive_bp_factor.jl:146-179, src/bp_core.jl:95-109)
NOT in tree (TensorTrains.jl 0.12, `PeriodicTensorTrain`): compress!, orthogonalize_right!/left!, normalize!,
  normalization, marginals, _compose.  [R] = restated by analogy, NOT from source: the sweeps are taken to be CYCLIC like
  the in-tree periodic mpem2 (src/mpems.jl:134-152: every site is split by an SVD, the carry of the last one is folded
  into the FIRST core), so that the boundary bond is truncated like every other one - an open-chain sweep over a train
  with a boundary bond would never truncate it, and the bonds of the reference's own loopy periodic test
  (test/periodic.jl:87, TruncBond(10)) would grow without bound (measured here: 2, 8, 80, ... per sweep).  Normalisation and
  marginals close the product with a trace.  The lossless regime of all of these is pinned by enumeration
  (tests/test_oracle.py::test_periodic_oracle_*); which subspaces a TRUNCATING sweep keeps on a ring is what cannot be
  pinned from the reference tree - the blocker is `TensorTrains.orthogonalize_right!/left!(::PeriodicTensorTrain)`
  (call sites src/recursive_bp_factor.jl:127,156,174) - see DESIGN.md section 7.
Only tests/ may import this module."""
from __future__ import annotations

import numpy as np

from . import mpbp as O
from .tensor_trains import TensorTrain, TruncThresh, _rescale, _reshape1, _reshapeas, normalize_eachmatrix


def _summed(a):
    return _reshape1(a).sum(axis=2)


def orthogonalize_right(C, svd_trunc):
    """[R] cyclic right sweep: t = L..1, M[m,(n,x)] = core t, SVD, core t <- V^T, carry U diag(lambda) into core t-1 -
    for t = 1 into the LAST core (its right bond is the boundary bond)."""
    L = len(C)
    logc = 0.0
    for t in range(L - 1, -1, -1):
        Ct = _reshape1(C[t])
        q = Ct.shape[2]
        M, logc = _rescale(Ct.reshape(Ct.shape[0], -1, order="F"), logc)
        U, lam, V = svd_trunc(M)
        k = len(lam)
        C[t] = _reshapeas(V.T.reshape(k, -1, q, order="F"), C[t])
        tp = (t - 1) % L
        Cm = _reshape1(C[tp])
        C[tp] = _reshapeas(np.transpose(np.tensordot(Cm, U * lam, axes=([1], [0])), (0, 2, 1)), C[tp])
    C.logz -= logc
    return C


def orthogonalize_left(C, svd_trunc):
    """[R] cyclic left sweep, the pattern of the in-tree periodic mpem2: t = 1..L, M[(m,x),n], core t <- U, carry
    diag(lambda) V^T into core t+1 - for t = L into the FIRST core."""
    L = len(C)
    logc = 0.0
    for t in range(L):
        Ct = _reshape1(C[t])
        q = Ct.shape[2]
        M, logc = _rescale(np.transpose(Ct, (0, 2, 1)).reshape(-1, Ct.shape[1], order="F"), logc)
        U, lam, V = svd_trunc(M)
        k = len(lam)
        C[t] = _reshapeas(np.transpose(U.reshape(-1, q, k, order="F"), (0, 2, 1)), C[t])
        tn = (t + 1) % L
        C[tn] = _reshapeas(np.tensordot((V * lam).T, _reshape1(C[tn]), axes=([1], [0])), C[tn])
    C.logz -= logc
    return C


def compress(A, svd_trunc, is_orthogonal="none"):
    """compress! on a periodic train: the same composition as for open chains (tensor_trains.compress)."""
    if is_orthogonal == "none":
        orthogonalize_right(A, TruncThresh(0.0))
        orthogonalize_left(A, svd_trunc)
    elif is_orthogonal == "left":
        orthogonalize_right(A, svd_trunc)
    else:
        orthogonalize_left(A, svd_trunc)
    return A


def _chain_products(A):
    """prefix[t] = prod_{s<t} S_s, suffix[t] = prod_{s>t} S_s of the x-summed cores, with their log scales."""
    L = len(A)
    S = [_summed(a) for a in A]
    d0 = S[0].shape[0]
    pre, lpre = [np.eye(d0)], [0.0]
    for t in range(L):
        m = pre[-1] @ S[t]
        s = np.max(np.abs(m))
        s = s if (np.isfinite(s) and s != 0) else 1.0
        pre.append(m / s)
        lpre.append(lpre[-1] + np.log(s))
    suf, lsuf = [None] * (L + 1), [0.0] * (L + 1)
    suf[L] = np.eye(d0)
    for t in range(L - 1, -1, -1):
        m = S[t] @ suf[t + 1]
        s = np.max(np.abs(m))
        s = s if (np.isfinite(s) and s != 0) else 1.0
        suf[t] = m / s
        lsuf[t] = lsuf[t + 1] + np.log(s)
    return pre, lpre, suf, lsuf


def log_trace(A) -> float:
    """log sum_x tr prod_t A[t][:, :, x_t]  (z not included)."""
    pre, lpre, _, _ = _chain_products(A)
    tr = float(np.trace(pre[-1]))
    return np.log(tr) + lpre[-1] if tr > 0 else -np.inf


def normalization_log(A) -> float:
    return log_trace(A) - A.logz


def normalize(A) -> float:
    """[R] normalize!: sum_x A(x) = 1, z = 1; returns the log of the previous normalisation."""
    logZ = normalization_log(A)
    f = np.exp(-log_trace(A) / len(A))
    for t in range(len(A)):
        A[t] = A[t] * f
    A.logz = 0.0
    return logZ


def marginals(A):
    """[R] p_t(x) ∝ tr( prod_{s<t} S_s  A[t][:, :, x]  prod_{s>t} S_s )."""
    pre, _, suf, _ = _chain_products(A)
    out = []
    for t in range(len(A)):
        a = _reshape1(A[t])
        p = np.einsum("am,mnx,na->x", pre[t], a, suf[t + 1])
        out.append((p / p.sum()).reshape(A[t].shape[2:], order="F"))
    return out


def twovar_marginals(A):
    """joint of sites t < u (physical axes of t, then of u), normalised."""
    L = len(A)
    pre, _, suf, _ = _chain_products(A)
    S = [_summed(a) for a in A]
    out = [[None] * L for _ in range(L)]
    for t in range(L):
        a = _reshape1(A[t])
        mid = np.einsum("am,mnx->xan", pre[t], a)
        for u in range(t + 1, L):
            b = _reshape1(A[u])
            p = np.einsum("xam,mny,na->xy", mid, b, suf[u + 1])
            out[t][u] = (p / p.sum()).reshape(tuple(A[t].shape[2:]) + tuple(A[u].shape[2:]), order="F")
            mid = np.einsum("xam,mn->xan", mid, S[u])
            mid = mid / max(np.max(np.abs(mid)), 1e-300)
    return out


def evaluate(A, x) -> float:
    M = np.eye(A[0].shape[0])
    for t in range(len(A)):
        xt = x[t] if isinstance(x[t], (tuple, list)) else (x[t],)
        M = M @ A[t][(slice(None), slice(None)) + tuple(xt)]
    return float(np.trace(M)) * np.exp(-A.logz)


def compose_sum(A, B, cB):
    """[R] _compose(x -> x cB, A, B) = A + cB B: block diagonal at EVERY site (the boundary bond is a bond like the others)."""
    L = len(A)
    sa, sb = np.exp(-A.logz / L), np.exp(-B.logz / L)
    cores = []
    for t in range(L):
        a, b = _reshape1(A[t]) * sa, _reshape1(B[t]) * sb * (cB if t == 0 else 1.0)
        c = np.zeros((a.shape[0] + b.shape[0], a.shape[1] + b.shape[1], a.shape[2]))
        c[: a.shape[0], : a.shape[1]] = a
        c[a.shape[0]:, a.shape[1]:] = b
        cores.append(_reshapeas(c, A[t]))
    return TensorTrain(cores, 0.0)


class PeriodicMPEM3:
    """src/mpems.jl:96-108"""

    def __init__(self, tensors, logz=0.0):
        assert tensors[0].shape[0] == tensors[-1].shape[1]
        assert all(a.shape[2] == a.shape[4] for a in tensors)
        self.tensors, self.logz = tensors, float(logz)

    def __len__(self):
        return len(self.tensors)

    def __getitem__(self, t):
        return self.tensors[t]


def evaluate_mpem3(B: PeriodicMPEM3, x):
    """src/mpems.jl:113-122"""
    L = len(B)
    M = np.eye(B[0].shape[0])
    for t in range(L):
        M = M @ B[t][:, :, x[t][0], x[t][1], x[(t + 1) % L][0]]
    return float(np.trace(M)) * np.exp(-B.logz)


def mpem2(B: PeriodicMPEM3) -> TensorTrain:
    """src/mpems.jl:124-155"""
    L = len(B)
    qi, qj, qi1 = B[0].shape[2], B[0].shape[3], B[0].shape[4]
    C = [None] * L
    logc = 0.0
    B0 = B[0]
    M = np.transpose(B0, (2, 3, 0, 1, 4)).reshape(qi * qj * B0.shape[0], B0.shape[1] * qi1, order="F")
    for t in range(L):
        mt = np.max(np.abs(M))
        if np.isfinite(mt) and mt != 0:
            M = M / mt
            logc += np.log(mt)
        U, lam, Vt = np.linalg.svd(M, full_matrices=False)
        m = len(lam)
        C[t] = np.transpose(U.reshape(qi, qj, -1, m, order="F"), (2, 3, 0, 1))
        Vtr = Vt.reshape(m, -1, qi1, order="F")                       # Vt[m, n, x_i^{t+1}]
        if t < L - 1:
            Bnew = np.einsum("m,mlx,lnxyz->mnxyz", lam, Vtr, B[t + 1], optimize=True)
            M = np.transpose(Bnew, (2, 3, 0, 1, 4)).reshape(qi * qj * Bnew.shape[0], Bnew.shape[1] * qi1, order="F")
        else:
            C[0] = np.einsum("m,mkx,knxy->mnxy", lam, Vtr, C[0], optimize=True)      # x_i^{T+2} = x_i^1
    return TensorTrain(C, B.logz - logc)


def _f_bp_partial(A, wi, phi_i, d, prob_name, qj, j) -> PeriodicMPEM3:
    """src/recursive_bp_factor.jl:89-101"""
    q = len(phi_i[0])
    B = []
    for t in range(len(A)):
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
    return PeriodicMPEM3(B, A.logz)


def flat_periodic(T, d, *q):
    A = TensorTrain([np.ones((d, d) + tuple(q)) for _ in range(T + 1)])
    normalize(A)
    return A


def periodic_mpbp(g, w, q, T, d=1, phi=None, psi=None):
    """src/mpbp.jl:399-409"""
    N = g.nv()
    phi = [[np.ones(q[i]) for _ in range(T + 1)] for i in range(N)] if phi is None else phi
    psi = [[np.ones((q[i], q[j])) for _ in range(T + 1)] for (i, j, _) in g.edges()] if psi is None else psi
    mu = [flat_periodic(T, d, q[i], q[j]) for (i, j, _) in g.edges()]
    b = [flat_periodic(T, d, q[i]) for i in range(N)]
    return O.MPBP(g, w, phi, psi, mu, b, np.zeros(N))


def set_msg(bp, mu_j, edge_id, damp, svd_trunc):
    """src/recursive_bp_factor.jl:168-179"""
    mu_old = bp.mu[edge_id]
    logz = normalize(mu_j)
    if damp > 0:
        mu_j = compose_sum(mu_j, mu_old, damp / (1 - damp))
        compress(mu_j, svd_trunc)
        normalize(mu_j)
    bp.mu[edge_id] = mu_j
    return logz


def op_kron_compress(wi, a, b, T, svd_trunc):
    """The `op` of compute_prob_ys (src/recursive_bp_factor.jl:118-131) on periodic trains: the Kronecker product acts on
    the boundary bond as on any other; compress! is the periodic one [R]."""
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
        B3 = np.einsum("yabx,ipax,jqbx->ijpqyx", Pyy, b1, b2, optimize=True)
        sh = B3.shape
        cores.append(B3.reshape(sh[0] * sh[1], sh[2] * sh[3], sh[4], sh[5], order="F"))
    Bout = TensorTrain(cores, B1.logz + B2.logz)
    compress(Bout, svd_trunc)
    normalize_eachmatrix(Bout)
    return Bout, d1 + d2


def compute_prob_ys(wi, qi, mu_in, psi_out, T, svd_trunc):
    """src/recursive_bp_factor.jl:104-143 with the periodic `op`."""
    B = [O.prob_xy_apply(wi, qi, mu_in[k], psi_out[k], k, T) for k in range(len(psi_out))]
    dest, full = O.cavity(B, lambda a, b: op_kron_compress(wi, a, b, T, svd_trunc), O.init_train(wi, qi, T))
    return [dd[0] for dd in dest], full[0]


def onebpiter(bp, i, svd_trunc, damp=0.0):
    """src/recursive_bp_factor.jl:146-165 on periodic trains."""
    g = bp.g
    ein, eout = g.inedges(i), g.outedges(i)
    wi, phi_i, di = bp.w[i], bp.phi[i], len(ein)
    C, full = compute_prob_ys(wi, bp.nstates(i), [bp.mu[e[2]] for e in ein], [bp.psi[e[2]] for e in eout], bp.T, svd_trunc)
    sumlogz = 0.0
    for j, e in enumerate(eout):
        B = _f_bp_partial(C[j], wi, phi_i, di - 1, "prob_y_partial", bp.nstates(e[1]), j + 1)
        mu_j = compress(mpem2(B), svd_trunc, is_orthogonal="left")
        normalize_eachmatrix(mu_j)
        sumlogz += set_msg(bp, mu_j, e[2], damp, svd_trunc)
    B = _f_bp_partial(full, wi, phi_i, di, "prob_y_dummy", 1, 1)
    bp.b[i] = O.marginalize(mpem2(B))
    logzi = normalize(bp.b[i])
    bp.f[i] = (di / 2 - 1) * logzi - 0.5 * sumlogz


def iterate(bp, maxiter, svd_trunc, damp=0.0, jacobi=False):
    """sweeps in index order (reference iterate! with shuffle_nodes=false); jacobi: all nodes from the messages at entry."""
    for _ in range(maxiter):
        if jacobi:
            old = [m.copy() for m in bp.mu]
            new = {}
            for i in range(bp.g.nv()):
                bp.mu = [m.copy() for m in old]
                onebpiter(bp, i, svd_trunc, damp)
                for e in bp.g.outedges(i):
                    new[e[2]] = bp.mu[e[2]]
            bp.mu = [new.get(k, old[k]) for k in range(len(old))]
        else:
            for i in range(bp.g.nv()):
                onebpiter(bp, i, svd_trunc, damp)


def beliefs(bp):
    return [marginals(b) for b in bp.b]


def pair_beliefs(bp):
    """src/mpbp.jl:202-235 with bp_core.jl:95-109 (the pair train is the generic Kronecker product; its marginals and
    normalisation close with a trace)."""
    g = bp.g
    b = [None] * g.ne()
    logz = np.zeros(g.nv())
    for j in range(g.N):
        dj = g.colptr[j + 1] - g.colptr[j]
        for k in range(g.colptr[j], g.colptr[j + 1]):
            ij, ji = k, g.rev[k]
            A = O.pair_belief_as_mpem(bp.mu[ij], bp.mu[ji], bp.psi[ij])
            logz[j] += (1 / dj - 0.5) * log_trace(A)
            b[ij] = marginals(A)
    return b, logz
