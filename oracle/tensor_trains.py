"""Tensor-train algebra: restatement of TensorTrains.jl v0.12 semantics (TEST INFRASTRUCTURE).

TensorTrains.jl is a registry dependency of the reference (``Project.toml:26,48``), not in
its tree.  Contracts restated here are the ones the reference relies on at its call sites
(``src/recursive_bp_factor.jl:127,128,156,157,162,171-175``, ``src/mpbp.jl:77,129-135,237``,
``src/bp_core.jl:105-109``, ``src/mpems.jl:63,93``), see SURVEY.md Appendix A.

Conventions
-----------
* a core is ``A[t][m, n, x1, x2, ...]`` (bond, bond, physical...), Julia index order kept;
* every reshape that the Julia code writes as a TensorCast group ``(a,b)`` is column-major
  with ``a`` fastest  ->  ``order='F'`` here;
* the represented function is ``prod_t A[t][:,:,x_t] / z`` (``src/mpems.jl:63``), ``z`` is kept
  as ``logz = log z`` (``z > 0`` always on this path: it is a product of max-abs values).
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "TensorTrain", "TruncThresh", "TruncBond", "TruncBondMax", "TruncBondThresh",
    "orthogonalize_right", "orthogonalize_left", "compress", "normalize_eachmatrix",
    "normalization_log", "normalize", "accumulate_L", "accumulate_R", "marginals",
    "twovar_marginals", "compose_sum", "flat_tt", "rand_tt", "evaluate",
]


class TensorTrain:
    """``TensorTrain{F,N}``: list of cores + scalar ``z`` (stored as ``logz``)."""

    def __init__(self, tensors, logz: float = 0.0):
        self.tensors = [np.asarray(a, dtype=np.float64) for a in tensors]
        self.logz = float(logz)
        for a, b in zip(self.tensors[:-1], self.tensors[1:]):
            if a.shape[1] != b.shape[0]:
                raise ValueError("Matrix indices for matrix product non compatible")

    def __len__(self):
        return len(self.tensors)

    def __getitem__(self, t):
        return self.tensors[t]

    def __setitem__(self, t, v):
        self.tensors[t] = v

    def __iter__(self):
        return iter(self.tensors)

    def copy(self):
        return TensorTrain([a.copy() for a in self.tensors], self.logz)

    @property
    def bonds(self):
        return [self.tensors[0].shape[0]] + [a.shape[1] for a in self.tensors]


# ----------------------------------------------------------------------------------------
# SVD truncation functors (TensorTrains.jl ``SVDTrunc``; constructor arities as used in
# reference test/sis_small_tree.jl:21,94,103, test/glauber_small_tree.jl:29-30)
# ----------------------------------------------------------------------------------------

def _svd(M):
    """Thin SVD through LAPACK gesdd (numpy default) - the driver Julia's ``svd`` uses."""
    U, s, Vt = np.linalg.svd(M, full_matrices=False)
    return U, s, Vt.T


def _findlast_above(lam, thr):
    idx = np.nonzero(lam > thr)[0]
    # Julia's findlast would return `nothing` (=> error) when no value passes; keep >=1.
    return int(idx[-1]) + 1 if idx.size else 1


class TruncThresh:
    """Keep singular values ``λ_k > ε·‖λ‖₂``."""

    def __init__(self, eps: float):
        self.eps = float(eps)

    def __call__(self, M):
        U, lam, V = _svd(M)
        k = _findlast_above(lam, self.eps * np.linalg.norm(lam))
        return U[:, :k], lam[:k], V[:, :k]

    def __repr__(self):
        return f"TruncThresh({self.eps})"


class TruncBond:
    """Keep the first ``min(len, mprime)`` singular values."""

    def __init__(self, mprime: int):
        self.mprime = int(mprime)

    def __call__(self, M):
        U, lam, V = _svd(M)
        k = min(len(lam), self.mprime)
        return U[:, :k], lam[:k], V[:, :k]

    def __repr__(self):
        return f"TruncBond({self.mprime})"


class TruncBondMax:
    """As ``TruncBond`` and records the worst relative truncation error seen."""

    def __init__(self, mprime: int):
        self.mprime = int(mprime)
        self.maxerr = 0.0

    def __call__(self, M):
        U, lam, V = _svd(M)
        k = min(len(lam), self.mprime)
        tot = float(np.sum(lam ** 2))
        if tot > 0:
            err = float(np.sqrt(np.sum(lam[k:] ** 2) / tot))
            self.maxerr = max(self.maxerr, err)
        return U[:, :k], lam[:k], V[:, :k]

    def __repr__(self):
        return f"TruncBondMax({self.mprime})"


class TruncBondThresh:
    """``min`` of the bond cap and the threshold rule (``ε`` defaults to 0)."""

    def __init__(self, mprime: int, eps: float = 0.0):
        self.mprime = int(mprime)
        self.eps = float(eps)

    def __call__(self, M):
        U, lam, V = _svd(M)
        k = min(_findlast_above(lam, self.eps * np.linalg.norm(lam)), self.mprime)
        return U[:, :k], lam[:k], V[:, :k]

    def __repr__(self):
        return f"TruncBondThresh({self.mprime}, {self.eps})"


# ----------------------------------------------------------------------------------------
# sweeps
# ----------------------------------------------------------------------------------------

def _reshape1(a):
    return a.reshape(a.shape[0], a.shape[1], -1, order="F")


def _reshapeas(a3, like):
    return a3.reshape((a3.shape[0], a3.shape[1]) + tuple(like.shape[2:]), order="F")


def _rescale(M, logc):
    mt = np.max(np.abs(M)) if M.size else 0.0
    if np.isfinite(mt) and mt != 0.0:
        M = M / mt
        logc += np.log(mt)
    return M, logc


def orthogonalize_right(C: TensorTrain, svd_trunc=TruncThresh(1e-6)) -> TensorTrain:
    """TensorTrains ``orthogonalize_right!``: sweep t = L..2, ``M[m,(n,x)]`` SVD, core[t] <- Vᵀ,
    carry ``core[t-1]·U·diag(λ)``; every step rescales M by its max-abs into ``z``."""
    L = len(C)
    CT = _reshape1(C[L - 1])
    q = CT.shape[2]
    M = CT.reshape(CT.shape[0], -1, order="F")
    D = CT
    logc = 0.0
    for t in range(L - 1, 0, -1):
        M, logc = _rescale(M, logc)
        U, lam, V = svd_trunc(M)
        k = len(lam)
        At = V.T.reshape(k, -1, q, order="F")
        C[t] = _reshapeas(At, C[t])
        Cm = _reshape1(C[t - 1])
        D = np.transpose(np.tensordot(Cm, U * lam, axes=([1], [0])), (0, 2, 1))
        M = D.reshape(D.shape[0], -1, order="F")
    C[0] = _reshapeas(D, C[0])
    C.logz -= logc
    return C


def orthogonalize_left(C: TensorTrain, svd_trunc=TruncThresh(1e-6)) -> TensorTrain:
    """TensorTrains ``orthogonalize_left!``: mirror image, ``M[(m,x),n]`` (m fastest)."""
    L = len(C)
    C0 = _reshape1(C[0])
    q = C0.shape[2]
    M = np.transpose(C0, (0, 2, 1)).reshape(-1, C0.shape[1], order="F")
    D = C0
    logc = 0.0
    for t in range(0, L - 1):
        M, logc = _rescale(M, logc)
        U, lam, V = svd_trunc(M)
        k = len(lam)
        At = np.transpose(U.reshape(-1, q, k, order="F"), (0, 2, 1))
        C[t] = _reshapeas(At, C[t])
        Cp = _reshape1(C[t + 1])
        D = np.tensordot((V * lam).T, Cp, axes=([1], [0]))
        M = np.transpose(D, (0, 2, 1)).reshape(-1, D.shape[1], order="F")
    C[L - 1] = _reshapeas(D, C[L - 1])
    C.logz -= logc
    return C


def compress(A: TensorTrain, svd_trunc=TruncThresh(1e-6), is_orthogonal: str = "none") -> TensorTrain:
    """TensorTrains ``compress!`` (call sites recursive_bp_factor.jl:127,156,174)."""
    if is_orthogonal == "none":
        orthogonalize_right(A, TruncThresh(0.0))
        orthogonalize_left(A, svd_trunc)
    elif is_orthogonal == "left":
        orthogonalize_right(A, svd_trunc)
    elif is_orthogonal == "right":
        orthogonalize_left(A, svd_trunc)
    else:
        raise ValueError("is_orthogonal must be one of none/left/right")
    return A


# ----------------------------------------------------------------------------------------
# normalisations
# ----------------------------------------------------------------------------------------

def normalize_eachmatrix(A: TensorTrain) -> float:
    """Divide every core by its max-abs, fold the product into z; returns log of the product
    (reference call sites recursive_bp_factor.jl:128,157; return value unused there)."""
    logc = 0.0
    for t in range(len(A)):
        mm = np.max(np.abs(A[t])) if A[t].size else 0.0
        if np.isfinite(mm) and mm != 0.0:
            A[t] = A[t] / mm
            logc += np.log(mm)
    A.logz -= logc
    return logc


def _summed(a):
    return _reshape1(a).sum(axis=2)


def accumulate_L(A: TensorTrain):
    """Left environments of the x-summed cores, rescaled; returns ``(l, logz)`` where
    ``logz = log Σ_x ∏_t A[t][:,:,x_t]`` (z NOT included) and ``l[t]`` ∝ ∏_{s<=t}."""
    l = []
    v = np.ones((1, A[0].shape[0])) if A[0].shape[0] == 1 else None
    if v is None:
        raise ValueError("open-chain tensor train expected")
    logz = 0.0
    for t in range(len(A)):
        v = v @ _summed(A[t])
        s = np.max(np.abs(v))
        if np.isfinite(s) and s != 0.0:
            v = v / s
            logz += np.log(s)
        l.append(v.copy())
    logz += np.log(abs(float(v[0, 0]))) if v[0, 0] != 0 else -np.inf
    return l, logz


def accumulate_R(A: TensorTrain):
    r = [None] * len(A)
    v = np.ones((A[len(A) - 1].shape[1], 1))
    logz = 0.0
    for t in range(len(A) - 1, -1, -1):
        v = _summed(A[t]) @ v
        s = np.max(np.abs(v))
        if np.isfinite(s) and s != 0.0:
            v = v / s
            logz += np.log(s)
        r[t] = v.copy()
    logz += np.log(abs(float(v[0, 0]))) if v[0, 0] != 0 else -np.inf
    return r, logz


def normalization_log(A: TensorTrain) -> float:
    """log of ``normalization(A)`` = log( Σ_x A(x) ) with z included."""
    _, logZ = accumulate_L(A)
    return logZ - A.logz


def normalize(A: TensorTrain) -> float:
    """``normalize!``: rescale so that Σ_x A(x) = 1 and z = 1; returns log of the previous
    normalisation as a real float (reference uses it so: recursive_bp_factor.jl:162-163,171)."""
    logZ = normalization_log(A)
    _, logP = accumulate_L(A)
    L = len(A)
    f = np.exp(-logP / L)
    for t in range(L):
        A[t] = A[t] * f
    A.logz = 0.0
    return logZ


def marginals(A: TensorTrain):
    """Per-site marginals ``p_t(x) ∝ L_{t-1}·A[t][:,:,x]·R_{t+1}``, each normalised to 1
    (reference: mpbp.jl:237 ``beliefs``, bp_core.jl:108)."""
    l, _ = accumulate_L(A)
    r, _ = accumulate_R(A)
    out = []
    L = len(A)
    for t in range(L):
        lv = l[t - 1] if t > 0 else np.ones((1, 1))
        rv = r[t + 1] if t < L - 1 else np.ones((1, 1))
        a = _reshape1(A[t])
        p = np.einsum("m,mnx,n->x", lv[0], a, rv[:, 0])
        p = p / p.sum()
        out.append(p.reshape(A[t].shape[2:], order="F"))
    return out


def twovar_marginals(A: TensorTrain, maxdist=None):
    """``[t][u]`` joint of sites t<u (physical axes of t then u), normalised (mpbp.jl:251,272)."""
    L = len(A)
    maxdist = L if maxdist is None else maxdist
    l, _ = accumulate_L(A)
    r, _ = accumulate_R(A)
    out = [[None] * L for _ in range(L)]
    for t in range(L):
        lv = l[t - 1][0] if t > 0 else np.ones(1)
        a = _reshape1(A[t])
        mid = np.einsum("m,mnx->xn", lv, a)          # [x_t, n]
        for u in range(t + 1, min(L, t + maxdist + 1)):
            rv = r[u + 1][:, 0] if u < L - 1 else np.ones(1)
            b = _reshape1(A[u])
            p = np.einsum("xm,mny,n->xy", mid, b, rv)
            p = p / p.sum()
            out[t][u] = p.reshape(tuple(A[t].shape[2:]) + tuple(A[u].shape[2:]), order="F")
            mid = mid @ _summed(A[u])
            s = np.max(np.abs(mid))
            if s > 0:
                mid = mid / s
    return out


def compose_sum(A: TensorTrain, B: TensorTrain, cB: float) -> TensorTrain:
    """``_compose(x->x*cB, A, B)``: train of ``A(x) + cB·B(x)`` by block-diagonal direct sum
    (reference recursive_bp_factor.jl:173).  The result has z = 1."""
    L = len(A)
    assert len(B) == L
    sa = np.exp(-A.logz / L)
    sb = np.exp(-B.logz / L)
    cores = []
    for t in range(L):
        a = _reshape1(A[t]) * sa
        b = _reshape1(B[t]) * sb
        q = a.shape[2]
        if t == 0:
            c = np.concatenate([a, b * cB], axis=1)
        elif t == L - 1:
            c = np.concatenate([a, b], axis=0)
        else:
            c = np.zeros((a.shape[0] + b.shape[0], a.shape[1] + b.shape[1], q))
            c[: a.shape[0], : a.shape[1]] = a
            c[a.shape[0]:, a.shape[1]:] = b
        if L == 1:
            c = a + cB * b
        cores.append(_reshapeas(c, A[t]))
    return TensorTrain(cores, 0.0)


# ----------------------------------------------------------------------------------------
# constructors / evaluation
# ----------------------------------------------------------------------------------------

def flat_tt(bondsizes, *q) -> TensorTrain:
    """Constant cores of the given bond sizes (reference mpems.jl:20, mpbp.jl:66)."""
    cores = [np.ones((bondsizes[t], bondsizes[t + 1]) + tuple(q)) for t in range(len(bondsizes) - 1)]
    A = TensorTrain(cores)
    normalize(A)
    return A


def rand_tt(bondsizes, *q, rng=None) -> TensorTrain:
    rng = np.random.default_rng(0) if rng is None else rng
    cores = [rng.random((bondsizes[t], bondsizes[t + 1]) + tuple(q)) for t in range(len(bondsizes) - 1)]
    return TensorTrain(cores)


def evaluate(A: TensorTrain, x) -> float:
    """``evaluate(A, x)``: x[t] is a tuple of 0-based physical indices of site t."""
    M = np.ones((1, 1))
    for t in range(len(A)):
        xt = x[t] if isinstance(x[t], (tuple, list)) else (x[t],)
        M = M @ A[t][(slice(None), slice(None)) + tuple(xt)]
    return float(M[0, 0]) * np.exp(-A.logz)
