"""CPU restatement of the DEVICE algorithm for `op` (TEST INFRASTRUCTURE; second, stronger CPU baseline).

The HIP engine does not run the reference's two SVD sweeps (``compress!`` = ``orthogonalize_right!(TruncThresh(0))``
then ``orthogonalize_left!(svd_trunc)``, restated in ``oracle/tensor_trains.py``).  It computes the same truncated
function by (``csrc/engine.h``, ``csrc/v2_kernels.h``):

* sweep 1: triangular factors ``Lf_t`` with ``Lf_t Lf_t^T = X_t X_t^T``, ``X_t = A_t (I x Lf_{t+1})``, from an R-only QR
  of ``Y_t = X_t^T`` assembled from the two factor trains (the product core is never formed);
* sweep 2: ``N_t = C_{t-1} A_t``, ``M_t = N_t Lf_{t+1}``, left singular vectors of ``M_t``, the ``SVDTrunc`` rule on its
  singular values, carry ``C_t = U^T N_t``.

This module states exactly that with LAPACK (``geqrf`` / ``gesdd`` through numpy) so that (i) a CPU test pins the claimed
function equivalence against the reference-algorithm oracle with binding truncation, and (ii) ``bench.py`` can time
"the device's own algorithm on the host cores" next to the reference-algorithm port.  It follows the oracle's
conventions: cores ``[m, n, y, x]``, ``(m1, m2)`` groups with ``m1`` fastest, ``logz = log z``.
"""
from __future__ import annotations

import numpy as np

from .tensor_trains import TensorTrain, TruncBond, TruncBondMax, TruncBondThresh, TruncThresh


def _keep(svd_trunc, lam):
    """number of singular values the SVDTrunc functor keeps (the functors of tensor_trains.py, applied to values)."""
    n = len(lam)
    if isinstance(svd_trunc, TruncThresh):
        idx = np.nonzero(lam > svd_trunc.eps * np.linalg.norm(lam))[0]
        return int(idx[-1]) + 1 if idx.size else 1
    if isinstance(svd_trunc, TruncBondThresh):
        idx = np.nonzero(lam > svd_trunc.eps * np.linalg.norm(lam))[0]
        return min(int(idx[-1]) + 1 if idx.size else 1, svd_trunc.mprime)
    if isinstance(svd_trunc, TruncBondMax):
        k = min(n, svd_trunc.mprime)
        tot = float(np.sum(lam ** 2))
        if tot > 0:
            svd_trunc.maxerr = max(svd_trunc.maxerr, float(np.sqrt(np.sum(lam[k:] ** 2) / tot)))
        return k
    if isinstance(svd_trunc, TruncBond):
        return min(n, svd_trunc.mprime)
    raise TypeError(svd_trunc)


def _chol_shifted(G, shift):
    """upper-triangular R with R^T R = G + s I, s = shift * trace-free scale of G; the shift grows tenfold until LAPACK's
    ``potrf`` accepts the matrix (a rank-deficient Gram matrix has non-positive pivots at rounding level)."""
    n = G.shape[0]
    scale = float(np.abs(np.diag(G)).max())
    if not scale > 0:
        return np.zeros_like(G), 0.0
    s = shift * scale
    for _ in range(40):
        try:
            return np.linalg.cholesky(G + s * np.eye(n)).T, s
        except np.linalg.LinAlgError:
            s = max(s * 10.0, 1e-18 * scale)
    raise np.linalg.LinAlgError("shifted Cholesky did not succeed")


def gauge_factor(Y, gauge="householder"):
    """R-only factor of ``Y`` (``R^T R = Y^T Y``) in the forms the round-4 experiment compares
    (``tools/gram_gauge_experiment.py``, ``profiles/r04_gram_gauge_errors.txt``):

    * ``householder`` - LAPACK ``geqrf`` (what the HIP engine's blocked Householder QR is checked against);
    * ``gram`` - single pass: Cholesky of the Gram matrix ``Y^T Y`` (+ the smallest shift ``potrf`` accepts);
    * ``cholqr2`` - shifted CholeskyQR2 (Fukaya, Kannan, Nakatsukasa, Yamamoto, Yanagisawa 2020): ``R1`` from the Gram
      matrix with the shift ``11 (m n + n (n + 1)) u |Y|^2``, ``Q1 = Y R1^-1``, ``R2`` from the Gram matrix of ``Q1``,
      ``R = R2 R1``;  ``cholqr3`` - one more pass on ``Q2 = Q1 R2^-1``.
    """
    if gauge == "householder":
        return np.linalg.qr(Y, mode="r")
    from scipy.linalg import solve_triangular
    m, n = Y.shape
    u = np.finfo(float).eps / 2
    G = Y.T @ Y
    if gauge == "gram":
        return _chol_shifted(G, 0.0)[0]
    R1, _ = _chol_shifted(G, 11.0 * (m * n + n * (n + 1)) * u)
    if not R1.any():
        return R1
    R = R1
    Q = solve_triangular(R1, Y.T, trans="T", lower=False).T               # Q1 = Y R1^-1
    for _ in range({"cholqr2": 1, "cholqr3": 2}[gauge]):
        Rk, _ = _chol_shifted(Q.T @ Q, 0.0)
        R = Rk @ R
        Q = solve_triangular(Rk, Q.T, trans="T", lower=False).T
    return R


def op_kron_compress_qr(wi, a, b, T, svd_trunc, gauge="householder"):
    """Same inputs / outputs as ``oracle.mpbp.op_kron_compress`` (recursive_bp_factor.jl:118-131), device algorithm.
    ``gauge`` selects how the triangular factor of sweep 1 is computed (``gauge_factor``); the device ships
    ``householder``."""
    B1, d1 = a
    B2, d2 = b
    L = T + 1
    q = B1[0].shape[3]
    Pyy = []
    for t in range(L):
        w = wi[t]
        ny, ny1, ny2 = w.nstates(d1 + d2), B1[t].shape[2], B2[t].shape[2]
        P = np.zeros((ny, ny1, ny2, q))
        for y in range(ny):
            for y1 in range(ny1):
                for y2 in range(ny2):
                    for xi in range(q):
                        P[y, y1, y2, xi] = w.prob_yy(y + 1, y1 + 1, y2 + 1, xi + 1, d1, d2)
        Pyy.append(P)
    # ---- sweep 1: Lf_t (B_t x r_t), from the R factor of Y_t = X_t^T;  Lf_L = [[1]]
    Lf = [None] * (L + 1)
    Lf[L] = np.ones((1, 1))
    for t in range(L - 1, 0, -1):
        b1, b2 = B1[t], B2[t]
        a_, an = b1.shape[0], b1.shape[1]
        b_, bn = b2.shape[0], b2.shape[1]
        r1 = Lf[t + 1].shape[1]
        ny, ny1 = Pyy[t].shape[0], b1.shape[2]
        Lf3 = Lf[t + 1].reshape(an, bn, r1, order="F")                       # [(n1, n2), k]
        # Z[m1, y1, xi, n2, k] = sum_n1 A1[m1, n1, y1, xi] Lf[(n1, n2), k]   (one matrix product)
        Z = (np.transpose(b1, (0, 2, 3, 1)).reshape(-1, an) @ Lf3.reshape(an, bn * r1)).reshape(a_, ny1, q, bn, r1)
        Y = np.empty((r1, ny, q, a_, b_))
        for xi in range(q):
            # E[m2, y, n2, y1] = sum_y2 pyy[y, y1, y2, xi] A2[m2, n2, y2, xi];  Y_xi[k, y, m1, m2] = sum_(n2, y1) E Z
            E = np.einsum("yab,MNb->MyNa", Pyy[t][:, :, :, xi], b2[:, :, :, xi])
            Zx = np.transpose(Z[:, :, xi], (2, 1, 0, 3)).reshape(bn * ny1, a_ * r1)          # [(n2, y1), (m1, k)]
            Yx = (E.reshape(b_ * ny, bn * ny1) @ Zx).reshape(b_, ny, a_, r1)                  # [m2, y, m1, k]
            Y[:, :, xi] = np.transpose(Yx, (3, 1, 2, 0))
        Y = Y.reshape(r1 * ny * q, a_ * b_, order="F")
        R = gauge_factor(Y, gauge)
        mx = np.abs(R).max()
        if mx > 0 and np.isfinite(mx):
            R = R / mx
        Lf[t] = R.T.copy()
    # ---- sweep 2
    C = np.ones((1, 1))
    logc = 0.0
    cores = []
    for t in range(L):
        b1, b2 = B1[t], B2[t]
        a_, an = b1.shape[0], b1.shape[1]
        b_, bn = b2.shape[0], b2.shape[1]
        kc = C.shape[0]
        ny = Pyy[t].shape[0]
        ny1 = b1.shape[2]
        C3 = C.reshape(kc, a_, b_, order="F")
        # T1[k, m2, n1, y1, xi] = sum_m1 C[k, (m1, m2)] A1[m1, n1, y1, xi]
        T1 = (np.transpose(C3, (0, 2, 1)).reshape(kc * b_, a_) @ b1.reshape(a_, -1)).reshape(kc, b_, an, ny1, q)
        N = np.empty((kc, ny, q, an, bn))
        for xi in range(q):
            E = np.einsum("yab,MNb->NyMa", Pyy[t][:, :, :, xi], b2[:, :, :, xi])              # [n2, y, m2, y1]
            Tx = np.transpose(T1[:, :, :, :, xi], (1, 3, 0, 2)).reshape(b_ * ny1, kc * an)     # [(m2, y1), (k, n1)]
            Nx = (E.reshape(bn * ny, b_ * ny1) @ Tx).reshape(bn, ny, kc, an)                  # [n2, y, k, n1]
            N[:, :, xi] = np.transpose(Nx, (2, 1, 3, 0))
        N = N.reshape(kc * ny * q, an * bn, order="F")
        mx = np.abs(N).max()
        if mx > 0 and np.isfinite(mx):
            N = N / mx
            logc += np.log(mx)
        if t == L - 1:
            cores.append(N[:, 0].reshape(kc, 1, ny, q, order="F"))
            break
        M = N @ Lf[t + 1]
        U, lam, _ = np.linalg.svd(M, full_matrices=False)
        kp = _keep(svd_trunc, lam)
        U = U[:, :kp]
        cores.append(np.transpose(U.reshape(kc, ny, q, kp, order="F"), (0, 3, 1, 2)))
        C = U.T @ N
    out = TensorTrain(cores, B1.logz + B2.logz - logc)
    # normalize_eachmatrix!
    for t in range(L):
        mm = np.abs(out[t]).max()
        if mm > 0 and np.isfinite(mm):
            out[t] = out[t] / mm
            out.logz -= np.log(mm)
    return out, d1 + d2
