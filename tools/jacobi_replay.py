"""Round-4 question: can a PRECONDITIONER cut the ~10 sweeps per call of the one-sided Jacobi of the truncating sweep
(wg::jacobi_rsv on R1^T, R1 = R-only QR of M_t^T)?  The left singular vectors of M_t are the right singular vectors of R1,
and - up to the known permutation - of R2 = R factor of R1 P for ANY column permutation P (an R-only QR loses a left factor
only), so Jacobi may run on R2^T instead; P = columns by decreasing norm is free, P = the pivot order of QR with column
pivoting is the Drmac-Veselic preconditioner.  This tool replays the kernel's tournament (same pairing, thresholds, null test
and deflation) in numpy on (a) factors of random saturated trains and (b) factors captured from a BP run of the oracle, and
counts sweeps.  Result (profiles/r04_jacobi_preconditioner.txt): on random factors the sweep count halves; on BP factors -
numerical rank ~2/3, singular values over 20 decades, where the deflation of null columns already does the work - it RISES
(8.7 -> 10.8).  Built on the device once (k_precond_perm + a second batched QR), measured slower, removed.
Test infrastructure (imports oracle/).  usage: python tools/jacobi_replay.py [bond=20]"""
import os
import sys

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import networkx as nx  # noqa: E402
import numpy as np  # noqa: E402

import oracle.device_algorithm as DA  # noqa: E402
import oracle.factors as OF  # noqa: E402
import oracle.mpbp as O  # noqa: E402
import oracle.tensor_trains as OT  # noqa: E402


def jacobi_sweeps(X, maxsweeps=60):
    """sweeps of the one-sided cyclic Jacobi of wg::jacobi_rsv on the columns of X (round-robin tournament, rotate when
    cos^2 > 1e-30 and both columns are non-null, converged when the largest cos^2 of a sweep < 1e-16, null columns leave)"""
    A = X.copy()
    nul, tol = 1e-28 * (A * A).sum(), 1e-15
    act = list(range(A.shape[1]))
    for sweep in range(maxsweeps):
        nact = len(act)
        ne = (nact + 1) & ~1
        worst = 0.0
        for rnd in range(ne - 1):
            ps, qs = [], []
            for pi in range(ne // 2):
                p, q = (ne - 1, rnd) if pi == 0 else ((rnd + pi) % (ne - 1), (rnd + ne - 1 - pi) % (ne - 1))
                p, q = min(p, q), max(p, q)
                if q < nact:
                    ps.append(act[p]); qs.append(act[q])
            ps, qs = np.array(ps), np.array(qs)
            x, y = A[:, ps], A[:, qs]
            al, be, ga = (x * x).sum(0), (y * y).sum(0), (x * y).sum(0)
            rot = (ga * ga > (tol * tol) * (al * be)) & (al > nul) & (be > nul)
            if rot.any():
                worst = max(worst, float((ga[rot] ** 2 / (al[rot] * be[rot])).max()))
                d, g2 = be - al, 2 * ga
                t = np.copysign(g2, d * ga) / (np.abs(d) + np.sqrt(d * d + g2 * g2))
                c = np.where(rot, 1 / np.sqrt(1 + t * t), 1.0)
                s = np.where(rot, c * t, 0.0)
                A[:, ps], A[:, qs] = c * x - s * y, s * x + c * y
        if worst < 1e-16:
            return sweep + 1
        nrm = (A[:, act] ** 2).sum(0)
        act = [a for a, v in zip(act, nrm) if v > nul]
        if len(act) < 2:
            return sweep + 1
    return -1


def pivot_order(R):
    """pivot sequence of QR with column pivoting on R (from the Gram matrix: pivoted Cholesky)"""
    G = R.T @ R
    n = G.shape[0]
    d, perm, alive = np.diag(G).copy(), [], np.ones(n, bool)
    for _ in range(n):
        j = int(np.argmax(np.where(alive, d, -1)))
        if d[j] <= 0:
            break
        perm.append(j); alive[j] = False
        col = G[:, j] / np.sqrt(d[j])
        G = G - np.outer(col, col)
        d = np.diag(G).copy()
    return perm + [i for i in range(n) if alive[i]]


def report(tag, mats, Mb):
    rows = []
    for Mx in mats:
        R1 = np.linalg.qr(Mx.T, mode="r")
        byn = np.argsort(-(R1 * R1).sum(0))
        sv = np.linalg.svd(Mx, compute_uv=False)
        rows.append((jacobi_sweeps(R1.T), jacobi_sweeps(np.linalg.qr(R1[:, byn], mode="r").T),
                     jacobi_sweeps(np.linalg.qr(R1[:, pivot_order(R1)], mode="r").T), sv[Mb - 1] / sv[0], int((sv > 1e-14 * sv[0]).sum())))
    r = np.array(rows)
    print(f"{tag}: {len(mats)} factors {mats[0].shape[0]} x {mats[0].shape[1]}; sweeps on R1^T (shipped) {r[:, 0].mean():.1f}, on R2^T with columns "
          f"by norm {r[:, 1].mean():.1f}, in pivot order {r[:, 2].mean():.1f}; sigma_{Mb} / sigma_1 median {np.median(r[:, 3]):.1e}, numerical rank "
          f"(1e-14) median {int(np.median(r[:, 4]))} of {mats[0].shape[0]}", flush=True)


def capture(fn, rows_min, cols_min):
    caps, orig = [], np.linalg.svd

    def spy(Mx, *a, **k):
        if Mx.ndim == 2 and Mx.shape[0] >= rows_min and Mx.shape[1] >= cols_min:
            caps.append(Mx.copy())
        return orig(Mx, *a, **k)
    np.linalg.svd = spy
    try:
        fn()
    finally:
        np.linalg.svd = orig
    return caps


def main():
    Mb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(5)
    T = 12
    w = OF.SISFactor(0.1, 0.05)
    wi = [w] * (T + 1)

    def rand_train():
        prof = [min(Mb, 4 ** min(t, T + 1 - t)) for t in range(T + 2)]
        return OT.TensorTrain([rng.random((prof[t], prof[t + 1], 2, 2)) * np.exp(-3 * rng.random((prof[t], prof[t + 1], 1, 1))) for t in range(T + 1)])
    mats = capture(lambda: [DA.op_kron_compress_qr(wi, (rand_train(), 1), (rand_train(), 1), T, OT.TruncBond(Mb)) for _ in range(3)], 2 * Mb, 16 * Mb)
    report("(a) random saturated trains", mats, Mb)
    N, T = 8, 30
    A = nx.to_numpy_array(nx.random_regular_graph(3, N, seed=0), nodelist=range(N))
    phi = [[np.array([0.9, 0.1]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    ref_op = O.op_kron_compress
    O.op_kron_compress = lambda wi_, a, b, T_, tr: DA.op_kron_compress_qr(wi_, a, b, T_, tr)
    try:
        bp = O.mpbp(O.IndexedBiDiGraph(A), [[OF.SISFactor(0.1, 0.05)] * (T + 1)] * N, [2] * N, T, phi=phi)
        O.iterate(bp, maxiter=3, svd_trunc=OT.TruncBond(Mb), tol=0.0, shuffle_nodes=False, jacobi=True)
        mats = capture(lambda: O.onebpiter(bp, 0, OT.TruncBond(Mb), 0.0), 4 * Mb, 16 * Mb)
    finally:
        O.op_kron_compress = ref_op
    report(f"(b) BP factors (SIS, 3-regular N = 8, T = {T}, TruncBond({Mb}), fourth sweep, node 0)", mats[::max(1, len(mats) // 30)], Mb)


if __name__ == "__main__":
    main()
