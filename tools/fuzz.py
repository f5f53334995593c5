"""Randomised parity check of the device path against the numpy oracle: random small graphs, models (SIS, SIRS,
homogeneous / +-J Glauber), chain lengths, bond caps, truncation rules, damping and schedules.  Test infrastructure
(imports oracle/); usage: python tools/fuzz.py [n_cases] [seed] [big]."""
import os
import sys

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import mpbp_amd as M
import oracle.factors as OF
import oracle.mpbp as O
import oracle.tensor_trains as OT

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = len(sys.argv) > 3 and sys.argv[3] == "big"      # bond caps 10..18: the 2- and 4-panel QR paths of the 512-thread engine
rng = np.random.default_rng(seed)


def random_graph(N):
    while True:
        A = np.triu((rng.random((N, N)) < rng.uniform(0.2, 0.6)).astype(float), 1)
        A = A + A.T
        if A.sum() > 0:
            return A


worst, bad = 0.0, 0
for case in range(n_cases):
    N = int(rng.integers(3, 9))
    T = int(rng.integers(1, 7))
    Mb = int(rng.integers(2, 10))
    if big:
        N, T, Mb = int(rng.integers(4, 7)), int(rng.integers(4, 9)), int(rng.integers(10, 19))
    A = random_graph(N)
    model = rng.choice(["sis", "sirs", "glauber_h", "glauber_pmj"]) if not big else rng.choice(["sis", "glauber_h"])
    damp = float(rng.choice([0.0, 0.0, 0.3]))
    kind = rng.choice(["bond", "bondmax", "thresh", "bondthresh"])
    sweeps = int(rng.integers(1, 4)) if not big else 3
    if model == "sis":
        q, par = 2, (rng.uniform(0.05, 0.6), rng.uniform(0.05, 0.6), rng.uniform(0, 0.2))
        w = [[M.SISFactor(*par)] * (T + 1)] * N
        ow = [[OF.SISFactor(*par)] * (T + 1)] * N
    elif model == "sirs":
        q, par = 3, (rng.uniform(0.1, 0.6), rng.uniform(0.1, 0.6), rng.uniform(0.1, 0.5), rng.uniform(0, 0.1))
        w = [[M.SIRSFactor(*par)] * (T + 1)] * N
        ow = [[OF.SIRSFactor(*par)] * (T + 1)] * N
        Mb = min(Mb, 6)
    else:
        q = 2
        J = A * (rng.uniform(0.2, 1.0) if model == "glauber_h" else 0.7 * rng.choice([-1.0, 1.0], size=A.shape))
        J = np.triu(J, 1); J = J + J.T
        h = (np.full(N, rng.uniform(-0.5, 0.5)) if model == "glauber_h" else np.zeros(N))
        beta = rng.uniform(0.3, 1.2)
        w = M.glauber_factors(A != 0, J, h, beta, T)
        ow = OF.glauber_factors(A != 0, J, h, beta, T)
    phi = [[rng.uniform(0.1, 1.0, size=q) if (t == 0 or rng.random() < 0.15) else np.ones(q) for t in range(T + 1)] for _ in range(N)]
    tr = {"bond": (M.TruncBond(Mb), OT.TruncBond(Mb)), "bondmax": (M.TruncBondMax(Mb), OT.TruncBondMax(Mb)),
          "thresh": (M.TruncThresh(1e-3), OT.TruncThresh(1e-3)), "bondthresh": (M.TruncBondThresh(Mb, 1e-4), OT.TruncBondThresh(Mb, 1e-4))}[kind]
    cap = 32 if kind == "thresh" else Mb
    try:
        bp = M.mpbp(M.IndexedBiDiGraph(A), w, q, T, phi=phi, max_bond=cap)
        obp = O.mpbp(O.IndexedBiDiGraph(A), ow, [q] * N, T, phi=phi)
        for s in range(sweeps):
            M.iterate(bp, maxiter=1, svd_trunc=tr[0], tol=0.0, damp=damp)
            O.iterate(obp, maxiter=1, svd_trunc=tr[1], tol=0.0, shuffle_nodes=False, jacobi=True, damp=damp)
        b, ob = np.array(M.beliefs(bp)), np.array(O.beliefs(obp))
        e1 = np.abs(b - ob).max()
        f, of = M.bethe_free_energy(bp), O.bethe_free_energy(obp)
        e2 = abs(f - of) / max(1.0, abs(of))
        pb, opb = M.pair_beliefs(bp)[0], O.pair_beliefs(obp)[0]
        e3 = max(np.abs(np.array(x) - np.array(y)).max() for x, y in zip(pb, opb))
        err = max(e1, e2, e3)
    except Exception as ex:          # noqa: BLE001
        err = float("inf")
        print(f"case {case}: EXCEPTION {type(ex).__name__}: {str(ex)[:200]}")
    worst = max(worst, err)
    flag = "" if err < 1e-6 else "   <-- MISMATCH"
    bad += bool(flag)
    print(f"case {case:3d}: {model:11s} N={N} T={T} Mb={Mb} {kind:10s} damp={damp:.1f} sweeps={sweeps}  err={err:.2e}{flag}", flush=True)
print(f"worst error {worst:.2e}; mismatches {bad} / {n_cases}")
sys.exit(1 if bad else 0)
