"""Host-side factors: the `RecursiveBPFactor` interface of the reference
(src/recursive_bp_factor.jl:6-27) and the shipped models (src/Models/epidemics/sis_bp.jl,
sirs_bp.jl, sis_heterogeneous_bp.jl; src/Models/glauber/glauber_bp.jl), evaluated into the dense
tables that `mpbp_set_factor` (include/mpbp_hip.h) takes.  States are 1-based as in the reference.

A new model only has to implement `nstates`, `prob_y`, `prob_xy`, `prob_yy` (and optionally
`prob_y0`), exactly the reference's minimal interface.
"""
from __future__ import annotations

import numpy as np

SUSCEPTIBLE, INFECTIOUS, RECOVERED = 1, 2, 3


def potts2spin(x):
    return 3 - 2 * x


class BPFactor:
    """src/bp_core.jl:1-10: a factor only has to be callable as `w(x_next, x_neighbours, x)` (1-based states).  A factor
    that is not a `RecursiveBPFactor` is updated by the exhaustive-trace path (`f_bp`, src/bp_core.jl:18-93): the host
    evaluates the functor into the dense table that `mpbp_set_generic_factor` (include/mpbp_hip.h) takes."""

    def __call__(self, xnext, xnbrs, xi):
        raise NotImplementedError("Not implemented")

    def key(self):
        return (type(self).__name__, id(self))

    def generic_table(self, deg: int, q: int):
        """w[x_next + q (x + q (x_1 + q (x_2 + ...)))], 0-based, first index fastest."""
        tab = np.zeros((q, q) + (q,) * deg)
        for cfg in np.ndindex(*((q,) * deg)):
            xs = [int(v) + 1 for v in cfg]
            for x in range(q):
                for xn in range(q):
                    tab[(xn, x) + tuple(cfg)] = self(xn + 1, xs, x + 1)
        return tab.ravel(order="F")


class RecursiveBPFactor(BPFactor):
    """src/recursive_bp_factor.jl:6-27"""

    def nstates(self, l: int) -> int:
        raise NotImplementedError("Not implemented")

    def prob_y(self, xnext, x, y, d):
        raise NotImplementedError("Not implemented")

    def prob_xy(self, yk, xk, xi, k=None):
        raise NotImplementedError("Not implemented")

    def prob_yy(self, y, y1, y2, xi, d1=None, d2=None):
        raise NotImplementedError("Not implemented")

    def prob_y0(self, y, xi):
        return float(y == 1)

    # dense tables in the layouts of include/mpbp_hip.h (first index fastest)
    def tables(self, deg: int, q: int):
        ny = [int(self.nstates(l)) for l in range(deg + 1)]
        py = np.zeros((q, q, ny[deg]))
        for xn in range(q):
            for x in range(q):
                for y in range(ny[deg]):
                    py[xn, x, y] = self.prob_y(xn + 1, x + 1, y + 1, deg)
        ny1 = ny[1] if deg > 0 else 1
        pxy = np.zeros((ny1, q, q, max(deg, 0)))
        for k in range(deg):
            for y in range(ny1):
                for xk in range(q):
                    for xi in range(q):
                        pxy[y, xk, xi, k] = self.prob_xy(y + 1, xk + 1, xi + 1, k + 1)
        blocks = []
        for d1 in range(deg + 1):
            for d2 in range(deg - d1 + 1):
                b = np.zeros((ny[d1 + d2], ny[d1], ny[d2], q))
                for y in range(ny[d1 + d2]):
                    for y1 in range(ny[d1]):
                        for y2 in range(ny[d2]):
                            for xi in range(q):
                                b[y, y1, y2, xi] = self.prob_yy(y + 1, y1 + 1, y2 + 1, xi + 1, d1, d2)
                blocks.append(b.ravel(order="F"))
        py0 = np.zeros((ny[0], q))
        for y in range(ny[0]):
            for xi in range(q):
                py0[y, xi] = self.prob_y0(y + 1, xi + 1)
        return (np.asarray(ny, dtype=np.int32), py.ravel(order="F"), pxy.ravel(order="F"),
                np.concatenate(blocks) if blocks else np.zeros(0), py0.ravel(order="F"))

    def key(self):
        """Hashable identity used to evaluate the tables once per distinct (factor, degree)."""
        return (type(self).__name__, id(self))

    def __call__(self, xnext, xnbrs, xi):
        """`w(x_i^{t+1}, x_∂i^t, x_i^t)`: the transition probability, by folding the neighbours one at a time
        through `prob_xy` / `prob_yy` and closing with `prob_y` - the recursion that defines a RecursiveBPFactor
        (src/recursive_bp_factor.jl:29-47, `RecursiveTraceFactor`).  States are 1-based."""
        d = len(xnbrs)
        py = np.array([self.prob_y0(y, xi) for y in range(1, self.nstates(0) + 1)])
        for k, xk in enumerate(xnbrs):
            pk = np.array([self.prob_xy(y, xk, xi, k + 1) for y in range(1, self.nstates(1) + 1)])
            ny = self.nstates(k + 1)
            new = np.zeros(ny)
            for y in range(ny):
                for y1 in range(len(py)):
                    if py[y1] == 0.0:
                        continue
                    for y2 in range(len(pk)):
                        new[y] += self.prob_yy(y + 1, y1 + 1, y2 + 1, xi, k, 1) * py[y1] * pk[y2]
            py = new
        return float(sum(self.prob_y(xnext, xi, y + 1, d) * py[y] for y in range(len(py))))


class SISFactor(RecursiveBPFactor):
    """src/Models/epidemics/sis_bp.jl:4-15,18,61-78"""

    def __init__(self, lam, rho, alpha=0.0):
        assert 0 <= lam <= 1 and 0 <= rho <= 1 and 0 <= alpha <= 1
        self.lam, self.rho, self.alpha = float(lam), float(rho), float(alpha)

    def key(self):
        return ("SIS", self.lam, self.rho, self.alpha)

    def nstates(self, l):
        return 1 if l == 0 else 2

    def prob_y(self, xnext, x, y, d):
        w = (y == SUSCEPTIBLE) * (1 - self.alpha)
        if xnext == INFECTIOUS:
            return (x == INFECTIOUS) * (1 - self.rho) + (x == SUSCEPTIBLE) * (1 - w)
        return (x == INFECTIOUS) * self.rho + (x == SUSCEPTIBLE) * w

    def prob_xy(self, yk, xk, xi, k=None):
        lam = self.lam
        return (yk == INFECTIOUS) * lam * (xk == INFECTIOUS) + (yk == SUSCEPTIBLE) * (1 - lam * (xk == INFECTIOUS))

    def prob_yy(self, y, y1, y2, xi, d1=None, d2=None):
        return 1.0 * ((y == INFECTIOUS) == ((y1 == INFECTIOUS) or (y2 == INFECTIOUS)))


class SIS_heterogeneousFactor(SISFactor):
    """src/Models/epidemics/sis_heterogeneous_bp.jl:4-74: `lam[k]` per incoming neighbour."""

    def __init__(self, lam, rho, alpha=0.0):
        self.lamv = [float(v) for v in lam]
        self.rho, self.alpha = float(rho), float(alpha)

    def key(self):
        return ("SIShet", tuple(self.lamv), self.rho, self.alpha)

    def prob_xy(self, yk, xk, xi, k=None):
        lam = self.lamv[k - 1]
        return (yk == INFECTIOUS) * lam * (xk == INFECTIOUS) + (yk == SUSCEPTIBLE) * (1 - lam * (xk == INFECTIOUS))


class SIRSFactor(RecursiveBPFactor):
    """src/Models/epidemics/sirs_bp.jl:3-44 (q = 3)"""

    def __init__(self, lam, rho, sigma, alpha=0.0):
        self.lam, self.rho, self.sigma, self.alpha = float(lam), float(rho), float(sigma), float(alpha)

    def key(self):
        return ("SIRS", self.lam, self.rho, self.sigma, self.alpha)

    def nstates(self, l):
        return 1 if l == 0 else 2

    def prob_y(self, xnext, x, y, d):
        w = (y == SUSCEPTIBLE) * (1 - self.alpha)
        if xnext == INFECTIOUS:
            return (x == INFECTIOUS) * (1 - self.rho) + (x == SUSCEPTIBLE) * (1 - w)
        if xnext == SUSCEPTIBLE:
            return (x == RECOVERED) * self.sigma + (x == SUSCEPTIBLE) * w
        return (x == INFECTIOUS) * self.rho + (x == RECOVERED) * (1 - self.sigma)

    def prob_xy(self, yk, xk, xi, k=None):
        lam = self.lam
        return (yk == INFECTIOUS) * lam * (xk == INFECTIOUS) + (yk == SUSCEPTIBLE) * (1 - lam * (xk == INFECTIOUS))

    def prob_yy(self, y, y1, y2, xi, d1=None, d2=None):
        return 1.0 * ((y == INFECTIOUS) == ((y1 == INFECTIOUS) or (y2 == INFECTIOUS)))


class HomogeneousGlauberFactor(RecursiveBPFactor):
    """src/Models/glauber/glauber_bp.jl:22-44"""

    def __init__(self, J, h, beta=1.0):
        self.betaJ, self.betah = float(J) * beta, float(h) * beta

    def key(self):
        return ("HG", self.betaJ, self.betah)

    def nstates(self, l):
        return l + 1

    def prob_y(self, xnext, x, z, d):
        y = 2 * z - 2 - d
        E = -potts2spin(xnext) * (self.betaJ * y + self.betah)
        return 1 / (1 + np.exp(2 * E))

    def prob_xy(self, yk, xk, xi, k=None):
        return float(yk != xk)

    def prob_yy(self, y, y1, y2, xi, d1=None, d2=None):
        return float(y == y1 + y2 - 1)


class PMJGlauberFactor(RecursiveBPFactor):
    """src/Models/glauber/glauber_bp.jl:58-80"""

    def __init__(self, signs, J, h, beta=1.0):
        self.signs = [int(s) for s in signs]
        self.betaJ, self.betah = float(J) * beta, float(h) * beta

    def key(self):
        return ("PMJ", tuple(self.signs), self.betaJ, self.betah)

    def nstates(self, d):
        return 2 * d + 1

    def prob_y(self, xnext, x, y, d):
        E = -potts2spin(xnext) * (self.betaJ * (y - d - 1) + self.betah)
        return 1 / (1 + np.exp(2 * E))

    def prob_xy(self, yk, xk, xi, k=None):
        return float(yk == potts2spin(xk) * self.signs[k - 1] + 2)

    def prob_yy(self, y, y1, y2, xi, d1=None, d2=None):
        return float(y == y1 + y2 - 1)


class IntegerGlauberFactor(RecursiveBPFactor):
    """src/Models/glauber/glauber_bp.jl:144-170"""

    def __init__(self, J, h, beta, K=None):
        self.J = [int(j) for j in J]
        self.h, self.beta = float(h), float(beta)
        self.K = sum(abs(j) for j in self.J) + 1 if K is None else int(K)

    def key(self):
        return ("IG", tuple(self.J), self.h, self.beta, self.K)

    def nstates(self, l):
        return 2 * self.K - 1

    def prob_y(self, xnext, x, y, d):
        E = -potts2spin(xnext) * self.beta * (y - self.K + self.h)
        return 1 / (1 + np.exp(2 * E))

    def prob_xy(self, yk, xk, xi, k=None):
        return float(yk == potts2spin(xk) * self.J[k - 1] + self.K)

    def prob_yy(self, y, y1, y2, xi, d1=None, d2=None):
        return float(y + self.K == y1 + y2)

    def prob_y0(self, y, xi):
        return float(y == self.K)


class DampedFactor(RecursiveBPFactor):
    """src/recursive_bp_factor.jl:183-206"""

    def __init__(self, w, p):
        assert 0 <= p <= 1
        self.w, self.p = w, float(p)

    def key(self):
        return ("Damped", self.w.key(), self.p)

    def nstates(self, l):
        return self.w.nstates(l)

    def prob_xy(self, *a):
        return self.w.prob_xy(*a)

    def prob_yy(self, *a):
        return self.w.prob_yy(*a)

    def prob_y0(self, y, xi):
        return self.w.prob_y0(y, xi)

    def prob_y(self, xnext, x, y, d):
        return (1 - self.p) * self.w.prob_y(xnext, x, y, d) + self.p * (xnext == x)


def glauber_factors(A, J, h, beta, T):
    """src/Models/glauber/glauber_bp.jl:121-142 for the recursive factor types."""
    N = A.shape[0]
    Jnz = J[np.nonzero(np.triu(J, 1))]
    absconst = bool(np.all(np.abs(Jnz) == abs(Jnz[0]))) if Jnz.size else True
    homog = bool(np.all(Jnz == Jnz[0])) if Jnz.size else True
    out = []
    for i in range(N):
        nb = np.nonzero(A[:, i])[0]
        Ji = [J[k, i] for k in nb]
        if absconst:
            J0 = 0.0 if len(nb) == 0 else Ji[0]
            w = HomogeneousGlauberFactor(J0, h[i], beta) if homog else \
                PMJGlauberFactor([int(np.sign(v)) for v in Ji], beta * abs(J0), beta * h[i], 1.0)
        elif all(float(v).is_integer() for v in Ji):
            w = IntegerGlauberFactor([int(v) for v in Ji], h[i], beta)
        else:
            w = GenericGlauberFactor(Ji, h[i], beta)
        out.append([w] * (T + 1))
    return out


class GenericGlauberFactor(BPFactor):
    """src/Models/glauber/glauber_bp.jl:1-20: arbitrary real couplings, no recursive structure."""

    def __init__(self, J, h, beta):
        self.betaJ = [float(j) * beta for j in J]
        self.betah = float(h) * beta

    def __call__(self, xnext, xnbrs, xi):
        assert len(xnbrs) == len(self.betaJ)
        hji = sum(J * potts2spin(xj) for xj, J in zip(xnbrs, self.betaJ))
        E = -potts2spin(xnext) * (hji + self.betah)
        return 1.0 / (1.0 + np.exp(2.0 * E))


class GenericFactor(BPFactor):
    """src/test_factors.jl:41-45: wraps any factor and hides its recursive structure, which forces the
    exhaustive-trace update (used by the reference's tests to compare the two paths)."""

    def __init__(self, w):
        self.w = w

    def __call__(self, xnext, xnbrs, xi):
        return self.w(xnext, xnbrs, xi)
