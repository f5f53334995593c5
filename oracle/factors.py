"""Factor (model) definitions - restatement of the reference's ``src/Models`` tables and of the
``RecursiveBPFactor`` interface (TEST INFRASTRUCTURE).

States are 1-based integers exactly as in the reference (SIS: 1=S, 2=I; Glauber: 1=spin +1,
2=spin -1), so every formula can be read side by side with the Julia source.
"""
from __future__ import annotations

import numpy as np

SUSCEPTIBLE, INFECTIOUS, RECOVERED = 1, 2, 3


def potts2spin(x):
    """glauber.jl:2"""
    return 3 - 2 * x


class BPFactor:
    """bp_core.jl:10 - anything callable as ``w(x_next, x_neighbours, x)``."""

    recursive = False

    def __call__(self, xnext, xneigh, x):
        raise NotImplementedError


class RecursiveBPFactor(BPFactor):
    """recursive_bp_factor.jl:6-27 interface + optional generic methods :33-61."""

    recursive = True

    def nstates(self, l):
        raise NotImplementedError

    def prob_y(self, xnext, x, y, d):
        raise NotImplementedError

    def prob_xy(self, yk, xk, xi, k=None):
        raise NotImplementedError

    def prob_yy(self, y, y1, y2, xi, d1=None, d2=None):
        raise NotImplementedError

    def prob_y0(self, y, xi):
        return float(y == 1)                                   # recursive_bp_factor.jl:27

    def __call__(self, xnext, xneigh, x):
        """recursive_bp_factor.jl:33-46 generic functor."""
        d = len(xneigh)
        Pyy = [float(self.prob_y0(y, x)) for y in range(1, self.nstates(0) + 1)]
        for k in range(1, d + 1):
            Pyy = [sum(self.prob_yy(y, y1, y2, x, 1, k - 1) * self.prob_xy(y1, xneigh[k - 1], x, k) * Pyy[y2 - 1]
                       for y1 in range(1, self.nstates(1) + 1) for y2 in range(1, len(Pyy) + 1))
                   for y in range(1, self.nstates(k) + 1)]
        return sum(Pyy[y - 1] * self.prob_y(xnext, x, y, d) for y in range(1, len(Pyy) + 1))

    def prob_y_partial(self, xnext, x, xk, y1, d, k):
        """recursive_bp_factor.jl:49-54"""
        return sum(self.prob_y(xnext, x, y, d + 1) * self.prob_xy(y2, xk, x, k) * self.prob_yy(y, y1, y2, x, d, 1)
                   for y in range(1, self.nstates(d + 1) + 1) for y2 in range(1, self.nstates(1) + 1))

    def prob_y_dummy(self, xnext, x, xk, y1, d, j):
        """recursive_bp_factor.jl:59-61"""
        return self.prob_y(xnext, x, y1, d)


# ------------------------------------------------------------------------------ epidemics

class SISFactor(RecursiveBPFactor):
    """sis_bp.jl:4-15,18,20-40,61-78"""

    def __init__(self, lam, rho, alpha=0.0):
        assert 0 <= lam <= 1 and 0 <= rho <= 1 and 0 <= alpha <= 1
        self.lam, self.rho, self.alpha = float(lam), float(rho), float(alpha)

    def nstates(self, l):
        return 1 if l == 0 else 2

    def __call__(self, xnext, xneigh, x):
        if x == INFECTIOUS:
            return self.rho if xnext == SUSCEPTIBLE else 1 - self.rho
        p = (1 - self.alpha) * (1 - self.lam) ** sum(1 for xj in xneigh if xj == INFECTIOUS)
        return p if xnext == SUSCEPTIBLE else 1 - p

    def prob_y(self, xnext, x, y, d):
        z = 1.0
        w = (y == SUSCEPTIBLE) * (1 - self.alpha)
        if xnext == INFECTIOUS:
            return (x == INFECTIOUS) * (1 - self.rho) + (x == SUSCEPTIBLE) * (1 - z * w)
        return (x == INFECTIOUS) * self.rho + (x == SUSCEPTIBLE) * z * w

    def prob_xy(self, yk, xk, xi, k=None):
        lam = self.lam
        return (yk == INFECTIOUS) * lam * (xk == INFECTIOUS) + (yk == SUSCEPTIBLE) * (1 - lam * (xk == INFECTIOUS))

    def prob_yy(self, y, y1, y2, xi, d1=None, d2=None):
        return 1.0 * ((y == INFECTIOUS) == ((y1 == INFECTIOUS) or (y2 == INFECTIOUS)))


class SISHeterogeneousFactor(RecursiveBPFactor):
    """sis_heterogeneous_bp.jl:4-74 (``λ[k]`` = infection probability from the k-th neighbour)."""

    def __init__(self, lam, rho, alpha=0.0):
        self.lam = [float(v) for v in lam]
        self.rho, self.alpha = float(rho), float(alpha)

    def nstates(self, l):
        return 1 if l == 0 else 2

    def __call__(self, xnext, xneigh, x):
        if x == INFECTIOUS:
            return self.rho if xnext == SUSCEPTIBLE else 1 - self.rho
        p = 1 - self.alpha
        for xj, lj in zip(xneigh, self.lam):
            p *= 1 - lj * (xj == INFECTIOUS)
        return p if xnext == SUSCEPTIBLE else 1 - p

    def prob_y(self, xnext, x, y, d):
        w = (y == SUSCEPTIBLE) * (1 - self.alpha)
        if xnext == INFECTIOUS:
            return (x == INFECTIOUS) * (1 - self.rho) + (x == SUSCEPTIBLE) * (1 - w)
        return (x == INFECTIOUS) * self.rho + (x == SUSCEPTIBLE) * w

    def prob_xy(self, yk, xk, xi, k=None):
        lam = self.lam[k - 1]
        return (yk == INFECTIOUS) * lam * (xk == INFECTIOUS) + (yk == SUSCEPTIBLE) * (1 - lam * (xk == INFECTIOUS))

    def prob_yy(self, y, y1, y2, xi, d1=None, d2=None):
        return 1.0 * ((y == INFECTIOUS) == ((y1 == INFECTIOUS) or (y2 == INFECTIOUS)))


class SIRSFactor(RecursiveBPFactor):
    """sirs_bp.jl:3-44 (q = 3).  The functor is the generic recursive one (the reference's own
    functor lives in sirs.jl, which is a container file; the generic recursion is equivalent)."""

    def __init__(self, lam, rho, sigma, alpha=0.0):
        self.lam, self.rho, self.sigma, self.alpha = float(lam), float(rho), float(sigma), float(alpha)

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


# -------------------------------------------------------------------------------- glauber

class GenericGlauberFactor(BPFactor):
    """glauber_bp.jl:1-20 (non recursive -> exhaustive-trace update)."""

    def __init__(self, J, h, beta):
        self.betaJ = [float(j) * beta for j in J]
        self.betah = float(h) * beta

    def __call__(self, xnext, xneigh, x):
        assert len(xneigh) == len(self.betaJ)
        hji = sum(J * potts2spin(xj) for xj, J in zip(xneigh, self.betaJ))
        E = -potts2spin(xnext) * (hji + self.betah)
        return 1 / (1 + np.exp(2 * E))


class HomogeneousGlauberFactor(RecursiveBPFactor):
    """glauber_bp.jl:22-56"""

    def __init__(self, J, h, beta):
        self.betaJ, self.betah = float(J) * beta, float(h) * beta

    def nstates(self, l):
        return l + 1

    def prob_y(self, xnext, x, z, d):
        y = 2 * z - 2 - d
        hji = self.betaJ * y + self.betah
        E = -potts2spin(xnext) * hji
        return 1 / (1 + np.exp(2 * E))

    def prob_xy(self, yk, xk, xi, k=None):
        return float(yk != xk)

    def prob_yy(self, y, y1, y2, xi, d1=None, d2=None):
        return float(y == y1 + y2 - 1)

    def __call__(self, xnext, xneigh, x):
        hji = self.betaJ * sum(potts2spin(xj) for xj in xneigh)
        E = -potts2spin(xnext) * (hji + self.betah)
        return 1 / (1 + np.exp(2 * E))


class PMJGlauberFactor(RecursiveBPFactor):
    """glauber_bp.jl:58-92 (±J couplings)."""

    def __init__(self, signs, J, h, beta):
        self.signs = [int(s) for s in signs]
        self.betaJ, self.betah = float(J) * beta, float(h) * beta

    def nstates(self, d):
        return 2 * d + 1

    def prob_y(self, xnext, x, y, d):
        ht = y - d - 1
        bh = self.betaJ * ht + self.betah
        E = -potts2spin(xnext) * bh
        return 1 / (1 + np.exp(2 * E))

    def prob_xy(self, yk, xk, xi, k=None):
        return float(yk == potts2spin(xk) * self.signs[k - 1] + 2)

    def prob_yy(self, y, y1, y2, xi, d1=None, d2=None):
        return float(y == y1 + y2 - 1)

    def __call__(self, xnext, xneigh, x):
        hji = self.betaJ * sum(s * potts2spin(xj) for xj, s in zip(xneigh, self.signs))
        E = -potts2spin(xnext) * (hji + self.betah)
        return 1 / (1 + np.exp(2 * E))


class IntegerGlauberFactor(RecursiveBPFactor):
    """glauber_bp.jl:144-179"""

    def __init__(self, J, h, beta, K=None):
        self.J = [int(j) for j in J]
        self.h, self.beta = float(h), float(beta)
        self.K = sum(abs(j) for j in self.J) + 1 if K is None else int(K)

    def nstates(self, l):
        return 2 * self.K - 1

    def prob_y(self, xnext, x, y, d):
        ht = y - self.K
        bh = self.beta * (ht + self.h)
        E = -potts2spin(xnext) * bh
        return 1 / (1 + np.exp(2 * E))

    def prob_xy(self, yk, xk, xi, k=None):
        return float(yk == potts2spin(xk) * self.J[k - 1] + self.K)

    def prob_yy(self, y, y1, y2, xi, d1=None, d2=None):
        return float(y + self.K == y1 + y2)

    def prob_y0(self, y, xi):
        return float(y == self.K)

    def __call__(self, xnext, xneigh, x):
        ht = sum(Jk * potts2spin(xk) for Jk, xk in zip(self.J, xneigh))
        bh = self.beta * (ht + self.h)
        E = -potts2spin(xnext) * bh
        return 1 / (1 + np.exp(2 * E))


def glauber_factors(A, J, h, beta, T):
    """glauber_bp.jl:121-142: choose the factor type per node from the coupling structure.
    ``J`` is the symmetric coupling matrix, ``A`` its sparsity (neighbours ascending)."""
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
            if homog:
                w = HomogeneousGlauberFactor(J0, h[i], beta)
            else:
                w = PMJGlauberFactor([int(np.sign(v)) for v in Ji], beta * abs(J0), beta * h[i], 1.0)
        elif all(float(v).is_integer() for v in Ji):
            w = IntegerGlauberFactor([int(v) for v in Ji], h[i], beta)
        else:
            w = GenericGlauberFactor(Ji, h[i], beta)
        out.append([w] * (T + 1))
    return out


# ---------------------------------------------------------------------- wrappers (tests)

class DampedFactor(RecursiveBPFactor):
    """recursive_bp_factor.jl:183-206"""

    def __init__(self, w, p):
        assert 0 <= p <= 1
        self.w, self.p = w, float(p)

    def nstates(self, l):
        return self.w.nstates(l)

    def prob_xy(self, *a):
        return self.w.prob_xy(*a)

    def prob_yy(self, *a):
        return self.w.prob_yy(*a)

    def prob_y0(self, y, xi):
        return self.w.prob_y0(y, xi)

    def __call__(self, xnext, xneigh, x):
        return (1 - self.p) * self.w(xnext, xneigh, x) + self.p * (xnext == x)

    def prob_y(self, xnext, x, y, d):
        return (1 - self.p) * self.w.prob_y(xnext, x, y, d) + self.p * (xnext == x)


class RecursiveTraceFactor(RecursiveBPFactor):
    """test_factors.jl:5-19: any BPFactor as a recursive one with ``N^d`` accumulated states."""

    def __init__(self, w, N):
        self.w, self.N = w, int(N)

    def nstates(self, d):
        return self.N ** d

    def prob_y(self, xnext, x, y, d):
        digits = []
        v = y - 1
        for _ in range(d):
            digits.append(v % self.N)
            v //= self.N
        xs = [dg + 1 for dg in reversed(digits)]
        return self.w(xnext, xs, x)

    def prob_xy(self, yk, xk, xi, k=None):
        return float(yk == xk)

    def prob_yy(self, y, y1, y2, xi, d1=None, d2=None):
        return float(y - 1 == (y1 - 1) + (y2 - 1) * self.nstates(d1))

    def __call__(self, xnext, xneigh, x):
        return self.w(xnext, xneigh, x)


class RestrictedRecursiveBPFactor(RecursiveBPFactor):
    """test_factors.jl:26-34: forces the generic ``prob_y_partial`` and functor."""

    def __init__(self, w):
        self.w = w

    def nstates(self, l):
        return self.w.nstates(l)

    def prob_y(self, *a):
        return self.w.prob_y(*a)

    def prob_xy(self, *a):
        return self.w.prob_xy(*a)

    def prob_yy(self, *a):
        return self.w.prob_yy(*a)

    def prob_y0(self, y, xi):
        return self.w.prob_y0(y, xi)


class GenericFactor(BPFactor):
    """test_factors.jl:41-45: forces the exhaustive-trace update."""

    def __init__(self, w):
        self.w = w

    def __call__(self, xnext, xneigh, x):
        return self.w(xnext, xneigh, x)
