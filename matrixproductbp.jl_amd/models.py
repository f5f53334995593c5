"""Problem containers of the reference's Models module (src/Models/epidemics/sis.jl:1-33,
src/Models/glauber/glauber.jl:6-63): they only hold the graph, rates and observations and build the
factor lists; `mpbp(model)` mirrors src/Models/epidemics/sis_bp.jl:42-46 / glauber_bp.jl:95-101."""
from __future__ import annotations

import numpy as np

from .factors import SISFactor, glauber_factors
from .mpbp import IndexedBiDiGraph, mpbp as _mpbp


class SIS:
    """SIS(g, λ, ρ, T; ψ, γ, α, ϕ) - src/Models/epidemics/sis.jl:24-29"""

    def __init__(self, A, lam, rho, T, gamma=0.5, alpha=0.0, phi=None, psi=None):
        self.A = np.asarray(A)
        self.N = self.A.shape[0]
        self.lam, self.rho, self.alpha, self.T = float(lam), float(rho), float(alpha), int(T)
        g = np.full(self.N, gamma) if np.isscalar(gamma) else np.asarray(gamma)
        self.phi = [[np.array([1 - g[i], g[i]]) if t == 0 else np.ones(2) for t in range(T + 1)]
                    for i in range(self.N)] if phi is None else phi
        self.psi = psi

    def factors(self):
        w = SISFactor(self.lam, self.rho, self.alpha)      # sis_factors: fill(SISFactor, T+1) per node
        return [[w] * (self.T + 1) for _ in range(self.N)]

    def mpbp(self, **kw):
        return _mpbp(IndexedBiDiGraph(self.A), self.factors(), 2, self.T, phi=self.phi, psi=self.psi, **kw)


class Ising:
    """src/Models/glauber/glauber.jl:6-30 (J given as the symmetric coupling matrix)."""

    def __init__(self, J, h, beta=1.0):
        self.J = np.asarray(J, dtype=float)
        self.h = np.asarray(h, dtype=float)
        self.beta = float(beta)
        self.A = self.J != 0


class Glauber:
    """src/Models/glauber/glauber.jl:43-63"""

    def __init__(self, ising: Ising, T, phi=None, psi=None):
        self.ising, self.T = ising, int(T)
        N = ising.A.shape[0]
        self.phi = [[np.ones(2) for _ in range(T + 1)] for _ in range(N)] if phi is None else phi
        self.psi = psi

    def mpbp(self, **kw):
        w = glauber_factors(self.ising.A, self.ising.J, self.ising.h, self.ising.beta, self.T)
        return _mpbp(IndexedBiDiGraph(self.ising.A), w, 2, self.T, phi=self.phi, psi=self.psi, **kw)
