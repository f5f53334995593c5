"""Brute-force enumeration - restatement of the reference's ``src/exact.jl:5-119`` (TEST
INFRASTRUCTURE).  Exponential in N·(T+1); only for the tiny graphs of the reference's tests."""
from __future__ import annotations

import itertools

import numpy as np


def exact_prob(bp, periodic=False):
    """exact.jl:5-41.  Returns ``(p, Z, states)`` with ``p`` an array of shape
    ``[q_1]*(T+1) + [q_2]*(T+1) + ...`` (node-major, time inside) and Z the partition function."""
    g = bp.g
    N, T = g.nv(), bp.T
    qs = [bp.nstates(i) for i in range(N)]
    shape = []
    for i in range(N):
        shape += [qs[i]] * (T + 1)
    logp = np.zeros(shape)
    edges = g.edges()
    nbrs = [g.neighbors(i) for i in range(N)]
    for idx in itertools.product(*[range(s) for s in shape]):
        X = np.array(idx).reshape(N, T + 1)
        lp = 0.0
        for i in range(N):
            lp += np.log(bp.phi[i][0][X[i, 0]])
            for t in range(T):
                lp += np.log(bp.w[i][t](X[i, t + 1] + 1, [X[k, t] + 1 for k in nbrs[i]], X[i, t] + 1))
                lp += np.log(bp.phi[i][t + 1][X[i, t + 1]])
            if periodic:
                lp += np.log(bp.w[i][T](X[i, 0] + 1, [X[k, T] + 1 for k in nbrs[i]], X[i, T] + 1))
        for (i, j, ij) in edges:
            for t in range(T + 1):
                lp += 0.5 * np.log(bp.psi[ij][t][X[i, t], X[j, t]])
        logp[idx] = lp
    m = logp.max()
    logZ = m + np.log(np.sum(np.exp(logp - m)))
    return np.exp(logp - logZ), float(np.exp(logZ))


def exact_marginals(bp, p):
    """exact.jl:43-74: ``out[i][t][x]``"""
    N, T = bp.g.nv(), bp.T
    out = []
    for i in range(N):
        row = []
        for t in range(T + 1):
            ax = i * (T + 1) + t
            other = tuple(a for a in range(p.ndim) if a != ax)
            row.append(p.sum(axis=other))
        out.append(row)
    return out


def exact_pair_marginals(bp, p):
    """exact.jl:76-119: ``out[e][t][x_i, x_j]`` for every directed edge e = (i->j)."""
    T = bp.T
    out = []
    for (i, j, ij) in bp.g.edges():
        row = []
        for t in range(T + 1):
            a, b = i * (T + 1) + t, j * (T + 1) + t
            other = tuple(c for c in range(p.ndim) if c not in (a, b))
            m = p.sum(axis=other)
            row.append(m if a < b else m.T)
        out.append(row)
    return out


def exact_autocorrelations(f, bp, p):
    """exact.jl:161-186"""
    N, T = bp.g.nv(), bp.T
    out = []
    for i in range(N):
        r = np.zeros((T + 1, T + 1))
        for u in range(T + 1):
            for t in range(u):
                a, b = i * (T + 1) + t, i * (T + 1) + u
                other = tuple(c for c in range(p.ndim) if c not in (a, b))
                m = p.sum(axis=other)
                r[t, u] = sum(f(x + 1, i) * f(y + 1, i) * m[x, y] for x in range(m.shape[0]) for y in range(m.shape[1]))
        out.append(r)
    return out
