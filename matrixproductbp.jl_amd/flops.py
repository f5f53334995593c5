"""Flop models of one node update on a given bond profile (used by bench.py for the roofline).

`executed`  : what the device engine actually runs (structured contractions + R-only Householder QR
              + small QR/Jacobi), GEMM = 2mnk, QR(m x n, R only) = 2 n^2 (m - n/3) for m >= n.
`reference` : the reference algorithm's dense operations on the same shapes: Kronecker
              build, two SVD sweeps per `op` with thin-SVD cost 6*max*min^2 + 20*min^3, carry GEMMs,
              mpem2 + final truncating sweep.
Both take the bond profile b[0..L] of the trains (messages and cavity intermediates share it once the
cap binds) and the degree; physical dims: messages q*q, cavity trains ny*q."""
from __future__ import annotations


def _qr(m, n):
    if m >= n:
        return 2.0 * n * n * (m - n / 3.0)
    return 2.0 * m * m * (n - m / 3.0)


def _svd(m, n):
    mx, mn = max(m, n), min(m, n)
    return 6.0 * mx * mn * mn + 20.0 * mn ** 3


def op_flops(b1, b2, bout, ny1, ny2, ny, q):
    """one `op` (Kronecker + compress!): returns (executed, reference)."""
    L = len(b1) - 1
    p = ny * q
    ex = ref = 0.0
    # sweep 1 (t = L-1 .. 1)
    r = 1
    rref = 1
    for t in range(L - 1, 0, -1):
        a, an, b, bn = b1[t], b1[t + 1], b2[t], b2[t + 1]
        Bm = a * b
        ex += 2.0 * (a * ny1 * q) * an * (bn * r)            # Y1
        ex += q * 2.0 * (b * ny) * (bn * ny1) * (a * r)      # Y2
        ex += _qr(r * p, Bm)
        r = min(r * p, Bm)
        # reference: Kronecker build, SVD of Bm x (rref*p), carry GEMM into core t-1
        ref += 2.0 * Bm * (an * bn) * p * ny1 * ny2 / max(ny, 1)
        ref += _svd(Bm, rref * p)
        knew = min(Bm, rref * p)
        ref += 2.0 * (b1[t - 1] * b2[t - 1]) * Bm * knew * p
        rref = knew
    # sweep 2 (t = 0 .. L-2)
    kc = 1
    rdims = [1] * (L + 1)
    rr = 1
    for t in range(L - 1, 0, -1):
        rr = min(rr * p, b1[t] * b2[t])
        rdims[t] = rr
    for t in range(L - 1):
        a, an, b, bn = b1[t], b1[t + 1], b2[t], b2[t + 1]
        Bn = an * bn
        Rr = kc * p
        r1 = rdims[t + 1]
        ex += 2.0 * (an * ny1 * q) * a * (kc * b)            # N1
        ex += q * 2.0 * (bn * ny) * (b * ny1) * (an * kc)    # N2
        ex += 2.0 * r1 * Bn * Rr                             # Mt^T
        ex += _qr(r1, Rr)
        ex += 8.0 * 10.0 * min(r1, Rr) * Rr * Rr             # ~8 Jacobi sweeps
        kp = bout[t + 1]
        ex += 2.0 * kp * Rr * Bn                             # carry
        ref += _svd(Rr, r1) + 2.0 * kp * r1 * rdims[t + 2] * p if t + 2 <= L else 0.0
        kc = kp
    return ex, ref


def finalize_flops(b, bout, q):
    """mpem2 + compress!(:left) of one message: returns (executed, reference)."""
    L = len(b) - 1
    p = q * q
    cb = [1] + [q * x for x in b[1:L]] + [1]
    ones = [1] * (L + 1)
    ex, _ = op_flops(cb[::-1], ones, bout[::-1], p, 1, p, 1)
    ref = 0.0
    for t in range(L - 1):
        ref += _svd(p * b[t], q * b[t + 1]) + 2.0 * min(p * b[t], q * b[t + 1]) * q * b[t + 1] * b[t + 2 if t + 2 <= L else L] * p
        ref += _svd(min(p * b[t], q * b[t + 1]), bout[t + 1] * p)
    return ex, ref


def node_update_flops(bmsg, deg, q, ny):
    """Whole onebpiter! of a degree-`deg` node whose trains all have bond profile `bmsg`
    (ny(l) given as a function).  Returns dict with executed/reference totals and the cavity-op share."""
    init = [1] * len(bmsg)
    ex_ops = ref_ops = 0.0

    def op(b1, d1, b2, d2):
        nonlocal ex_ops, ref_ops
        bo = [1] + [min(bmsg[t], b1[t] * b2[t]) for t in range(1, len(bmsg) - 1)] + [1]
        e, r = op_flops(b1, b2, bo, ny(d1), ny(d2), ny(d1 + d2), q)
        ex_ops += e
        ref_ops += r
        return bo, d1 + d2

    z = deg
    src = [(list(bmsg), 1)] * z
    if z == 1:
        op(*src[0], init, 0)
    elif z >= 2:
        dest = [src[0]]
        for k in range(1, z):
            dest.append(op(*dest[-1], *src[k]))
        op(*dest[-1], init, 0)
        right = (init, 0)
        for i in range(z - 1, 0, -1):
            op(*dest[i - 1], *right)
            right = op(*src[i], *right)
    ex_fin = ref_fin = 0.0
    for _ in range(z):
        e, r = finalize_flops(list(bmsg), list(bmsg), q)
        ex_fin += e
        ref_fin += r
    return {"executed_ops": ex_ops, "reference_ops": ref_ops, "executed_total": ex_ops + ex_fin,
            "reference_total": ref_ops + ref_fin}
