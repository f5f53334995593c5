"""Device building blocks (MFMA GEMM with index maps, R-only blocked Householder QR, one-sided Jacobi)
against numpy, through the C-ABI self-test entry points."""
import ctypes as C

import numpy as np
import pytest

import mpbp_amd

pytestmark = pytest.mark.gpu


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("M,N,K", [(16, 16, 4), (20, 37, 20), (80, 300, 20), (40, 129, 40), (97, 200, 33), (200, 64, 400)])
def test_gemm_mfma_f64_asymmetric(M, N, K):
    rng = np.random.default_rng(1)
    A = np.asfortranarray(rng.standard_normal((M, K)))
    B = np.asfortranarray(rng.standard_normal((K, N)))
    Cc = np.zeros((M, N), order="F")
    rc = mpbp_amd._lib.lib().mpbp_selftest_gemm(0, M, N, K, _dp(A), _dp(B), _dp(Cc))
    assert rc == 0
    ref = A @ B
    assert np.abs(Cc - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()) * K


# (672, 150), (1568, 400), (992, 464), (2048, 208): an odd number of 32-row stages (the last 64-row reflector stage of the
# cooperative trailing pass is half empty), 4k+1 trailing tiles (odd tile updated row-parallel), the register-panel limit
@pytest.mark.parametrize("rows,cols", [(4, 16), (16, 16), (64, 16), (100, 37), (256, 64), (1600, 400), (80, 400), (33, 80),
                                       (672, 150), (1568, 400), (992, 464), (2048, 208)])
def test_qr_r_only(rows, cols):
    rng = np.random.default_rng(2)
    A = np.asfortranarray(rng.standard_normal((rows, cols)))
    if rows > 40:
        A[:, 3] = A[:, 1] * 2.0          # exact rank deficiency must not break Householder
    k = min(rows, cols)
    R = np.zeros((k, cols), order="F")
    rc = mpbp_amd._lib.lib().mpbp_selftest_qr(0, rows, cols, _dp(A), _dp(R))
    assert rc == 0
    # R is unique up to row signs: compare Gram matrices and |R| with numpy's
    G = A.T @ A
    assert np.abs(R.T @ R - G).max() < 1e-11 * np.abs(G).max()
    Rn = np.linalg.qr(A, mode="r")
    sgn = np.sign(np.diag(Rn[:k, :k])) * np.sign(np.diag(R[:k, :k]))
    sgn[sgn == 0] = 1
    assert np.abs(R * sgn[:, None] - Rn[:k]).max() < 1e-9 * np.abs(Rn).max() or rows > 40


@pytest.mark.parametrize("m,n", [(8, 8), (40, 80), (80, 80), (20, 33)])
def test_jacobi_right_singular_vectors(m, n):
    rng = np.random.default_rng(3)
    A = rng.standard_normal((m, n)) @ np.diag(np.logspace(0, -9, n))
    Af = np.asfortranarray(A.copy())
    sig = np.zeros(n)
    V = np.zeros((n, n), order="F")
    rc = mpbp_amd._lib.lib().mpbp_selftest_svd(0, m, n, _dp(Af), _dp(sig), _dp(V))
    assert rc == 0
    s_ref = np.linalg.svd(A, compute_uv=False)
    s = np.sort(sig)[::-1]
    assert np.abs(s[:len(s_ref)] - s_ref).max() < 1e-12 * s_ref[0]
    assert np.abs(V.T @ V - np.eye(n)).max() < 1e-12
    # A V has orthogonal columns with norms sigma
    W = A @ V
    assert np.abs(W.T @ W - np.diag(sig ** 2)).max() < 1e-12 * s_ref[0] ** 2


@pytest.mark.parametrize("variant", ["random", "duplicate_column", "tiny_columns"])
def test_qr_small_and_degenerate_shapes(variant):
    """Shapes the engine meets during the first sweeps (bonds 1, 4, 16: a few rows, rows < cols, partial panels)
    and rank-deficient inputs: R^T R must reproduce A^T A."""
    import ctypes as C
    L = mpbp_amd._lib.lib()
    rng = np.random.default_rng(11)
    shapes = [(1, 1), (1, 4), (2, 2), (2, 8), (3, 7), (4, 1), (4, 5), (4, 16), (4, 20), (4, 64), (8, 4), (8, 64), (12, 40),
              (16, 4), (16, 64), (20, 17), (40, 20), (80, 40), (100, 33), (160, 40), (320, 80), (400, 80), (640, 100)]
    for (r, c) in shapes:
        A = np.asfortranarray(rng.standard_normal((r, c)))
        if variant == "duplicate_column" and c > 2:
            A[:, 1] = A[:, 0]
        if variant == "tiny_columns":
            A[:, c // 2:] *= 1e-150
        R = np.zeros((min(r, c), c), order="F")
        rc = L.mpbp_selftest_qr(0, r, c, A.ctypes.data_as(C.POINTER(C.c_double)), R.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == 0 and np.isfinite(R).all(), (r, c)
        G1, G2 = R.T @ R, A.T @ A
        assert np.abs(G1 - G2).max() <= 1e-12 * max(np.abs(G2).max(), 1e-300), (r, c, variant)


def _qr_batched(A3, force_tall):
    """A3: [nprob, rows, cols] -> R [nprob, kmax, cols] through the grid-level batched QR (v2_kernels.h)."""
    nprob, rows, cols = A3.shape
    k = min(rows, cols)
    Af = np.concatenate([np.asfortranarray(A3[p]).ravel(order="F") for p in range(nprob)])
    R = np.zeros(nprob * k * cols)
    ms = C.c_double(0.0)
    rc = mpbp_amd._lib.lib().mpbp_selftest_qr_batched(0, rows, cols, nprob, int(force_tall), _dp(Af), _dp(R), C.byref(ms))
    assert rc == 0
    return np.stack([R[p * k * cols:(p + 1) * k * cols].reshape((k, cols), order="F") for p in range(nprob)]), ms.value


@pytest.mark.parametrize("force_tall", [0, 1])
def test_qr_batched_shapes(force_tall):
    """Grid-level QR of the batched gauge sweep: register-panel path and column-step ("tall") path over the shapes of the
    first sweeps (bonds 1, 4, 16), wide / square / tall matrices, rank-deficient and badly scaled inputs."""
    rng = np.random.default_rng(21)
    shapes = [(1, 1), (1, 4), (2, 8), (4, 16), (4, 20), (4, 64), (8, 4), (12, 40), (16, 64), (20, 17), (40, 20), (64, 64),
              (80, 40), (100, 33), (128, 200), (160, 40), (320, 80), (400, 80), (640, 100), (1000, 130), (1600, 400)]
    for (r, c) in shapes:
        for variant in ("random", "duplicate_column", "tiny_columns"):
            A = rng.standard_normal((3, r, c))
            if variant == "duplicate_column" and c > 2:
                A[:, :, 1] = A[:, :, 0]
            if variant == "tiny_columns":
                A[:, :, c // 2:] *= 1e-150
            R, _ = _qr_batched(A, force_tall)
            assert np.isfinite(R).all(), (r, c, variant)
            for p in range(3):
                G1, G2 = R[p].T @ R[p], A[p].T @ A[p]
                assert np.abs(G1 - G2).max() <= 1e-12 * max(np.abs(G2).max(), 1e-300), (r, c, variant, p)


@pytest.mark.parametrize("rows,cols", [(2100, 96), (4500, 160), (6400, 400), (3000, 900), (23400, 900), (21600, 900)])
def test_qr_batched_tall(rows, cols):
    """Rows beyond one workgroup's register panel (BASELINE configs[2..4] shapes, reduced): column-step panels over
    several row chunks, multi-chunk block-reflector updates.  23400 x 900 and 21600 x 900 are the full-size Y_t of the
    degree-12 / degree-11 hubs of configs[2] (900 (z+1) 2 rows)."""
    rng = np.random.default_rng(22)
    A = rng.standard_normal((2, rows, cols)) * np.logspace(0, -12, cols)[None, None, :]
    R, _ = _qr_batched(A, 0)
    for p in range(2):
        Rn = np.linalg.qr(A[p], mode="r")
        sgn = np.sign(np.diag(Rn)) * np.sign(np.diag(R[p]))
        sgn[sgn == 0] = 1
        assert np.abs(R[p] * sgn[:, None] - Rn).max() < 1e-11 * np.abs(Rn).max()


def test_qr_batched_tree_many_nodes_two_per_cu_and_ragged_tail():
    """The communication-avoiding form (csrc/cq_kernels.h) with more level-0 nodes than CUs (several rounds of node
    factorisations per launch), upper levels inside the fused update + factorisation launches, a last node of 32 rows
    (rows = 2112 + 32) and rank-deficient / badly scaled columns; every problem against LAPACK."""
    rng = np.random.default_rng(24)
    for (r, c, nprob) in [(4128, 192, 40), (2144, 320, 36)]:
        A = rng.standard_normal((nprob, r, c)) * np.logspace(0, -10, c)[None, None, :]
        A[1, :, 5] = A[1, :, 4]                     # a dependent column
        A[2, :, c // 2:] *= 1e-140                  # numerically null trailing half
        R, _ = _qr_batched(A, 0)
        assert np.isfinite(R).all()
        for p in range(nprob):
            G1, G2 = R[p].T @ R[p], A[p].T @ A[p]
            assert np.abs(G1 - G2).max() <= 1e-12 * max(np.abs(G2).max(), 1e-300), (r, c, p)
        for p in (0, 7, nprob - 1):
            Rn = np.linalg.qr(A[p], mode="r")
            sgn = np.sign(np.diag(Rn)) * np.sign(np.diag(R[p]))
            sgn[sgn == 0] = 1
            assert np.abs(R[p] * sgn[:, None] - Rn).max() < 1e-11 * np.abs(Rn).max(), (r, c, p)


def test_qr_batched_sequence_on_shared_scratch_caqr_then_lookahead():
    """Consecutive time steps of a gauge sweep share one per-problem scratch.  A step taken by the communication-avoiding
    form keeps its node slots in that scratch; a LATER step on the same scratch can fall to the look-ahead form (more rows:
    the node slots no longer fit), whose cooperative panel kernel trusts arrival counters zeroed once per sweep.  The slots
    used to run over the second scratch copy's counters (round-3 advisor); the two headers now sit in front of both bodies.
    4128 x 320 -> tree (slots reach into the second copy), 8192 x 320 -> look-ahead, twice; every R against LAPACK."""
    rng = np.random.default_rng(26)
    rows, cols = [4128, 8192, 4128, 8192], 320
    mats = [rng.standard_normal((r, cols)) * np.logspace(0, -8, cols)[None, :] for r in rows]
    A = np.concatenate([np.asfortranarray(m).ravel(order="F") for m in mats])
    R = np.zeros(sum(min(r, cols) * cols for r in rows))
    path = np.zeros(len(rows), dtype=np.int32)
    rws = np.array(rows, dtype=np.int32)
    lib = mpbp_amd._lib.lib()
    rc = lib.mpbp_selftest_qr_batched_seq(0, len(rows), rws.ctypes.data_as(C.POINTER(C.c_int32)), cols, _dp(A), _dp(R),
                                          path.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 0, lib.mpbp_last_error(None)
    assert list(path) == [1, 2, 1, 2], path          # the sequence really alternates between the two forms
    off = 0
    for s, m in enumerate(mats):
        k = min(rows[s], cols)
        Rs = R[off:off + k * cols].reshape((k, cols), order="F")
        off += k * cols
        Rn = np.linalg.qr(m, mode="r")
        sgn = np.sign(np.diag(Rn)) * np.sign(np.diag(Rs))
        sgn[sgn == 0] = 1
        assert np.abs(Rs * sgn[:, None] - Rn).max() < 1e-11 * np.abs(Rn).max(), s


def test_qr_batched_many_problems_fused_update():
    """Enough problems for the wave-per-tile-pair fused trailing update (k_trail4f) and the one-launch register panels
    with Gram + T (k_fpanel), the path a configs[1]-sized batch takes in the batched gauge sweep."""
    rng = np.random.default_rng(23)
    for (r, c) in [(200, 150), (352, 208)]:
        A = rng.standard_normal((600, r, c)) * np.logspace(0, -8, c)[None, None, :]
        R, _ = _qr_batched(A, 0)
        for p in (0, 17, 599):
            Rn = np.linalg.qr(A[p], mode="r")
            sgn = np.sign(np.diag(Rn)) * np.sign(np.diag(R[p]))
            sgn[sgn == 0] = 1
            assert np.abs(R[p] * sgn[:, None] - Rn).max() < 1e-11 * np.abs(Rn).max(), (r, c, p)


@pytest.mark.parametrize("m,n", [(4, 4), (9, 7), (40, 40), (80, 33), (360, 360), (660, 500)])
def test_jacobi_grid_singular_values(m, n):
    """The multi-launch one-sided Jacobi of the batched truncating sweep (k_jac_round / k_jac_check; configs[2]'s
    660-column factors) on lower-triangular matrices with singular values spread over 22 decades, as the engine meets
    them: converges in a few sweeps to LAPACK's singular values."""
    rng = np.random.default_rng(31)
    U, _ = np.linalg.qr(rng.standard_normal((m, m)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.logspace(0, -22, n)
    Mx = (U[:, :n] * s) @ V.T
    # JA = R2^T with R2 the triangular / trapezoidal factor of M^T (M: m x n), as k_svd_trunc builds it
    A = np.asfortranarray(np.linalg.qr(Mx.T, mode="r").T.copy())
    sig = np.zeros(n)
    sw = C.c_int32(0)
    rc = mpbp_amd._lib.lib().mpbp_selftest_jacobi_grid(0, A.shape[0], n, _dp(A), _dp(sig), 30, C.byref(sw))
    assert rc == 0
    assert 0 < sw.value <= 14, sw.value
    ref = np.linalg.svd(A, compute_uv=False)
    got = np.sort(sig)[::-1]
    assert np.abs(got - ref).max() < 1e-13 * ref[0]


# shapes: one lone block (n <= nb), two blocks, an odd number of blocks with a ragged last block, configs[3]'s 160 x 160
# factor (eight blocks of 20), configs[2] hub factors (360 x 360; 660 x 500: block pairs of 14 columns just fit the LDS)
@pytest.mark.parametrize("m,n", [(4, 4), (9, 7), (40, 40), (80, 33), (96, 96), (161, 130), (160, 160), (360, 360), (660, 500)])
def test_jacobi_block_singular_values_and_orthogonality(m, n):
    """The two-level (block) one-sided Jacobi of the batched truncating sweep (v2::k_jac_block: column blocks, block pairs
    rotated LDS resident, one launch per round of the block tournament) on the matrices the engine meets - transposed
    triangular factors with singular values over 22 decades: LAPACK's singular values, within the sweep budget."""
    rng = np.random.default_rng(33)
    U, _ = np.linalg.qr(rng.standard_normal((m, m)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.logspace(0, -22, n)
    Mx = (U[:, :n] * s) @ V.T
    A = np.asfortranarray(np.linalg.qr(Mx.T, mode="r").T.copy())
    sig = np.zeros(n)
    sw = C.c_int32(0)
    lib = mpbp_amd._lib.lib()
    rc = lib.mpbp_selftest_jacobi_block(0, A.shape[0], n, _dp(A), _dp(sig), 30, C.byref(sw))
    assert rc == 0, lib.mpbp_last_error(None)
    assert 0 < sw.value <= 14, sw.value
    ref = np.linalg.svd(A, compute_uv=False)
    got = np.sort(sig)[::-1]
    assert np.abs(got - ref).max() < 1e-13 * ref[0]
