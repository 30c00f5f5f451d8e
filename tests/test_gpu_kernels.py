"""-m gpu: every hand-written HIP kernel against numpy/scipy (FP64, tolerance 1e-12 relative unless
stated), called through the C ABI of libgeneopc.so."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from geneo4petsc_amd import _lib
    return _lib.load()          # raises if the HIP library is missing: no fallback


def _rand_csr(n, density, seed, long_row=None):
    rng = np.random.default_rng(seed)
    nnz = max(1, int(density * n * n))
    rows = rng.integers(0, n, size=nnz)           # (sp.random is quadratic in n on this scipy)
    cols = rng.integers(0, n, size=nnz)
    a = sp.csr_matrix((rng.random(nnz) - 0.5, (rows, cols)), shape=(n, n)) + sp.diags(rng.random(n) + 1.0)
    if long_row is not None:
        a = a.tolil()
        a[long_row, :] = rng.random(n)
        a = a.tocsr()
    a = a.tocsr()
    a.sum_duplicates()
    a.sort_indices()
    return a


def _band_csr(n, per_row, half_width, seed):
    """random entries within `half_width` columns of the diagonal (what a locally numbered mesh operator looks like)"""
    rng = np.random.default_rng(seed)
    nnz = n * per_row
    rows = rng.integers(0, n, size=nnz)
    cols = np.clip(rows + rng.integers(-half_width, half_width + 1, size=nnz), 0, n - 1)
    a = sp.csr_matrix((rng.random(nnz) - 0.5, (rows, cols)), shape=(n, n)) + sp.diags(rng.random(n) + 1.0)
    a = a.tocsr()
    a.sum_duplicates()
    a.sort_indices()
    return a


def test_backend_is_hip(lib):
    assert lib.GeneoBackendName() == b"hip-gfx950"


def test_mfma_lane_map(lib):
    assert lib.GeneoSelfTestMFMA() == 0


@pytest.mark.parametrize("kind", [1, 0, 21, 31])     # sliced (default policy), LDS row blocks, cached / non-temporal stream
@pytest.mark.parametrize("n,density,long_row", [(1, 1.0, None), (257, 0.02, None), (5000, 0.002, None),
                                                (4000, 0.001, 17), (70000, 0.0001, None), (130, 0.6, None)])
def test_spmv(lib, n, density, long_row, kind):
    from geneo4petsc_amd.pc import Spmv
    a = _rand_csr(n, density, 1, long_row)
    x = np.random.default_rng(2).random(n) - 0.5
    lib.GeneoSetSpmvKind(kind)
    try:
        y = Spmv(a, lib).apply(x)
    finally:
        lib.GeneoSetSpmvKind(1)
    np.testing.assert_allclose(y, a @ x, rtol=1e-13, atol=1e-13)


def test_spmv_large_matrix_with_long_rows(lib):
    """>= 2^18 rows: rows longer than 64 leave the sliced layout and go through the long-row kernel;
    below that size they stay in the slices (4000-row case above, also fused epilogues)."""
    from geneo4petsc_amd.pc import Spmv
    n = (1 << 18) + 77
    rng = np.random.default_rng(21)
    rows = np.concatenate([np.repeat(np.arange(n), 3), np.full(900, 5), np.full(3000, n - 2)])
    cols = rng.integers(0, n, size=len(rows))
    a = sp.csr_matrix((rng.random(len(rows)) - 0.5, (rows, cols)), shape=(n, n))
    a.sum_duplicates()
    a.sort_indices()
    x = rng.random(n) - 0.5
    np.testing.assert_allclose(Spmv(a, lib).apply(x), a @ x, rtol=1e-12, atol=1e-12)


def test_spmv_empty_rows(lib):
    from geneo4petsc_amd.pc import Spmv
    a = sp.csr_matrix(([1.0, 2.0], ([0, 3], [1, 2])), shape=(5, 5))
    y = Spmv(a, lib).apply(np.arange(1.0, 6.0))
    np.testing.assert_allclose(y, a @ np.arange(1.0, 6.0))


def test_spmv_laplacian_7pt(lib):
    from geneo4petsc_amd import decomp
    from geneo4petsc_amd.pc import Spmv
    a = decomp.global_matrix(decomp.grid_mesh(n=40, dim=3))
    x = np.random.default_rng(3).random(a.shape[0])
    # rows of the Laplacian nearly cancel: compare with an absolute floor scaled by |A||x|
    np.testing.assert_allclose(Spmv(a, lib).apply(x), a @ x, rtol=1e-13, atol=1e-14 * float(abs(a).max()) * 7)


@pytest.mark.parametrize("m", [1, 16, 20, 32, 64])
def test_spmm(lib, m):
    from geneo4petsc_amd.pc import Spmv
    a = _rand_csr(3000, 0.003, 4)
    rng = np.random.default_rng(5)
    X = rng.random((3000, m)) - 0.5
    pre, post = rng.random(3000) + 0.5, rng.random(3000) + 0.5
    h = Spmv(a, lib)
    np.testing.assert_allclose(h.spmm(X), a @ X, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(h.spmm(X, pre, post), post[:, None] * (a @ (pre[:, None] * X)), rtol=1e-12,
                               atol=1e-13)


def test_spmm_slice_schedules_natural_blobs_strips(lib, monkeypatch, capfd):
    """Traversal order of the sliced SpMM (backend_hip.hip: blob_schedule / strip_schedule): compact blobs of the slice graph,
    or -- one far offset per block, a 3-D stencil in natural order -- strips of a grid plane walked through all planes.  A
    schedule is a permutation of the slices and nothing else: the products of the natural order, bit for bit, for two
    blocks of different plane sizes in one flat matrix (two runs of equal far offset).  The tile and strip sizes are lowered
    so that a 40 x 40 x 36 and a 32 x 32 x 40 block reach the schedules the 187^3 blocks of the benchmark take."""
    from geneo4petsc_amd.pc import Spmv

    def stencil(nx, ny, nz, seed):
        def t(k):
            e = np.ones(k)
            return sp.diags([e[:-1], e, e[:-1]], [-1, 0, 1])
        a = (sp.kron(sp.kron(sp.identity(nz), sp.identity(ny)), t(nx)) + sp.kron(sp.kron(sp.identity(nz), t(ny)), sp.identity(nx))
             + sp.kron(sp.kron(t(nz), sp.identity(ny)), sp.identity(nx))).tocsr()
        a.sort_indices()
        a.data = np.random.default_rng(seed).random(a.nnz) - 0.5
        return a
    a = sp.block_diag([stencil(40, 40, 36, 1), stencil(32, 32, 40, 2)], format="csr")
    a.sort_indices()
    n = a.shape[0]
    X = np.random.default_rng(3).random((n, 32)) - 0.5
    monkeypatch.setenv("GENEO_SPMM_TILE", "16")
    monkeypatch.setenv("GENEO_SPMM_STRIP", "256")
    monkeypatch.setenv("GENEO_DEBUG_SCHED", "1")
    out = {}
    for mode in ("natural", "blob", "strip"):
        if mode == "strip":
            monkeypatch.delenv("GENEO_SPMM_SCHED", raising=False)      # auto: strips where one far offset is found
        else:
            monkeypatch.setenv("GENEO_SPMM_SCHED", mode)
        capfd.readouterr()
        h = Spmv(a, lib)
        out[mode] = h.spmm(X)
        err = capfd.readouterr().err
        assert ("[sched] strip" in err) == (mode == "strip"), (mode, err[-300:])
    np.testing.assert_allclose(out["natural"], a @ X, rtol=1e-12, atol=1e-13)
    np.testing.assert_array_equal(out["blob"], out["natural"])
    np.testing.assert_array_equal(out["strip"], out["natural"])


@pytest.mark.parametrize("density", [0.002, 0.012])
@pytest.mark.parametrize("m", [1, 16, 32, 40])
def test_fused_multigrid_epilogues(lib, m, density):
    """RES / ADD / JAC / PRE epilogues on the SpMV (m = 1: sliced kernel for short rows, lanes-per-row kernel for
    ragged ones, avg >= 20 nnz/row) and SpMM launches vs the unfused algebra."""
    from geneo4petsc_amd.pc import Spmv
    n = 5000
    a = _rand_csr(n, density, 9)
    rng = np.random.default_rng(10)
    shape = (n,) if m == 1 else (n, m)
    X, B, Z = rng.random(shape) - 0.5, rng.random(shape) - 0.5, rng.random(shape) - 0.5
    dinv, w = rng.random(n) + 0.5, 0.61
    d = dinv if m == 1 else dinv[:, None]
    h = Spmv(a, lib)
    tol = dict(rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(h.fused(1, X=X, B=B)[0], B - a @ X, **tol)
    np.testing.assert_allclose(h.fused(2, X=X, Z=Z)[0], Z + a @ X, **tol)
    np.testing.assert_allclose(h.fused(3, X=X, B=B, dinv=dinv, w=w)[0], X + w * d * (B - a @ X), **tol)
    y, z = h.fused(4, B=B, dinv=dinv, w=w)
    np.testing.assert_allclose(z, w * d * B, **tol)
    np.testing.assert_allclose(y, B - a @ (w * d * B), **tol)
    np.testing.assert_allclose(h.fused(5, X=X, B=B, Z=Z, dinv=dinv, w=w)[0], w * d * (Z + B) + a @ X, **tol)   # EPI_POST


@pytest.mark.parametrize("n,per_row", [(5000, 30), (8200, 22), (640, 45)])
@pytest.mark.parametrize("m", [1, 16, 32])
def test_wide_slice_kernels(lib, m, n, per_row):
    """k_spmv_sell_wide / k_spmm_sell_wide (one workgroup per slice: coarse Galerkin operators and restrictions of
    20-60 entries per row on more than 2^18 rows in production): plain product, row/column scalings and the four
    multigrid epilogues.  Kind 101 keeps small ragged matrices on the slices instead of the lanes-per-row kernel."""
    from geneo4petsc_amd.pc import Spmv
    a = _rand_csr(n, per_row / n, 21)
    assert np.diff(a.indptr).max() <= 64 and a.nnz >= 16 * n
    rng = np.random.default_rng(22)
    shape = (n,) if m == 1 else (n, m)
    X, B, Z = rng.random(shape) - 0.5, rng.random(shape) - 0.5, rng.random(shape) - 0.5
    dinv, w = rng.random(n) + 0.5, 0.61
    d = dinv if m == 1 else dinv[:, None]
    tol = dict(rtol=1e-12, atol=1e-13)
    lib.GeneoSetSpmvKind(101)
    try:
        h = Spmv(a, lib)
        if m == 1:
            np.testing.assert_allclose(h.apply(X), a @ X, **tol)
        else:
            pre, post = rng.random(n) + 0.5, rng.random(n) + 0.5
            np.testing.assert_allclose(h.spmm(X), a @ X, **tol)
            np.testing.assert_allclose(h.spmm(X, pre, post), post[:, None] * (a @ (pre[:, None] * X)), **tol)
        np.testing.assert_allclose(h.fused(1, X=X, B=B)[0], B - a @ X, **tol)
        np.testing.assert_allclose(h.fused(2, X=X, Z=Z)[0], Z + a @ X, **tol)
        np.testing.assert_allclose(h.fused(3, X=X, B=B, dinv=dinv, w=w)[0], X + w * d * (B - a @ X), **tol)
        y, z = h.fused(4, B=B, dinv=dinv, w=w)
        np.testing.assert_allclose(z, w * d * B, **tol)
        np.testing.assert_allclose(y, B - a @ (w * d * B), **tol)
        np.testing.assert_allclose(h.fused(5, X=X, B=B, Z=Z, dinv=dinv, w=w)[0], w * d * (Z + B) + a @ X, **tol)
    finally:
        lib.GeneoSetSpmvKind(1)


@pytest.mark.parametrize("n,per_row,kind,half_width", [(5000, 8, 1, 20000), (300000, 6, 1, 20000), (300000, 6, 1, 150000),
                                                       (5000, 30, 101, 2000), (640, 45, 101, 300)])
@pytest.mark.parametrize("span_max", [None, 2500, 450, 40])
def test_single_precision_companion(lib, n, per_row, kind, half_width, span_max, monkeypatch):
    """k_spmv_sell_lp (wave-per-slice and workgroup-per-slice forms): the product and the four epilogues against the
    FP64 algebra on the matrix ROUNDED to float (the companion stores float values, arithmetic is FP64: agreement to
    1e-12), and against the unrounded matrix to single precision.  span_max lowers the column span a 16-bit offset may
    cover (GENEO_LP_SPAN_MAX, 65535 by default) so that these small matrices take the two-base layout (k_lp_base: slices
    of the 2000- and 300-wide bands at 2500 / 450) and the 32-bit fallback (40) as well: same results in every layout."""
    from geneo4petsc_amd.pc import Spmv
    if span_max is not None:
        monkeypatch.setenv("GENEO_LP_SPAN_MAX", str(span_max))
    a = _band_csr(n, per_row, half_width, 31)
    assert np.diff(a.indptr).max() <= 64
    a32 = a.copy()
    a32.data = a32.data.astype(np.float32).astype(np.float64)
    rng = np.random.default_rng(32)
    X, B, Z = rng.random(n) - 0.5, rng.random(n) - 0.5, rng.random(n) - 0.5
    dinv, w = rng.random(n) + 0.5, 0.61
    tol = dict(rtol=1e-12, atol=1e-13)
    lib.GeneoSetSpmvKind(kind)
    try:
        h = Spmv(a, lib)
        y = h.fused_single(0, X=X)[0]
        np.testing.assert_allclose(y, a32 @ X, **tol)
        np.testing.assert_allclose(y, a @ X, rtol=0, atol=1e-6 * np.abs(a @ X).max())
        np.testing.assert_allclose(h.fused_single(1, X=X, B=B)[0], B - a32 @ X, **tol)
        np.testing.assert_allclose(h.fused_single(2, X=X, Z=Z)[0], Z + a32 @ X, **tol)
        np.testing.assert_allclose(h.fused_single(3, X=X, B=B, dinv=dinv, w=w)[0], X + w * dinv * (B - a32 @ X), **tol)
        y, z = h.fused_single(4, B=B, dinv=dinv, w=w)
        np.testing.assert_allclose(z, w * dinv * B, **tol)
        np.testing.assert_allclose(y, B - a32 @ (w * dinv * B), **tol)
        np.testing.assert_allclose(h.fused_single(5, X=X, B=B, Z=Z, dinv=dinv, w=w)[0], w * dinv * (Z + B) + a32 @ X, **tol)
    finally:
        lib.GeneoSetSpmvKind(1)


def test_spmv_with_16bit_column_offsets(lib):
    """Once a matrix has its companion, the FP64 SpMV of a large matrix reads the companion's 16-bit column offsets
    (10 bytes per entry) with the FP64 values: same result as the 32-bit path to rounding."""
    from geneo4petsc_amd.pc import Spmv
    n = 700000
    a = _band_csr(n, 6, 20000, 51)
    x = np.random.default_rng(52).random(n) - 0.5
    h = Spmv(a, lib)
    y32 = h.apply(x)
    h.fused_single(0, X=x)            # builds the companion
    y16 = h.apply(x)
    np.testing.assert_allclose(y32, a @ x, rtol=1e-13, atol=1e-13)
    np.testing.assert_array_equal(y16, y32)     # same entries in the same order: bitwise


def test_two_column_bases_for_slices_wider_than_16_bits(lib, capfd, monkeypatch):
    """A 64-row slice of one 187^3 block per GPU (the metric's layout: 368^3 on 8 GPUs) spans 2 x 187^2 + 64 = 70 002
    columns -- more than one 16-bit base covers.  Such slices carry TWO bases (k_lp_base: the first ks entries of every
    row count from the first, the others from the second), here on a 7-point pattern with that plane distance and with
    the short rows of block faces (padding behind the row's entries).  The FP64 SpMV over the 16-bit offsets equals the
    32-bit-column kernel to the bit, the single-precision companion equals the float-rounded matrix to 1e-12, and the
    debug line says that the two-base layout (not the 32-bit fallback) is what ran."""
    from geneo4petsc_amd.pc import Spmv
    n, nx, plane = 700000, 187, 187 * 187
    rng = np.random.default_rng(77)
    i = np.arange(n)
    rows, cols = [i], [i]
    for off, keep in ((-plane, i >= plane), (plane, i + plane < n), (-nx, (i % plane) >= nx), (nx, (i % plane) < plane - nx),
                      (-1, i % nx != 0), (1, i % nx != nx - 1)):
        keep = keep & (i + off >= 0) & (i + off < n)
        rows.append(i[keep]); cols.append(i[keep] + off)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    a = sp.csr_matrix((rng.random(len(rows)) + 0.5, (rows, cols)), shape=(n, n)).tocsr()
    a.sort_indices()
    assert a.nnz * 12 > 48e6           # large enough for the non-temporal FP64 kernel that reads the 16-bit offsets
    a32 = a.copy()
    a32.data = a32.data.astype(np.float32).astype(np.float64)
    x, b = rng.random(n) - 0.5, rng.random(n) - 0.5
    monkeypatch.setenv("GENEO_DEBUG", "1")
    h = Spmv(a, lib)
    y32 = h.apply(x)
    ylp = h.fused_single(0, X=x)[0]            # builds the companion
    y16 = h.apply(x)
    err = capfd.readouterr().err
    assert "slices with two column bases" in err and "32-bit columns kept" not in err, err
    np.testing.assert_allclose(y32, a @ x, rtol=1e-13, atol=1e-13)
    np.testing.assert_array_equal(y16, y32)
    np.testing.assert_allclose(ylp, a32 @ x, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(h.fused_single(1, X=x, B=b)[0], b - a32 @ x, rtol=1e-12, atol=1e-13)


def test_single_precision_companion_keeps_32bit_columns_for_wide_slices(lib):
    """A slice whose columns two bases cannot cover either (row 0: columns 0 and n - 1 with the other rows' diagonal
    entries and padding between them) keeps its 32-bit columns (float values only)."""
    from geneo4petsc_amd.pc import Spmv
    n = 70000
    a = (sp.diags(np.arange(1.0, n + 1)) + sp.csr_matrix((np.full(1, 0.25), ([0], [n - 1])), shape=(n, n))).tocsr()
    x = np.random.default_rng(3).random(n)
    np.testing.assert_allclose(Spmv(a, lib).fused_single(0, X=x)[0], a @ x, rtol=1e-12)


@pytest.mark.parametrize("p,q", [(16, 16), (32, 32), (48, 48), (96, 96), (64, 32), (192, 192), (20, 12)])
@pytest.mark.parametrize("mfma", [1, 0])
def test_gram(lib, p, q, mfma):
    from geneo4petsc_amd.pc import block_kernel
    if not mfma and p * q > 256 * 40:
        pytest.skip("FMA twin covers p*q <= 10240")
    suboff = np.array([0, 1500, 1500 + 1024, 1500 + 1024 + 3333, 6000], dtype=np.int32)
    rng = np.random.default_rng(6)
    S, T = rng.random((6000, p)) - 0.5, rng.random((6000, q)) - 0.5
    lib.GeneoSetMFMA(mfma)
    try:
        G, _ = block_kernel(0, suboff, S, T, lib)
    finally:
        lib.GeneoSetMFMA(1)
    for s in range(4):
        a, b = suboff[s], suboff[s + 1]
        np.testing.assert_allclose(G[s], S[a:b].T @ T[a:b], rtol=1e-12, atol=1e-11)


@pytest.mark.parametrize("p,q", [(64, 96), (32, 96), (128, 64), (24, 12)])
@pytest.mark.parametrize("mfma", [1, 0])
def test_gram_two_left_blocks(lib, p, q, mfma):
    """gram2: [(A W)^T S ; (B W)^T S] in one pass over S (the reduced Gram rows of LOBPCG, p = 64, q = 96)."""
    from geneo4petsc_amd.pc import block_kernel
    rng = np.random.default_rng(41)
    suboff = np.array([0, 700, 1500, 1501, 4000], dtype=np.int32)
    S, T = rng.random((4000, p)) - 0.5, rng.random((4000, q)) - 0.5
    lib.GeneoSetMFMA(mfma)
    try:
        g, _ = block_kernel(2, suboff, S, T, lib)
    finally:
        lib.GeneoSetMFMA(1)
    for s in range(4):
        r = slice(suboff[s], suboff[s + 1])
        np.testing.assert_allclose(g[s], S[r].T @ T[r], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("p,q", [(16, 16), (32, 32), (48, 32), (96, 64), (64, 64), (192, 128), (96, 32), (20, 12)])
@pytest.mark.parametrize("mfma", [1, 0])
def test_block_mul(lib, p, q, mfma):
    from geneo4petsc_amd.pc import block_kernel
    suboff = np.array([0, 777, 777 + 2048, 5000], dtype=np.int32)
    rng = np.random.default_rng(7)
    S = rng.random((5000, p)) - 0.5
    Cm = rng.random((3, p, q)) - 0.5          # asymmetric on purpose (catches row/col swaps)
    lib.GeneoSetMFMA(mfma)
    try:
        Y, _ = block_kernel(1, suboff, S, Cm, lib)
    finally:
        lib.GeneoSetMFMA(1)
    for s in range(3):
        a, b = suboff[s], suboff[s + 1]
        np.testing.assert_allclose(Y[a:b], S[a:b] @ Cm[s], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n,m,k,density", [(3000, 2500, 700, 0.004), (500, 400, 240, 0.05), (64, 64, 64, 0.5), (5, 7, 3, 0.6)])
def test_device_sparse_products(lib, n, m, k, density):
    """C = A B and A^T on the device (multigrid set-up): sorted columns, values vs scipy."""
    from geneo4petsc_amd.pc import sparse_product
    rng = np.random.default_rng(31)
    a = sp.random(n, m, density=density, random_state=rng, format="csr") + sp.eye(n, m, format="csr")
    b = sp.random(m, k, density=min(1.0, 4 * density), random_state=rng, format="csr") + sp.eye(m, k, format="csr")
    a.sort_indices(); b.sort_indices()
    c = sparse_product(a, b, lib)
    ref = (a @ b).tocsr(); ref.sort_indices()
    assert c is not None and c.nnz >= ref.nnz          # structural zeros from cancellation are kept
    for i in range(n):                                   # columns strictly increasing inside every row
        r = c.indices[c.indptr[i]:c.indptr[i + 1]]
        assert np.all(np.diff(r) > 0)
    assert abs(c - ref).max() <= 1e-13 * max(1.0, abs(ref).max())
    t = sparse_product(a, None, lib)
    assert abs(t - a.T.tocsr()).max() == 0.0
    for i in range(m):
        r = t.indices[t.indptr[i]:t.indptr[i + 1]]
        assert np.all(np.diff(r) > 0)


def test_device_sparse_product_forms_agree_bitwise(lib):
    """Numeric pass of the sparse products: hash accumulators (default) == owner-computes scan (round 2), to the bit, on a
    multigrid-like chain A P0, A P, R (A P) -- the sums are taken in the same (k, l) order in both."""
    from geneo4petsc_amd.pc import sparse_product
    n = 20
    e = np.ones(n)
    t = sp.diags([-e[:-1], 2.0001 * e, -e[:-1]], [-1, 0, 1])
    i = sp.identity(n)
    a = (sp.kron(sp.kron(t, i), i) + sp.kron(sp.kron(i, t), i) + sp.kron(sp.kron(i, i), t)).tocsr()
    a.data *= 1.0 + 0.37 * np.sin(np.arange(a.nnz))           # no two products alike
    agg = (np.arange(n ** 3) // 3) // 1
    agg = np.minimum(agg // 3, n ** 3 // 9 - 1)
    p0 = sp.csr_matrix((np.ones(n ** 3), (np.arange(n ** 3), agg)), shape=(n ** 3, n ** 3 // 9))
    P = (p0 - 0.6 * sp.diags(1.0 / a.diagonal()) @ (a @ p0)).tocsr()
    P.sort_indices()
    R = P.T.tocsr()
    R.sort_indices()
    ap = (a @ P).tocsr()
    ap.sort_indices()
    dense_left = sp.csr_matrix(np.random.default_rng(5).random((3, 300)))      # rows longer than the prefix table holds
    dense_right = sp.random(300, 200, density=0.3, random_state=np.random.default_rng(6), format="csr")
    dense_right.sort_indices()
    for x, y in ((a, p0), (a, P), (R, ap), (dense_left, dense_right)):
        y = y.tocsr()
        y.sort_indices()
        assert lib.GeneoSetKernelVariant(b"spgemm_fill_scan", 1) == 0
        try:
            old = sparse_product(x, y, lib)
        finally:
            lib.GeneoSetKernelVariant(b"spgemm_fill_scan", 0)
        new = sparse_product(x, y, lib)
        assert lib.GeneoSetKernelVariant(b"spgemm_small_rows", 0) == 0      # short rows through the hash table as well
        try:
            mid = sparse_product(x, y, lib)
        finally:
            lib.GeneoSetKernelVariant(b"spgemm_small_rows", 1)
        assert old is not None and new is not None and mid is not None
        for other in (new, mid):
            assert np.array_equal(old.indptr, other.indptr) and np.array_equal(old.indices, other.indices)
            assert np.array_equal(old.data.view(np.uint64), other.data.view(np.uint64))
        ref = (x @ y).tocsr()
        assert abs(new - ref).max() <= 1e-13 * abs(ref).max()
    assert lib.GeneoSetKernelVariant(b"no such switch", 1) == 1


@pytest.mark.parametrize("kind,p", [(2, 64), (0, 96)])
def test_gram_streaming_form_agrees_bitwise(lib, kind, p):
    """LOBPCG's two Gram shapes (W rows: p = 32 + 32; full: p = 96; q = 96): k_gram_flat == k_gram_mfma to the bit, ragged
    chunks included (same sequence of 4-row MFMA steps per output tile)."""
    from geneo4petsc_amd.pc import block_kernel
    rng = np.random.default_rng(43)
    suboff = np.array([0, 700, 1500, 1501, 4000, 4000 + 1024 * 3 + 5], dtype=np.int32)
    nrow = int(suboff[-1])
    S, T = rng.random((nrow, p)) - 0.5, rng.random((nrow, 96)) - 0.5
    assert lib.GeneoSetKernelVariant(b"gram_flat", 0) == 0
    try:
        g_old, _ = block_kernel(kind, suboff, S, T, lib)
    finally:
        lib.GeneoSetKernelVariant(b"gram_flat", 1)
    g_new, _ = block_kernel(kind, suboff, S, T, lib)
    assert np.array_equal(g_old.view(np.uint64), g_new.view(np.uint64))
    for s in range(len(suboff) - 1):
        r = slice(suboff[s], suboff[s + 1])
        np.testing.assert_allclose(g_new[s], S[r].T @ T[r], rtol=1e-12, atol=1e-12)


def test_device_sparse_product_reports_overflow(lib):
    """More than 256 distinct columns in one output row: the kernels say so and the caller uses the host product."""
    from geneo4petsc_amd.pc import sparse_product
    a = sp.csr_matrix(np.ones((4, 300)))
    b = sp.eye(300, format="csr")
    assert sparse_product(a, b, lib) is None
    a2 = sp.csr_matrix(np.ones((2, 200)))
    c = sparse_product(a2, sp.eye(200, format="csr"), lib)
    assert c is not None and abs(c - a2).max() == 0.0


def test_fused_lobpcg_update(lib):
    """k_lobpcg_update32: [X' P'] = S C for S, AS, BS with the [P W] product shared between X' and P', and the next
    residual block, against numpy.  C has the structure core.cpp gives it (P columns = X columns on the P / W rows for
    kept pairs, zero otherwise); two subdomains with ragged sizes (chunk tails, a partial 32-row slab)."""
    import ctypes as C
    rng = np.random.default_rng(11)
    suboff = np.array([0, 1500, 1500 + 1061], dtype=np.int32)
    n, ns = int(suboff[-1]), 2
    S, AS, BS = (rng.random((n, 96)) - 0.5 for _ in range(3))
    Cm = np.zeros((ns, 96, 64))
    keep = (rng.random((ns, 32)) > 0.3).astype(np.float64)
    Cm[:, :, :32] = rng.random((ns, 96, 32)) - 0.5
    Cm[:, 32:, 32:] = Cm[:, 32:, :32] * keep[:, None, :]
    lam = rng.random((ns, 32)) + 0.1
    mask = (rng.random((ns, 32)) > 0.2).astype(np.float64)
    T, AT, BT = (np.zeros((n, 96)) for _ in range(3))
    R = np.zeros((n, 32))
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.POINTER(C.c_double))
    args = [S, AS, BS, Cm, keep, lam, mask]
    keepalive = [np.ascontiguousarray(a) for a in args]
    rc = lib.GeneoTestLobpcgUpdate(ns, suboff.ctypes.data_as(C.POINTER(C.c_int)), *[p(a) for a in keepalive],
                                   p(T), p(AT), p(BT), p(R))
    assert rc == 0, lib.PCGenEOGetError(None).decode()
    for s in range(ns):
        rows = slice(suboff[s], suboff[s + 1])
        for src, out in ((S, T), (AS, AT), (BS, BT)):
            np.testing.assert_allclose(out[rows, :64], src[rows] @ Cm[s], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(R[rows], mask[s] * ((AS[rows] @ Cm[s])[:, :32] - lam[s] * (BS[rows] @ Cm[s])[:, :32]),
                                   rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("m", [16, 32, 64])
def test_spmm_dual_two_operators_one_pass(lib, m):
    """bk::spmm_dual (LOBPCG's A W and B W in one pass over W): B's values laid out on A's sliced pattern (pattern(B)
    contained in pattern(A): A_Neu inside A_Dir), both products bit-identical to the separate launches; a B with an entry
    outside A's pattern is refused."""
    import ctypes as C
    from geneo4petsc_amd.pc import Spmv, DeviceVector
    g = 64                              # 7-point pattern (slices 7 entries wide: the wave-per-slice kernels), random values
    e = np.ones(g)
    t = sp.diags([e[:-1], e, e[:-1]], [-1, 0, 1])
    i3 = sp.identity(g)
    a = (sp.kron(sp.kron(t, i3), i3) + sp.kron(sp.kron(i3, t), i3) + sp.kron(sp.kron(i3, i3), t)).tocsr()
    a.sort_indices()
    a.data = np.random.default_rng(7).random(a.nnz) - 0.5
    n = a.shape[0]
    keep = np.random.default_rng(8).random(a.nnz) < 0.8
    b = a.copy()
    b.data = np.where(keep, np.random.default_rng(9).random(a.nnz) - 0.5, 0.0)
    b.eliminate_zeros()
    ha, hb = Spmv(a, lib), Spmv(b, lib)
    ld = 96
    X = np.random.default_rng(10).random((n, ld)) - 0.5
    xd = DeviceVector.from_host(lib, X.ravel())
    y1, y2 = DeviceVector(lib, n * ld), DeviceVector(lib, n * ld)
    assert lib.GeneoSpmmDualTest(ha.h, hb.h, xd.ptr, ld, y1.ptr, y2.ptr, ld, m) == 0
    Y1 = y1.to_host().reshape(n, ld)[:, :m]
    Y2 = y2.to_host().reshape(n, ld)[:, :m]
    np.testing.assert_array_equal(Y2, ha.spmm(np.ascontiguousarray(X[:, :m])))
    np.testing.assert_array_equal(Y1, hb.spmm(np.ascontiguousarray(X[:, :m])))
    np.testing.assert_allclose(Y1, b @ X[:, :m], rtol=1e-12, atol=1e-13)
    assert lib.GeneoSpmmDualTest(hb.h, ha.h, xd.ptr, ld, y1.ptr, y2.ptr, ld, m) == 2      # A is not inside B's pattern


def test_lean_lobpcg_update_and_residual(lib):
    """The two kernels of LOBPCG's lean iteration (core.cpp, `lean`): the basis-only update k_lobpcg_update32<1> -- bit-identical
    to the S part of the three-operand kernel -- and the residual block from X alone, k_spmm_sell_dual<.., .., 1>:
    R = mask .* (A X - B X diag(lam)) per subdomain, equal to what the separate products give (ragged subdomain boundaries
    inside 64-row slices)."""
    import ctypes as C
    from geneo4petsc_amd.pc import Spmv, DeviceVector
    rng = np.random.default_rng(21)
    suboff = np.array([0, 1500, 1500 + 1061], dtype=np.int32)
    n, ns = int(suboff[-1]), 2
    S, AS, BS = (rng.random((n, 96)) - 0.5 for _ in range(3))
    Cm = np.zeros((ns, 96, 64))
    keep = (rng.random((ns, 32)) > 0.3).astype(np.float64)
    Cm[:, :, :32] = rng.random((ns, 96, 32)) - 0.5
    Cm[:, 32:, 32:] = Cm[:, 32:, :32] * keep[:, None, :]
    lam = rng.random((ns, 32)) + 0.1
    mask = (rng.random((ns, 32)) > 0.2).astype(np.float64)
    T, AT, BT, T1 = (np.zeros((n, 96)) for _ in range(4))
    R = np.zeros((n, 32))
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    arrs = [np.ascontiguousarray(a) for a in (S, AS, BS, Cm, keep, lam, mask)]
    assert lib.GeneoTestLobpcgUpdate(ns, ip(suboff), *[p(a) for a in arrs], p(T), p(AT), p(BT), p(R)) == 0
    null = C.POINTER(C.c_double)()
    rc = lib.GeneoTestLobpcgUpdate(ns, ip(suboff), p(arrs[0]), null, null, p(arrs[3]), p(arrs[4]), null, null, p(T1), null, null, null)
    assert rc == 0, lib.PCGenEOGetError(None).decode()
    np.testing.assert_array_equal(T1[:, :64], T[:, :64])
    # residual from X alone on a 7-point pattern with B inside A's pattern; two subdomains cut inside a slice
    g = 24
    e = np.ones(g)
    t = sp.diags([e[:-1], e, e[:-1]], [-1, 0, 1])
    i3 = sp.identity(g)
    a = (sp.kron(sp.kron(t, i3), i3) + sp.kron(sp.kron(i3, t), i3) + sp.kron(sp.kron(i3, i3), t)).tocsr()
    a.sort_indices()
    a.data = rng.random(a.nnz) - 0.5
    b = a.copy()
    b.data = np.where(rng.random(a.nnz) < 0.8, rng.random(a.nnz) - 0.5, 0.0)
    b.eliminate_zeros()
    na = a.shape[0]
    ha, hb = Spmv(a, lib), Spmv(b, lib)
    for m in (16, 32):
        so = np.array([0, 5001, 9000, na], dtype=np.int32)
        lam3 = np.ascontiguousarray(rng.random((3, m)) + 0.1)
        mask3 = np.ascontiguousarray((rng.random((3, m)) > 0.2).astype(np.float64))
        ld = 96
        X = rng.random((na, ld)) - 0.5
        xd = DeviceVector.from_host(lib, X.ravel())
        rd = DeviceVector(lib, na * m)
        rc = lib.GeneoSpmmDualResidualTest(ha.h, hb.h, xd.ptr, ld, rd.ptr, m, m, 3, ip(so), p(lam3), p(mask3))
        assert rc == 0, lib.PCGenEOGetError(None).decode()
        Rd = rd.to_host().reshape(na, m)
        ax, bx = ha.spmm(np.ascontiguousarray(X[:, :m])), hb.spmm(np.ascontiguousarray(X[:, :m]))
        for sd in range(3):
            rows = slice(so[sd], so[sd + 1])
            np.testing.assert_allclose(Rd[rows], mask3[sd] * (ax[rows] - lam3[sd] * bx[rows]), rtol=0, atol=4e-16 * 8)
            assert not Rd[rows][:, mask3[sd] == 0.0].any()


def test_library_threads_run_on_the_librarys_device(lib):
    """The HIP current device is per host thread and a new thread starts on device 0: threads the library starts (the
    level-1 set-up on its side stream, upload helpers) must be bound to the device of the thread that configured the
    library -- on a node with several visible GPUs rank r > 0 would otherwise allocate and launch on GPU 0.  With one
    visible GPU this is a sanity check of the hook; with more, the LAST device is selected first so that a thread left on
    device 0 is caught."""
    nd = lib.GeneoDeviceCount()
    assert nd >= 1
    want = lib.GeneoSetDevice(nd - 1)
    try:
        assert want == nd - 1
        assert lib.GeneoCurrentDevice() == want
        assert lib.GeneoThreadDeviceCheck() == want
    finally:
        assert lib.GeneoSetDevice(0) == 0
    assert lib.GeneoThreadDeviceCheck() == 0


@pytest.mark.parametrize("per_row", [1, 3, 4, 6, 7, 8, 9, 11, 12, 13, 15])
def test_single_precision_companion_two_latency_form_is_bit_identical(lib, per_row):
    """k_spmv_sell_lp, slices of <= 16 entries per row (the fine-level operators of the benchmark are 7 wide, the
    post-smoothing matrices of the V-cycle 12): all (col, val) loads of the slice, then all gathers -- two dependent
    latencies instead of up to eight (lp_row_sum_fixed, lp_row_sum_p16) -- with the products summed in the order of the
    4-step loop they replace: bit-identical for every width, plain product and epilogues, one and two column bases."""
    from geneo4petsc_amd.pc import Spmv
    n = 70000
    rng = np.random.default_rng(per_row)
    offs = sorted(set([0] + list(rng.choice(np.arange(-30000, 30000), size=per_row - 1, replace=False)))) if per_row > 1 else [0]
    a = sp.diags([rng.random(n - abs(o)) - 0.5 for o in offs], offs, format="csr")
    X, B, Z = rng.random(n) - 0.5, rng.random(n) - 0.5, rng.random(n) - 0.5
    dinv, w = rng.random(n) + 0.5, 0.61
    h = Spmv(a, lib)
    res = []
    for on in (1, 0):
        assert lib.GeneoSetKernelVariant(b"lp_fixed", on) == 0
        out = [h.fused_single(0, X=X)[0], h.fused_single(1, X=X, B=B)[0], h.fused_single(2, X=X, Z=Z)[0],
               h.fused_single(3, X=X, B=B, dinv=dinv, w=w)[0], h.fused_single(5, X=X, B=B, Z=Z, dinv=dinv, w=w)[0]]
        out += list(h.fused_single(4, B=B, dinv=dinv, w=w))
        res.append(out)
    lib.GeneoSetKernelVariant(b"lp_fixed", 1)
    for y1, y0 in zip(*res):
        np.testing.assert_array_equal(y1, y0)
    a32 = a.copy()
    a32.data = a32.data.astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(res[0][0], a32 @ X, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("per_row", [1, 2, 5, 7, 8])
@pytest.mark.parametrize("companion", [False, True])
def test_fp64_spmv_two_latency_form_is_bit_identical(lib, per_row, companion):
    """k_spmv_sell<4, ..> on slices of <= 8 entries per row (spmv_row_sum_fixed: all loads, then all gathers, the products
    summed in the 4-step loop's order): the same bits as the loop it replaces, with 32-bit columns and with the companion's
    16-bit column offsets (the form the inner PCG of the local solves launches)."""
    from geneo4petsc_amd.pc import Spmv
    n = 300000                           # large enough for the non-temporal / 16-bit-column policy of the solver
    rng = np.random.default_rng(40 + per_row)
    offs = sorted(set([0] + list(rng.choice(np.arange(-20000, 20000), size=per_row - 1, replace=False)))) if per_row > 1 else [0]
    a = sp.diags([rng.random(n - abs(o)) - 0.5 for o in offs], offs, format="csr")
    x = rng.random(n) - 0.5
    h = Spmv(a, lib)
    if companion:
        h.fused_single(0, X=x)           # builds the companion: the FP64 SpMV then reads its 16-bit column offsets
    out = []
    for on in (1, 0):
        assert lib.GeneoSetKernelVariant(b"lp_fixed", on) == 0
        out.append(h.apply(x))
    lib.GeneoSetKernelVariant(b"lp_fixed", 1)
    np.testing.assert_array_equal(out[0], out[1])
    np.testing.assert_allclose(out[0], a @ x, rtol=1e-12, atol=1e-13)
