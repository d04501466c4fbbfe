"""ORACLE (test infrastructure, not product code): the value side of the streamed real butterfly
(`fac_streamer`, BASELINE config 5).

`butterfly_amd/streamer_structure.py` follows the reference's merge-and-split recursion on block
shapes; the one numerical step is a callback.  Here that callback is restated in numpy:

  bfGetTruncatedSvd           reference src/linalg.c:1002-1082 (LAPACKE_dgesvd via src/mat_dense_real.c:1139-1200;
                              numpy's gesdd computes the same singular values to rounding)
  bfTruncSpecGetNumTerms      src/linalg.c:26-35: keep s_k while s_k >= tol * s_0
  W := S V^T (bfMatScaleRows) src/fac.c:690-698, 797-803
  conversion of a block matrix to dense before the SVD   src/mat_dense_real.c:711-799

plus the synthetic input the survey's probe of the real reference streamed (SURVEY.md section 8(c):
Fibonacci-sphere points, 3-D lattice plane waves cos(kappa k_j . x_i + 0.3), monotone frequency labels),
restated from the survey's description so that node counts can be compared with what it recorded.

Parity status: UNPINNED by reference fixtures (the reference's tests hold none for this path); the
structure this produces at N = 4096 / 1024 columns is held to the survey's recorded probe numbers in
tests/test_streamer_structure.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import numpy as np

from butterfly_amd import streamer_structure as ss


def densify(mat):
    """bfMatDenseRealNewFromMatrix (src/mat_dense_real.c:711-799) on the block algebra."""
    if isinstance(mat, ss.Dense):
        if mat.a is None:
            raise ValueError("value-free leaf")
        return np.asarray(mat.a, dtype=np.float64)
    if isinstance(mat, ss.Identity):
        return np.eye(mat.m)
    if isinstance(mat, ss.Product):
        out = None
        for f in mat.factors:
            a = densify(f)
            out = a if out is None else out @ a
        return out
    out = np.zeros((mat.m, mat.n))
    if isinstance(mat, ss.BlockDiag):
        for k, b in enumerate(mat.blocks):
            out[mat.ro[k]:mat.ro[k] + b.m, mat.co[k]:mat.co[k] + b.n] = densify(b)
    elif isinstance(mat, ss.BlockDense):
        for p in range(mat.nbr):
            for q in range(mat.nbc):
                b = mat.blocks[p * mat.nbc + q]
                out[mat.ro[p]:mat.ro[p] + b.m, mat.co[q]:mat.co[q] + b.n] = densify(b)
    elif isinstance(mat, ss.BlockCoo):
        for i, j, b in zip(mat.i0s, mat.j0s, mat.blocks):
            out[i:i + b.m, j:j + b.n] += densify(b)
    else:
        raise TypeError(type(mat))
    return out


def apply(mat, x):
    """y = mat @ x by recursion over the block algebra (what bfMatMulVec computes, in numpy)."""
    if isinstance(mat, ss.Dense):
        return mat.a @ x
    if isinstance(mat, ss.Identity):
        return np.array(x, copy=True)
    if isinstance(mat, ss.Product):
        for f in reversed(mat.factors):
            x = apply(f, x)
        return x
    y = np.zeros(mat.m)
    if isinstance(mat, ss.BlockDiag):
        for k, b in enumerate(mat.blocks):
            y[mat.ro[k]:mat.ro[k] + b.m] = apply(b, x[mat.co[k]:mat.co[k] + b.n])
    elif isinstance(mat, ss.BlockDense):
        for p in range(mat.nbr):
            for q in range(mat.nbc):
                b = mat.blocks[p * mat.nbc + q]
                y[mat.ro[p]:mat.ro[p] + b.m] += apply(b, x[mat.co[q]:mat.co[q] + b.n])
    else:
        for i, j, b in zip(mat.i0s, mat.j0s, mat.blocks):
            y[i:i + b.m] += apply(b, x[j:j + b.n])
    return y


def _blas_threads(n):
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=n)
    except Exception:               # no threadpoolctl: run with whatever the BLAS does
        import contextlib
        return contextlib.nullcontext()


class SvdFactorizer:
    """Truncated SVD with a relative tolerance, as the streamer calls it."""

    def __init__(self, tol=1e-3, record=None):
        self.tol = float(tol)
        self.record = record            # optional list: (rows, cols, depth, col_node, k) per SVD, for fitting a rank model

    def svd(self, block, row_node, col_node, tree):
        a = densify(block)
        with _blas_threads(2):      # hundreds of small SVDs: BLAS thread teams only fight each other (and other processes)
            return self._svd(a, row_node, col_node, tree)

    def _svd(self, a, row_node, col_node, tree):
        try:
            u, s, vt = np.linalg.svd(a, full_matrices=False)            # gesdd
        except np.linalg.LinAlgError:                                   # the reference's own driver (dgesvd) never gives up
            import scipy.linalg
            u, s, vt = scipy.linalg.svd(a, full_matrices=False, lapack_driver="gesvd")
        k = 0
        while k < len(s) and s[k] >= self.tol * s[0]:       # src/linalg.c:29-30
            k += 1
        if self.record is not None:
            self.record.append((a.shape[0], a.shape[1], int(tree.depth[row_node]), int(col_node), k))
        U = ss.Dense(a.shape[0], k, np.ascontiguousarray(u[:, :k]))
        W = ss.Dense(k, a.shape[1], np.ascontiguousarray(s[:k, None] * vt[:k]))
        return k, len(s), U, W


# ---------------------------------------------------------------------------------------------------
# the survey's synthetic eigenband input
# ---------------------------------------------------------------------------------------------------
fibonacci_sphere = ss.fibonacci_sphere


def half_space_lattice(num):
    """Integer wave vectors of a half space, sorted by length (ties in scan order p, q, r), at least `num`."""
    R = 1
    while int(2.0 * np.pi / 3 * R * R * R) < 2 * num:
        R += 1
    R += 2
    ks = []
    for p in range(-R, R + 1):
        for q in range(-R, R + 1):
            for r in range(0, R + 1):
                if r == 0 and (q < 0 or (q == 0 and p <= 0)):
                    continue
                n2 = p * p + q * q + r * r
                if n2 > R * R:
                    continue
                ks.append((np.sqrt(float(n2)), p, q, r))
    return ks


def probe_problem(n, num_cols):
    """(points, wave vectors [num_cols, 3], kappa, frequency labels, wmax) of the survey's streamer probe.
    The lattice is sorted by |k| only; ties keep an order the survey did not record (C qsort is not
    stable), so column order inside a shell of equal |k| is this function's own (scan order)."""
    pts = fibonacci_sphere(n)
    h = np.sqrt(4 * np.pi / n)
    wmax = 0.5 * np.pi / h
    ks = half_space_lattice(num_cols)
    ks.sort(key=lambda t: t[0])             # stable: scan order within a shell
    if len(ks) < num_cols:
        raise ValueError("lattice too small")
    kap = wmax / ks[num_cols - 1][0] * 0.999
    kv = np.array([[p, q, r] for _, p, q, r in ks[:num_cols]], dtype=np.float64)
    w = wmax * (np.arange(num_cols) + 0.5) / num_cols
    return pts, kv, kap, w, wmax


def stream_probe_case(n, num_cols, tol=1e-3, freq_depth=None, record=None, min_rows=20, min_cols=20):
    """Run the streamer on the probe problem.  Returns (streamer, Phi) with Phi the dense N x J matrix,
    rows in row-tree order, columns in streaming order."""
    pts, kv, kap, w, wmax = probe_problem(n, num_cols)
    tree = ss.Octree(pts, 1)
    if freq_depth is None:
        freq_depth = tree.max_depth - 3         # examples/covariance/lbo_cov.c:97-98
    fz = SvdFactorizer(tol, record)
    st = ss.Streamer(tree, freq_depth, fz, min_rows, min_cols, 0)
    x = pts[tree.perm]
    phi = np.cos(kap * (x @ kv.T) + 0.3)
    j0 = 0
    while not st.is_done():
        leaf = st.current_col_node()
        a, b, left, right = st.cols.leaf_interval(leaf, 0.0, wmax)
        lo = -1e300 if left else a
        hi = 1e300 if right else b
        j1 = j0
        while j1 < num_cols and lo <= w[j1] < hi:
            j1 += 1
        if j1 == j0:
            raise RuntimeError("empty leaf band")
        st.feed(ss.Dense(n, j1 - j0, phi[:, j0:j1]))
        j0 = j1
    return st, phi[:, :j0]


# ---------------------------------------------------------------------------------------------------
# Laplace-Beltrami eigenvectors of the sphere (what examples/covariance streams, without a mesh)
# ---------------------------------------------------------------------------------------------------
def sphere_lbo_problem(n, lmax):
    """(points, Phi [n, (lmax+1)^2], freqs): real spherical harmonics up to degree lmax sampled at the
    Fibonacci points -- the Laplace-Beltrami eigenfunctions of the unit sphere, eigenvalue l (l + 1), i.e.
    what bfLboFeedFacStreamerNextEigenband (src/lbo.c:70-150) would stream for a fine mesh of the sphere,
    with freqs = sqrt(eigenvalue) (src/lbo.c:20-30), ascending."""
    from scipy.special import sph_harm_y
    pts = fibonacci_sphere(n)
    theta = np.arccos(np.clip(pts[:, 2], -1, 1))         # polar
    phi = np.arctan2(pts[:, 1], pts[:, 0])               # azimuth
    cols, freqs = [], []
    for l in range(lmax + 1):
        for m in range(-l, l + 1):
            y = sph_harm_y(l, abs(m), theta, phi)
            if m == 0:
                v = y.real
            elif m > 0:
                v = np.sqrt(2.0) * y.real
            else:
                v = np.sqrt(2.0) * y.imag
            cols.append(v)
            freqs.append(np.sqrt(l * (l + 1.0)))
    return pts, np.stack(cols, axis=1), np.asarray(freqs)


def stream_columns(pts, phi, freqs, wmax, freq_depth, tol=1e-3, record=None, min_rows=20, min_cols=20, max_cols=None, wmin=0.0):
    """The streaming loop of examples/covariance/lbo_cov.c:139-143 + src/lbo.c:70-150: feed, leaf by leaf of
    the frequency tree over [wmin, wmax], the columns whose frequency falls in the leaf's bracket
    ([a, b), open-ended at both ends of the tree: src/lbo.c:41-68); stop when the tree is exhausted or, as
    lbo_cov.c:141 does with `numEigs`, once `max_cols` columns went in.  Rows of phi in file order.
    Returns (streamer, Phi in tree order restricted to the streamed columns)."""
    tree = ss.Octree(pts, 1)
    st = ss.Streamer(tree, freq_depth, SvdFactorizer(tol, record), min_rows, min_cols, 0)
    a_phi = np.ascontiguousarray(phi[tree.perm])
    j0 = 0
    J = phi.shape[1]
    while not st.is_done():
        leaf = st.current_col_node()
        a, b, left, right = st.cols.leaf_interval(leaf, wmin, wmax)        # bfIntervalTreeInitEmpty(tree, a, b, 2, depth)
        lo = -np.inf if left else a
        hi = np.inf if right else b
        j1 = j0
        while j1 < J and lo <= freqs[j1] < hi:
            j1 += 1
        st.feed(ss.Dense(len(pts), j1 - j0, a_phi[:, j0:j1]))
        j0 = j1
        if max_cols is not None and j0 >= max_cols:
            break
    return st, a_phi[:, :j0]
