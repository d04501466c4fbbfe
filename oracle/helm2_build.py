"""ORACLE (test infrastructure, not product code).

Numpy restatement of the *value* side of the reference's Helmholtz butterfly
builder, used only to manufacture real operands and known answers for parity
tests (the engine itself never builds factorizations):

  kernel matrix (single layer, (i/4) H0(k r), 0 on r == 0)   reference src/helm2.c:93-125
  S' kernel matrix ((i/4) k H1(k r)/r n_tgt.(x_tgt-x_src))    src/helm2.c:126-171
  proxy-circle sampling                                       src/circle.c:12-35
  re-expansion matrix Z_equiv \\ Z_orig                        src/helm2.c:321-365
  truncated-SVD least squares (rtol = max(m,n) eps)           src/mat_dense_complex.c:1767-1849
  seeded complex normal RHS (xoshiro256+ / Box-Muller)        src/rand.c:19-76,
                                                              src/splitmix64.c, src/xoshiro256plus.c

H0 is evaluated with scipy's j0/y0 (the reference uses libm's, src/bessel.c:284-296);
the two agree to a few ulp, far below the butterfly's own 1e-10 truncation.

PARITY STATUS: see oracle/bfref.h ("parity unpinned" by reference goldens; pinned
by dense-kernel known answers and the survey's ||y||^2 checksums).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import j0, j1, y0, y1

EPS_MACH = 2.220446049250313e-16
TWO_PI = 6.283185307179586


def sample_circle(cx, cy, r, count):
    """bfCircle2SamplePoints, src/circle.c:12-35."""
    scale = TWO_PI / float(count)
    theta = scale * np.arange(count)
    return np.stack([r * np.cos(theta) + cx, r * np.sin(theta) + cy], axis=1)


def resolve_points(spec, tree_points, tgt_tree_points=None):
    if spec[0] == "node":
        return tree_points[spec[1]:spec[2]]
    if spec[0] == "tnode":                      # a range of a separate target tree's points
        return tgt_tree_points[spec[1]:spec[2]]
    if spec[0] == "circle":
        return sample_circle(spec[1], spec[2], spec[3], spec[4])
    raise ValueError(spec)


def kernel_matrix(k, src, tgt):
    """get_S_kernel_matrix, src/helm2.c:93-125: rows = targets, cols = sources."""
    r = np.hypot(tgt[:, None, 0] - src[None, :, 0], tgt[:, None, 1] - src[None, :, 1])
    z = np.zeros(r.shape, dtype=np.complex128)
    nz = r != 0
    kr = k * r[nz]
    z[nz] = 0.25j * (j0(kr) + 1j * y0(kr))
    return z


def kernel_matrix_sp(k, src, tgt, ntgt):
    """get_Sp_kernel_matrix, src/helm2.c:126-171: (i/4) k H1(k r)/r n_tgt.(x_tgt - x_src), 0 on r == 0."""
    dx = tgt[:, None, 0] - src[None, :, 0]
    dy = tgt[:, None, 1] - src[None, :, 1]
    r = np.hypot(dx, dy)
    z = np.zeros(r.shape, dtype=np.complex128)
    nz = r != 0
    dot = (ntgt[:, None, 0] * dx + ntgt[:, None, 1] * dy)[nz]
    kr = k * r[nz]
    z[nz] = 0.25j * k * (j1(kr) + 1j * y1(kr)) / r[nz] * dot
    return z


def kernel_matrix_d(k, src, tgt, nsrc):
    """get_D_kernel_matrix, src/helm2.c:173-218: (i/4) k H1(k r)/r n_src.(x_tgt - x_src), 0 on r == 0."""
    dx = tgt[:, None, 0] - src[None, :, 0]
    dy = tgt[:, None, 1] - src[None, :, 1]
    r = np.hypot(dx, dy)
    z = np.zeros(r.shape, dtype=np.complex128)
    nz = r != 0
    dot = (nsrc[None, :, 0] * dx + nsrc[None, :, 1] * dy)[nz]
    kr = k * r[nz]
    z[nz] = 0.25j * k * (j1(kr) + 1j * y1(kr)) / r[nz] * dot
    return z


def resolve_normals(spec, normals):
    """Unit normals of a point set: stored ones for tree points, radial for a sampled circle
    (bfCircle2SampleUnitNormals, src/circle.c:36-58)."""
    if spec[0] == "node":
        return normals[spec[1]:spec[2]]
    theta = (TWO_PI / float(spec[4])) * np.arange(spec[4])
    return np.stack([np.cos(theta), np.sin(theta)], axis=1)


def layer_matrix(layer_pot, k, src_spec, tgt_spec, tree_points, normals, alpha=0.0, beta=0.0, tgt_tree_points=None):
    """bfHelm2GetKernelMatrix (src/helm2.c:281-318) for S, D and the combined field alpha S + beta D
    (get_S_plus_D_kernel_matrix, :220-279)."""
    src, tgt = resolve_points(src_spec, tree_points), resolve_points(tgt_spec, tree_points, tgt_tree_points)
    if layer_pot == "S":
        return kernel_matrix(k, src, tgt)
    d = kernel_matrix_d(k, src, tgt, resolve_normals(src_spec, normals))
    if layer_pot == "D":
        return d
    assert layer_pot == "combined"
    return alpha * kernel_matrix(k, src, tgt) + beta * d


def lstsq_truncated(lhs, rhs):
    """bfMatDenseComplexDenseComplexLstSq, src/mat_dense_complex.c:1767-1849."""
    m, n = lhs.shape
    u, s, vh = np.linalg.svd(lhs, full_matrices=False)
    tol = max(m, n) * EPS_MACH * s[0] + EPS_MACH
    below = np.nonzero(s < tol)[0]
    kk = int(below[0]) if len(below) else len(s)
    tmp = u[:, :kk].conj().T @ rhs
    tmp /= s[:kk, None]
    return vh[:kk].conj().T @ tmp


def reexpansion_matrix(k, src_orig, src_equiv, tgt):
    """bfHelm2GetReexpansionMatrix, src/helm2.c:321-365."""
    z_orig = kernel_matrix(k, src_orig, tgt)
    z_equiv = kernel_matrix(k, src_equiv, tgt)
    return lstsq_truncated(z_equiv, z_orig)


# Kapur-Rokhlin weights as the reference tabulates them (src/quadrature.c:12-41)
KR_WEIGHTS = {
    2: [1.825748064736159, -1.325748064736159],
    6: [4.967362978287758, -16.20501504859126, 25.85153761832639, -22.22599466791883, 9.930104998037539, -1.817995878141594],
    10: [7.832432020568779, -4.565161670374749, 1.452168846354677, -2.901348302886379, 3.870862162579900, -3.523821383570681,
         2.172421547519342, -8.707796087382991, 2.053584266072635, -2.166984103403823],
}


def kr_factors(order, orig_tgt, orig_src, n):
    """1 + w_KR[d-1] where the original indices are d = 1..order apart on the closed curve, else 1:
    adding w_KR[p] K(i, j) at j = i +- (p+1) mod n (bf_get_KR_corr_block_spmat, src/quadrature.c:126-166)
    is multiplying K(i, j) by that factor."""
    fwd = (np.asarray(orig_tgt, dtype=np.int64)[:, None] - np.asarray(orig_src, dtype=np.int64)[None, :]) % n
    d = np.minimum(fwd, n - fwd)
    f = np.ones(d.shape)
    w = KR_WEIGHTS[order]
    for p in range(order):
        f[d == p + 1] += w[p]
    return f


def leaf_values(desc, k, tree_points, layer_pot="S", normals=None, col_weights=None, self_value=0.0, kr_order=0,
                orig_index=None, alpha=0.0, beta=0.0, tgt_tree_points=None, tgt_normals=None):
    """Evaluate every dense leaf's recipe -> {node: complex128 array}.

    layer_pot "Sp": kernel leaves (evaluation factor src/fac_helm2.c:403-509, dense near field
    :745-760) use S' with the target normals; re-expansions use the proxy potential S
    (BF_PROXY_LAYER_POT, include/bf/layer_pot.h:63-69).  col_weights / self_value fold the
    reference's bfMatScaleCols and bfMatAddInplace(c I) (examples/simple/helm2_bie.c:109-121) into
    the values: the operand is  self_value I + K diag(col_weights)."""
    def scaled(z, src_spec):
        if col_weights is not None and src_spec[0] == "node":
            z = z * col_weights[src_spec[1]:src_spec[2]][None, :]
        return z

    out = {}
    for node, rc in desc.recipe.items():
        if rc[0] == "kernel":
            src, tgt = resolve_points(rc[1], tree_points), resolve_points(rc[2], tree_points, tgt_tree_points)
            if layer_pot == "Sp":
                assert rc[2][0] in ("node", "tnode")
                nt = normals if rc[2][0] == "node" else tgt_normals
                z = kernel_matrix_sp(k, src, tgt, nt[rc[2][1]:rc[2][2]])
            else:
                z = layer_matrix(layer_pot, k, rc[1], rc[2], tree_points, normals, alpha, beta, tgt_tree_points)
            if kr_order and rc[1][0] == "node" and rc[2][0] == "node":
                z = z * kr_factors(kr_order, orig_index[rc[2][1]:rc[2][2]], orig_index[rc[1][1]:rc[1][2]], len(tree_points))
            z = scaled(z, rc[1])
            if rc[1][0] == "node" and rc[2][0] == "node":
                # target point == source point: the identity term of the system matrix
                same = (np.arange(rc[2][1], rc[2][2])[:, None] == np.arange(rc[1][1], rc[1][2])[None, :])
                z = np.where(same, self_value, z)
        elif rc[0] == "reexp":
            # proxy potential: S for S and S', the potential itself for D and the combined field
            proxy = "S" if layer_pot in ("S", "Sp") else layer_pot
            z_eq = layer_matrix(proxy, k, rc[2], rc[3], tree_points, normals, alpha, beta, tgt_tree_points)
            z_or = layer_matrix(proxy, k, rc[1], rc[3], tree_points, normals, alpha, beta, tgt_tree_points)
            z = lstsq_truncated(z_eq, scaled(z_or, rc[1]))
        else:
            raise ValueError(rc)
        assert z.shape == (desc.rows[node], desc.cols[node]), (z.shape, desc.rows[node], desc.cols[node], rc[0])
        out[node] = np.ascontiguousarray(z)
    return out


# --------------------------------------------------------------------------
# the reference's PRNG, to reproduce bfMatDenseComplexNewRandn after bfSeed(0)
# --------------------------------------------------------------------------
_M64 = (1 << 64) - 1


class Xoshiro256Plus:
    def __init__(self, seed: int):
        # bfSeed, src/rand.c:19-33: splitmix64(seed) -> 4 state words
        x = seed & _M64
        s = []
        for _ in range(4):
            x = (x + 0x9E3779B97F4A7C15) & _M64
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
            s.append(z ^ (z >> 31))
        self.s = s

    def next(self) -> int:
        s = self.s
        result = (s[0] + s[3]) & _M64
        t = (s[1] << 17) & _M64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = ((s[3] << 45) | (s[3] >> 19)) & _M64
        return result

    def uniform(self, n):
        # bfRealUniform1, src/rand.c:43-47
        return np.array([(self.next() >> 11) * (2.0 ** -53) for _ in range(n)])


def complex_randn(n, seed=0):
    """bfComplexRandn(n) after bfSeed(seed): Box-Muller over interleaved re/im
    (src/rand.c:53-76)."""
    g = Xoshiro256Plus(seed)
    m = 2 * n
    x = g.uniform(2 * (m // 2))
    u0, u1 = x[0::2].copy(), x[1::2].copy()
    mag = np.sqrt(-2 * np.log(u0))
    theta = TWO_PI * u1
    x[0::2] = mag * np.cos(theta)
    x[1::2] = mag * np.sin(theta)
    return x[0::2] + 1j * x[1::2]
