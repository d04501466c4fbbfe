"""Pins the CPU oracle (oracle/bfref.c + oracle/helm2_build.py).

The reference's own tests hold no fixture for the apply path and the
reference cannot be built here, so the oracle is pinned by
  (1) ||y||^2 checksums the survey recorded from the real reference
      (same points, same quadtree permutation, same seeded RHS), and
  (2) analytic known answers: the dense kernel matrix applied to the same
      vector (the reference examples' own acceptance check,
      examples/simple/bf_all_blocks.c:149-153), committed under tests/golden/.
"""
import json
import os

import numpy as np
import pytest

from butterfly_amd import helm2_structure as hs
from oracle import bfref, helm2_build as hb
from fixtures import load_fixture
import randgraph

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))


def test_prng_restatement_statistics():
    """bfSeed(0) + bfComplexRandn (src/rand.c:19-76): deterministic, unit variance."""
    a = hb.complex_randn(4096, 0)
    b = hb.complex_randn(4096, 0)
    assert np.array_equal(a, b)
    assert abs(np.mean(np.abs(a) ** 2) - 2.0) < 0.1          # re, im ~ N(0,1) each
    assert not np.array_equal(a, hb.complex_randn(4096, 1))
    g = hb.Xoshiro256Plus(0)
    first = [g.next() for _ in range(3)]
    assert all(0 <= v < 2 ** 64 for v in first) and len(set(first)) == 3


def test_reference_checksum_n4096_k100():
    """||A_BF x||^2 recorded from the real reference: 29105.487524831169."""
    stats = json.load(open(os.path.join(GOLD, "survey_probe_stats.json")))
    want = [c for c in stats["cases"] if (c["n"], c["k"]) == (4096, 100)][0]["y_norm2"]
    n, k = 4096, 100
    pts = hs.circle_points(n)
    desc, root, perm = hs.helm2_multilevel_structure(pts, k, recipes=True, exact_sift=True)
    tp = pts[perm]
    vals = hb.leaf_values(desc, k, tp)
    x = hb.complex_randn(n, 0)
    y = bfref.mat_mul(bfref.from_desc(desc, vals), x)
    got = float(np.vdot(y, y).real)
    assert abs(got - want) / want < 1e-10
    y_dense = hb.kernel_matrix(k, tp, tp) @ x
    assert abs(float(np.vdot(y_dense, y_dense).real) - want) / want < 1e-10
    assert rel(y, y_dense) < 1e-10


def test_golden_one_block():
    desc, vals, ex = load_fixture(os.path.join(GOLD, "helm2_one_block_n2048_k128.npz"))
    y = bfref.mat_mul(bfref.from_desc(desc, vals), ex["x"])
    assert rel(y, ex["y_oracle"]) < 1e-14
    assert rel(y, ex["y_dense"]) < 1e-11
    # the stored factor values are what the builder restatement produces today
    pts = hs.circle_points(2048)
    d2, root, perm, sn, tn = hs.single_product_structure(pts, float(ex["k"]), tuple(ex["src_path"]), tuple(ex["tgt_path"]))
    v2 = hb.leaf_values(d2, float(ex["k"]), pts[perm])
    assert d2.rows == desc.rows and d2.cols == desc.cols
    y2 = bfref.mat_mul(bfref.from_desc(d2, v2), ex["x"])
    assert rel(y2, ex["y_oracle"]) < 1e-12
    # Factor by factor.  A KERNEL leaf is a table of Hankel values: entry-wise agreement.  A REEXP leaf is the truncated
    # least-squares solution X of Z_equiv X = Z_orig (src/helm2.c:321-365, src/mat_dense_complex.c:1767-1849): its
    # components along singular vectors with sigma ~ max(m, n) eps sigma_max are amplified rounding noise (they move with
    # the BLAS thread count: 3e-4 entry-wise on one factor under OPENBLAS_NUM_THREADS=2), so X itself is only determined
    # to eps * sigma_max / sigma_cut ~ 1 / max(m, n) -- but Z_equiv X, the field the equivalent sources radiate and the
    # only thing the factorization uses, is backward stable: || Z_equiv (X2 - X1) || <= c eps || Z_orig ||.
    tp = pts[perm]
    for node, rc in d2.recipe.items():
        if rc[0] == "kernel":
            assert np.allclose(v2[node], vals[node], rtol=0, atol=1e-12 * np.abs(vals[node]).max()), node
        else:
            z_eq = hb.kernel_matrix(float(ex["k"]), hb.resolve_points(rc[2], tp), hb.resolve_points(rc[3], tp))
            z_or = hb.kernel_matrix(float(ex["k"]), hb.resolve_points(rc[1], tp), hb.resolve_points(rc[3], tp))
            assert np.linalg.norm(z_eq @ (v2[node] - vals[node])) <= 1e-10 * np.linalg.norm(z_or), node


def test_golden_multilevel_vectors():
    z = np.load(os.path.join(GOLD, "helm2_multilevel_n2048_k128_vectors.npz"))
    n, k = int(z["n"]), float(z["k"])
    pts = hs.circle_points(n)
    desc, root, perm = hs.helm2_multilevel_structure(pts, k, recipes=True)
    assert desc.leaf_elems() == int(z["leaf_elems"])
    assert sum(desc.meta["stats"]["products"].values()) == int(z["num_products"])
    vals = hb.leaf_values(desc, k, pts[perm])
    y = bfref.mat_mul(bfref.from_desc(desc, vals), z["x"])
    assert rel(y, z["y_oracle"]) < 1e-12
    assert rel(y, z["y_dense"]) < 1e-10


def test_golden_real_nested():
    desc, vals, ex = load_fixture(os.path.join(GOLD, "real_nested_small.npz"))
    y = bfref.mat_mul_vec(bfref.from_desc(desc, vals), ex["x"])
    assert rel(y, ex["y_oracle"]) < 1e-14
    assert rel(y, ex["y_dense"]) < 1e-13
    assert rel(randgraph.densify(desc, vals, desc.root) @ ex["x"], ex["y_dense"]) < 1e-13


def test_oracle_block_types_against_numpy():
    """Each container type on its own, built through the reference-named
    constructors, multi-RHS."""
    rng = np.random.default_rng(5)
    def c(m, n):
        return rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))
    a, b, d = c(5, 7), c(3, 4), c(6, 2)
    x = c(13, 3)
    diag = bfref.block_diag([bfref.dense_complex(a), bfref.dense_complex(b), bfref.dense_complex(d)])
    dense = np.zeros((14, 13), dtype=complex)
    dense[:5, :7] = a; dense[5:8, 7:11] = b; dense[8:, 11:] = d
    assert diag.shape == (14, 13)
    assert rel(bfref.mat_mul(diag, x), dense @ x) < 1e-14
    # coo with an empty block row and two blocks in one row
    e, f = c(4, 6), c(4, 3)
    coo = bfref.block_coo([0, 4, 9], [0, 6, 9], [0, 0], [0, 1], [bfref.dense_complex(e), bfref.dense_complex(f)])
    full = np.zeros((9, 9), dtype=complex)
    full[:4, :6] = e; full[:4, 6:] = f
    xx = c(9, 2)
    assert rel(bfref.mat_mul(coo, xx) + 1, full @ xx + 1) < 1e-14
    # product of dense grid and diag
    g = [c(2, 5), c(2, 4), c(3, 5), c(3, 4)]
    grid = bfref.block_dense([0, 2, 5], [0, 5, 9], [bfref.dense_complex(v) for v in g])
    gd = np.block([[g[0], g[1]], [g[2], g[3]]])
    prod = bfref.product([grid, coo])
    assert rel(bfref.mat_mul(prod, xx), gd @ (full @ xx)) < 1e-13
    assert prod.num_bytes() == 16 * (sum(v.size for v in g) + e.size + f.size)


def test_oracle_rmul_walks_a_product_of_dense_factors_in_order():
    """bfMatRmul(P, X) = X F0 F1 F2 (src/mat_product.c:282-310 down to bfMatDenseComplexRmul's one zgemm,
    src/mat_dense_complex.c:1075-1133); a mismatched left operand and a factor type without the slot (every block type: the
    reference's own chain is a NULL call there) are errors."""
    rng = np.random.default_rng(17)
    c = lambda m, n: rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))
    f = [c(9, 6), c(6, 11), c(11, 4)]
    prod = bfref.product([bfref.dense_complex(v) for v in f])
    x = c(5, 9)
    assert rel(bfref.mat_rmul(prod, x), x @ f[0] @ f[1] @ f[2]) < 1e-14
    y = c(3, 6)
    assert rel(bfref.mat_rmul(bfref.dense_complex(f[1]), y), y @ f[1]) < 1e-14
    with pytest.raises(RuntimeError):
        bfref.mat_rmul(prod, c(5, 8))
    diag = bfref.block_diag([bfref.dense_complex(c(4, 4)), bfref.dense_complex(c(5, 5))])
    with pytest.raises(RuntimeError):
        bfref.mat_rmul(bfref.product([diag]), c(2, 9))


@pytest.mark.parametrize("seed", range(6))
def test_oracle_transpose_is_the_conjugate_transpose_for_complex_leaves(seed):
    """bfMatTranspose through the restated slots (Product :409-420, BlockDiag mat_block_diag.c:603-624, BlockDense
    mat_block_dense.c:950-986, Identity, DenseComplex :1475-1478 = bfMatConjTrans): dense complex leaves end up flagged
    TRANS | CONJ and getCblasTranspose (:27-35) turns that into CblasConjTrans, so the transposed operator multiplies as
    A^H; extents swap; twice is the identity.  BlockCoo has no Transpose slot (a NULL call in the reference)."""
    rng = np.random.default_rng(7300 + seed)
    desc, vals = randgraph.random_operand(rng, depth=int(rng.integers(1, 4)), size_hint=40, cplx=True, coo=False)
    m, n = desc.rows[desc.root], desc.cols[desc.root]
    A = bfref.from_desc(desc, vals, typed=True)
    dense = randgraph.densify(desc, vals, desc.root)
    x = rng.standard_normal((m, 3)) + 1j * rng.standard_normal((m, 3))
    xf = rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))
    bfref.mat_transpose(A)
    assert A.shape == (n, m)
    assert rel(bfref.mat_mul(A, x) + 1, dense.conj().T @ x + 1) < 1e-13
    bfref.mat_transpose(A)
    assert A.shape == (m, n)
    assert rel(bfref.mat_mul(A, xf) + 1, dense @ xf + 1) < 1e-13
    coo = bfref.block_coo([0, 4], [0, 6], [0], [0], [bfref.dense_complex(np.ones((4, 6), dtype=complex))])
    with pytest.raises(RuntimeError, match="BfError 3"):
        bfref.mat_transpose(coo)


def test_oracle_error_behaviour():
    """Shape mismatch -> NULL + error code, as the reference (mat_block_coo.c:391-392)."""
    a = bfref.dense_complex(np.ones((3, 4), dtype=complex))
    with pytest.raises(RuntimeError):
        bfref.mat_mul(a, np.ones(5, dtype=complex))
