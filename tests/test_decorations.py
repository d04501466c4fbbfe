"""System-matrix decorations: what
bfMatBlockDenseAddInplace leaves in the graph when a sparse correction lands on
a butterfly block -- BfMatSum(product, BfMatCooComplex) -- plus BfMatDiagReal
terms (reference src/mat_block_dense.c:458-510, src/mat_sum.c:54-83,
src/mat_coo_complex.c:212-262, src/mat_diag_real.c)."""
import numpy as np
import pytest

from butterfly_amd import _capi
from butterfly_amd.operator import HipOperator
from oracle import bfref
import plan_emulator


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))


def decorated_complex_graph(rng):
    """2 x 2 block grid: [ Sum(P, C)  D01 ; D10  Sum(D11, C2) ] with P a 2-factor product."""
    def c(m, n):
        return rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))
    f0, f1 = c(40, 9), c(9, 50)
    d01, d10, d11 = c(40, 30), c(25, 50), c(25, 30)
    ri = np.array([0, 3, 3, 17, 39]); ci = np.array([1, 0, 44, 9, 49]); cv = c(5, 1)[:, 0]
    ri2 = np.array([2, 2, 24]); ci2 = np.array([5, 6, 29]); cv2 = c(3, 1)[:, 0]
    coo = np.zeros((40, 50), dtype=complex); np.add.at(coo, (ri, ci), cv)
    coo2 = np.zeros((25, 30), dtype=complex); np.add.at(coo2, (ri2, ci2), cv2)
    dense = np.block([[f0 @ f1 + coo, d01], [d10, d11 + coo2]])
    P = bfref.product([bfref.dense_complex(f0), bfref.dense_complex(f1)])
    s00 = bfref.mat_sum([P, bfref.coo_complex(40, 50, ri, ci, cv)])
    s11 = bfref.mat_sum([bfref.dense_complex(d11), bfref.coo_complex(25, 30, ri2, ci2, cv2)])
    G = bfref.block_dense([0, 40, 65], [0, 50, 80], [s00, bfref.dense_complex(d01), bfref.dense_complex(d10), s11])
    return G, dense


def test_oracle_sum_and_coo_against_numpy_and_the_reference_quirk():
    rng = np.random.default_rng(4)
    G, dense = decorated_complex_graph(rng)
    x = rng.standard_normal((80, 2)) + 1j * rng.standard_normal((80, 2))
    assert rel(bfref.mat_mul(G, x), dense @ x) < 1e-13
    # as written, the reference's COO product assigns: with two entries in one row the last one wins
    lib = bfref.load()
    lib.bfrefCooComplexAssignQuirk(1)
    try:
        coo = bfref.coo_complex(3, 3, [1, 1], [0, 2], [2.0, 5.0])
        y = bfref.mat_mul(coo, np.array([1.0, 1.0, 1.0], dtype=complex))
        assert np.allclose(y, [0, 5, 0])
    finally:
        lib.bfrefCooComplexAssignQuirk(0)
    y = bfref.mat_mul(bfref.coo_complex(3, 3, [1, 1], [0, 2], [2.0, 5.0]), np.ones(3, dtype=complex))
    assert np.allclose(y, [0, 7, 0])


def test_plan_walks_sum_and_coo_terms():
    rng = np.random.default_rng(4)
    G, dense = decorated_complex_graph(rng)
    x = rng.standard_normal((80, 3)) + 1j * rng.standard_normal((80, 3))
    op = HipOperator.from_bfmat(G.ptr.value, flags=_capi.FLAG_PLAN_ONLY | _capi.FLAG_ADJOINT)
    assert op.stats()["numStages"] == 2
    assert rel(plan_emulator.run_plan(op, x), dense @ x) < 1e-13
    xt = rng.standard_normal(65) + 1j * rng.standard_normal(65)
    assert rel(plan_emulator.run_plan(op, xt, transpose=True), dense.T @ xt) < 1e-13


def test_plan_walks_real_diag_terms():
    """Real operand: BlockDiag of (dense, BfMatDiagReal); MulVec and RmulVec."""
    rng = np.random.default_rng(9)
    a = rng.standard_normal((12, 7))
    d = rng.standard_normal(9)
    G = bfref.block_diag([bfref.dense_real(a), bfref.diag_real(9, 9, d)])
    dense = np.zeros((21, 16)); dense[:12, :7] = a; dense[12:, 7:] = np.diag(d)
    x = rng.standard_normal(16)
    assert rel(bfref.mat_mul_vec(G, x), dense @ x) < 1e-14
    op = HipOperator.from_bfmat(G.ptr.value, flags=_capi.FLAG_PLAN_ONLY | _capi.FLAG_ADJOINT)
    assert rel(plan_emulator.run_plan(op, x), dense @ x) < 1e-14
    v = rng.standard_normal(21)
    assert rel(plan_emulator.run_plan(op, v, transpose=True), dense.T @ v) < 1e-14
    assert rel(bfref.mat_rmul_vec(G, v), dense.T @ v) < 1e-14


@pytest.mark.gpu
def test_decorated_graph_on_gpu():
    rng = np.random.default_rng(4)
    G, dense = decorated_complex_graph(rng)
    x = rng.standard_normal((80, 2)) + 1j * rng.standard_normal((80, 2))
    op = HipOperator.from_bfmat(G.ptr.value, flags=_capi.FLAG_ADJOINT)
    assert rel(op.apply_host(x), bfref.mat_mul(G, x)) < 1e-12
    xt = rng.standard_normal(65) + 1j * rng.standard_normal(65)
    assert rel(op.apply_transpose_host(xt), dense.T @ xt) < 1e-12
    op.close()
