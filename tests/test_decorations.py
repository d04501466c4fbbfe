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


def test_sparse_terms_are_folded_into_covering_dense_leaves():
    """Sum(whole operator, COO correction + c I) handed over at the ROOT (what a caller gets who builds the
    Kapur-Rokhlin correction and the 1/2 I as terms of their own): every entry that lies over a dense leaf is
    added into that leaf's values when the arena is packed; only entries over butterflied blocks stay
    1 x 1 leaves.  Before this, all of them were (3 M one-element pieces at N = 262144)."""
    rng = np.random.default_rng(21)

    def c(m, n):
        return rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))
    f0, f1 = c(40, 9), c(9, 50)
    d01, d10, d11a, d11b = c(40, 30), c(25, 50), c(25, 12), c(25, 18)
    P = bfref.product([bfref.dense_complex(f0), bfref.dense_complex(f1)])
    nested = bfref.block_dense([0, 25], [0, 12, 30], [bfref.dense_complex(d11a), bfref.dense_complex(d11b)])
    G = bfref.block_dense([0, 40, 65], [0, 50, 80], [P, bfref.dense_complex(d01), bfref.dense_complex(d10), nested])
    dense = np.block([[f0 @ f1, d01], [d10, np.hstack([d11a, d11b])]])
    # entries: 3 over the product block, the rest over dense leaves (one of them nested two levels down), one doubled
    ri = np.array([0, 5, 39, 0, 39, 41, 64, 64, 50, 50]); ci = np.array([0, 49, 7, 50, 79, 3, 49, 79, 55, 55])
    cv = c(len(ri), 1)[:, 0]
    coo = np.zeros((65, 80), dtype=complex); np.add.at(coo, (ri, ci), cv)
    diag = rng.standard_normal(65)
    eye = np.zeros((65, 80)); eye[np.arange(65), np.arange(65)] = diag
    S = bfref.mat_sum([G, bfref.coo_complex(65, 80, ri, ci, cv), bfref.diag_real(65, 80, diag)])
    want = (dense + coo + eye)
    x = rng.standard_normal((80, 2)) + 1j * rng.standard_normal((80, 2))
    # (the oracle, like the reference, has no DiagReal x DenseComplex product: the dense answer is the check)
    op = HipOperator.from_bfmat(S.ptr.value, flags=_capi.FLAG_PLAN_ONLY | _capi.FLAG_ADJOINT)
    st = op.stats()
    # 6 dense leaves + the 1 x 1 leaves that could not be folded: 3 COO entries and 40 diagonal entries over the product
    assert st["numLeaves"] == 6 + 3 + 40, st["numLeaves"]
    assert rel(plan_emulator.run_plan(op, x), want @ x) < 1e-13
    xt = rng.standard_normal(65) + 1j * rng.standard_normal(65)
    assert rel(plan_emulator.run_plan(op, xt, transpose=True), want.T @ xt) < 1e-13


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


@pytest.mark.gpu
def test_kr_style_decoration_costs_nothing_at_n65536():
    """A Kapur-Rokhlin-shaped correction (12 entries per row next to the diagonal, 786 432 entries) plus c I,
    summed onto an ORACLE-built N = 65536 operator at the root: all but a few entries land in dense near-field
    leaves, so the decorated operator keeps the undecorated one's plan and apply time (within a few %)."""
    import time
    import torch
    from butterfly_amd import helm2_structure as hs
    n, k = 65536, 100.0
    desc, _ = hs.native_multilevel_structure(hs.circle_points(n), k)
    G = bfref.from_desc(desc, None, seed=3)                       # host-valued oracle graph (synthetic stream)
    rng = np.random.default_rng(8)
    rows = np.repeat(np.arange(n), 12)
    cols = (rows + np.tile(np.concatenate([np.arange(-6, 0), np.arange(1, 7)]), n)) % n
    vals = (rng.standard_normal(len(rows)) + 1j * rng.standard_normal(len(rows))) * 1e-2
    G2 = bfref.from_desc(desc, None, seed=3)
    S = bfref.mat_sum([G2, bfref.coo_complex(n, n, rows, cols, vals), bfref.diag_real(n, n, np.full(n, 0.5))])
    plain = HipOperator.from_bfmat(G.ptr.value)
    deco = HipOperator.from_bfmat(S.ptr.value)
    sp, sd = plain.stats(), deco.stats()
    # entries over dense near-field leaves are folded; the few next to a block corner that fall into a
    # butterflied block stay 1 x 1 terms (the reference keeps those as MatSum(product, coo) too,
    # src/mat_block_dense.c:486-497): a fraction of a percent of the 851 968 entries
    left = sd["numLeaves"] - sp["numLeaves"]
    assert 0 <= left <= 0.005 * (len(rows) + n), (left, sp, sd)
    x = torch.from_numpy(rng.standard_normal(n) + 1j * rng.standard_normal(n)).cuda()

    def ms(op):
        y = op.apply_device(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            op.apply_device(x, y)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 30 * 1e3, y
    tp, yp = ms(plain)
    td, yd = ms(deco)
    assert td <= 1.05 * tp, (tp, td)
    # y_deco = y_plain + (C + I/2) x, with C applied in numpy
    xs = x.cpu().numpy()
    corr = np.zeros(n, dtype=complex)
    np.add.at(corr, rows, vals * xs[cols])
    want = yp.cpu().numpy() + corr + 0.5 * xs
    assert rel(yd.cpu().numpy(), want) < 1e-12
    plain.close(); deco.close()
