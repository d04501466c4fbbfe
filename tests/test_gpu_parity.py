"""GPU parity tests proper: the HIP path, called through the C-ABI, against the
CPU oracle (oracle/bfref.c) on identical operands.

Tolerance (north_star "stated fp64 tolerance"):
  fp64 / complex128: ||y_gpu - y_oracle||_2 / ||y_oracle||_2 <= 1e-12
  (the reference itself is only reproducible to ~1e-16*kappa between BLAS
  thread counts); vs the dense kernel matrix the bound
  is the butterfly's own truncation error, <= 1e-9.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))


@pytest.mark.parametrize("n,k", [(1024, 100), (4096, 100)])
def test_helm2_host_apply_matches_oracle_and_dense(helm2_cases, n, k):
    from butterfly_amd.operator import HipOperator
    from oracle import bfref, helm2_build as hb
    desc, tp, vals = helm2_cases(n, k)
    A = bfref.from_desc(desc, vals)
    x = hb.complex_randn(n, 0)
    y_ref = bfref.mat_mul(A, x)
    op = HipOperator.from_desc(desc, vals)
    y = op.apply_host(x)
    assert rel(y, y_ref) <= TOL
    y_dense = hb.kernel_matrix(k, tp, tp) @ x
    assert rel(y, y_dense) <= 1e-9
    assert op.num_bytes() == A.num_bytes()     # bfMatNumBytes semantics
    op.close()


def test_dropin_vtable_shim(helm2_cases):
    """bfMatMul(A_hip, X) through the oracle's own virtual dispatch: the shim's
    Mul slot runs the device path and allocates its result via X's EmptyLike."""
    from butterfly_amd.operator import HipOperator
    from oracle import bfref, helm2_build as hb
    n, k = 4096, 100
    desc, tp, vals = helm2_cases(n, k)
    A = bfref.from_desc(desc, vals)
    op = HipOperator.from_bfmat(A.ptr.value)          # walks the BfMat graph
    a_hip = op.as_bfmat()
    lib = bfref.load()
    assert lib.bfMatGetNumRows(a_hip) == n and lib.bfMatGetNumCols(a_hip) == n
    rng = np.random.default_rng(1)
    for nrhs in (1, 3):
        x = rng.standard_normal((n, nrhs)) + 1j * rng.standard_normal((n, nrhs))
        X = bfref.dense_complex(x)
        r = lib.bfMatMul(a_hip, X.ptr)
        assert r
        Y = bfref.Mat(r)
        assert Y.type == 19 and Y.shape == (n, nrhs)
        y_ref = bfref.mat_mul(A, x)
        assert rel(Y.to_numpy(), y_ref) <= TOL
    p = C.c_void_p(a_hip)
    lib.bfMatDelete(C.byref(p))
    op.close()


def test_apply_can_be_captured_in_a_hip_graph():
    """bfhipApplyDevice / bfhipApplyTransposeDevice enqueue kernels on the caller's stream and nothing else (no allocation, no
    synchronisation, no host-side state that changes between calls): a caller may capture them in a HIP graph and replay it; the
    replay is the eager result bit for bit.  (Measured at N = 65536: 1.0675 ms eager, 1.0725 ms replayed -- seven 150 us launches
    are not launch-bound, and a kernel boundary costs the same in a graph.)"""
    import torch
    from butterfly_amd import _capi, helm2_structure as hs
    from butterfly_amd.operator import HipOperator
    n = 8192
    desc, perm = hs.native_multilevel_structure(hs.circle_points(n), n / 16)
    op = HipOperator.from_desc(desc, None, seed=3, flags=_capi.FLAG_ADJOINT, max_rhs=3)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for shape in ((n,), (n, 3)):
            x = torch.randn(shape, dtype=torch.complex128, device="cuda")
            y, z = torch.empty_like(x), torch.empty_like(x)
            op.apply_device(x, y); op.apply_transpose_device(x, z)
            s.synchronize()
            y0, z0 = y.clone(), z.clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                op.apply_device(x, y)
                op.apply_transpose_device(x, z)
            y.zero_(); z.zero_()
            g.replay()
            s.synchronize()
            assert torch.equal(y, y0) and torch.equal(z, z0)
            x.mul_(2.0)                      # the graph reads the buffers it was captured with
            g.replay()
            s.synchronize()
            assert torch.equal(y, 2.0 * y0)
    op.close()


def test_rmul_slot_of_the_shim(helm2_cases):
    """bfMatRmul(A_hip, X) = X A through the oracle's own dispatch (src/mat.c:195-197): the adjoint plan applied to the rows of X;
    as the last factor of an oracle Product (bfMatProductRmul walks the factors in order, src/mat_product.c:282-310); after
    bfMatTranspose the object stands for A^H, as the reference's transposed complex leaves do; a left operand of the wrong
    width and an operator without an adjoint plan raise the reference's error state and return NULL."""
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    n, k = 2048, 128
    desc, tp, vals = helm2_cases(n, k)
    A = bfref.from_desc(desc, vals)
    dense = bfref.mat_mul(A, np.eye(n, dtype=complex))
    op = HipOperator.from_bfmat(A.ptr.value, flags=_capi.FLAG_ADJOINT)
    a_hip = op.as_bfmat()
    h = _handle(a_hip, (n, n))
    rng = np.random.default_rng(44)
    c = lambda m, q: rng.standard_normal((m, q)) + 1j * rng.standard_normal((m, q))
    for m in (1, 5):
        x = c(m, n)
        z = bfref.mat_rmul(h, x)
        assert z.shape == (m, n) and rel(z, x @ dense) <= 1e-11
    # X F0 A_hip: the shim as the last factor of a reference product
    f0 = c(37, n) / np.sqrt(n)
    x = c(4, 37)
    first = bfref.mat_rmul(bfref.dense_complex(f0), x)                     # X F0
    assert rel(bfref.mat_rmul(h, first), x @ f0 @ dense) <= 1e-11
    # wrong width
    lib = bfref.load()
    lib.bfClearError()
    with pytest.raises(RuntimeError):
        bfref.mat_rmul(h, c(2, n - 1))
    # transposed object: X A^H
    bfref.mat_transpose(h)
    x = c(3, n)
    assert rel(bfref.mat_rmul(h, x), x @ dense.conj().T) <= 1e-11
    bfref.mat_transpose(h)
    assert rel(bfref.mat_rmul(h, x), x @ dense) <= 1e-11
    p = C.c_void_p(a_hip)
    lib.bfMatDelete(C.byref(p))
    op.close()
    # without an adjoint plan: NOT_IMPLEMENTED, NULL
    op = HipOperator.from_bfmat(A.ptr.value)
    a_hip = op.as_bfmat()
    lib.bfClearError()
    with pytest.raises(RuntimeError):
        bfref.mat_rmul(_handle(a_hip, (n, n)), c(2, n))
    p = C.c_void_p(a_hip)
    lib.bfMatDelete(C.byref(p))
    op.close()


def test_shim_takes_a_column_strided_right_hand_side(helm2_cases):
    """A BfMatDenseComplex whose colStride is not 1 (every other column of a wider matrix: what a column-range view of
    the reference looks like, src/mat_dense_complex.c:648-672) through the shim's Mul slot: gathered, applied, and
    returned as a fresh packed matrix -- the refusal SURVEY section 8(a) A9 suggested is no longer needed."""
    from butterfly_amd.operator import HipOperator
    from oracle import bfref, helm2_build as hb
    n, k, nrhs = 4096, 100, 3
    desc, tp, vals = helm2_cases(n, k)
    A = bfref.from_desc(desc, vals)
    op = HipOperator.from_bfmat(A.ptr.value)
    a_hip = op.as_bfmat()
    lib = bfref.load()
    rng = np.random.default_rng(8)
    wide = rng.standard_normal((n, 2 * nrhs)) + 1j * rng.standard_normal((n, 2 * nrhs))
    X = bfref.dense_complex(wide)
    # BfMatDenseComplex: numCols at byte 24, rowStride at 32, colStride at 40 (include/bfhip_abi.h; tests/test_abi_layout.py)
    base = X.ptr if isinstance(X.ptr, int) else X.ptr.value
    C.c_size_t.from_address(base + 24).value = nrhs
    C.c_size_t.from_address(base + 40).value = 2
    r = lib.bfMatMul(a_hip, X.ptr)
    assert r
    Y = bfref.Mat(r)
    assert Y.shape == (n, nrhs)
    assert rel(Y.to_numpy(), bfref.mat_mul(A, np.ascontiguousarray(wide[:, ::2]))) <= TOL
    C.c_size_t.from_address(base + 24).value = 2 * nrhs        # restore before the oracle frees it
    C.c_size_t.from_address(base + 40).value = 1
    p = C.c_void_p(a_hip)
    lib.bfMatDelete(C.byref(p))
    op.close()


def test_synthetic_operand_matches_oracle():
    """Structure-exact, value-synthetic operand: the device generates the same
    values the oracle builds on the host (include/bfhip_synth.h)."""
    from butterfly_amd import helm2_structure as hs
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    n, k = 8192, 512
    pts = hs.circle_points(n)
    desc, root, perm = hs.helm2_multilevel_structure(pts, k)
    A = bfref.from_desc(desc, None, seed=42)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    y_ref = bfref.mat_mul(A, x)
    op = HipOperator.from_desc(desc, None, seed=42)
    y = op.apply_host(x)
    assert rel(y, y_ref) <= TOL
    op.close()


def test_device_apply_torch(helm2_cases):
    import torch
    from butterfly_amd.operator import HipOperator
    from oracle import bfref, helm2_build as hb
    n, k = 4096, 100
    desc, tp, vals = helm2_cases(n, k)
    A = bfref.from_desc(desc, vals)
    op = HipOperator.from_desc(desc, vals)
    x = hb.complex_randn(n, 0)
    xd = torch.from_numpy(x).cuda()
    yd = op.apply_device(xd)
    torch.cuda.synchronize()
    y_ref = bfref.mat_mul(A, x)
    assert rel(yd.cpu().numpy(), y_ref) <= TOL
    # run-to-run reproducible (single owner per row, fixed summation order)
    yd2 = op.apply_device(xd)
    torch.cuda.synchronize()
    assert torch.equal(yd, yd2)
    op.close()


# ---------------------------------------------------------------------------
# committed golden fixtures (tests/golden/, generator: make_golden.py)
# ---------------------------------------------------------------------------
def _gold(name):
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)


def test_golden_one_block_on_gpu():
    from butterfly_amd.operator import HipOperator
    from fixtures import load_fixture
    desc, vals, ex = load_fixture(_gold("helm2_one_block_n2048_k128.npz"))
    op = HipOperator.from_desc(desc, vals)
    y = op.apply_host(ex["x"])
    assert rel(y, ex["y_oracle"]) <= TOL
    assert rel(y, ex["y_dense"]) <= 1e-11
    op.close()


def test_golden_multilevel_vectors_on_gpu(helm2_cases):
    from butterfly_amd.operator import HipOperator
    z = np.load(_gold("helm2_multilevel_n2048_k128_vectors.npz"))
    desc, tp, vals = helm2_cases(int(z["n"]), float(z["k"]))
    op = HipOperator.from_desc(desc, vals)
    y = op.apply_host(z["x"])
    assert rel(y, z["y_oracle"]) <= TOL
    assert rel(y, z["y_dense"]) <= 1e-10
    op.close()


def test_golden_real_nested_on_gpu_f64_and_f32():
    """Real (fac_streamer-like) operand through MulVec semantics; fp32 is the
    build's extension (reference has no single precision, include/bf/def.h:31-35):
    tolerance 2e-5 relative, fp64 1e-12."""
    from butterfly_amd.operator import HipOperator
    from fixtures import load_fixture
    desc, vals, ex = load_fixture(_gold("real_nested_small.npz"))
    op = HipOperator.from_desc(desc, vals)
    assert rel(op.apply_host(ex["x"]), ex["y_oracle"]) <= TOL
    op.close()
    op32 = HipOperator.from_desc(desc, vals, demote_to_f32=True)
    assert rel(op32.apply_host(ex["x"]), ex["y_oracle"]) <= 2e-5
    op32.close()


# ---------------------------------------------------------------------------
# operand zoo: nesting, Identity leaves, ragged sizes, empty block rows
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(8))
def test_random_nested_real_graphs_on_gpu(seed):
    import randgraph
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(1000 + seed)
    desc, vals = randgraph.random_real_operand(rng, depth=int(rng.integers(1, 5)), size_hint=int(rng.integers(8, 200)))
    x = rng.standard_normal(desc.cols[desc.root])
    A = bfref.from_desc(desc, vals)
    want = bfref.mat_mul_vec(A, x)
    op = HipOperator.from_bfmat(A.ptr.value)
    assert rel(op.apply_host(x) + 1, want + 1) <= TOL
    op.close()


@pytest.mark.parametrize("seed", range(4))
def test_random_nested_complex_graphs_on_gpu_multi_rhs(seed):
    import randgraph
    from butterfly_amd.operator import HipOperator
    rng = np.random.default_rng(2000 + seed)
    desc, vals = randgraph.random_operand(rng, depth=3, size_hint=int(rng.integers(20, 400)), cplx=True)
    n = desc.cols[desc.root]
    x = rng.standard_normal((n, 5)) + 1j * rng.standard_normal((n, 5))
    want = randgraph.densify(desc, vals, desc.root) @ x
    # compiled for 1 RHS: items of <= 64 rows, one wave each (GEMV / one-wave matrix-core kernels);
    # compiled for a block of RHS: items of <= 128 rows on the workgroup-cooperative kernel (any nrhs)
    for max_rhs in (1, 5):
        op = HipOperator.from_desc(desc, vals, max_rhs=max_rhs)
        assert rel(op.apply_host(x) + 1, want + 1) <= TOL
        assert rel(op.apply_host(np.ascontiguousarray(x[:, 2])) + 1, want[:, 2] + 1) <= TOL
        op.close()


def test_mulvec_shim_real_operator():
    """bfMatMulVec(A_hip, v) for a square real operator through the vtable shim."""
    import randgraph
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(77)
    desc, vals = randgraph.random_operand(rng, depth=3, size_hint=120, cplx=False, m=150, n=150)
    A = bfref.from_desc(desc, vals)
    x = rng.standard_normal(150)
    want = bfref.mat_mul_vec(A, x)
    op = HipOperator.from_bfmat(A.ptr.value)
    a_hip = C.c_void_p(op.as_bfmat())
    got = bfref.mat_mul_vec(type("H", (), {"ptr": a_hip, "shape": (150, 150)})(), x)
    assert rel(got, want) <= TOL
    bfref.load().bfMatDelete(C.byref(a_hip))
    op.close()


def test_wide_tall_leaves_and_zero_fill_on_gpu():
    from butterfly_amd import helm2_structure as hs
    from butterfly_amd.operator import HipOperator
    rng = np.random.default_rng(7)
    d = hs.Desc(dtype=0)
    vals = {}
    a = d.add(hs.NODE_DENSE, 150, 700); vals[a] = rng.standard_normal((150, 700)) + 1j * rng.standard_normal((150, 700))
    b = d.add(hs.NODE_DENSE, 700, 3); vals[b] = rng.standard_normal((700, 3)) + 1j * rng.standard_normal((700, 3))
    p = d.add(hs.NODE_PRODUCT, 150, 3, [(a, 0, 0), (b, 0, 0)])
    # place the product in the middle of a larger, otherwise empty block matrix: rows 0..99 and 250..299 must come out zero
    d.root = d.add(hs.NODE_BLOCK, 300, 10, [(p, 100, 4)], hs.BF_TYPE_BLOCK_COO)
    x = rng.standard_normal((10, 2)) + 1j * rng.standard_normal((10, 2))
    want = np.zeros((300, 2), dtype=complex)
    want[100:250] = vals[a] @ (vals[b] @ x[4:7])
    op = HipOperator.from_desc(d, vals)
    y = op.apply_host(x)
    assert np.all(y[:100] == 0) and np.all(y[250:] == 0)
    assert rel(y[100:250], want[100:250]) <= TOL
    op.close()


@pytest.mark.parametrize("nrhs", [3, 17, 64, 80])
def test_rhs_block_kernel_segments(nrhs):
    """The matrix-core kernel runs an item as one or more *segments* (contiguous leaf columns reading one vector, at most
    2304 of them): an operator compiled for ONE right-hand side has 16-row items of up to 4096 columns, so applying it
    to a block of right-hand sides splits items at the table capacity; a product under a dense block mixes pieces that
    read x with pieces that read an intermediate in one row group; 37- and 150-row groups leave ragged slabs, 17 and 80
    right-hand sides ragged tiles and a second 64-wide pass.  All against numpy."""
    from butterfly_amd import helm2_structure as hs
    from butterfly_amd.operator import HipOperator
    rng = np.random.default_rng(70 + nrhs)
    cz = lambda m, n: (rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))) / np.sqrt(n)
    d = hs.Desc(dtype=0)
    vals = {}
    wide = d.add(hs.NODE_DENSE, 37, 6000); vals[wide] = cz(37, 6000)           # one row group, 6000 columns of x
    a = d.add(hs.NODE_DENSE, 150, 300); vals[a] = cz(150, 300)
    b = d.add(hs.NODE_DENSE, 300, 500); vals[b] = cz(300, 500)
    prod = d.add(hs.NODE_PRODUCT, 150, 500, [(a, 0, 0), (b, 0, 0)])
    near = d.add(hs.NODE_DENSE, 150, 700); vals[near] = cz(150, 700)           # same rows as the product's last factor: reads x
    d.root = d.add(hs.NODE_BLOCK, 187, 6000, [(wide, 0, 0), (prod, 37, 100), (near, 37, 900)], hs.BF_TYPE_BLOCK_COO)
    x = rng.standard_normal((6000, nrhs)) + 1j * rng.standard_normal((6000, nrhs))
    want = np.zeros((187, nrhs), dtype=complex)
    want[:37] = vals[wide] @ x
    want[37:] = vals[a] @ (vals[b] @ x[100:600]) + vals[near] @ x[900:1600]
    for max_rhs in (1, nrhs):
        op = HipOperator.from_desc(d, vals, max_rhs=max_rhs)
        assert rel(op.apply_host(x), want) <= TOL, max_rhs
        op.close()


def test_rhs_block_kernel_row_tails_of_every_height():
    """The 64-RHS kernel runs the ragged last slab of an item as quarter slabs on v_mfma_f64_4x4x4 (tails of 1 - 12 rows; 13 - 16 stay
    a full slab): row groups of every height from 1 to 37, two column widths (one a ragged k-step), 64 and 70 right-hand sides (the
    partial second block runs the full-slab passes), against numpy; and componentwise for real data stored as complex (Gauss's
    error bound does not depend on which instruction formed the products)."""
    from butterfly_amd import helm2_structure as hs
    from butterfly_amd.operator import HipOperator
    rng = np.random.default_rng(4711)
    for ncols in (40, 61):
        d = hs.Desc(dtype=0)
        vals, blocks, r0 = {}, [], 0
        heights = list(range(1, 38))
        for i, m in enumerate(heights):
            leaf = d.add(hs.NODE_DENSE, m, ncols)
            vals[leaf] = (rng.standard_normal((m, ncols)) + 1j * rng.standard_normal((m, ncols))) / np.sqrt(ncols)
            blocks.append((leaf, r0, i * ncols))
            r0 += m
        d.root = d.add(hs.NODE_BLOCK, r0, len(heights) * ncols, blocks, hs.BF_TYPE_BLOCK_DIAG)
        dense = np.zeros((r0, len(heights) * ncols), dtype=complex)
        for leaf, rr, cc in blocks:
            dense[rr:rr + vals[leaf].shape[0], cc:cc + ncols] = vals[leaf]
        for nrhs in (64, 70):
            x = rng.standard_normal((dense.shape[1], nrhs)) + 1j * rng.standard_normal((dense.shape[1], nrhs))
            op = HipOperator.from_desc(d, vals, max_rhs=nrhs)
            y = op.apply_host(x)
            want = dense @ x
            assert rel(y, want) <= TOL
            # row by row: no tail row is dropped or duplicated
            assert np.max(np.abs(y - want) / (np.abs(dense) @ np.abs(x) + 1e-300)) <= 1e-14
            op.close()


def test_row_sharded_operators_on_gpu(helm2_cases):
    from butterfly_amd import helm2_structure as hs
    from butterfly_amd.operator import HipOperator
    from oracle import bfref, helm2_build as hb
    n, k = 4096, 100
    desc, tp, vals = helm2_cases(n, k)
    x = hb.complex_randn(n, 0)
    y_ref = bfref.mat_mul(bfref.from_desc(desc, vals), x)
    nrb = len(desc.meta["top_rows"])
    parts = []
    for b, e in ((0, 4), (4, 9), (9, nrb)):
        op = HipOperator.from_desc(desc, vals, row_blocks=(b, e))
        parts.append(op.apply_host(x))
        op.close()
    assert rel(np.concatenate(parts), y_ref) <= TOL


def test_apply_argument_errors():
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from fixtures import load_fixture
    desc, vals, ex = load_fixture(_gold("helm2_one_block_n2048_k128.npz"))
    op = HipOperator.from_desc(desc, vals)
    lib = _capi.load()
    x = np.ascontiguousarray(ex["x"])
    y = np.empty(op.shape[0], dtype=complex)
    assert lib.bfhipApply(op.handle, x.ctypes.data, 1, 0, y.ctypes.data, 1) == 1      # nrhs = 0
    assert lib.bfhipApply(op.handle, None, 1, 1, y.ctypes.data, 1) == 1              # NULL X
    assert lib.bfhipApply(op.handle, x.ctypes.data, 1, 2, y.ctypes.data, 2) == 1     # ldx < nrhs
    with pytest.raises(ValueError):
        op.apply_host(np.zeros(3, dtype=complex))
    op.close()


def test_stage_profile_reports_every_stage(helm2_cases):
    import torch
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    desc, tp, vals = helm2_cases(4096, 100)
    x = torch.randn(4096, dtype=torch.complex128, device="cuda")
    # the one-launch executor is not in the product library (make experimental; tests/experimental_checks.py)
    with pytest.raises(_capi.BfhipError) as ei:
        HipOperator.from_desc(desc, vals, flags=_capi.FLAG_PROFILE | _capi.FLAG_FLOW)
    assert ei.value.code == 3      # BF_ERROR_NOT_IMPLEMENTED
    for flags, one_launch in ((_capi.FLAG_PROFILE, False),):
        op = HipOperator.from_desc(desc, vals, flags=flags)
        assert op.flow_status()[0] == one_launch
        for _ in range(3):
            op.apply_device(x)
        ms, launches, nbytes = op.stage_profile()
        st = op.stats()
        assert len(ms) == st["numStages"]
        if one_launch:      # the whole plan is one dependency-driven launch: reported under stage 0, with the bytes of all stages
            assert launches[0] == 3 and ms[0] > 0 and not launches[1:].any() and not nbytes[1:].any()
        else:
            assert all(launches == 3) and all(ms > 0)
        assert int(nbytes.sum()) == st["leafBytes"] + 16 * (st["vecElemsRead"] + st["vecElemsWritten"])
        op.close()


@pytest.mark.parametrize("nrhs", [2, 3, 16, 20, 33, 64, 70])
def test_rhs_block_kernel_matches_oracle(helm2_cases, nrhs):
    """nrhs >= 2 runs the MFMA (v_mfma_f64_16x16x4_f64) stage kernel; RHS counts
    that are not multiples of 16 / 64 exercise its tile masking."""
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    n, k = 2048, 128
    desc, tp, vals = helm2_cases(n, k)
    rng = np.random.default_rng(nrhs)
    x = rng.standard_normal((n, nrhs)) + 1j * rng.standard_normal((n, nrhs))
    y_ref = bfref.mat_mul(bfref.from_desc(desc, vals), x)
    op = HipOperator.from_desc(desc, vals, max_rhs=nrhs)
    y = op.apply_host(x)
    assert rel(y, y_ref) <= TOL
    # column q of a block apply equals the single-RHS apply of column q
    y1 = op.apply_host(np.ascontiguousarray(x[:, 1]))
    assert rel(y[:, 1], y1) <= TOL
    op.close()


# ---------------------------------------------------------------------------
# adjoint apply (RmulVec of the reference): the plan of A^T over the same arena
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("nrhs", [20, 64])
def test_exact_complex_flag_keeps_a_small_imaginary_part_accurate(nrhs):
    """BFHIP_FLAG_EXACT_COMPLEX: the matrix-core kernels form every complex product with its four real multiplications, as
    cblas_zgemm's recurrence does (reference src/mat_dense_complex.c:1704-1765), instead of Gauss's three.  Nearly real data --
    A = Ar + i d Ai, X = Xr + i d Xi with d = 1e-9 -- have a result whose imaginary part is of size d next to a real part of
    size 1.  With the flag each part's rounding error is bounded by eps times ITS OWN sum of absolute products: the imaginary
    part agrees with the oracle's zgemm to <= 1e-12 of its own size.  Without it (the default) the imaginary part is
    T3 - T1 - T2, a difference of sums of size |A||X|: its error is bounded by eps * sum (|Ar| + |Ai|)(|Xr| + |Xi|) -- 1e-13 of
    the REAL part's size here, i.e. ~1e-5 of its own -- while the result stays normwise within the 1e-12 of every other test.
    Purely real data stored as complex give an exactly zero imaginary part either way (T3 and T1 are then the same sums)."""
    import torch
    import randgraph
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(404)
    desc, vals = randgraph.random_operand(rng, depth=3, size_hint=300, cplx=True)
    d = 1e-9
    vals = {k: np.ascontiguousarray(v.real + 1j * d * v.imag) for k, v in vals.items()}
    n = desc.cols[desc.root]
    x = rng.standard_normal((n, nrhs)) + 1j * d * rng.standard_normal((n, nrhs))
    y_ref = bfref.mat_mul(bfref.from_desc(desc, vals), x)
    re_scale, im_scale = np.abs(y_ref.real).max(), np.abs(y_ref.imag).max()
    assert im_scale < 1e-6 * re_scale
    xd = torch.from_numpy(x).cuda()
    err = {}
    for name, flag in (("gauss", 0), ("exact", _capi.FLAG_EXACT_COMPLEX)):
        op = HipOperator.from_desc(desc, vals, max_rhs=nrhs, flags=flag)
        y = op.apply_device(xd).cpu().numpy()
        assert rel(y, y_ref) <= TOL                                     # normwise: both
        err[name] = np.abs(y.imag - y_ref.imag).max()
        # purely real data: an exactly zero imaginary part, with or without the flag
        xr = torch.from_numpy(x.real.astype(np.complex128)).cuda()
        opr = HipOperator.from_desc(desc, {k: np.ascontiguousarray(v.real.astype(np.complex128)) for k, v in vals.items()}, max_rhs=nrhs, flags=flag)
        assert not opr.apply_device(xr).cpu().numpy().imag.any()
        opr.close(); op.close()
    assert err["exact"] <= 1e-12 * im_scale                              # a few ulps of the imaginary part itself
    assert err["gauss"] <= 1e-13 * re_scale                              # eps |A||X|: ulps of the REAL part's size
    assert err["gauss"] > 100 * err["exact"]                             # ... which is what the flag is for


@pytest.mark.parametrize("seed", range(6))
def test_transposed_apply_random_real_graphs_on_gpu(seed):
    import randgraph
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(3000 + seed)
    desc, vals = randgraph.random_real_operand(rng, depth=int(rng.integers(1, 5)), size_hint=int(rng.integers(8, 300)))
    A = bfref.from_desc(desc, vals)
    x = rng.standard_normal(desc.rows[desc.root])
    want = bfref.mat_rmul_vec(A, x)
    op = HipOperator.from_bfmat(A.ptr.value, flags=_capi.FLAG_ADJOINT)
    assert rel(op.apply_transpose_host(x) + 1, want + 1) <= TOL
    xf = rng.standard_normal(desc.cols[desc.root])
    assert rel(op.apply_host(xf) + 1, bfref.mat_mul_vec(A, xf) + 1) <= TOL
    op.close()
    op32 = HipOperator.from_desc(desc, vals, flags=_capi.FLAG_ADJOINT, demote_to_f32=True)
    assert rel(op32.apply_transpose_host(x) + 1, want + 1) <= 2e-5
    op32.close()


def test_transposed_apply_helm2_on_gpu(helm2_cases):
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import helm2_build as hb
    n, k = 4096, 100
    desc, tp, vals = helm2_cases(n, k)
    op = HipOperator.from_desc(desc, vals, flags=_capi.FLAG_ADJOINT)
    x = hb.complex_randn(n, 3)
    y = op.apply_transpose_host(x)
    assert rel(y, hb.kernel_matrix(k, tp, tp).T @ x) <= 1e-9
    assert rel(y, op.apply_host(x)) <= 1e-9              # S^T = S for the single-layer kernel
    op.close()


def test_rmulvec_shim_real_operator():
    """cov_matvec-style use: z = Phi (Phi^T v) through the vtable shim's RmulVec + MulVec
    (examples/covariance/lbo_cov.c:48-60)."""
    import randgraph
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(78)
    desc, vals = randgraph.random_operand(rng, depth=3, size_hint=120, cplx=False, m=140, n=140)
    A = bfref.from_desc(desc, vals)
    v = rng.standard_normal(140)
    want = bfref.mat_mul_vec(A, bfref.mat_rmul_vec(A, v))
    op = HipOperator.from_bfmat(A.ptr.value, flags=_capi.FLAG_ADJOINT)
    a_hip = C.c_void_p(op.as_bfmat())
    h = type("H", (), {"ptr": a_hip, "shape": (140, 140)})()
    got = bfref.mat_mul_vec(h, bfref.mat_rmul_vec(h, v))
    assert rel(got, want) <= TOL
    bfref.load().bfMatDelete(C.byref(a_hip))
    op.close()


def test_save_load_round_trip(tmp_path, helm2_cases):
    """bfhipSave / bfhipLoad: the loaded operator applies bit-identically (forward
    and adjoint), without the BfMat graph; bad files are FILE_ERRORs."""
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref, helm2_build as hb
    n, k = 4096, 100
    desc, tp, vals = helm2_cases(n, k)
    op = HipOperator.from_desc(desc, vals, flags=_capi.FLAG_ADJOINT)
    x = hb.complex_randn(n, 2)
    y, yt = op.apply_host(x), op.apply_transpose_host(x)
    path = tmp_path / "op.bfhip"
    op.save(path)
    st = op.stats()
    op.close()
    assert path.stat().st_size > st["leafBytes"]
    op2 = HipOperator.load(path)
    assert op2.shape == (n, n) and op2.stats()["leafBytes"] == st["leafBytes"] and op2.num_bytes() == st["leafBytes"]
    assert np.array_equal(op2.apply_host(x), y)
    assert np.array_equal(op2.apply_transpose_host(x), yt)
    assert rel(y, bfref.mat_mul(bfref.from_desc(desc, vals), x)) <= TOL
    op2.close()
    with pytest.raises(_capi.BfhipError) as e:
        HipOperator.load(tmp_path / "missing.bfhip")
    assert e.value.code == 6
    bad = tmp_path / "bad.bfhip"
    bad.write_bytes(b"NOTANOPERATOR" * 10)
    with pytest.raises(_capi.BfhipError) as e:
        HipOperator.load(bad)
    assert e.value.code == 6
    trunc = tmp_path / "trunc.bfhip"
    trunc.write_bytes(path.read_bytes()[:100000])
    with pytest.raises(_capi.BfhipError) as e:
        HipOperator.load(trunc)
    assert e.value.code == 6
    # a file is untrusted input: an index table that points outside the arena / the vectors is refused before
    # any kernel could dereference it.  Layout: 88-byte file header, 40-byte plan header, 56-byte stage
    # header, then stage 0's items {u32 pieceBegin, numPieces, outOff, mrFlags}
    raw = bytearray(path.read_bytes())
    for off, what in ((184 + 8, "item.outOff"), (184 + 0, "item.pieceBegin"), (184 + 12, "item.mrFlags")):
        bad_bytes = bytearray(raw)
        bad_bytes[off:off + 4] = (0xFFFFFFF0).to_bytes(4, "little")
        corrupt = tmp_path / "corrupt.bfhip"
        corrupt.write_bytes(bytes(bad_bytes))
        with pytest.raises(_capi.BfhipError) as e:
            HipOperator.load(corrupt)
        assert e.value.code == 6, what
    # 64-bit offsets near 2^64: `dataOff + extent` / `destOff + numRows` wrap to a small value, so the
    # bounds checks must be subtractions.  Walk the forward plan: per stage a 56-byte header {numItems, numPieces,
    # leafElems, vecIn, vecOut, numReduce, ...}, items (16 B), pieces (24 B, dataOff first), then per reduce a 40-byte
    # header {destOff, numRows, numIntervals, numSrc, destSpace} + rowInterval[numRows] + ivBegin[numIntervals + 1] + srcBias[numSrc]
    u64 = lambda b, o: int.from_bytes(b[o:o + 8], "little")
    num_stages, pos = u64(raw, 88), 128
    piece0 = reduce0 = None
    for _ in range(num_stages):
        ni, npc, nred = u64(raw, pos), u64(raw, pos + 8), u64(raw, pos + 40)
        pos += 56 + 16 * ni
        if piece0 is None and npc:
            piece0 = pos
        pos += 24 * npc
        for _ in range(nred):
            if reduce0 is None:
                reduce0 = pos
            rows, niv, nsrc = u64(raw, pos + 8), u64(raw, pos + 16), u64(raw, pos + 24)
            pos += 40 + 4 * rows + 4 * (niv + 1) + 8 * nsrc
    assert piece0 is not None and reduce0 is not None
    for off, what in ((piece0, "piece.dataOff"), (reduce0, "reduce.destOff")):
        bad_bytes = bytearray(raw)
        bad_bytes[off:off + 8] = (0xFFFFFFFFFFFFFFF0).to_bytes(8, "little")
        corrupt = tmp_path / "corrupt64.bfhip"
        corrupt.write_bytes(bytes(bad_bytes))
        with pytest.raises(_capi.BfhipError) as e:
            HipOperator.load(corrupt)
        assert e.value.code == 6, what


def test_item_class_flags_of_a_file_are_checked(tmp_path):
    """Round 2's item classes are promises the kernels rely on (a MERGED / SMALL item's dense pieces are one contiguous
    block of bounded width, SMALL items close the list, a ROWMAJOR item has at most two lane granules of rows): a file
    that breaks one of them is a FILE_ERROR, a faithful one loads and applies bit-identically."""
    import randgraph
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    rng = np.random.default_rng(12)
    d, vals, dense = randgraph.narrow_items_operand(rng)
    op = HipOperator.from_desc(d, vals, flags=_capi.FLAG_ADJOINT)
    x, v = rng.standard_normal(dense.shape[1]), rng.standard_normal(dense.shape[0])
    y, z = op.apply_host(x), op.apply_transpose_host(v)
    path = tmp_path / "narrow.bfhip"
    op.save(path)
    op.close()
    op2 = HipOperator.load(path)
    assert np.array_equal(op2.apply_host(x), y) and np.array_equal(op2.apply_transpose_host(v), z)
    op2.close()
    raw = bytearray(path.read_bytes())
    # 88-byte file header, 40-byte plan header, 56-byte stage header {numItems, numPieces, ...}, then stage 0's items
    num_items = int.from_bytes(raw[128:136], "little")
    items = np.frombuffer(bytes(raw[184:184 + 16 * num_items]), dtype=_capi.ITEM_DTYPE)
    small = np.nonzero(items["mrFlags"] & (1 << 19))[0]
    merged = np.nonzero((items["mrFlags"] & (1 << 18)) != 0)[0]
    assert len(small) >= 10 and len(merged) >= 1 and small[-1] == num_items - 1

    def flip(index, xor=0, set_rows=None):
        b = bytearray(raw)
        off = 184 + 16 * int(index) + 12
        f = int.from_bytes(b[off:off + 4], "little") ^ xor
        if set_rows is not None:
            f = (f & ~0xFFFF) | set_rows
        b[off:off + 4] = f.to_bytes(4, "little")
        bad = tmp_path / "flags.bfhip"
        bad.write_bytes(bytes(b))
        with pytest.raises(_capi.BfhipError) as e:
            HipOperator.load(bad)
        assert e.value.code == 6
    flip(0, xor=1 << 19)                 # a SMALL item at the head of the list
    flip(small[-1], set_rows=40)         # a small item taller than two lane granules
    flip(merged[0], xor=1 << 17)         # MERGED and ROWMAJOR at once
    flip(0, xor=1 << 21)                 # an unknown flag
    # the class is a scheduling promise, not a layout: the first small item without its flag simply runs as an ordinary item
    b = bytearray(raw)
    off = 184 + 16 * int(small[0]) + 12
    b[off:off + 4] = (int.from_bytes(b[off:off + 4], "little") ^ (1 << 19)).to_bytes(4, "little")
    ok = tmp_path / "unflagged.bfhip"
    ok.write_bytes(bytes(b))
    op3 = HipOperator.load(ok)
    assert rel(op3.apply_host(x), y) <= TOL
    op3.close()


@pytest.mark.parametrize("seed", range(3))
def test_deeply_nested_real_graphs_forward_and_transposed(seed):
    """The survey's fac_streamer sample nests all three block types and Identity leaves up to 9 deep
    (SURVEY.md section 8(c)): graphs of depth 9 through the BfMat walker, f64 and f32, A x and A^T x."""
    import randgraph
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(7000 + seed)
    for _ in range(50):                                          # draw until the graph is a real zoo
        desc, vals = randgraph.random_real_operand(rng, depth=9, size_hint=700)
        kinds = np.asarray(desc.kind)
        if (kinds == 1).sum() >= 2 and (kinds == 3).sum() >= 3 and len(kinds) > 80:
            break
    assert (kinds == 1).sum() >= 2 and len(kinds) > 80          # Identity leaves inside a deep nest
    m, n = desc.rows[desc.root], desc.cols[desc.root]
    x, w = rng.standard_normal(n), rng.standard_normal(m)
    A = bfref.from_desc(desc, vals)
    want, want_t = bfref.mat_mul_vec(A, x), bfref.mat_rmul_vec(A, w)
    op = HipOperator.from_bfmat(A.ptr.value, flags=_capi.FLAG_ADJOINT)
    assert rel(op.apply_host(x) + 1, want + 1) <= TOL
    assert rel(op.apply_transpose_host(w) + 1, want_t + 1) <= TOL
    op.close()
    op32 = HipOperator.from_bfmat(A.ptr.value, demote_to_f32=True)
    assert rel(op32.apply_host(x) + 1, want + 1) <= 2e-5
    op32.close()


def _handle(ptr, shape):
    """What oracle.bfref.mat_mul_vec / mat_rmul_vec need of an operand: .ptr and .shape."""
    return type("H", (), {"ptr": ptr, "shape": shape})()


@pytest.mark.parametrize("m,n", [(300, 70), (64, 211)])
def test_rectangular_cov_matvec_through_the_shim(m, n):
    """cov_matvec's sequence on a RECTANGULAR real operator Phi (N x m eigenvector band):
    tmp = bfMatRmulVec(Phi, v); z = bfMatMulVec(Phi, Gamma^2 tmp)
    (examples/covariance/lbo_cov.c:48-60), dispatched by the oracle's own bfMatRmulVec / bfMatMulVec
    on the shim.  The results are sized by the operator (src/mat_block_dense.c:574-590,
    src/mat_block_coo.c:427-444) and freed by the oracle's bfVecDelete."""
    import randgraph
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(1000 + m)
    desc, vals = randgraph.random_operand(rng, depth=4, size_hint=120, cplx=False, m=m, n=n)
    A = bfref.from_desc(desc, vals)
    gamma2 = rng.random(n) ** 2
    v = rng.standard_normal(m)
    t_ref = bfref.mat_rmul_vec(A, v)
    z_ref = bfref.mat_mul_vec(A, gamma2 * t_ref)
    op = HipOperator.from_bfmat(A.ptr.value, flags=_capi.FLAG_ADJOINT)
    a_hip = C.c_void_p(op.as_bfmat())
    h = _handle(a_hip, (m, n))
    t = bfref.mat_rmul_vec(h, v)
    assert t.shape == (n,) and rel(t, t_ref) <= TOL
    z = bfref.mat_mul_vec(h, gamma2 * t)
    assert z.shape == (m,) and rel(z, z_ref) <= TOL
    # shape errors surface as NULL + the reference's error state (bfSetError), not as a crash
    lib = bfref.load()
    lib.bfClearError()
    with pytest.raises(RuntimeError, match="BfError 8"):     # BF_ERROR_INCOMPATIBLE_SHAPES
        bfref.mat_mul_vec(h, np.zeros(n + 1))
    lib.bfMatDelete(C.byref(a_hip))
    op.close()


@pytest.mark.parametrize("cplx,values", [(False, True), (False, False), (True, True), (True, False)])
def test_packed_adjoint_runs_the_transposed_expression_on_the_forward_kernels(cplx, values, tmp_path):
    """BFHIP_FLAG_ADJOINT_PACKED: A^T gets its own packed copy of the leaves and its plan is a FORWARD plan of the transposed
    expression (blocks placed at (col0, row0), products reversed, every leaf transposed -- reference
    src/mat_product.c:409-420).  Same result as the shared-leaf adjoint plan and as the oracle's bfMatRmulVec (real) /
    the dense transpose (complex), with host-valued and with synthetic leaves, rectangular, several right-hand sides."""
    import randgraph
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(900 + 2 * cplx + values)
    m, n = 260, 190
    desc, vals = randgraph.random_operand(rng, depth=4, size_hint=110, cplx=cplx, m=m, n=n)
    if not values:
        vals = None
    A = bfref.from_desc(desc, vals, seed=11)
    packed = HipOperator.from_desc(desc, vals, seed=11, flags=_capi.FLAG_ADJOINT_PACKED)
    shared = HipOperator.from_desc(desc, vals, seed=11, flags=_capi.FLAG_ADJOINT)
    assert packed.stats()["arenaBytes"] >= 2 * shared.stats()["leafBytes"] and packed.stats()["leafBytes"] == shared.stats()["leafBytes"]
    for nrhs in (1, 3):
        shape = (m,) if nrhs == 1 else (m, nrhs)
        v = rng.standard_normal(shape) + (1j * rng.standard_normal(shape) if cplx else 0)
        if cplx:
            dense = bfref.mat_mul(A, np.eye(n, dtype=np.complex128))
            want = dense.T @ v
        else:
            want = np.stack([bfref.mat_rmul_vec(A, v if nrhs == 1 else v[:, q]) for q in range(nrhs)], axis=-1).reshape((n,) if nrhs == 1 else (n, nrhs))
        got_p, got_s = packed.apply_transpose_host(v), shared.apply_transpose_host(v)
        assert rel(got_p, want) <= TOL and rel(got_s, want) <= TOL
        # the forward apply is untouched by either
        w = rng.standard_normal((n,) if nrhs == 1 else (n, nrhs)) + (1j * rng.standard_normal((n,) if nrhs == 1 else (n, nrhs)) if cplx else 0)
        assert np.array_equal(packed.apply_host(w), shared.apply_host(w))
    # round 5: both arenas travel in the file; the loaded operator applies A and A^T bit for bit like the saved one
    packed.save(tmp_path / "p.bfhip")
    back = HipOperator.load(tmp_path / "p.bfhip")
    assert back.stats()["arenaBytes"] == packed.stats()["arenaBytes"]
    v = rng.standard_normal(m) + (1j * rng.standard_normal(m) if cplx else 0)
    w = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
    assert np.array_equal(back.apply_transpose_host(v), packed.apply_transpose_host(v))
    assert np.array_equal(back.apply_host(w), packed.apply_host(w))
    back.close()
    # a file whose header claims a packed adjoint but whose second plan is a transposed one is refused
    raw = bytearray(open(tmp_path / "p.bfhip", "rb").read())
    shared.save(tmp_path / "s.bfhip")
    raw_s = bytearray(open(tmp_path / "s.bfhip", "rb").read())
    raw_s[36:40] = (1).to_bytes(4, "little")              # FileHeader.reserved: "packed" on a shared-leaf file
    open(tmp_path / "bad.bfhip", "wb").write(bytes(raw_s))
    with pytest.raises(_capi.BfhipError) as ei:
        HipOperator.load(tmp_path / "bad.bfhip")
    assert ei.value.code == 6
    packed.close(); shared.close()


@pytest.mark.parametrize("m,n", [(300, 70), (96, 96)])
def test_transpose_slot_flips_the_shim_between_its_two_plans(m, n):
    """bfMatTranspose (slot 63, reference src/mat.c:271-273) on the shim, driven by the oracle's dispatcher: in place, like
    bfMatProductTranspose (src/mat_product.c:409-420).  Afterwards bfMatMulVec is what the oracle's bfMatRmulVec gives on
    the original operator and the other way round, GetNumRows / GetNumCols answer for A^T, and twice is the identity."""
    import randgraph
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(4242 + m)
    desc, vals = randgraph.random_operand(rng, depth=4, size_hint=120, cplx=False, m=m, n=n)
    A = bfref.from_desc(desc, vals)
    v, w = rng.standard_normal(m), rng.standard_normal(n)
    lib = bfref.load()
    op = HipOperator.from_bfmat(A.ptr.value, flags=_capi.FLAG_ADJOINT)
    a_hip = C.c_void_p(op.as_bfmat())
    assert (lib.bfMatGetNumRows(a_hip), lib.bfMatGetNumCols(a_hip)) == (m, n)
    lib.bfMatTranspose(a_hip)
    assert (lib.bfMatGetNumRows(a_hip), lib.bfMatGetNumCols(a_hip)) == (n, m)
    ht = _handle(a_hip, (n, m))
    assert rel(bfref.mat_mul_vec(ht, v), bfref.mat_rmul_vec(A, v)) <= TOL           # A^T v
    assert rel(bfref.mat_rmul_vec(ht, w), bfref.mat_mul_vec(A, w)) <= TOL           # w^T A^T = (A w)^T
    lib.bfClearError()
    with pytest.raises(RuntimeError, match="BfError 8"):                              # shapes are those of A^T now
        bfref.mat_mul_vec(ht, np.zeros(m + 1))
    # a view taken of the transposed object is transposed too (GetView is a shallow copy)
    lib.bfMatTranspose(a_hip)
    assert (lib.bfMatGetNumRows(a_hip), lib.bfMatGetNumCols(a_hip)) == (m, n)
    h = _handle(a_hip, (m, n))
    assert rel(bfref.mat_mul_vec(h, w), bfref.mat_mul_vec(A, w)) <= TOL
    assert rel(bfref.mat_rmul_vec(h, v), bfref.mat_rmul_vec(A, v)) <= TOL
    lib.bfMatDelete(C.byref(a_hip))
    op.close()
    # without an adjoint plan the slot raises the reference's error state and leaves the object as it was
    op = HipOperator.from_bfmat(A.ptr.value)
    a_hip = C.c_void_p(op.as_bfmat())
    lib.bfClearError()
    lib.bfMatTranspose(a_hip)
    assert lib.bfGetError() == 3                                                      # BF_ERROR_NOT_IMPLEMENTED
    lib.bfClearError()
    assert (lib.bfMatGetNumRows(a_hip), lib.bfMatGetNumCols(a_hip)) == (m, n)
    assert rel(bfref.mat_mul_vec(_handle(a_hip, (m, n)), w), bfref.mat_mul_vec(A, w)) <= TOL
    lib.bfMatDelete(C.byref(a_hip))
    op.close()


def test_transposed_complex_shim_multiplies_by_the_conjugate_transpose(helm2_cases):
    """bfMatTranspose on a complex operator: bfMatMul(A_hip, X) is then A^H X, as in the reference -- its dense complex
    leaves transpose by bfMatConjTrans (src/mat_dense_complex.c:1475-1478) and multiply through CblasConjTrans (:27-35).
    Checked against the dense matrix obtained column by column from the oracle; the library's own transposed apply
    (bfhipApplyTranspose) stays the plain transpose."""
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref, helm2_build as hb
    n, k = 1024, 64
    desc, tp, vals = helm2_cases(n, k)
    A = bfref.from_desc(desc, vals)
    dense = bfref.mat_mul(A, np.eye(n, dtype=np.complex128))
    x = hb.complex_randn(n * 3, 5).reshape(n, 3)
    lib = bfref.load()
    op = HipOperator.from_desc(desc, vals, flags=_capi.FLAG_ADJOINT, max_rhs=3)
    a_hip = C.c_void_p(op.as_bfmat())
    h = _handle(a_hip, (n, n))
    lib.bfMatTranspose(a_hip)
    assert rel(bfref.mat_mul(h, x), dense.conj().T @ x) <= 1e-11
    assert rel(bfref.mat_mul(h, x[:, ::2]), dense.conj().T @ x[:, ::2]) <= 1e-11      # a column-strided right-hand side
    lib.bfMatTranspose(a_hip)
    assert rel(bfref.mat_mul(h, x), dense @ x) <= 1e-11
    assert rel(op.apply_transpose_host(x), dense.T @ x) <= 1e-11
    lib.bfMatDelete(C.byref(a_hip))
    op.close()


@pytest.mark.parametrize("seed", range(4))
def test_transpose_of_complex_graphs_equals_the_oracles_transpose(seed):
    """Graphs the reference can transpose (Product / BlockDiag / BlockDense / Identity / DenseComplex; BlockCoo has no slot),
    complex: (a) the oracle transposes its graph in place (leaves flagged TRANS | CONJ) and the shim, wrapped around the
    UNtransposed graph and transposed through its own slot, multiplies to the same result; (b) an operator compiled from
    the ALREADY transposed graph -- the walker reads flagged dense leaves as conjugates with their strides swapped
    (src/mat_dense_complex.c:27-35, 503-511) -- is A^H, and its own transposed apply gives conj(A) back."""
    import randgraph
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(9100 + seed)
    desc, vals = randgraph.random_operand(rng, depth=int(rng.integers(1, 4)), size_hint=60, cplx=True, coo=False)
    m, n = desc.rows[desc.root], desc.cols[desc.root]
    dense = randgraph.densify(desc, vals, desc.root)
    A = bfref.from_desc(desc, vals, typed=True)
    x = rng.standard_normal((m, 3)) + 1j * rng.standard_normal((m, 3))
    xf = rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))
    lib = bfref.load()
    op = HipOperator.from_bfmat(A.ptr.value, flags=_capi.FLAG_ADJOINT, max_rhs=3)
    a_hip = C.c_void_p(op.as_bfmat())
    h = _handle(a_hip, (m, n))
    assert rel(bfref.mat_mul(h, xf) + 1, bfref.mat_mul(A, xf) + 1) <= TOL
    lib.bfMatTranspose(a_hip)
    bfref.mat_transpose(A)
    want = bfref.mat_mul(A, x)
    assert rel(want + 1, dense.conj().T @ x + 1) <= TOL
    assert rel(bfref.mat_mul(h, x) + 1, want + 1) <= TOL
    # (b) compiled from the transposed graph
    op2 = HipOperator.from_bfmat(A.ptr.value, flags=_capi.FLAG_ADJOINT, max_rhs=3)
    assert op2.shape == (n, m)
    assert rel(op2.apply_host(x) + 1, want + 1) <= TOL
    assert rel(op2.apply_transpose_host(xf) + 1, dense.conj() @ xf + 1) <= TOL
    op2.close()
    lib.bfMatDelete(C.byref(a_hip))
    op.close()


@pytest.mark.parametrize("demote", [False, True])
def test_fused_covariance_products_match_the_oracle_sequence(demote):
    """sample_z and cov_matvec of examples/covariance/lbo_cov.c:36-60 as ONE device call each
    (bfhipCovSampleDevice / bfhipCovMatvecDevice) against the same sequence run step by step through the oracle's
    dispatch: bfVecRealPermute scatters (out[perm[i]] = in[i], src/vec_real.c:312-329), GammaLam is a diagonal applied
    once (sample) or twice (matvec)."""
    import torch
    import randgraph
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(77)
    m, n = 211, 93
    desc, vals = randgraph.random_operand(rng, depth=4, size_hint=120, cplx=False, m=m, n=n)
    A = bfref.from_desc(desc, vals)
    gam = rng.random(n) + 0.1
    row_perm = rng.permutation(m)
    rev = np.empty(m, dtype=np.int64); rev[row_perm] = np.arange(m)      # bfPermGetReversePerm: the inverse permutation

    def permute(x, perm):                 # bfVecRealPermute
        out = np.empty_like(x); out[perm] = x
        return out
    w, v = rng.standard_normal(n), rng.standard_normal(m)
    z_sample = permute(bfref.mat_mul_vec(A, gam * w), row_perm)
    t = bfref.mat_rmul_vec(A, permute(v, rev))
    z_cov = permute(bfref.mat_mul_vec(A, gam * (gam * t)), row_perm)
    op = HipOperator.from_bfmat(A.ptr.value, flags=_capi.FLAG_ADJOINT, demote_to_f32=demote)
    dt = torch.float32 if demote else torch.float64
    dev = torch.device("cuda", 0)
    tg = torch.from_numpy(gam).to(dev).to(dt)
    tp = torch.from_numpy(row_perm.astype(np.int64)).to(dev)
    tr = torch.from_numpy(rev).to(dev)
    tol = 3e-5 if demote else TOL
    got = op.cov_sample_device(tg, tp, torch.from_numpy(w).to(dev).to(dt)).cpu().numpy()
    assert rel(got, z_sample) <= tol
    got = op.cov_matvec_device(tg, tp, tr, torch.from_numpy(v).to(dev).to(dt)).cpu().numpy()
    assert rel(got, z_cov) <= tol
    # identity diagonal and permutations: plain A A^T v
    got = op.cov_matvec_device(None, None, None, torch.from_numpy(v).to(dev).to(dt)).cpu().numpy()
    assert rel(got, bfref.mat_mul_vec(A, bfref.mat_rmul_vec(A, v))) <= tol
    op.close()
    # complex operators and operators without the transposed plan are refused, not crashed
    op2 = HipOperator.from_bfmat(A.ptr.value)
    with pytest.raises(_capi.BfhipError):
        op2.cov_matvec_device(None, None, None, torch.from_numpy(v).to(dev))
    op2.close()


def test_shim_nested_inside_an_oracle_block_dense(helm2_cases):
    """The device operator as ONE block of a reference container: bfMatBlockDenseMul takes every
    block through bfMatGet(block, BF_POLICY_VIEW) = the block's GetView slot, multiplies, and
    deletes the view (src/mat_block_dense.c:541-563, :1043-1061)."""
    from butterfly_amd.operator import HipOperator
    from oracle import bfref, helm2_build as hb
    n, k = 1024, 100
    desc, tp, vals = helm2_cases(n, k)
    A = bfref.from_desc(desc, vals)
    op = HipOperator.from_bfmat(A.ptr.value)
    rng = np.random.default_rng(5)
    m2 = 37
    d01 = rng.standard_normal((n, m2)) + 1j * rng.standard_normal((n, m2))
    d10 = rng.standard_normal((m2, n)) + 1j * rng.standard_normal((m2, n))
    d11 = rng.standard_normal((m2, m2)) + 1j * rng.standard_normal((m2, m2))
    x = rng.standard_normal((n + m2, 2)) + 1j * rng.standard_normal((n + m2, 2))
    y0 = bfref.mat_mul(A, x[:n]) + d01 @ x[n:]
    y1 = d10 @ x[:n] + d11 @ x[n:]
    want = np.vstack([y0, y1])

    class Borrowed(bfref.Mat):           # the container steals the pointer like any other block
        pass
    shim = Borrowed(op.as_bfmat(owns=True))
    grid = bfref.block_dense([0, n, n + m2], [0, n, n + m2],
                             [shim, bfref.dense_complex(d01), bfref.dense_complex(d10), bfref.dense_complex(d11)])
    got = bfref.mat_mul(grid, x)
    assert rel(got, want) <= TOL
    del grid                              # bfMatDelete on the container deletes the shim (and with it the operator)


def test_adjoint_never_reads_past_a_short_column_block():
    """with a leaf height that is not a multiple of 16 the transposed kernel's last
    16-unit block used to read into whatever follows the piece (times x = 0: NaN if those bits are
    NaN).  Poisoned neighbour: a second leaf full of NaN is packed right behind the first."""
    from butterfly_amd import _capi
    from butterfly_amd.helm2_structure import BF_TYPE_BLOCK_DIAG, Desc, NODE_BLOCK, NODE_DENSE
    from butterfly_amd.operator import HipOperator
    rng = np.random.default_rng(3)
    for dtype, cplx in ((0, True), (1, False)):
        d = Desc(dtype=dtype)
        a = d.add(NODE_DENSE, 37, 19)
        b = d.add(NODE_DENSE, 5, 5)
        d.root = d.add(NODE_BLOCK, 42, 24, [(a, 0, 0), (b, 37, 19)], BF_TYPE_BLOCK_DIAG)
        va = rng.standard_normal((37, 19)) + (1j * rng.standard_normal((37, 19)) if cplx else 0)
        vb = np.full((5, 5), np.nan) + (1j * np.nan if cplx else 0)
        for demote in ((False,) if cplx else (False, True)):
            op = HipOperator.from_desc(d, {a: va, b: vb}, flags=_capi.FLAG_ADJOINT, demote_to_f32=demote)
            x = rng.standard_normal(42) + (1j * rng.standard_normal(42) if cplx else 0)
            y = op.apply_transpose_host(x)
            want = va.T @ x[:37]
            assert np.all(np.isfinite(y[:19])), "adjoint picked up the poisoned neighbour"
            assert rel(y[:19], want) <= (2e-5 if demote else TOL)
            assert np.all(np.isnan(y[19:]))       # the NaN leaf's own outputs
            op.close()


def test_sharded_apply_from_plain_c(tmp_path):
    """examples/sharded_apply.c: the multi-GPU step of include/bfhip.h (RCCL communicator, local stages,
    in-place ncclAllGather + segment reordering / ncclAllReduce) driven from plain C with a 1-rank
    communicator; exits 0 iff all three shard modes reproduce the unsharded apply bit for bit."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "butterfly_amd", "csrc")
    exe = str(tmp_path / "sharded_apply")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "sharded_apply.c"), "-L", lib, "-lbfhip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    p = subprocess.run([exe, "1", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("bit-identical") == 3, p.stdout          # block rows, (row, col) blocks, row ranges
    assert p.stdout.count("adjoint-equal") == 3 and p.stdout.count("gmres-equal") == 1, p.stdout      # the adjoint of each, and GMRES over the sharded step


@pytest.mark.parametrize("mode", ["rows", "blocks"])
def test_rccl_sharded_apply_one_rank_matches_plain_apply(helm2_cases, mode):
    """dist.RcclShardedApply (the C-ABI's sharded step as bench.py drives it) with a 1-rank communicator:
    identical to the plain device apply, for 1 and 3 right-hand sides."""
    import torch
    from butterfly_amd import helm2_structure as hs
    from butterfly_amd.dist import RcclShardedApply, ShardLayout
    from butterfly_amd.operator import HipOperator
    from oracle import helm2_build as hb
    n, k = 4096, 100
    desc, tp, vals = helm2_cases(n, k)
    op = HipOperator.from_desc(desc, vals, max_rhs=3)
    top_rows = desc.meta["top_rows"]
    layout = ShardLayout(top_rows, [0] * len(top_rows), 1)
    for nrhs in (1, 3):
        x = hb.complex_randn(n * nrhs, 3).reshape(n, nrhs) if nrhs > 1 else hb.complex_randn(n, 3)
        xd = torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
        want = op.apply_device(xd).cpu().numpy()
        step = RcclShardedApply(layout, 0, op, 0, nrhs=nrhs, mode=mode)
        got = step(xd)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), want)
        loc, coll = step.last_times()
        assert loc > 0 and coll >= 0
        step.close()
    op.close()


@pytest.mark.parametrize("nrhs", [1, 5])
def test_row_range_shards_are_the_one_gpu_result_bit_for_bit(nrhs):
    """BfhipOptions.rowBegin/rowEnd with the cuts of bfhipRowPartition (one level or more below the top-level row blocks,
    reference src/fac_helm2.c:814-858): every one of 8 ranks' operators, run on this GPU one after another, produces exactly
    its rows of the unsharded apply -- same row groups, same items, same order of additions -- and together they hold
    1.0x - 1.1x the operator's leaves (the first-applied factor of a split block row is replicated).  Also against the
    oracle, and through the C-ABI's sharded step (1-rank communicator, the rank's segment gathered into place)."""
    import torch
    from butterfly_amd import helm2_structure as hs
    from butterfly_amd.dist import RcclShardedApply, ShardLayout, row_partition
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    n, k, world = 16384, 1024.0, 8
    desc, perm = hs.native_multilevel_structure(hs.circle_points(n), k)
    rng = np.random.default_rng(77)
    shape = (n,) if nrhs == 1 else (n, nrhs)
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2)
    xd = torch.from_numpy(x).cuda()
    full = HipOperator.from_desc(desc, None, seed=5, max_rhs=nrhs)
    y = full.apply_device(xd).clone()
    total = full.stats()["leafElems"]
    full.close()
    y_ref = bfref.mat_mul(bfref.from_desc(desc, None, seed=5), x if nrhs > 1 else x[:, None]).reshape(shape)
    assert rel(y.cpu().numpy(), y_ref) <= TOL
    cuts, loads = row_partition(desc, world)
    kept = 0
    for r in range(world):
        op = HipOperator.from_desc(desc, None, seed=5, max_rhs=nrhs, row_range=(cuts[r], cuts[r + 1]))
        st = op.stats()
        assert st["numRows"] == cuts[r + 1] - cuts[r] and st["leafElems"] == loads[r]
        kept += st["leafElems"]
        got = op.apply_device(xd)
        torch.cuda.synchronize()
        assert torch.equal(got, y[cuts[r]:cuts[r + 1]]), r
        op.close()
    assert total <= kept <= 1.12 * total
    assert max(loads) <= 1.10 * (kept / world)             # N = 16384 offers coarse places to cut (1.07 here; 1.02 at N = 262144)
    # the same operator behind the sharded step: a one-rank world owns everything, cut into 3 segments to be put in place
    op = HipOperator.from_desc(desc, None, seed=5, max_rhs=nrhs)
    step = RcclShardedApply(ShardLayout([cuts[2], cuts[5] - cuts[2], n - cuts[5]], [0, 0, 0], 1), 0, op, 0, nrhs=nrhs, mode="rows")
    got = step(xd)
    torch.cuda.synchronize()
    assert torch.equal(got, y)
    step.close()
    op.close()


@pytest.mark.parametrize("nrhs", [1, 3])
def test_shared_row_ranges_are_summed_after_the_gather(nrhs):
    """"rowsum" sharding (BfhipShardSpec.segGlobalOff): ranks that share a block row by columns each send a partial result
    for it and every rank adds the partials of a range in list order after the ONE all-gather (bfSumSegmentsKernel).
    Exercised with a 1-rank communicator whose operator yields block row 0 as TWO partials (even / odd column blocks)
    stacked on top of the other rows: the summed result equals the plain apply to rounding, is the same run to run, and a
    range that would not tile the rows is refused."""
    import torch
    from butterfly_amd import _capi, helm2_structure as hs
    from butterfly_amd.dist import RcclShardedApply, ShardLayout, rowsum_partition
    from butterfly_amd.operator import HipOperator
    n, k = 8192, 512.0
    desc, perm = hs.native_multilevel_structure(hs.circle_points(n), k)
    rng = np.random.default_rng(21)
    shape = (n,) if nrhs == 1 else (n, nrhs)
    xd = torch.from_numpy((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2)).cuda()
    full = HipOperator.from_desc(desc, None, seed=6, max_rhs=nrhs)
    want = full.apply_device(xd).clone()
    full.close()
    top_rows, trb = desc.meta["top_rows"], desc.top_row_block
    ch = desc.children[desc.root]
    m0 = top_rows[0]
    new, col = [], 0
    for i, (c, r0, c0) in enumerate(ch):
        if trb[i] == 0:
            new.append((c, r0 + (m0 if col % 2 else 0), c0))       # odd column blocks of block row 0: a second partial below the first
            col += 1
        else:
            new.append((c, r0 + m0, c0))                           # every other row moves down by one copy of block row 0
    root = desc.add(hs.NODE_BLOCK, n + m0, n, new, hs.BF_TYPE_BLOCK_DENSE)
    op = HipOperator.from_desc(desc, None, root=root, seed=6, max_rhs=nrhs)
    segs = [(0, 0), (0, 0)] + [(rb, 0) for rb in range(1, len(top_rows))]
    step = RcclShardedApply(ShardLayout(top_rows, [0] * len(top_rows), 1, segments=segs), 0, op, 0, nrhs=nrhs, mode="rowsum")
    got = step(xd).clone()
    torch.cuda.synchronize()
    assert rel(got.cpu().numpy(), want.cpu().numpy()) <= 1e-14
    assert torch.equal(got[m0:], want[m0:])                        # rows with one owner are copied, not summed
    assert torch.equal(step(xd), got)                              # fixed order of additions: reproducible
    step.close()
    # the partition bench.py deals to 8 ranks: every block row whole or shared by two ranks, loads within 3 % of the mean
    bowner, loads, rsegs = rowsum_partition(desc, 8)
    assert max(loads) <= 1.035 * sum(loads) / 8 and len(rsegs) <= 2 * len(top_rows)
    # ranges that do not tile the rows are refused at create time
    bad = ShardLayout(top_rows, [0] * len(top_rows), 1, segments=segs)
    bad.row_offsets = bad.row_offsets.copy()
    bad.row_offsets[2] += 1                                        # block row 2 would start one row late: a gap and an overlap
    with pytest.raises(_capi.BfhipError) as e:
        RcclShardedApply(bad, 0, op, 0, nrhs=nrhs, mode="rowsum")
    assert e.value.code == 1
    op.close()


@pytest.mark.parametrize("dtype,demote", [(0, False), (1, False), (1, True)])
def test_long_contractions_on_gpu(dtype, demote):
    """Row groups cut into several groups (private slots + reduce) because their contraction is long: a
    40 x 40000 leaf, a 37000 x 21 leaf and a block column of 90 leaves, forward and transposed."""
    import randgraph
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(23 + dtype)
    d, vals, apply_t, val = randgraph.long_contraction_operand(rng, dtype)
    m, n = d.rows[d.root], d.cols[d.root]
    x, v = val(n, 1)[:, 0], val(m, 1)[:, 0]
    A = bfref.from_desc(d, vals)
    y_ref = bfref.mat_mul(A, x) if dtype == 0 else bfref.mat_mul_vec(A, x)
    op = HipOperator.from_desc(d, vals, flags=_capi.FLAG_ADJOINT, demote_to_f32=demote)
    tol = 3e-5 if demote else TOL
    assert rel(op.apply_host(x), y_ref) <= tol
    assert rel(op.apply_transpose_host(v), apply_t(v)) <= tol
    op.close()


@pytest.mark.parametrize("demote", [False, True])
@pytest.mark.parametrize("nrhs", [1, 3, 40])
def test_few_row_leaves_on_gpu(demote, nrhs):
    """Row-major few-row leaves (the pass-through W blocks of a streamed butterfly) through the real stage kernel and the
    transposed kernel, mixed with column-major leaves and Identity terms in the same groups; f64 and f32, 1, 3 and 40 RHS."""
    import randgraph
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(41 + nrhs)
    d, vals, dense = randgraph.few_row_operand(rng)        # (Identity terms off the block grid: the dense matrix is the answer)
    _check_against_dense(d, vals, dense, rng, demote, nrhs)
    # merged narrow items, small items four to a wavefront (their own launch), a zero fill between them
    d, vals, dense = randgraph.narrow_items_operand(rng)
    _check_against_dense(d, vals, dense, rng, demote, nrhs)


@pytest.mark.parametrize("demote", [False, True])
def test_transposed_chain_of_few_row_pieces_on_gpu(demote):
    """A block column of 70 - 300 few-row leaves: transposed items that are chains of hundreds of row-major and
    column-major pieces (shared by a workgroup), windows of more than 64 piece descriptors, leaves whose last forward
    task is too narrow for the row-major layout."""
    import randgraph
    rng = np.random.default_rng(99)
    for leaves, width in ((70, 900), (300, 2100)):
        d, vals, dense = randgraph.few_row_column_operand(rng, leaves, width)
        _check_against_dense(d, vals, dense, rng, demote, 1)
    d, vals, dense = randgraph.few_row_column_operand(rng, 90, 520)
    _check_against_dense(d, vals, dense, rng, demote, 3)


def _check_against_dense(d, vals, dense, rng, demote, nrhs):
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    m, n = dense.shape
    op = HipOperator.from_desc(d, vals, flags=_capi.FLAG_ADJOINT, demote_to_f32=demote, max_rhs=nrhs)
    tol = 3e-5 if demote else TOL
    x = rng.standard_normal((n, nrhs)) if nrhs > 1 else rng.standard_normal(n)
    v = rng.standard_normal((m, nrhs)) if nrhs > 1 else rng.standard_normal(m)
    y = op.apply_host(x)
    z = op.apply_transpose_host(v)
    assert rel(y, dense @ x) <= tol and rel(z, dense.T @ v) <= tol
    op.close()


def test_experimental_executors_in_their_own_build():
    """The executors that were measured and set aside (one dependency-driven launch, persistent ticket launch, per-item
    timeline) live in `make experimental` builds only (libbfhip_exp.so, -DBFHIP_EXPERIMENTAL): their bit-identity
    tests run here, in ONE child test run that loads that library instead of the product."""
    import subprocess
    import sys
    lib = os.path.join(ROOT, "butterfly_amd", "csrc", "libbfhip_exp.so")
    assert os.path.exists(lib), "make -C butterfly_amd/csrc experimental (also done by __graft_entry__.build())"
    env = dict(os.environ, BFHIP_LIB_PATH=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "experimental_checks.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
