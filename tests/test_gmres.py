"""GMRES: the production caller of the apply path (reference src/linalg.c:47-317).
CPU: the numpy restatement (oracle/linalg_ref.py) driven by the oracle's
bfMatMul solves a second-kind system to the dense answer.  GPU: the
device-resident solver follows the restatement iteration for iteration."""
import numpy as np
import pytest

from oracle import bfref, linalg_ref
import bie


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))


@pytest.fixture(scope="module")
def system():
    n, k = 2048, 128
    desc, root, vals, dense = bie.second_kind_case(n, k)
    A = bfref.from_desc(desc, vals, root=root)
    rng = np.random.default_rng(12)
    b = rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))
    return desc, root, vals, dense, A, b


def test_oracle_gmres_solves_second_kind_system(system):
    desc, root, vals, dense, A, b = system
    assert rel(bfref.mat_mul(A, b), dense @ b) < 1e-10          # I + alpha S through Identity leaves
    x, it, hist = linalg_ref.solve_gmres(lambda v: bfref.mat_mul(A, v), b[:, 0], tol=1e-10, max_num_iter=80)
    assert it < 80 and hist[-1] < 1e-10
    assert rel(x, np.linalg.solve(dense, b[:, 0])) < 1e-8
    # two right-hand sides at once, and the not-converged path (all maxNumIter vectors are used)
    x2, it2, hist2 = linalg_ref.solve_gmres(lambda v: bfref.mat_mul(A, v), b, tol=1e-10, max_num_iter=80)
    assert rel(x2, np.linalg.solve(dense, b)) < 1e-8
    x3, it3, hist3 = linalg_ref.solve_gmres(lambda v: bfref.mat_mul(A, v), b[:, 0], tol=1e-30, max_num_iter=5)
    assert it3 == 5 and len(hist3) == 5


def test_plan_of_decorated_operator_matches_oracle(system):
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    import plan_emulator
    desc, root, vals, dense, A, b = system
    op = HipOperator.from_desc(desc, vals, root=root, flags=_capi.FLAG_PLAN_ONLY)
    assert rel(plan_emulator.run_plan(op, b), bfref.mat_mul(A, b)) < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("nrhs", [1, 2])
def test_device_gmres_follows_the_restatement(system, nrhs, monkeypatch):
    from butterfly_amd.operator import HipOperator
    desc, root, vals, dense, A, b = system
    bb = b[:, 0] if nrhs == 1 else b
    op = HipOperator.from_desc(desc, vals, root=root, max_rhs=nrhs)
    x_ref, it_ref, hist = linalg_ref.solve_gmres(lambda v: bfref.mat_mul(A, v), bb, tol=1e-10, max_num_iter=80)
    # the reference's own order (modified Gram-Schmidt, one basis vector at a time): iteration for iteration
    monkeypatch.setenv("BFHIP_GMRES_MGS", "1")
    x, it, res = op.solve_gmres(bb, tol=1e-10, max_num_iter=80)
    assert it == it_ref
    assert abs(res - hist[-1]) <= 1e-6 * hist[-1] + 1e-16
    assert rel(x, x_ref) < 1e-9
    # default: batched CGS2 (7 launches per iteration whatever j): same Krylov space, the count may move by one
    monkeypatch.delenv("BFHIP_GMRES_MGS")
    x, it, res = op.solve_gmres(bb, tol=1e-10, max_num_iter=80)
    assert abs(it - it_ref) <= 1 and res < 1e-10
    assert rel(x, x_ref) < 1e-8
    assert rel(x, np.linalg.solve(dense, bb)) < 1e-8
    # warm start and an iteration cap: the not-converged path uses all maxNumIter vectors
    x0 = 0.5 * x_ref
    xr, itr, _ = linalg_ref.solve_gmres(lambda v: bfref.mat_mul(A, v), bb, X0=x0, tol=1e-30, max_num_iter=6)
    xg, itg, _ = op.solve_gmres(bb, x0=x0, tol=1e-30, max_num_iter=6)
    assert itg == itr == 6
    assert rel(xg, xr) < 1e-9
    op.close()


@pytest.mark.gpu
def test_device_gmres_on_resident_tensors(system):
    import torch
    from butterfly_amd.operator import HipOperator
    desc, root, vals, dense, A, b = system
    op = HipOperator.from_desc(desc, vals, root=root)
    bd = torch.from_numpy(np.ascontiguousarray(b[:, 0])).cuda()
    x, it, res = op.solve_gmres_device(bd, tol=1e-10, max_num_iter=80)
    assert res < 1e-10
    assert rel(x.cpu().numpy(), np.linalg.solve(dense, b[:, 0])) < 1e-8
    # errors mirror the reference's argument checks (linalg.c:81-92)
    from butterfly_amd import _capi
    with pytest.raises(_capi.BfhipError) as e:
        op.solve_gmres(b[:, 0], max_num_iter=0)
    assert e.value.code == 1
    op.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nrhs", [1, 2])
def test_replicated_vector_gmres_matches_the_device_solver(system, nrhs):
    """dist.sharded_solve_gmres (what a multi-GPU solve runs on every rank around the sharded apply),
    here with world size 1 on the GPU: same iteration count and solution as bfhipSolveGMRESDevice."""
    import torch
    from butterfly_amd.dist import sharded_solve_gmres
    from butterfly_amd.operator import HipOperator
    desc, root, vals, dense, A, b = system
    op = HipOperator.from_desc(desc, vals, root=root, max_rhs=nrhs)
    bd = torch.from_numpy(np.ascontiguousarray(b[:, 0] if nrhs == 1 else b)).cuda()
    x_dev, it_dev, res_dev = op.solve_gmres_device(bd, tol=1e-10, max_num_iter=80)
    x, it, res = sharded_solve_gmres(lambda v: op.apply_device(v), bd, tol=1e-10, max_num_iter=80)
    assert it == it_dev and abs(res - res_dev) <= 1e-6 * res_dev + 1e-16
    assert rel(x.cpu().numpy(), x_dev.cpu().numpy()) < 1e-10
    op.close()


@pytest.mark.gpu
def test_left_preconditioned_gmres(system, monkeypatch):
    """bfSolveGMRES(A, B, X0, tol, maxNumIter, &numIter, M) with a left preconditioner (src/linalg.c:90-97,131,159):
    M^{-1} = block-Jacobi inverse of the dense system, applied as a device operator of its own."""
    import torch
    from butterfly_amd import _capi, helm2_structure as hs
    from butterfly_amd.operator import HipOperator
    desc, root, vals, dense, A, b = system
    n = dense.shape[0]
    nb = 16
    d = hs.Desc(dtype=0)
    pv, ch = {}, []
    minv = np.zeros_like(dense)
    for i in range(nb):
        sl = slice(i * n // nb, (i + 1) * n // nb)
        blk = np.linalg.inv(dense[sl, sl])
        leaf = d.add(hs.NODE_DENSE, blk.shape[0], blk.shape[1])
        pv[leaf] = blk
        ch.append((leaf, sl.start, sl.start))
        minv[sl, sl] = blk
    d.root = d.add(hs.NODE_BLOCK, n, n, ch, hs.BF_TYPE_BLOCK_DIAG)
    op = HipOperator.from_desc(desc, vals, root=root)
    pre = HipOperator.from_desc(d, pv)
    bb = b[:, 0]
    x_ref, it_ref, hist = linalg_ref.solve_gmres(lambda v: bfref.mat_mul(A, v), bb, tol=1e-10, max_num_iter=80, msolve=lambda v: minv @ v)
    _, it_plain, _ = linalg_ref.solve_gmres(lambda v: bfref.mat_mul(A, v), bb, tol=1e-10, max_num_iter=80)
    assert it_ref <= it_plain
    bd = torch.from_numpy(np.ascontiguousarray(bb)).cuda()
    # the reference's order as an explicit per-call option (no environment involved) ...
    x, it, res = op.solve_gmres_device(bd, tol=1e-10, max_num_iter=80, precond=pre, orth="mgs")
    assert it == it_ref and abs(res - hist[-1]) <= 1e-6 * hist[-1] + 1e-16
    assert rel(x.cpu().numpy(), x_ref) < 1e-9
    # ... which wins over the environment override, both ways
    monkeypatch.setenv("BFHIP_GMRES_MGS", "1")
    x, it, res = op.solve_gmres_device(bd, tol=1e-10, max_num_iter=80, precond=pre, orth="cgs2")
    assert abs(it - it_ref) <= 1
    assert rel(x.cpu().numpy(), np.linalg.solve(dense, bb)) < 1e-8
    xe, ite, _ = op.solve_gmres_device(bd, tol=1e-10, max_num_iter=80, precond=pre)          # default: the environment decides
    assert ite == it_ref
    monkeypatch.delenv("BFHIP_GMRES_MGS")
    x2, it2, res = op.solve_gmres_device(bd, tol=1e-10, max_num_iter=80, precond=pre)
    assert it2 == it and torch.equal(x2, x)
    # a real preconditioner would be applied to the complex Krylov vectors as if they were real: TYPE_ERROR
    dr = hs.Desc(dtype=1)
    chr_ = []
    for s_ in range(0, n, 64):
        e_ = min(s_ + 64, n)
        chr_.append((dr.add(hs.NODE_IDENTITY, e_ - s_, e_ - s_), s_, s_))
    dr.root = dr.add(hs.NODE_BLOCK, n, n, chr_, hs.BF_TYPE_BLOCK_DIAG)
    pre_real = HipOperator.from_desc(dr, None)
    with pytest.raises(_capi.BfhipError) as e:
        op.solve_gmres_device(bd, precond=pre_real)
    assert e.value.code == 7
    pre_real.close()
    # shape / device mismatches are refused like linalg.c:92-97
    small = HipOperator.from_desc(d, pv, root=ch[0][0])
    with pytest.raises(Exception):
        op.solve_gmres_device(bd, precond=small)
    small.close(); pre.close(); op.close()
