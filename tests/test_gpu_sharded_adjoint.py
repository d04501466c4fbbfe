"""The adjoint, cov_matvec and GMRES of a SHARDED operator behind the C-ABI (include/bfhip.h "multi-GPU":
bfhipShardedApplyTransposeDevice / CovMatvecDevice / SolveGMRESDevice, bfhipShardedMatNew), on one GPU:

* the RCCL path with a 1-rank communicator (every collective is issued, on one rank);
* "every rank's partial, one after another, summed = the one-GPU adjoint": the operators the W ranks of a job would
  hold (row ranges from bfhipRowPartition; rowsum shares) are compiled and applied in turn, A_r^T to ITS entries of v,
  and the full-length partials are added -- what the closing ncclAllReduce computes.

Reference anchors: src/mat_product.c:312-345, src/mat_block_dense.c:696-758 (Rmul mirrors), examples/covariance/
lbo_cov.c:48-60 (cov_matvec), src/linalg.c:47-317 (bfSolveGMRES).  The 2- and 3-rank torch rendition of the same
steps runs on CPU over gloo (tests/test_dist_cpu.py)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-12


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))


@pytest.mark.parametrize("nrhs", [1, 3])
def test_every_ranks_adjoint_partial_sums_to_the_one_gpu_adjoint(nrhs):
    """N = 65536, k = 4096 (BASELINE configs[1]): 8 row ranges (bfhipRowPartition) and the 8 rowsum shares; each shard's
    operator applies A_r^T to its own rows of v; the sum is the whole operator's A^T v to rounding."""
    import torch
    from butterfly_amd import _capi, helm2_structure as hs
    from butterfly_amd.dist import ShardLayout, row_partition, rowsum_partition
    from butterfly_amd.operator import HipOperator
    n, k, world = 65536, 4096.0, 8
    desc, perm = hs.native_multilevel_structure(hs.circle_points(n), k)
    rng = np.random.default_rng(3)
    shape = (n,) if nrhs == 1 else (n, nrhs)
    v = torch.from_numpy((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2)).cuda()
    full = HipOperator.from_desc(desc, None, seed=5, max_rhs=nrhs, flags=_capi.FLAG_ADJOINT)
    want = full.apply_transpose_device(v).clone()
    y_full = full.apply_device(v).clone()
    full.close()
    # row ranges
    cuts, loads = row_partition(desc, world)
    z = torch.zeros_like(want)
    for r in range(world):
        op = HipOperator.from_desc(desc, None, seed=5, max_rhs=nrhs, flags=_capi.FLAG_ADJOINT, row_range=(cuts[r], cuts[r + 1]))
        assert op.stats()["leafElems"] == loads[r]
        assert torch.equal(op.apply_device(v), y_full[cuts[r]:cuts[r + 1]])        # the forward shard is still the bit-identical one
        z += op.apply_transpose_device(v[cuts[r]:cuts[r + 1]].contiguous())
        op.close()
    assert rel(z.cpu().numpy(), want.cpu().numpy()) <= 1e-13
    # rowsum shares (the 8-rank default): whole block rows + column shares of the shared ones
    top_rows = desc.meta["top_rows"]
    bowner, loads2, segs = rowsum_partition(desc, world)
    lay = ShardLayout(top_rows, [0] * len(top_rows), world, segments=segs)
    z = torch.zeros_like(want)
    for r in range(world):
        root, touched, rows = hs.shard_desc_children(desc, [i for i in range(len(bowner)) if bowner[i] == r])
        op = HipOperator.from_desc(desc, None, root=root, seed=5, max_rhs=nrhs, flags=_capi.FLAG_ADJOINT)
        idx = np.concatenate([np.arange(lay.row_offsets[rb], lay.row_offsets[rb] + top_rows[rb]) for rb in lay.blocks_of[r]])
        vr = v[torch.from_numpy(idx).cuda()].contiguous()
        assert vr.shape[0] == rows
        z += op.apply_transpose_device(vr)
        op.close()
    assert rel(z.cpu().numpy(), want.cpu().numpy()) <= 1e-13


@pytest.mark.parametrize("mode", ["rows", "blocks", "rowsum"])
def test_rccl_sharded_adjoint_one_rank(mode):
    """bfhipShardedApplyTransposeDevice with a 1-rank communicator: rows cut into three segments that are one run of v (read in
    place), the whole operator as (row, col) blocks, and the rowsum layout whose first block row arrives as TWO partials (this
    rank's entries of v are then NOT one run: the gather kernel) -- all equal the plain transposed apply."""
    import torch
    from butterfly_amd import _capi, helm2_structure as hs
    from butterfly_amd.dist import RcclShardedApply, ShardLayout
    from butterfly_amd.operator import HipOperator
    n, k = 8192, 512.0
    desc, perm = hs.native_multilevel_structure(hs.circle_points(n), k)
    rng = np.random.default_rng(21)
    top_rows, trb = desc.meta["top_rows"], desc.top_row_block
    full = HipOperator.from_desc(desc, None, seed=6, max_rhs=3, flags=_capi.FLAG_ADJOINT)
    for nrhs in (1, 3):
        shape = (n,) if nrhs == 1 else (n, nrhs)
        v = torch.from_numpy((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2)).cuda()
        want = full.apply_transpose_device(v).clone()
        if mode == "rowsum":
            ch = desc.children[desc.root]
            m0 = top_rows[0]
            new, col = [], 0
            for i, (c, r0, c0) in enumerate(ch):
                if trb[i] == 0:
                    new.append((c, r0 + (m0 if col % 2 else 0), c0))
                    col += 1
                else:
                    new.append((c, r0 + m0, c0))
            root = desc.add(hs.NODE_BLOCK, n + m0, n, new, hs.BF_TYPE_BLOCK_DENSE)
            op = HipOperator.from_desc(desc, None, root=root, seed=6, max_rhs=nrhs, flags=_capi.FLAG_ADJOINT)
            segs = [(0, 0), (0, 0)] + [(rb, 0) for rb in range(1, len(top_rows))]
            lay = ShardLayout(top_rows, [0] * len(top_rows), 1, segments=segs)
        else:
            op = full
            lay = ShardLayout([top_rows[0], sum(top_rows[1:5]), n - sum(top_rows[:5])], [0, 0, 0], 1) if mode == "rows" else ShardLayout(top_rows, [0] * len(top_rows), 1)
        step = RcclShardedApply(lay, 0, op, 0, nrhs=nrhs, mode=mode)
        got = step.apply_transpose(v)
        torch.cuda.synchronize()
        if mode == "rowsum":
            assert rel(got.cpu().numpy(), want.cpu().numpy()) <= 1e-14
        else:
            assert torch.equal(got, want)
        assert torch.equal(step.apply_transpose(v), got)
        step.close()
        if op is not full:
            op.close()
    # an operator without an adjoint plan is refused
    plain = HipOperator.from_desc(desc, None, seed=6)
    step = RcclShardedApply(ShardLayout(top_rows, [0] * len(top_rows), 1), 0, plain, 0, nrhs=1, mode="rows")
    with pytest.raises(_capi.BfhipError) as e:
        step.apply_transpose(torch.zeros(n, dtype=torch.complex128, device="cuda"))
    assert e.value.code == 1
    step.close()
    plain.close()
    full.close()


def test_rccl_sharded_cov_matvec_and_shim_real_operator():
    """cov_matvec over a sharded REAL operator (bfhipShardedCovMatvecDevice, 1-rank RCCL) against the oracle's own sequence
    (bfVecRealPermute, bfMatRmulVec, GammaLam twice, bfMatMulVec, bfVecRealPermute: examples/covariance/lbo_cov.c:48-60), and
    bfMatMulVec / bfMatRmulVec through the vtable shim of the sharded handle (bfhipShardedMatNew) driven by the oracle's
    virtual dispatch.  Rectangular operator, three row segments."""
    import torch
    import randgraph
    from butterfly_amd import _capi
    from butterfly_amd.dist import RcclShardedApply, ShardLayout, row_partition
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    rng = np.random.default_rng(33)
    desc, vals = randgraph.random_real_operand(rng, depth=3, size_hint=150)
    m, n = desc.rows[desc.root], desc.cols[desc.root]
    A = bfref.from_desc(desc, vals)
    cuts, _ = row_partition(desc, 3)
    gam = rng.random(n) + 0.1
    row_perm = rng.permutation(m)
    rev = np.empty(m, dtype=np.int64); rev[row_perm] = np.arange(m)

    def permute(x, perm):
        out = np.empty_like(x); out[perm] = x
        return out
    v, w = rng.standard_normal(m), rng.standard_normal(n)
    z_cov = permute(bfref.mat_mul_vec(A, gam * (gam * bfref.mat_rmul_vec(A, permute(v, rev)))), row_perm)
    op = HipOperator.from_bfmat(A.ptr.value, flags=_capi.FLAG_ADJOINT)
    step = RcclShardedApply(ShardLayout([cuts[1], cuts[2] - cuts[1], m - cuts[2]], [0, 0, 0], 1), 0, op, 0, nrhs=1, mode="rows")
    dev = torch.device("cuda", 0)
    got = step.cov_matvec(torch.from_numpy(gam).to(dev), torch.from_numpy(row_perm.astype(np.int64)).to(dev), torch.from_numpy(rev).to(dev),
                          torch.from_numpy(v).to(dev)).cpu().numpy()
    assert rel(got, z_cov) <= TOL
    got = step.apply_transpose(torch.from_numpy(v).to(dev)).cpu().numpy()
    assert rel(got, bfref.mat_rmul_vec(A, v)) <= TOL
    # the shim over the sharded handle, through the oracle's dispatch
    a_hip = C.c_void_p(step.mat_new())
    lib = bfref.load()
    assert lib.bfMatGetNumRows(a_hip) == m and lib.bfMatGetNumCols(a_hip) == n
    H = type("H", (), {"ptr": a_hip, "shape": (m, n)})()
    assert rel(bfref.mat_mul_vec(H, w), bfref.mat_mul_vec(A, w)) <= TOL
    assert rel(bfref.mat_rmul_vec(H, v), bfref.mat_rmul_vec(A, v)) <= TOL
    lib.bfMatDelete(C.byref(a_hip))
    step.close()
    op.close()


def test_rccl_sharded_gmres_one_rank_is_the_plain_solve():
    """bfhipShardedSolveGMRESDevice (1-rank RCCL, the operator cut into three row segments) takes the iterations of
    bfhipSolveGMRESOptsDevice on the same operator and returns the same solution bit for bit -- the recurrences are the same
    code around another matvec -- and follows the numpy restatement of bfSolveGMRES (oracle/linalg_ref.py) in the reference's
    Gram-Schmidt order.  Also bfMatMul through the sharded shim."""
    import torch
    from butterfly_amd import _capi
    from butterfly_amd.dist import RcclShardedApply, ShardLayout
    from butterfly_amd.operator import HipOperator
    from oracle import bfref, linalg_ref
    import bie
    n, k = 2048, 128
    desc, root, vals, dense = bie.second_kind_case(n, k)          # I + alpha S through Identity leaves: well conditioned
    A = bfref.from_desc(desc, vals, root=root)
    op = HipOperator.from_bfmat(A.ptr.value, max_rhs=2)
    lay = ShardLayout([700, 900, n - 1600], [0, 0, 0], 1)
    step = RcclShardedApply(lay, 0, op, 0, nrhs=2, mode="rows")
    rng = np.random.default_rng(4)
    for nrhs in (1, 2):
        b = rng.standard_normal((n, nrhs)) + 1j * rng.standard_normal((n, nrhs))
        bd = torch.from_numpy(b[:, 0].copy() if nrhs == 1 else b).cuda()
        for orth in (_capi.GMRES_ORTH_MGS, _capi.GMRES_ORTH_CGS2):
            x1, it1, res1 = op.solve_gmres_device(bd, tol=1e-9, max_num_iter=40, orth="mgs" if orth == _capi.GMRES_ORTH_MGS else "cgs2")
            x2, it2, res2 = step.solve_gmres(bd, tol=1e-9, max_num_iter=40, orthogonalization=orth)
            assert it1 == it2 and res1 == res2 and torch.equal(x1, x2)
            if orth == _capi.GMRES_ORTH_MGS:
                want, iters, hist = linalg_ref.solve_gmres(lambda X: bfref.mat_mul(A, X), b if nrhs > 1 else b[:, 0], tol=1e-9, max_num_iter=40)
                assert it2 == iters and rel(x2.cpu().numpy(), want) <= 1e-8
    # bfMatMul through the shim of the sharded handle
    a_hip = C.c_void_p(step.mat_new())
    lib = bfref.load()
    x = rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))
    X = bfref.dense_complex(x)
    r = lib.bfMatMul(a_hip, X.ptr)
    assert r
    assert rel(bfref.Mat(r).to_numpy(), bfref.mat_mul(A, x)) <= TOL
    lib.bfMatDelete(C.byref(a_hip))
    # a real operator is refused by the solver, a rectangular one too (src/linalg.c:85-87)
    step.close()
    op.close()
