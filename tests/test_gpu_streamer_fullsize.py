"""BASELINE configs[4] at FULL size on the GPU: the streamed real butterfly of examples/covariance, N = 1 048 576 octree
rows x 65 536 columns (lmax = 255), laid out by the native layout (bfhipStreamerLayoutCreate: the fac_streamer recursion
under the rank model, reference src/fac.c:1080-1294, examples/covariance/lbo_cov.c:120-143), values synthetic, applied in
fp32 (the build's extension, 49.5 GB) and fp64 (the reference's type, 98.9 GB).

The oracle cannot apply 99 GB in a test, so the full-size operator is tied down from both sides:
  * every factor of the product [Psi, W0, ..., W7], cut to a prefix of its top-level blocks (the sub-operators
    bench.py's cpu_baseline samples), is applied by the oracle's bfMatMulVec and by the device, fp64 <= 1e-12, fp32 <= 2e-5:
    the kernels at the real leaf shapes and item classes;
  * the full fp32 operator equals the chain of its nine factors applied one operator at a time (the stage schedule,
    intermediates and reduce passes of the 9-stage plan at full size);
  * linearity, <A x, v> = <x, A^T v> (the transposed plan over the same arena), fp32 against fp64 <= 2e-5;
  * bfhipSave / bfhipLoad: the loaded operator applies bit-identically (full size when the box has the disk for 50 GB,
    else the N = 262144 x 16384 operand)."""
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, LMAX = 1048576, 255


def rel(a, b):
    a, b = np.ravel(np.asarray(a, dtype=np.float64)), np.ravel(np.asarray(b, dtype=np.float64))
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def layout(n, lmax):
    from butterfly_amd import streamer_structure as ss
    pts = ss.fibonacci_sphere(n)
    fd = ss.octree_depth(pts) - 3                    # lbo_cov.c:97-98
    wmax = float(np.sqrt(lmax * (lmax + 1.0)) * 1.0001)
    counts, _ = ss.sphere_band_columns(wmax, fd)
    return ss.native_stream_structure(pts, wmax, fd, counts)


@pytest.fixture(scope="module")
def operand():
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs the 288 GB of an MI355X")
    desc, perm, stats = layout(N, LMAX)
    assert stats["numRows"] == N and stats["numCols"] == (LMAX + 1) ** 2 and stats["numFacs"] == 1 and stats["numW"] == 8
    assert stats["denseReal"] > 1_500_000 and stats["identity"] > 1_000_000 and 95e9 < stats["leafBytes"] < 103e9
    return dict(desc=desc, stats=stats, rng=np.random.default_rng(11))


def test_factor_prefixes_against_the_oracle(operand):
    """Of each of the 9 factors a prefix of its top-level blocks (~0.25 GB fp64 each): oracle bfMatMulVec vs the device."""
    from butterfly_amd import helm2_structure as hs
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    bfref.try_use_openblas()
    desc, rng = operand["desc"], operand["rng"]
    a = desc.arrays()
    sub_elems = desc.subtree_leaf_elems()
    prod = desc.children[desc.root][0][0]
    assert a["kind"][prod] == hs.NODE_PRODUCT
    factors = [f for f, _, _ in desc.children[prod]]
    assert len(factors) == 9
    checked = 0
    for f in factors:
        ch = desc.children[f]
        bk = int(a["blockKind"][f])
        one_col = all(c0 == 0 for _, _, c0 in ch) and bk == hs.BF_TYPE_BLOCK_DENSE
        assert bk == hs.BF_TYPE_BLOCK_DIAG or one_col
        kids, acc = [], 0
        for c, r0, c0 in ch:
            kids.append((c, r0, c0))
            acc += int(sub_elems[c]) * 8
            if acc >= 0.25e9:
                break
        last = kids[-1]
        m = last[1] + int(a["rows"][last[0]])
        n = (last[2] + int(a["cols"][last[0]])) if bk == hs.BF_TYPE_BLOCK_DIAG else int(a["cols"][f])
        sub = desc.add(hs.NODE_BLOCK, m, n, kids, bk)
        a = desc.arrays()
        x = rng.standard_normal(n)
        want = bfref.mat_mul_vec(bfref.from_desc(desc, None, seed=3, root=sub), x)
        for demote, tol in ((False, 1e-12), (True, 2e-5)):
            op = HipOperator.from_desc(desc, None, root=sub, seed=3, demote_to_f32=demote)
            assert rel(op.apply_host(x), want) <= tol, (f, demote)
            op.close()
        checked += acc
    assert checked > 1.5e9


def test_full_operator_fp32_fp64_linearity_adjoint_and_factor_chain(operand):
    import torch
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    desc, stats, rng = operand["desc"], operand["stats"], operand["rng"]
    J = stats["numCols"]
    op64 = HipOperator.from_desc(desc, None, seed=3, flags=_capi.FLAG_ADJOINT)
    st = op64.stats()
    assert st["leafBytes"] == stats["leafBytes"] and st["numStages"] == 9 and op64.shape == (N, J)
    x, z = torch.from_numpy(rng.standard_normal(J)).cuda(), torch.from_numpy(rng.standard_normal(J)).cuda()
    v = torch.from_numpy(rng.standard_normal(N)).cuda()
    y64 = op64.apply_device(x).clone()
    assert torch.isfinite(y64).all() and float(torch.linalg.norm(y64)) > 0
    yl = op64.apply_device(0.75 * x - 1.25 * z).clone()
    assert rel(yl.cpu().numpy(), (0.75 * y64 - 1.25 * op64.apply_device(z)).cpu().numpy()) <= 1e-12
    assert torch.equal(op64.apply_device(x), y64)                                  # run-to-run reproducible
    t64 = op64.apply_transpose_device(v).clone()
    lhs, rhs = float(torch.dot(y64, v)), float(torch.dot(x, t64))
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), float(torch.linalg.norm(y64) * torch.linalg.norm(v)) * 1e-3)
    op64.close()
    del yl
    torch.cuda.empty_cache()

    op32 = HipOperator.from_desc(desc, None, seed=3, flags=_capi.FLAG_ADJOINT, demote_to_f32=True)
    assert op32.stats()["leafBytes"] * 2 == stats["leafBytes"]
    x32, v32 = x.float(), v.float()
    y32 = op32.apply_device(x32).clone()
    assert rel(y32.cpu().numpy(), y64.cpu().numpy()) <= 2e-5                        # the fp32 extension against the reference's type
    t32 = op32.apply_transpose_device(v32).clone()
    assert rel(t32.cpu().numpy(), t64.cpu().numpy()) <= 2e-5
    lhs, rhs = float(torch.dot(y32.double(), v)), float(torch.dot(x, t32.double()))
    assert abs(lhs - rhs) <= 2e-5 * float(torch.linalg.norm(y32.double()) * torch.linalg.norm(v))
    # the nine factors one operator at a time: [Psi, W0, ..., W7] applied right to left (src/mat_product.c:247-280)
    prod = desc.children[desc.root][0][0]
    cur = x32
    for f, _, _ in reversed(desc.children[prod]):
        opf = HipOperator.from_desc(desc, None, root=f, seed=3, demote_to_f32=True)
        cur = opf.apply_device(cur).clone()
        opf.close()
    assert rel(cur.cpu().numpy(), y32.cpu().numpy()) <= 1e-5
    operand["op32"], operand["x32"], operand["y32"], operand["t32"], operand["v32"] = op32, x32, y32, t32, v32
    # the packed copy for A^T (what bench.py --adjoint runs: the transposed expression on the forward kernels, runs of narrow
    # pieces contracted as one block, >= 32768 items per stage) against the transposed kernels on the shared leaves
    opp = HipOperator.from_desc(desc, None, seed=3, flags=_capi.FLAG_ADJOINT_PACKED, demote_to_f32=True)
    tp = opp.apply_transpose_device(v32).clone()
    assert rel(tp.cpu().numpy(), t32.cpu().numpy()) <= 2e-5
    assert rel(tp.cpu().numpy(), t64.cpu().numpy()) <= 2e-5
    assert torch.equal(opp.apply_device(x32), y32)                                  # its forward plan is the same plan
    assert torch.equal(opp.apply_transpose_device(v32), tp)                         # reproducible
    opp.close()


def test_save_load_round_trip(operand, tmp_path):
    import torch
    from butterfly_amd import _capi
    from butterfly_amd.operator import HipOperator
    if "op32" not in operand:
        pytest.skip("the full-size operator was not built")
    op32, x32, y32, t32, v32 = (operand[k] for k in ("op32", "x32", "y32", "t32", "v32"))
    need = op32.stats()["arenaBytes"] * 1.05
    where = str(tmp_path)
    full = shutil.disk_usage(where).free > need + 5e9
    if full:
        path = os.path.join(where, "streamer_n1m_f32.bfhip")
        op32.save(path)
        op32.close()
        torch.cuda.empty_cache()
        back = HipOperator.load(path)
        assert torch.equal(back.apply_device(x32), y32) and torch.equal(back.apply_transpose_device(v32), t32)
        back.close()
        os.remove(path)
        return
    # not enough disk for 50 GB on this box: the N = 262144 x 16384 operand (4.7 GB) instead
    op32.close()
    desc, perm, stats = layout(262144, 127)
    op = HipOperator.from_desc(desc, None, seed=3, flags=_capi.FLAG_ADJOINT, demote_to_f32=True)
    rng = np.random.default_rng(2)
    x = torch.from_numpy(rng.standard_normal(stats["numCols"])).float().cuda()
    v = torch.from_numpy(rng.standard_normal(262144)).float().cuda()
    y, t = op.apply_device(x).clone(), op.apply_transpose_device(v).clone()
    path = os.path.join(where, "streamer_n262144_f32.bfhip")
    op.save(path)
    op.close()
    back = HipOperator.load(path)
    assert torch.equal(back.apply_device(x), y) and torch.equal(back.apply_transpose_device(v), t)
    back.close()
    os.remove(path)
