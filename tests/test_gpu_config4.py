"""BASELINE configs[3]: fac_helm2 N = 1 048 576, k = 65536 (919 GB of leaves), rows dealt to 8 GPUs with one
RCCL all-gather.  One GPU can hold any one rank's share (~125 GB: 1/8 of the operator plus the replicated
first-applied factors of the block row it shares): this module compiles the HEAVIEST rank's shard of the 8-way
row partition exactly as bench.py would on that rank, checks a top-level block of it against the oracle, ties the
rest together with linearity / reproducibility, and drives the closing collective of the C-ABI (in-place
ncclAllGather + segment reorder) with a 1-rank communicator."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 1048576
WORLD = 8
TOL = 1e-12


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))


@pytest.fixture(scope="module")
def shard():
    import torch
    from butterfly_amd import helm2_structure as hs
    from butterfly_amd.dist import block_weights, choose_mode, row_partition
    from butterfly_amd.operator import HipOperator
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs the 288 GB of an MI355X")
    desc, _ = hs.native_multilevel_structure(hs.circle_points(N), N / 16.0)
    assert choose_mode(desc, WORLD) == "rowsum" and choose_mode(desc, WORLD, "rows") == "rows"      # this module holds the ROWS shard: the bit-identical one
    bw = block_weights(desc)
    cuts, loads = row_partition(desc, WORLD)
    # 12 top-level row blocks on 8 ranks: 4 block rows are split, their first-applied factors held twice
    assert max(loads) / (sum(loads) / WORLD) < 1.04 and sum(loads) < 1.1 * sum(bw)
    heavy = int(np.argmax(loads))
    a, b = cuts[heavy], cuts[heavy + 1]
    from butterfly_amd import _capi
    op = HipOperator.from_desc(desc, None, device=0, seed=7, row_range=(a, b), flags=_capi.FLAG_ADJOINT)      # + the shard's adjoint plan (shared leaves)
    st = op.stats()
    assert st["leafBytes"] == loads[heavy] * 16 and st["leafBytes"] > 110e9 and st["numRows"] == b - a
    # the top-level blocks whose rows this rank holds completely
    mine = [i for i, (node, r0, c0) in enumerate(desc.children[desc.root]) if a <= r0 and r0 + int(desc.rows[node]) <= b]
    assert mine
    yield dict(desc=desc, op=op, mine=mine, bw=bw, rng=np.random.default_rng(5), a=a, b=b, cuts=cuts)
    op.close()


def test_a_top_level_block_of_the_heaviest_shard_matches_the_oracle(shard):
    import torch
    from butterfly_amd import helm2_structure as hs
    from oracle import bfref
    bfref.try_use_openblas()
    desc, op, mine, bw, rng, a = shard["desc"], shard["op"], shard["mine"], shard["bw"], shard["rng"], shard["a"]
    assert op.shape == (shard["b"] - a, N)
    i = min(mine, key=lambda j: bw[j])                     # the lightest whole block of this rank: ~1 GB, affordable for the oracle
    node, r0, c0 = desc.children[desc.root][i]
    m, n = int(desc.rows[node]), int(desc.cols[node])
    # no other block may share the column block AND the row block (then y[r0:r0+m] is A_rc x_c alone for x supported on c)
    assert sum(1 for j in range(len(bw)) if desc.children[desc.root][j][1] == r0 and desc.children[desc.root][j][2] == c0) == 1
    x = np.zeros(N, dtype=complex)
    x[c0:c0 + n] = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)
    y = op.apply_device(torch.from_numpy(x).cuda()).cpu().numpy()
    A = bfref.from_desc(desc, None, seed=7, root=hs.shard_desc_blocks(desc, [i]))
    want = bfref.mat_mul(A, x[:, None])[:, 0]
    assert rel(y[r0 - a:r0 - a + m], want[r0:r0 + m]) <= TOL, (i, m, n)


def test_linearity_and_reproducibility_of_the_shard(shard):
    import torch
    op, rng = shard["op"], shard["rng"]
    vec = lambda: torch.from_numpy((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / np.sqrt(2)).cuda()
    x, z = vec(), vec()
    a, b = 0.75 - 0.5j, -1.25 + 2.0j
    yx, yz = op.apply_device(x), op.apply_device(z)
    yl = op.apply_device(a * x + b * z)
    torch.cuda.synchronize()
    assert rel(yl.cpu().numpy(), (a * yx + b * yz).cpu().numpy()) <= TOL
    assert torch.equal(op.apply_device(x), yx)


def test_adjoint_of_the_heaviest_shard(shard):
    """The shard's adjoint plan (round 5: the transposed task list pruned by reachability from the rank's rows, over the same
    125 GB of leaves): <A_r x, v> = <x, A_r^T v> at full size ties it to the forward plan that the tests above hold to the
    oracle; through the C-ABI's sharded adjoint step (1-rank communicator: this rank's rows are its whole world) the
    all-reduce leaves the same vector."""
    import torch
    from butterfly_amd.dist import RcclShardedApply, ShardLayout
    op, rng = shard["op"], shard["rng"]
    rows = shard["b"] - shard["a"]
    x = torch.from_numpy((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / np.sqrt(2)).cuda()
    v = torch.from_numpy((rng.standard_normal(rows) + 1j * rng.standard_normal(rows)) / np.sqrt(2)).cuda()
    y = op.apply_device(x).clone()
    z = op.apply_transpose_device(v).clone()
    torch.cuda.synchronize()
    assert z.shape == (N,)
    lhs, rhs = torch.sum(y * v), torch.sum(x * z)
    assert abs(lhs - rhs) / abs(lhs) <= 1e-12
    assert torch.equal(op.apply_transpose_device(v), z)
    step = RcclShardedApply(ShardLayout([rows // 3, rows - rows // 3], [0, 0], 1), 0, op, 0, nrhs=1, mode="rows")
    assert torch.equal(step.apply_transpose(v), z)
    step.close()


def test_closing_collective_on_the_shard(shard):
    """bfhipShardedApplyDevice in "rows" mode on the heaviest shard: local stages + the in-place ncclAllGather of this
    rank's rows on the apply stream + the segment reorder (a 1-rank communicator whose world is this rank's row range,
    cut into two segments so that the reorder kernel has something to put in place)."""
    import torch
    from butterfly_amd.dist import RcclShardedApply, ShardLayout
    desc, op, rng = shard["desc"], shard["op"], shard["rng"]
    rows = shard["b"] - shard["a"]
    x = torch.from_numpy((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / np.sqrt(2)).cuda()
    want = op.apply_device(x).clone()
    step = RcclShardedApply(ShardLayout([rows // 3, rows - rows // 3], [0, 0], 1), 0, op, 0, nrhs=1, mode="rows")
    got = step(x)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    loc, coll = step.last_times()          # the events exist and are ordered; how long the stages take is bench.py's business, not a correctness test's
    assert loc > 0 and coll >= 0, (loc, coll)
    step.close()
