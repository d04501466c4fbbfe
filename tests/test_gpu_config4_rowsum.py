"""BASELINE configs[3] with the 8-rank DEFAULT partition: `rowsum` (whole top-level block rows + a column share of the block rows
that have to be shared; no leaf replicated, one all-gather, the two partials of a shared row added in list order).  The rows
partition is held to the oracle in tests/test_gpu_config4.py; this module compiles the heaviest RANK'S SHARE of the rowsum
partition of N = 1 048 576 (~115 GB) exactly as bench.py would on that rank and checks a whole top-level block of it against the
oracle, a shared block row's partial against linearity, and the share's adjoint against the transpose identity."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 1048576
WORLD = 8
TOL = 1e-12


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))


@pytest.fixture(scope="module")
def share():
    import torch
    from butterfly_amd import _capi, helm2_structure as hs
    from butterfly_amd.dist import ShardLayout, block_weights, choose_mode, rowsum_partition
    from butterfly_amd.operator import HipOperator
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs the 288 GB of an MI355X")
    desc, _ = hs.native_multilevel_structure(hs.circle_points(N), N / 16.0)
    assert choose_mode(desc, WORLD) == "rowsum"
    bw = block_weights(desc)
    bowner, loads, segs = rowsum_partition(desc, WORLD)
    assert max(loads) <= 1.005 * sum(loads) / WORLD and sum(loads) == sum(bw)          # nothing replicated, balanced to half a percent
    top_rows = desc.meta["top_rows"]
    lay = ShardLayout(top_rows, [0] * len(top_rows), WORLD, segments=segs)
    heavy = int(np.argmax(loads))
    kids = [i for i in range(len(bowner)) if bowner[i] == heavy]
    root, touched, rows = hs.shard_desc_children(desc, kids)
    assert touched == lay.blocks_of[heavy] and rows == lay.rows_of[heavy]
    op = HipOperator.from_desc(desc, None, root=root, device=0, seed=7, flags=_capi.FLAG_ADJOINT)
    st = op.stats()
    assert st["leafBytes"] == loads[heavy] * 16 and st["leafBytes"] > 100e9 and st["numRows"] == rows
    # local row offset of every block row this rank touches (its operator yields them in list order, compacted)
    local = {}
    pos = 0
    for rb in lay.blocks_of[heavy]:
        local[rb] = pos
        pos += top_rows[rb]
    shared_rows = sorted({rb for rb, _ in segs if sum(1 for b2, _ in segs if b2 == rb) > 1} & set(lay.blocks_of[heavy]))
    yield dict(desc=desc, op=op, kids=kids, bw=bw, lay=lay, local=local, shared_rows=shared_rows, rng=np.random.default_rng(6), rows=rows)
    op.close()


def test_a_whole_block_of_the_rowsum_share_matches_the_oracle(share):
    import torch
    from butterfly_amd import helm2_structure as hs
    from oracle import bfref
    bfref.try_use_openblas()
    desc, op, kids, bw, lay, local, rng = (share[k] for k in ("desc", "op", "kids", "bw", "lay", "local", "rng"))
    trb = desc.top_row_block
    i = min(kids, key=lambda j: bw[j])                      # the lightest top-level block of this rank: affordable for the oracle
    node, r0, c0 = desc.children[desc.root][i]
    m, n = int(desc.rows[node]), int(desc.cols[node])
    x = np.zeros(N, dtype=complex)
    x[c0:c0 + n] = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)
    y = op.apply_device(torch.from_numpy(x).cuda()).cpu().numpy()
    A = bfref.from_desc(desc, None, seed=7, root=hs.shard_desc_blocks(desc, [i]))
    want = bfref.mat_mul(A, x[:, None])[:, 0]
    lo = local[trb[i]] + (r0 - int(lay.row_offsets[trb[i]]))
    assert rel(y[lo:lo + m], want[r0:r0 + m]) <= TOL, (i, m, n)
    # x is supported on one column block: every other row of the share sees only blocks of that block column
    other = np.ones(len(y), dtype=bool)
    other[lo:lo + m] = False
    cols_hit = [j for j in kids if desc.children[desc.root][j][2] == c0 and j != i]
    if not cols_hit:
        assert not y[other].any()


def test_linearity_adjoint_and_the_shared_block_row(share):
    """The share holds a column share of (at least) one block row: its partial result for that row is one of the two terms the
    closing sum adds.  Linearity and reproducibility on the whole share; <A_r x, v> = <x, A_r^T v> for its adjoint plan."""
    import torch
    op, rng, rows = share["op"], share["rng"], share["rows"]
    assert share["shared_rows"], "the heaviest rank of the 8-way rowsum partition shares a block row"
    vec = lambda k: torch.from_numpy((rng.standard_normal(k) + 1j * rng.standard_normal(k)) / np.sqrt(2)).cuda()
    x, z = vec(N), vec(N)
    a, b = 0.75 - 0.5j, -1.25 + 2.0j
    yx, yz = op.apply_device(x).clone(), op.apply_device(z).clone()
    yl = op.apply_device(a * x + b * z)
    torch.cuda.synchronize()
    assert rel(yl.cpu().numpy(), (a * yx + b * yz).cpu().numpy()) <= TOL
    assert torch.equal(op.apply_device(x), yx)
    v = vec(rows)
    zt = op.apply_transpose_device(v)
    lhs, rhs = torch.sum(yx * v), torch.sum(x * zt)
    assert abs(lhs - rhs) / abs(lhs) <= 1e-12
