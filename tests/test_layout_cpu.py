"""The native layout (bfhip_layout.c, bfhipHelm2LayoutCreate) against the Python restatement of the
reference's structure logic (butterfly_amd/helm2_structure.py, itself pinned to the survey's probe
of the real reference in test_structure.py): identical descriptor arrays, recipes and permutation."""
import numpy as np
import pytest

from butterfly_amd import _capi
from butterfly_amd import helm2_structure as hs


def shapes():
    t = lambda n: 2 * np.pi * np.arange(n) / n
    yield "circle-4096-k100", hs.circle_points(4096), 100.0
    yield "circle-16384-k1024", hs.circle_points(16384), 1024.0
    yield "ellipse-6000-k150", np.stack([np.cos(t(6000)), 0.35 * np.sin(t(6000))], axis=1), 150.0
    yield "kite-5000-k80", np.stack([np.cos(t(5000)) + 0.65 * np.cos(2 * t(5000)) - 0.65, 1.5 * np.sin(t(5000))], axis=1) / 1.5, 80.0
    h = 2500
    yield "two-circles-5000-k120", np.concatenate([np.stack([0.5 * np.cos(t(h)) - 0.6, 0.5 * np.sin(t(h))], axis=1),
                                                   np.stack([0.25 * np.cos(t(h)) + 0.55, 0.25 * np.sin(t(h)) + 0.2], axis=1)]), 120.0
    rng = np.random.default_rng(3)
    yield "random-3000-k60", rng.random((3000, 2)), 60.0


@pytest.mark.parametrize("name,pts,k", list(shapes()), ids=[s[0] for s in shapes()])
def test_native_layout_equals_the_python_restatement(name, pts, k):
    desc, _, perm = hs.helm2_multilevel_structure(pts, k, recipes=True)
    lay = _capi.Helm2Layout(pts, k)
    assert np.array_equal(lay.perm, perm)
    assert np.array_equal(lay.tree_points, pts[perm])
    want = desc.arrays()
    got = lay.arrays()
    assert lay.num_nodes == desc.num_nodes and lay.root == desc.root and lay.dtype == desc.dtype
    for key in ("kind", "rows", "cols", "childBegin", "childNode", "childRow0", "childCol0", "blockKind"):
        assert np.array_equal(got[key], want[key]), key
    assert lay.top_row_block == desc.top_row_block
    ref = _capi.recipe_array(desc.recipe)
    assert lay.recipes.shape == ref.shape
    for f in ("node", "kind"):
        assert np.array_equal(lay.recipes[f], ref[f]), f
    for ps in ("src", "equiv", "tgt"):
        for f in ("kind", "count", "first", "cx", "cy", "r"):
            assert np.array_equal(lay.recipes[ps][f], ref[ps][f]), (ps, f)


def test_layout_argument_errors():
    lib = _capi.load()
    import ctypes as C
    h = C.c_void_p()
    pts = np.zeros((8, 2))                       # coincident points: refused, not an endless subdivision
    assert lib.bfhipHelm2LayoutCreate(pts.ctypes.data, 8, 1.0, C.byref(h)) == 1 and not h.value
    pts = hs.circle_points(64)
    assert lib.bfhipHelm2LayoutCreate(pts.ctypes.data, 64, 0.0, C.byref(h)) == 1
    assert lib.bfhipHelm2LayoutCreate(None, 64, 1.0, C.byref(h)) == 1
    assert lib.bfhipHelm2LayoutCreate(pts.ctypes.data, 1, 1.0, C.byref(h)) == 1


def test_array_backed_descriptor_shards_like_the_python_one():
    """bench.py runs on the native layout: weights, row shards and block shards must be the ones the
    Python descriptor gives (compared through the planner: identical stage / item / piece counts)."""
    from butterfly_amd.dist import assign_row_blocks, block_weights, choose_mode, row_block_weights
    from butterfly_amd.operator import HipOperator
    pts, k = hs.circle_points(8192), 512.0
    d_py, _, perm = hs.helm2_multilevel_structure(pts, k)
    d_c, perm_c = hs.native_multilevel_structure(pts, k)
    assert np.array_equal(perm, perm_c) and d_c.meta["top_rows"] == d_py.meta["top_rows"]
    assert d_c.meta["stats"] == d_py.meta["stats"] and d_c.leaf_elems() == d_py.leaf_elems()
    assert row_block_weights(d_c) == row_block_weights(d_py) and block_weights(d_c) == block_weights(d_py)
    assert choose_mode(d_c, 8) == choose_mode(d_py, 8)
    owner, _ = assign_row_blocks(row_block_weights(d_py), 3)
    mine = [rb for rb, o in enumerate(owner) if o == 1]
    stats = []
    for d in (d_py, d_c):
        root, nloc = hs.shard_desc(d, mine)
        broot = hs.shard_desc_blocks(d, [3, 17, 40, 41])
        row = []
        for r in (None, root, broot):
            op = HipOperator.from_desc(d, None, root=r, flags=_capi.FLAG_PLAN_ONLY)
            st = op.stats()
            row.append((nloc, st["numRows"], st["numCols"], st["numStages"], st["numItems"], st["numPieces"], st["leafElems"], st["tempElems"]))
            op.close()
        stats.append(row)
    assert stats[0] == stats[1]


def test_two_tree_layout_equals_the_python_restatement():
    """bfFacHelm2MakeMultilevel(helm, srcTree, tgtTree) with different trees (the evaluation butterfly of
    examples/multiple_scattering): rows follow the target quadtree."""
    n, m, k = 5000, 3500, 90.0
    t, u = 2 * np.pi * np.arange(n) / n, 2 * np.pi * np.arange(m) / m
    src = np.stack([np.cos(t), 0.6 * np.sin(t)], axis=1)
    tgt = np.stack([1.8 + 0.9 * np.cos(u), 0.4 + 0.7 * np.sin(u)], axis=1)
    desc, _, (ps, pt) = hs.helm2_multilevel_structure(src, k, recipes=True, tgt_points=tgt)
    lay = _capi.Helm2Layout(src, k, tgt)
    assert np.array_equal(lay.perm, ps) and np.array_equal(lay.tgt_perm, pt)
    assert np.array_equal(lay.tree_points, src[ps]) and np.array_equal(lay.tgt_tree_points, tgt[pt])
    want, got = desc.arrays(), lay.arrays()
    for key in ("kind", "rows", "cols", "childBegin", "childNode", "childRow0", "childCol0", "blockKind"):
        assert np.array_equal(got[key], want[key]), key
    assert int(got["rows"][lay.root]) == m and int(got["cols"][lay.root]) == n and desc.meta["stats"]["products"]
    ref = _capi.recipe_array(desc.recipe)
    for ps_ in ("src", "equiv", "tgt"):
        for f in ("kind", "count", "first", "cx", "cy", "r"):
            assert np.array_equal(lay.recipes[ps_][f], ref[ps_][f]), (ps_, f)
    assert set(np.unique(lay.recipes["tgt"]["kind"])) >= {_capi.PTS_TREE_TGT}


def test_single_pair_layout_equals_the_python_restatement():
    """bfFacHelm2MakeSingleLevel (examples/simple/bf_one_block.c): one butterfly for a node pair."""
    import ctypes as C
    pts, k = hs.circle_points(2048), 128.0
    src_path, tgt_path = (0, 0), (3, 1)
    desc, _, perm, sn, tn = hs.single_product_structure(pts, k, src_path, tgt_path)
    lay = _capi.Helm2Layout(pts, k, single=(src_path, tgt_path))
    assert np.array_equal(lay.perm, perm) and lay.root == desc.root
    want, got = desc.arrays(), lay.arrays()
    for key in ("kind", "rows", "cols", "childBegin", "childNode", "childRow0", "childCol0", "blockKind"):
        assert np.array_equal(got[key], want[key]), key
    assert int(got["rows"][lay.root]) == tn.npts and int(got["cols"][lay.root]) == sn.npts
    ref = _capi.recipe_array(desc.recipe)
    for ps in ("src", "equiv", "tgt"):
        for f in ("kind", "count", "first", "cx", "cy", "r"):
            assert np.array_equal(lay.recipes[ps][f], ref[ps][f]), (ps, f)
    lib, h = _capi.load(), C.c_void_p()
    bad = np.array([9, 0], dtype=np.uint32)
    ok = np.array([0, 0], dtype=np.uint32)
    assert lib.bfhipHelm2LayoutCreateSingle(pts.ctypes.data, len(pts), k, bad.ctypes.data, 2, ok.ctypes.data, 2, C.byref(h)) == 1
    assert lib.bfhipHelm2LayoutCreateSingle(pts.ctypes.data, len(pts), k, ok.ctypes.data, 2, ok.ctypes.data, 1, C.byref(h)) == 1
