"""The native layout of a streamed butterfly (butterfly_amd/csrc/bfhip_streamer_layout.c, bfhipStreamerLayoutCreate)
against the Python restatement of the reference's recursion (butterfly_amd/streamer_structure.py, itself held to the
survey's probe of the real reference in tests/test_streamer_structure.py): the same rank answers must give the same
flat descriptor, array for array -- octree order, node order (children before parents), offsets, block kinds."""
import ctypes as C

import numpy as np
import pytest

from butterfly_amd import _capi, streamer_structure as ss


def both(n, lmax, fd=None, points=None, **kw):
    pts = ss.fibonacci_sphere(n) if points is None else points
    tree = ss.Octree(pts, 1)
    fd = tree.max_depth - 3 if fd is None else fd
    wmax = float(np.sqrt(lmax * (lmax + 1.0)) * 1.0001)
    counts, _ = ss.sphere_band_columns(wmax, fd)
    model = ss.LboRankModel(tree.num_points, wmax, fd, counts, alpha=kw.get("alpha", 1.75), delta=kw.get("delta", 3.0))
    st = ss.stream_structure(tree, wmax, fd, counts, model=model, min_rows=kw.get("min_rows", 20), min_cols=kw.get("min_cols", 20),
                             max_cols=kw.get("max_cols"))
    g = st.get_mat()
    d, _ = ss.to_desc(g, with_values=False)
    nd, perm, stats = ss.native_stream_structure(pts, wmax, fd, counts, **kw)
    return tree, st, g, d, nd, perm, stats


@pytest.mark.parametrize("n,lmax,kw", [(4096, 31, {}), (16384, 31, {}), (16384, 63, {}), (8192, 47, dict(alpha=1.25, delta=2.0)),
                                       (8192, 31, dict(min_rows=12, min_cols=30)), (16384, 63, dict(max_cols=1500)), (3000, 23, {})])
def test_native_layout_equals_the_python_restatement(n, lmax, kw):
    tree, st, g, d, nd, perm, stats = both(n, lmax, **kw)
    a, b = d.arrays(), nd.arrays()
    assert d.root == nd.root and d.num_nodes == nd.num_nodes and nd.dtype == 1
    for key in a:
        assert a[key].dtype == b[key].dtype and np.array_equal(a[key], b[key]), key
    assert np.array_equal(perm, tree.perm)
    gs = ss.graph_stats(g)
    for key in ("product", "blockCoo", "blockDense", "blockDiag", "denseReal", "identity", "maxNest", "leafBytes"):
        assert gs[key] == stats[key], key
    assert stats["svds"] == st.stats["svds"] and stats["merges"] == st.stats["merges"] and stats["feeds"] == st.stats["feeds"]
    assert stats["numFacs"] == len(st.partial) and stats["numW"] == len(st.partial[-1].W) and stats["rowNodes"] == len(st.partial[-1].row_nodes)
    assert stats["octreeDepth"] == tree.max_depth == ss.octree_depth(ss.fibonacci_sphere(n))
    assert stats["numRows"] == n and stats["numCols"] == g.n


def test_native_layout_on_points_that_are_not_a_sphere():
    """a cloud with clusters and ties in the octant tests (points on the splitting planes go low, src/octree_node.c:105-140)"""
    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.standard_normal((3000, 3)) * 0.1 + 0.5, rng.uniform(-1, 1, (2000, 3)),
                          np.array([[0.0, 0.0, 0.0], [0.25, 0.0, -0.25], [0.5, 0.5, 0.5]])])
    tree, st, g, d, nd, perm, stats = both(len(pts), 23, points=pts)
    a, b = d.arrays(), nd.arrays()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    assert np.array_equal(perm, tree.perm)


def test_the_native_descriptor_compiles_and_applies_like_the_python_one():
    """plan-only compile + the numpy plan interpreter on both descriptors: one operand"""
    import plan_emulator
    from butterfly_amd.operator import HipOperator
    tree, st, g, d, nd, perm, stats = both(4096, 31)
    x = np.random.default_rng(0).standard_normal(g.n)
    ya = plan_emulator.run_plan(HipOperator.from_desc(d, None, seed=4, flags=_capi.FLAG_PLAN_ONLY), x)
    yb = plan_emulator.run_plan(HipOperator.from_desc(nd, None, seed=4, flags=_capi.FLAG_PLAN_ONLY), x)
    assert np.array_equal(ya, yb)
    assert int(nd.subtree_leaf_elems()[nd.root]) * 8 == stats["leafBytes"]


def test_bad_arguments_are_refused():
    lib = _capi.load()
    pts = ss.fibonacci_sphere(64)
    h = C.c_void_p()
    spec = _capi.BfhipStreamerSpec()
    spec.structSize = C.sizeof(spec)
    spec.colDepth, spec.wmax = 1, 4.0
    assert lib.bfhipStreamerLayoutCreate(pts.ctypes.data, 64, C.byref(spec), C.byref(h)) == 1        # no band columns
    bands = np.array([5, 7], dtype=np.uint64)
    spec.bandColumns = bands.ctypes.data
    spec.structSize = 8
    assert lib.bfhipStreamerLayoutCreate(pts.ctypes.data, 64, C.byref(spec), C.byref(h)) == 1        # struct too small
    spec.structSize = C.sizeof(spec)
    assert lib.bfhipStreamerLayoutCreate(None, 64, C.byref(spec), C.byref(h)) == 1
    assert lib.bfhipStreamerLayoutCreate(pts.ctypes.data, 64, C.byref(spec), C.byref(h)) == 0 and h
    lib.bfhipStreamerLayoutFree(C.byref(h))
    assert not h
    dup = np.zeros((10, 3))                                       # coincident points never separate: refused, not an endless recursion
    with pytest.raises(_capi.BfhipError):
        ss.native_stream_structure(dup, 4.0, 1, [5, 7])
