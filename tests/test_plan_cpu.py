"""Host logic of the engine, without a GPU: the flattened plan (scheduling,
row groups, packing, reduce intervals) is read back through the inspection
C-ABI and interpreted in numpy (tests/plan_emulator.py), then compared with
the CPU oracle on the same operand."""
import ctypes as C
import os

import numpy as np
import pytest

from butterfly_amd import _capi, helm2_structure as hs
from butterfly_amd.operator import HipOperator
from oracle import bfref, helm2_build as hb
from fixtures import load_fixture
import plan_emulator
import randgraph

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PLAN = dict(flags=_capi.FLAG_PLAN_ONLY)


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))


@pytest.mark.parametrize("n,k", [(1024, 100), (2048, 128)])
def test_helm2_plan_matches_oracle(helm2_cases, n, k):
    desc, tp, vals = helm2_cases(n, k)
    A = bfref.from_desc(desc, vals)
    x = hb.complex_randn(n, 0)
    y_ref = bfref.mat_mul(A, x)
    op = HipOperator.from_desc(desc, vals, **PLAN)
    assert rel(plan_emulator.run_plan(op, x), y_ref) < 1e-13
    st = op.stats()
    assert st["leafElems"] == desc.leaf_elems() and st["leafBytes"] == A.num_bytes() == op.num_bytes()
    assert st["arenaBytes"] == st["leafBytes"]          # complex128: no padding
    # the drop-in route: walk the BfMat graph instead of the descriptor
    op2 = HipOperator.from_bfmat(A.ptr.value, **PLAN)
    assert op2.stats()["numStages"] == st["numStages"]
    assert rel(plan_emulator.run_plan(op2, x), y_ref) < 1e-13
    # multi-RHS
    rng = np.random.default_rng(0)
    xm = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
    assert rel(plan_emulator.run_plan(op, xm), bfref.mat_mul(A, xm)) < 1e-13


def test_synthetic_values_identical_host_and_oracle():
    desc, root, perm = hs.helm2_multilevel_structure(hs.circle_points(2048), 128)
    x = hb.complex_randn(2048, 3)
    y_ref = bfref.mat_mul(bfref.from_desc(desc, None, seed=99), x)
    op = HipOperator.from_desc(desc, None, seed=99, **PLAN)
    assert rel(plan_emulator.run_plan(op, x), y_ref) < 1e-13
    op_other = HipOperator.from_desc(desc, None, seed=100, **PLAN)
    assert rel(plan_emulator.run_plan(op_other, x), y_ref) > 0.1


def test_golden_one_block_plan():
    desc, vals, ex = load_fixture(os.path.join(GOLD, "helm2_one_block_n2048_k128.npz"))
    op = HipOperator.from_desc(desc, vals, **PLAN)
    y = plan_emulator.run_plan(op, ex["x"])
    assert op.stats()["numStages"] == 3
    assert rel(y, ex["y_oracle"]) < 1e-13 and rel(y, ex["y_dense"]) < 1e-11


@pytest.mark.parametrize("seed", range(12))
def test_random_nested_real_graphs(seed):
    """Arbitrary nesting of all container types, Identity leaves, ragged
    sizes, empty block rows (zero fill): real operands through bfMatMulVec."""
    rng = np.random.default_rng(1000 + seed)
    desc, vals = randgraph.random_real_operand(rng, depth=int(rng.integers(1, 5)), size_hint=int(rng.integers(8, 200)))
    n = desc.cols[desc.root]
    x = rng.standard_normal(n)
    want = randgraph.densify(desc, vals, desc.root) @ x
    A = bfref.from_desc(desc, vals)
    assert rel(bfref.mat_mul_vec(A, x) + 1, want + 1) < 1e-12
    for demote in (False, True):
        op = HipOperator.from_desc(desc, vals, demote_to_f32=demote, **PLAN)
        y = plan_emulator.run_plan(op, x)
        assert rel(y + 1, want + 1) < (2e-5 if demote else 1e-12)
    op3 = HipOperator.from_bfmat(A.ptr.value, **PLAN)
    assert rel(plan_emulator.run_plan(op3, x) + 1, want + 1) < 1e-12


@pytest.mark.parametrize("seed", range(6))
def test_random_nested_complex_graphs(seed):
    rng = np.random.default_rng(2000 + seed)
    desc, vals = randgraph.random_operand(rng, depth=3, size_hint=int(rng.integers(20, 400)), cplx=True)
    n = desc.cols[desc.root]
    x = rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))
    want = randgraph.densify(desc, vals, desc.root) @ x
    op = HipOperator.from_desc(desc, vals, **PLAN)
    assert rel(plan_emulator.run_plan(op, x) + 1, want + 1) < 1e-12


def test_wide_and_tall_leaves_are_split():
    """cols > xcap (256) are cut into column pieces, rows > 64 into items."""
    rng = np.random.default_rng(7)
    d = hs.Desc(dtype=0)
    vals = {}
    a = d.add(hs.NODE_DENSE, 150, 700); vals[a] = rng.standard_normal((150, 700)) + 1j * rng.standard_normal((150, 700))
    b = d.add(hs.NODE_DENSE, 700, 3); vals[b] = rng.standard_normal((700, 3)) + 1j * rng.standard_normal((700, 3))
    d.root = d.add(hs.NODE_PRODUCT, 150, 3, [(a, 0, 0), (b, 0, 0)])
    x = rng.standard_normal(3) + 1j * rng.standard_normal(3)
    op = HipOperator.from_desc(d, vals, **PLAN)
    st = op.stats()
    assert st["numStages"] == 2
    # wide leaf: 700 columns -> 16-row items (128 KiB cap) of 3 column pieces; tall narrow leaf: 64-row items
    assert st["numItems"] == 10 + 11         # ceil(150/16) + ceil(700/64)
    assert st["numPieces"] == 10 * 3 + 11
    assert rel(plan_emulator.run_plan(op, x), vals[a] @ (vals[b] @ x)) < 1e-13


@pytest.mark.parametrize("dtype", [0, 1])
def test_long_contractions_are_cut_into_several_groups(dtype):
    """A row group whose contraction is long -- a very wide leaf, a tall leaf under the transposed plan, a
    block column of many leaves (fac_streamer's W factors) -- is cut into several groups over the same
    output rows: private slots + one deterministic reduce, forward and transposed."""
    rng = np.random.default_rng(17 + dtype)
    cplx = dtype == 0
    d, vals, apply_t, val = randgraph.long_contraction_operand(rng, dtype)
    m, n = d.rows[d.root], d.cols[d.root]
    x = val(n, 1)[:, 0]
    v = val(m, 1)[:, 0]
    A = bfref.from_desc(d, vals)
    y_ref = bfref.mat_mul(A, x) if cplx else bfref.mat_mul_vec(A, x)
    op = HipOperator.from_desc(d, vals, flags=_capi.FLAG_PLAN_ONLY | _capi.FLAG_ADJOINT)
    lib = _capi.load()
    sv = _capi.BfhipStageView(); sv.structSize = C.sizeof(sv)
    _capi.check(lib.bfhipPlanGetStage(op.handle, 0, C.byref(sv)))
    assert sv.numReduce == 1                                  # the wide leaf's 40 rows are summed from several slots
    _capi.check(lib.bfhipPlanGetStage(op.handle, 1, C.byref(sv)))
    assert sv.numReduce == 1                                  # transposed: the tall leaf's and the block column's columns
    assert rel(plan_emulator.run_plan(op, x), y_ref) < 1e-12
    dense_t = apply_t(v)
    assert rel(plan_emulator.run_plan(op, v, transpose=True), dense_t) < 1e-12
    op.close()


@pytest.mark.parametrize("demote", [False, True])
def test_few_row_leaves_are_stored_row_major(demote):
    """Row groups of <= 2 lane granules of rows and >= 128 columns (what a streamed butterfly's pass-through W blocks look
    like: ~5 rows x thousands of columns) are packed row-major with zero-padded row ends; forward and transposed plans
    read them as such, next to ordinary column-major leaves in the same block columns."""
    rng = np.random.default_rng(31)
    d, vals, dense = randgraph.few_row_operand(rng)
    op = HipOperator.from_desc(d, vals, flags=_capi.FLAG_PLAN_ONLY | _capi.FLAG_ADJOINT, demote_to_f32=demote)
    lib = _capi.load()
    sv = _capi.BfhipStageView(); sv.structSize = C.sizeof(sv)
    _capi.check(lib.bfhipPlanGetStage(op.handle, 0, C.byref(sv)))
    items = np.frombuffer((C.c_char * (int(sv.numItems) * 16)).from_address(sv.items), dtype=_capi.ITEM_DTYPE)
    rm = (items["mrFlags"] & plan_emulator.BF_ITEM_ROWMAJOR) != 0
    rows = items["mrFlags"] & 0xFFFF
    epl = 4 if demote else 2
    assert rm.any() and (rows[rm] <= 2 * epl).all() and (rows[~rm & (items["numPieces"] > 0)] > 2 * epl).any()
    x, v = rng.standard_normal(dense.shape[1]), rng.standard_normal(dense.shape[0])
    tol = 2e-6 if demote else 1e-13
    assert rel(plan_emulator.run_plan(op, x), dense @ x) < tol
    assert rel(plan_emulator.run_plan(op, v, transpose=True), dense.T @ v) < tol
    # no padded rows are stored for the few-row leaves: the arena holds (almost) exactly the leaf elements
    st = op.stats()
    assert st["arenaBytes"] <= 1.02 * st["leafBytes"]
    op.close()


@pytest.mark.parametrize("demote", [False, True])
def test_narrow_items_are_merged_and_small_ones_close_the_list(demote):
    """What the inner factors of a streamed butterfly are made of: row nodes of 1 - 8 rows whose terms are an Identity
    and a leaf of a few dozen columns, next to ordinary leaves.  Items whose dense pieces span <= 256 columns are flagged
    MERGED (one contiguous block), those of <= 2 lane granules of rows and < 384 columns SMALL -- and the small ones are
    the END of the stage's item list, behind the zero fills (they get their own launch, four to a wavefront)."""
    rng = np.random.default_rng(5)
    d, vals, dense = randgraph.narrow_items_operand(rng)
    op = HipOperator.from_desc(d, vals, flags=_capi.FLAG_PLAN_ONLY | _capi.FLAG_ADJOINT, demote_to_f32=demote)
    lib = _capi.load()
    sv = _capi.BfhipStageView(); sv.structSize = C.sizeof(sv)
    _capi.check(lib.bfhipPlanGetStage(op.handle, 0, C.byref(sv)))
    items = np.frombuffer((C.c_char * (int(sv.numItems) * 16)).from_address(sv.items), dtype=_capi.ITEM_DTYPE)
    small = (items["mrFlags"] & plan_emulator.BF_ITEM_SMALL) != 0
    merged = (items["mrFlags"] & plan_emulator.BF_ITEM_MERGED) != 0
    epl = 4 if demote else 2
    first = int(np.argmax(small))
    assert small.sum() >= 10 and small[first:].all() and not small[:first].any()
    assert ((items["mrFlags"][small] & 0xFFFF) <= 2 * epl).all()
    assert merged[~small & (items["numPieces"] > 0)].all()          # every ordinary item here is < 256 columns wide
    assert (items["numPieces"][:first] == 0).any()                  # the zero fill sits before the small items
    x, v = rng.standard_normal(300), rng.standard_normal(dense.shape[0])
    tol = 2e-6 if demote else 1e-13
    assert rel(plan_emulator.run_plan(op, x), dense @ x) < tol
    assert rel(plan_emulator.run_plan(op, v, transpose=True), dense.T @ v) < tol
    op.close()


@pytest.mark.parametrize("demote", [False, True])
def test_transposed_plan_of_a_block_column_of_few_row_leaves(demote):
    """A block column of a streamed butterfly's W factor under the transposed plan: chains of dozens of few-row pieces
    per item (row-major and column-major ones over the same outputs), leaves of 2100 columns whose last forward task
    (52 columns) is too narrow for the row-major layout."""
    rng = np.random.default_rng(98)
    for leaves, width in ((70, 900), (40, 2100)):
        d, vals, dense = randgraph.few_row_column_operand(rng, leaves, width)
        op = HipOperator.from_desc(d, vals, flags=_capi.FLAG_PLAN_ONLY | _capi.FLAG_ADJOINT, demote_to_f32=demote)
        x, v = rng.standard_normal(dense.shape[1]), rng.standard_normal(dense.shape[0])
        tol = 2e-6 if demote else 1e-13
        assert rel(plan_emulator.run_plan(op, x), dense @ x) < tol
        assert rel(plan_emulator.run_plan(op, v, transpose=True), dense.T @ v) < tol
        # the two tall leaves of the column get 16-column items of their own, at the head of the transposed stage's list
        lib = _capi.load()
        info = _capi.BfhipPlanInfo(); info.structSize = C.sizeof(info)
        _capi.check(lib.bfhipPlanGetInfo(op.handle, C.byref(info)))
        sv = _capi.BfhipStageView(); sv.structSize = C.sizeof(sv)
        _capi.check(lib.bfhipPlanGetStage(op.handle, int(info.numStages), C.byref(sv)))
        items = np.frombuffer((C.c_char * (int(sv.numItems) * 16)).from_address(sv.items), dtype=_capi.ITEM_DTYPE)
        narrow = (items["mrFlags"] & plan_emulator.BF_ITEM_TNARROW) != 0
        k = int(narrow.sum())
        assert 0 < k < len(items) and narrow[:k].all() and ((items["mrFlags"][:k] & 0xFFFF) <= 16).all()
        assert ((items["mrFlags"][k:] & 0xFFFF) > 16).any() and int(sv.numReduce) >= 1
        op.close()


def test_row_sharding_union_equals_full(helm2_cases):
    n, k = 2048, 128
    desc, tp, vals = helm2_cases(n, k)
    x = hb.complex_randn(n, 0)
    y_ref = bfref.mat_mul(bfref.from_desc(desc, vals), x)
    nrb = len(desc.meta["top_rows"])
    offs = np.concatenate([[0], np.cumsum(desc.meta["top_rows"])])
    # C-ABI sharding option: contiguous block-row ranges
    parts = []
    for b, e in ((0, 5), (5, nrb)):
        op = HipOperator.from_desc(desc, vals, row_blocks=(b, e), **PLAN)
        assert op.shape == (int(offs[e] - offs[b]), n)
        parts.append(plan_emulator.run_plan(op, x))
    assert rel(np.concatenate(parts), y_ref) < 1e-13
    # descriptor-level sharding: arbitrary subsets
    mine = [0, 3, 7, 11]
    root, m = hs.shard_desc(desc, mine)
    op = HipOperator.from_desc(desc, vals, root=root, **PLAN)
    want = np.concatenate([y_ref[offs[rb]:offs[rb + 1]] for rb in mine])
    assert rel(plan_emulator.run_plan(op, x), want) < 1e-13


def test_rhs_block_operators_keep_items_with_equal_inputs_together(helm2_cases):
    """Operators compiled for blocks of right-hand sides (complex128): the row chunks of a group and the sibling groups of a radix-4
    stage (reference src/fac_helm2.c:277-318) read the same input rows; the planner keeps them together in the item list (items of
    <= 32 rows), and the stage view's bundle table partitions the list into workgroups of <= 4 neighbours -- shared ones (bit 31
    clear): exactly four items with the same pieces' input rows, <= 32 rows and the same number of 16-row slabs; mixed ones: the
    rest.  The plan still computes A x (emulator), and most of a butterfly stage's work sits in runs of >= 2 equal neighbours."""
    n, k = 4096, 256
    desc, tp, vals = helm2_cases(n, k)
    x = hb.complex_randn(n, 3)
    op = HipOperator.from_desc(desc, vals, flags=_capi.FLAG_PLAN_ONLY, max_rhs=64)
    assert rel(plan_emulator.run_plan(op, x), bfref.mat_mul(bfref.from_desc(desc, vals), x)) < 1e-13
    lib = _capi.load()
    info = _capi.BfhipPlanInfo(); info.structSize = C.sizeof(info)
    _capi.check(lib.bfhipPlanGetInfo(op.handle, C.byref(info)))
    shared_items = total_items = 0
    for s in range(int(info.numStages)):
        sv = _capi.BfhipStageView(); sv.structSize = C.sizeof(sv)
        _capi.check(lib.bfhipPlanGetStage(op.handle, s, C.byref(sv)))
        items = np.frombuffer((C.c_char * (int(sv.numItems) * 16)).from_address(sv.items), dtype=_capi.ITEM_DTYPE)
        pieces = np.frombuffer((C.c_char * (int(sv.numPieces) * 24)).from_address(sv.pieces), dtype=_capi.PIECE_DTYPE)
        bb = np.frombuffer((C.c_char * ((int(sv.numBundles) + 1) * 4)).from_address(sv.bundleBegin), dtype=np.uint32)
        first = (bb & 0x7FFFFFFF).astype(np.int64)
        assert first[0] == 0 and first[-1] == len(items) and (np.diff(first) >= 1).all() and (np.diff(first) <= 4).all()
        mr = items["mrFlags"] & 0xFFFF
        assert (mr[items["numPieces"] > 0] <= 32).all()

        def key(i):
            p = pieces[items["pieceBegin"][i]:items["pieceBegin"][i] + items["numPieces"][i]]
            return (p["inOff"].tobytes(), p["ncols"].tobytes(), (p["flags"] & 3).tobytes())

        for b in range(len(bb) - 1):
            mem = range(first[b], first[b + 1])
            if bb[b] >> 31:
                continue
            assert len(mem) == 4 and len({key(i) for i in mem}) == 1 and len({bool(mr[i] > 16) for i in mem}) == 1 and all(0 < mr[i] <= 32 for i in mem)
            shared_items += 4
        total_items += int((items["numPieces"] > 0).sum())
        # equal inputs are neighbours: every key occupies ONE contiguous run of the list per cost bucket at most a few times
        keys = [key(i) for i in range(len(items)) if items["numPieces"][i]]
        runs = 1 + sum(keys[i] != keys[i - 1] for i in range(1, len(keys)))
        assert runs <= 3 * len(set(keys)), (s, runs, len(set(keys)))
    assert shared_items >= 0.1 * total_items, (shared_items, total_items)
    op.close()


# ---- error behaviour (mirrors the reference's BfError codes) ---------------
@pytest.mark.parametrize("seed", range(8))
def test_row_ranges_of_random_nested_graphs(seed):
    """BfhipOptions.rowBegin/rowEnd on arbitrary graphs (products nested in all three block types, Identity leaves,
    ragged sizes): the plan of rows [a, b) keeps what those rows depend on -- backward liveness over the task list --
    and produces exactly rows [a, b) of the whole operator.  Arbitrary ends may trim a leaf (equal to rounding); the ends
    bfhipRowPartition picks never do, and then the rows are the whole plan's bit for bit.  Real (f64, f32) and complex."""
    rng = np.random.default_rng(4200 + seed)
    cplx = seed % 2 == 1
    desc, vals = randgraph.random_operand(rng, depth=int(rng.integers(2, 5)), size_hint=int(rng.integers(60, 220)), cplx=cplx)
    m, n = desc.rows[desc.root], desc.cols[desc.root]
    x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
    want = randgraph.densify(desc, vals, desc.root) @ x
    for demote in ((False,) if cplx else (False, True)):
        full = HipOperator.from_desc(desc, vals, demote_to_f32=demote, **PLAN)
        y = plan_emulator.run_plan(full, x)
        tol = 2e-5 if demote else 1e-12
        assert rel(y + 1, want + 1) < tol
        kept = 0
        for _ in range(4):
            a = int(rng.integers(0, m))
            b = int(rng.integers(a + 1, m + 1))
            op = HipOperator.from_desc(desc, vals, demote_to_f32=demote, row_range=(a, b), **PLAN)
            st = op.stats()
            assert st["numRows"] == b - a and st["leafElems"] <= full.stats()["leafElems"]
            got = plan_emulator.run_plan(op, x)
            assert not np.isnan(got).any()              # every row has an owner or a zero fill
            assert rel(got + 1, y[a:b] + 1) < tol
            kept += st["leafElems"]
        # the cuts the library proposes: clean, covering, bit-identical
        for world in (2, 3):
            cuts = np.zeros(world + 1, dtype=np.uint64)
            loads = np.zeros(world, dtype=np.uint64)
            da = _capi.DescArrays(desc)
            rc = _capi.load().bfhipRowPartition(da.byref(), world, cuts.ctypes.data, loads.ctypes.data)
            if rc:                                      # an operand written by one tall leaf offers no place to cut
                assert rc == 1
                continue
            assert cuts[0] == 0 and cuts[-1] == m and (np.diff(cuts.astype(np.int64)) > 0).all()
            for r in range(world):
                op = HipOperator.from_desc(desc, vals, demote_to_f32=demote, row_range=(int(cuts[r]), int(cuts[r + 1])), **PLAN)
                assert op.stats()["leafElems"] == loads[r] or not cplx       # hulls are exact on these graphs unless a range skips a middle block
                assert op.stats()["leafElems"] <= loads[r]
                assert np.array_equal(plan_emulator.run_plan(op, x), y[int(cuts[r]):int(cuts[r + 1])])


@pytest.mark.parametrize("seed", range(8))
def test_transposed_plans_of_row_range_shards_add_up(seed):
    """The adjoint of a row-range shard (BFHIP_FLAG_ADJOINT + rowBegin / rowEnd): rank r holds A_r = rows [a_r, b_r) of A and
    applies A_r^T to ITS entries of v; the ranks' full-length results add up to A^T v (what the closing all-reduce of
    bfhipShardedApplyTransposeDevice computes).  Arbitrary nested graphs, real (f64, f32) and complex; arbitrary range ends
    (leaves that straddle them are entered part-way) and the clean cuts of bfhipRowPartition; BFHIP_FLAG_ADJOINT_PACKED on a
    shard falls back to the shared-leaf plan."""
    rng = np.random.default_rng(9100 + seed)
    cplx = seed % 2 == 1
    desc, vals = randgraph.random_operand(rng, depth=int(rng.integers(2, 5)), size_hint=int(rng.integers(60, 220)), cplx=cplx)
    m, n = desc.rows[desc.root], desc.cols[desc.root]
    v = rng.standard_normal(m) + (1j * rng.standard_normal(m) if cplx else 0)
    want = randgraph.densify(desc, vals, desc.root).T @ v
    for demote in ((False,) if cplx else (False, True)):
        tol = 2e-5 if demote else 1e-12
        for packed in (False, True):
            flags = _capi.FLAG_PLAN_ONLY | (_capi.FLAG_ADJOINT_PACKED if packed else _capi.FLAG_ADJOINT)
            # arbitrary ends
            ends = sorted({0, m} | {int(e) for e in rng.integers(1, m, size=3)} if m > 1 else {0, m})
            z = np.zeros(n, dtype=want.dtype)
            for a, b in zip(ends[:-1], ends[1:]):
                op = HipOperator.from_desc(desc, vals, demote_to_f32=demote, row_range=(a, b), flags=flags)
                info = _capi.BfhipPlanInfo()
                info.structSize = C.sizeof(info)
                _capi.check(_capi.load().bfhipPlanGetInfo(op.handle, C.byref(info)))
                assert int(info.reserved) == 0 and int(info.numStagesT) > 0          # the shared-leaf plan, packed flag or not
                zr = plan_emulator.run_plan(op, v[a:b], transpose=True)
                assert zr.shape == (n,) and not np.isnan(zr).any()
                # the forward plan of the same operator still yields rows [a, b)
                z += zr
            assert rel(z + 1, want + 1) < tol
        # the library's own cuts
        for world in (2, 3):
            cuts = np.zeros(world + 1, dtype=np.uint64)
            da = _capi.DescArrays(desc)
            if _capi.load().bfhipRowPartition(da.byref(), world, cuts.ctypes.data, None):
                continue
            z = np.zeros(n, dtype=want.dtype)
            for r in range(world):
                a, b = int(cuts[r]), int(cuts[r + 1])
                op = HipOperator.from_desc(desc, vals, demote_to_f32=demote, row_range=(a, b), flags=_capi.FLAG_PLAN_ONLY | _capi.FLAG_ADJOINT)
                z += plan_emulator.run_plan(op, v[a:b], transpose=True)
            assert rel(z + 1, want + 1) < tol


def test_row_range_arguments_are_checked(helm2_cases):
    desc, tp, vals = helm2_cases(1024, 100)
    for bad in ((5, 5), (9, 3), (0, 1025)):
        with pytest.raises(_capi.BfhipError) as e:
            HipOperator.from_desc(desc, vals, row_range=bad, **PLAN)
        assert e.value.code == 1
    # (round 5: the transposed plan of a shard exists -- its input is the shard's rows of v: test_transposed_plans_of_row_range_shards_add_up)
    HipOperator.from_desc(desc, vals, row_range=(0, 512), flags=_capi.FLAG_PLAN_ONLY | _capi.FLAG_ADJOINT).close()
    with pytest.raises(_capi.BfhipError) as e:
        HipOperator.from_desc(desc, vals, row_range=(0, 512), row_blocks=(0, 1), **PLAN)
    assert e.value.code == 1
    # an options struct of the round-2 size (no rowBegin / rowEnd) still compiles: all rows
    o = _capi.BfhipOptions()
    o.structSize, o.device, o.flags, o.rowEnd = 48, -1, _capi.FLAG_PLAN_ONLY, 77
    h = C.c_void_p()
    da = _capi.DescArrays(desc, leaf_values=vals)
    _capi.check(_capi.load().bfhipCompileDesc(da.byref(), C.byref(o), C.byref(h)))
    assert _capi.load().bfhipGetNumRows(h) == 1024
    _capi.load().bfhipFree(C.byref(h))


def _compile_bfmat(ptr):
    lib = _capi.load()
    h = C.c_void_p()
    o = _capi.BfhipOptions()
    o.structSize = C.sizeof(o)
    o.device = -1
    o.flags = _capi.FLAG_PLAN_ONLY
    rc = lib.bfhipCompile(C.c_void_p(ptr), C.byref(o), C.byref(h))
    if h:
        lib.bfhipFree(C.byref(h))
    return rc, lib.bfhipLastErrorMessage().decode()


def test_flagged_leaves_follow_the_reference_and_conj_alone_is_refused():
    """props of a dense leaf (the int at offset 8 of BfMat): TRANS (with or without CONJ) makes a complex leaf multiply as
    its conjugate transpose -- getCblasTranspose maps TRANS *or* CONJ to CblasConjTrans (src/mat_dense_complex.c:27-35), the
    extents follow TRANS (:503-511) -- and a real leaf as its transpose (src/mat_dense_real.c:20-28, 291-298); CONJ without
    TRANS would be ConjTrans with untransposed extents, which nothing in the reference produces: refused."""
    from butterfly_amd._capi import ERROR_NAMES
    rng = np.random.default_rng(77)
    TRANS, CONJ = 2, 4
    v = rng.standard_normal((5, 9)) + 1j * rng.standard_normal((5, 9))
    a = bfref.dense_complex(v)
    x = rng.standard_normal(5) + 1j * rng.standard_normal(5)
    for flags in (TRANS, TRANS | CONJ):
        C.c_int.from_address(a.ptr.value + 8).value |= flags
        op = HipOperator.from_bfmat(a.ptr.value, flags=_capi.FLAG_PLAN_ONLY)
        assert op.shape == (9, 5)
        assert rel(plan_emulator.run_plan(op, x), v.conj().T @ x) < 1e-14
        assert rel(bfref.mat_mul(a, x), v.conj().T @ x) < 1e-14
        C.c_int.from_address(a.ptr.value + 8).value &= ~(TRANS | CONJ)
    C.c_int.from_address(a.ptr.value + 8).value |= CONJ
    rc, msg = _compile_bfmat(a.ptr.value)
    assert ERROR_NAMES[rc] == "BF_ERROR_NOT_IMPLEMENTED" and "CONJ without TRANS" in msg
    C.c_int.from_address(a.ptr.value + 8).value &= ~CONJ
    w = rng.standard_normal((4, 7))
    r = bfref.dense_real(w)
    C.c_int.from_address(r.ptr.value + 8).value |= TRANS
    op = HipOperator.from_bfmat(r.ptr.value, flags=_capi.FLAG_PLAN_ONLY)
    assert op.shape == (7, 4)
    xr = rng.standard_normal(4)
    assert rel(plan_emulator.run_plan(op, xr), w.T @ xr) < 1e-14
    C.c_int.from_address(r.ptr.value + 8).value &= ~TRANS


def test_mixed_real_and_complex_is_a_type_error():
    from butterfly_amd._capi import ERROR_NAMES
    g = bfref.block_diag([bfref.dense_complex(np.ones((2, 2), dtype=complex)), bfref.dense_real(np.ones((2, 2)))])
    rc, msg = _compile_bfmat(g.ptr.value)
    assert ERROR_NAMES[rc] == "BF_ERROR_TYPE_ERROR"


def test_non_chaining_product_is_refused():
    from butterfly_amd._capi import ERROR_NAMES
    p = bfref.product([bfref.dense_complex(np.ones((3, 4), dtype=complex)), bfref.dense_complex(np.ones((5, 2), dtype=complex))])
    rc, msg = _compile_bfmat(p.ptr.value)
    assert ERROR_NAMES[rc] == "BF_ERROR_INCOMPATIBLE_SHAPES"


def test_null_and_bad_arguments():
    lib = _capi.load()
    h = C.c_void_p()
    assert lib.bfhipCompile(None, None, C.byref(h)) == 1
    d = _capi.BfhipDesc()
    d.structSize = 4
    assert lib.bfhipCompileDesc(C.byref(d), None, C.byref(h)) == 1
    assert lib.bfhipGetNumRows(None) == 0
    lib.bfhipFree(C.byref(h))          # freeing NULL is a no-op


def test_plan_only_operator_refuses_apply():
    desc, vals, ex = load_fixture(os.path.join(GOLD, "helm2_one_block_n2048_k128.npz"))
    op = HipOperator.from_desc(desc, vals, **PLAN)
    with pytest.raises(_capi.BfhipError) as e:
        op.apply_host(ex["x"])
    assert e.value.code == 2


def test_unsupported_node_type_is_a_type_error():
    """A BfMat whose GetType answers something outside the factorization zoo."""
    from butterfly_amd._capi import ERROR_NAMES
    a = bfref.identity(4)
    # the shim object itself reports BF_TYPE_MAT_FUNC (6): nesting it is refused
    op = HipOperator.from_bfmat(a.ptr.value, **PLAN)
    shim = op.as_bfmat()
    rc, msg = _compile_bfmat(shim)
    assert ERROR_NAMES[rc] == "BF_ERROR_TYPE_ERROR"
    p = C.c_void_p(shim)
    bfref.load().bfMatDelete(C.byref(p))


# ---- adjoint plan (A^T x): RmulVec of the reference --------------------------
ADJ = dict(flags=_capi.FLAG_PLAN_ONLY | _capi.FLAG_ADJOINT)


@pytest.mark.parametrize("seed", range(10))
def test_transposed_plan_random_real_graphs(seed):
    rng = np.random.default_rng(3000 + seed)
    desc, vals = randgraph.random_real_operand(rng, depth=int(rng.integers(1, 5)), size_hint=int(rng.integers(8, 300)))
    m = desc.rows[desc.root]
    x = rng.standard_normal(m)
    A = bfref.from_desc(desc, vals)
    want = bfref.mat_rmul_vec(A, x)                       # oracle: bfMatRmulVec
    assert rel(want + 1, randgraph.densify(desc, vals, desc.root).T @ x + 1) < 1e-12
    for demote in (False, True):
        op = HipOperator.from_desc(desc, vals, demote_to_f32=demote, **ADJ)
        assert rel(plan_emulator.run_plan(op, x, transpose=True) + 1, want + 1) < (2e-5 if demote else 1e-12)
        # the forward plan is untouched by the adjoint flag
        xf = rng.standard_normal(desc.cols[desc.root])
        assert rel(plan_emulator.run_plan(op, xf) + 1, bfref.mat_mul_vec(A, xf) + 1) < (2e-5 if demote else 1e-12)


ADJP = dict(flags=_capi.FLAG_PLAN_ONLY | _capi.FLAG_ADJOINT_PACKED)


@pytest.mark.parametrize("seed", range(8))
def test_packed_adjoint_plan_random_graphs(seed):
    """BFHIP_FLAG_ADJOINT_PACKED: the adjoint plan is a forward plan of the transposed expression (blocks at (col0, row0),
    products reversed, leaves transposed: reference src/mat_product.c:409-420) over an arena of its own; run by the plan
    interpreter with forward semantics on that arena, it is the oracle's bfMatRmulVec -- host-valued and synthetic leaves,
    real (fp64 and demoted) and complex."""
    rng = np.random.default_rng(8100 + seed)
    cplx = bool(seed % 2)
    desc, vals = randgraph.random_operand(rng, depth=int(rng.integers(1, 5)), size_hint=int(rng.integers(8, 260)), cplx=cplx)
    if seed % 4 >= 2:
        vals = None                                           # synthetic: the transposed leaves hold the SAME value stream, transposed
    m, n = desc.rows[desc.root], desc.cols[desc.root]
    A = bfref.from_desc(desc, vals, seed=5)
    x = rng.standard_normal(m) + (1j * rng.standard_normal(m) if cplx else 0)
    if cplx:
        want = bfref.mat_mul(A, np.eye(n, dtype=np.complex128)).T @ x
    else:
        want = bfref.mat_rmul_vec(A, x)
    for demote in ((False,) if cplx else (False, True)):
        op = HipOperator.from_desc(desc, vals, seed=5, demote_to_f32=demote, **ADJP)
        assert rel(plan_emulator.run_plan(op, x, transpose=True) + 1, want + 1) < (2e-5 if demote else 1e-12)
        xf = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
        fwd = bfref.mat_mul(A, xf) if cplx else bfref.mat_mul_vec(A, xf)
        assert rel(plan_emulator.run_plan(op, xf) + 1, fwd + 1) < (2e-5 if demote else 1e-12)


@pytest.mark.parametrize("seed", range(4))
def test_walker_reads_flagged_dense_leaves_as_conjugate_transposes(seed):
    """A graph the oracle has transposed in place (bfMatTranspose: dense complex leaves flagged TRANS | CONJ, containers
    restructured) compiles to A^H: the walker takes a flagged leaf as the conjugate of its stored values with the strides
    swapped (reference src/mat_dense_complex.c:27-35, 503-511, 1475-1478).  Plan interpreter on the CPU."""
    rng = np.random.default_rng(9300 + seed)
    desc, vals = randgraph.random_operand(rng, depth=int(rng.integers(1, 4)), size_hint=50, cplx=True, coo=False)
    m, n = desc.rows[desc.root], desc.cols[desc.root]
    dense = randgraph.densify(desc, vals, desc.root)
    A = bfref.from_desc(desc, vals, typed=True)
    bfref.mat_transpose(A)
    op = HipOperator.from_bfmat(A.ptr.value, flags=_capi.FLAG_PLAN_ONLY | _capi.FLAG_ADJOINT)
    assert op.shape == (n, m)
    x = rng.standard_normal(m) + 1j * rng.standard_normal(m)
    xf = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    assert rel(plan_emulator.run_plan(op, x) + 1, dense.conj().T @ x + 1) < 1e-12
    assert rel(plan_emulator.run_plan(op, xf, transpose=True) + 1, dense.conj() @ xf + 1) < 1e-12
    assert rel(bfref.mat_mul(A, x) + 1, dense.conj().T @ x + 1) < 1e-12


def test_transposed_plan_helm2(helm2_cases):
    n, k = 2048, 128
    desc, tp, vals = helm2_cases(n, k)
    x = hb.complex_randn(n, 4)
    dense_t = hb.kernel_matrix(k, tp, tp).T @ x            # plain transpose, no conjugation
    op = HipOperator.from_desc(desc, vals, **ADJ)
    y = plan_emulator.run_plan(op, x, transpose=True)
    assert rel(y, dense_t) < 1e-9
    # single-layer kernel on a symmetric point set: S^T = S, so A^T x must agree with A x to truncation accuracy
    assert rel(y, plan_emulator.run_plan(op, x)) < 1e-9


def test_transposed_plans_of_helm2_row_shards_add_up(helm2_cases):
    """A fac_helm2 operator dealt to 2, 3 and 5 ranks by bfhipRowPartition: the shards' transposed plans applied to their
    own rows of v add up to the one-operator A^T v (to rounding: the order of additions differs)."""
    from butterfly_amd.dist import row_partition
    desc, tp, vals = helm2_cases(2048, 128)
    v = hb.complex_randn(2048, 5)
    full = HipOperator.from_desc(desc, vals, **ADJ)
    want = plan_emulator.run_plan(full, v, transpose=True)
    for world in (2, 3, 5):
        cuts, loads = row_partition(desc, world)
        z = np.zeros_like(want)
        for r in range(world):
            op = HipOperator.from_desc(desc, vals, row_range=(cuts[r], cuts[r + 1]), **ADJ)
            assert op.stats()["leafElems"] == loads[r]
            z += plan_emulator.run_plan(op, v[cuts[r]:cuts[r + 1]], transpose=True)
        assert rel(z, want) < 1e-13


def test_transpose_needs_the_adjoint_flag():
    desc, vals, ex = load_fixture(os.path.join(GOLD, "helm2_one_block_n2048_k128.npz"))
    op = HipOperator.from_desc(desc, vals, **PLAN)
    with pytest.raises(_capi.BfhipError) as e:
        op.apply_transpose_host(np.zeros(op.shape[0], dtype=complex))
    assert e.value.code == 1
