"""Structure of a `fac_helm2` multilevel butterfly factorization, without values.

The engine's benchmark operands at N >= 262144 are *structure-exact,
value-random* (SURVEY.md section 8(d)): the reference's CPU builder would need
~20 min and ~60 GB of host RAM at N = 262144, and the cost of an apply does
not depend on the values.  This module re-derives, from the point set and the
wavenumber alone, every block shape the reference's builder would produce:

  quadtree (leaf size 1, square root bbox)   reference src/quadtree_node.c:17,123-282,292-294
  bounding circles / separation test          src/quadtree_node.c:321-330,393-401
  rank rule p = ceil(k r1 r2 / d + 15)        src/helm2.c:14-36 (C = 1, eps = 1e-15)
  level selection (bfFacHelm2Prepare)         src/fac_helm2.c:551-651
  first / inner / last factor block layout    src/fac_helm2.c:42-160, 222-401, 403-509
  dense threshold, recursion (HODBF)          src/fac_helm2.c:20, 860-941, 943-1002

and emits the flat expression descriptor `include/bfhip.h: BfhipDesc` consumes.
Optionally each dense leaf carries a *recipe* (which point sets its kernel /
re-expansion matrix is built from) so that `oracle/helm2_build.py` can fill in
real values for parity tests.

Nothing here is on the timed path; it is host-side operand preparation.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

NODE_DENSE, NODE_IDENTITY, NODE_BLOCK, NODE_PRODUCT = 0, 1, 2, 3
BF_TYPE_BLOCK_COO, BF_TYPE_BLOCK_DENSE, BF_TYPE_BLOCK_DIAG = 15, 16, 17

MAX_DENSE_MATRIX_SIZE = 128 * 128  # src/fac_helm2.c:20
LEAF_SIZE_THRESHOLD = 1            # src/quadtree_node.c:17
EPS_MACH = 2.220446049250313e-16   # include/bf/def.h:27


def circle_points(n: int) -> np.ndarray:
    """N equispaced points on the unit circle, file order = index
    (examples/simple/make_circle_test_data.py:12-14)."""
    theta = 2 * np.pi * np.arange(n) / n
    return np.stack([np.cos(theta), np.sin(theta)], axis=1)


# --------------------------------------------------------------------------
# quadtree
# --------------------------------------------------------------------------
class QNode:
    __slots__ = ("xmin", "ymin", "xmax", "ymax", "i0", "i1", "children", "depth",
                 "cx", "cy", "r", "_levels", "tree")

    def __init__(self, xmin, ymin, xmax, ymax, i0, i1, depth, tree=0):
        self.xmin, self.ymin, self.xmax, self.ymax = xmin, ymin, xmax, ymax
        self.i0, self.i1, self.depth = i0, i1, depth
        self.tree = tree      # 0: the source tree (or the only one), 1: a separate target tree
        self.children = []  # non-empty children in quadrant order 0..3
        # bounding circle: src/quadtree_node.c:321-330
        self.r = float(np.hypot(xmax - xmin, ymax - ymin)) / 2
        self.cx = (xmin + xmax) / 2
        self.cy = (ymin + ymax) / 2
        self._levels = None

    @property
    def npts(self):
        return self.i1 - self.i0

    @property
    def is_leaf(self):
        return not self.children


def _sift_exact(px, py, perm, i0, i1, sx, sy):
    """The reference's in-place sifting (src/quadtree_node.c:143-185), which is
    stable for members of each quadrant but not for the rest; reproduced
    literally so that `perm` equals the reference's permutation."""
    off = [i0, 0, 0, 0, i1]

    def member(q, p):
        if q == 0:
            return px[p] <= sx and py[p] <= sy
        if q == 1:
            return px[p] <= sx and py[p] > sy
        return px[p] > sx and py[p] <= sy

    for q in range(3):
        i = off[q]
        while i < i1 and member(q, perm[i]):
            i += 1
        j = i if i == i1 else i + 1
        while j < i1:
            if member(q, perm[j]) and not member(q, perm[i]):
                perm[i], perm[j] = perm[j], perm[i]
                i += 1
            j += 1
        off[q + 1] = i
    return off


def build_quadtree(points: np.ndarray, exact_sift: bool = False, tree: int = 0):
    """Returns (root, perm): perm[i] = index into `points` of the i-th point in
    tree order (BfTree.perm, include/bf/tree.h:30-38)."""
    n = len(points)
    px = np.ascontiguousarray(points[:, 0])
    py = np.ascontiguousarray(points[:, 1])
    xmin, xmax, ymin, ymax = px.min(), px.max(), py.min(), py.max()
    # bfBbox2RescaleToSquare, src/bbox.c:18-30
    w, h = xmax - xmin, ymax - ymin
    if w > h:
        c = (ymin + ymax) / 2
        ymin, ymax = w * (ymin - c) / h + c, w * (ymax - c) / h + c
    else:
        c = (xmin + xmax) / 2
        xmin, xmax = h * (xmin - c) / w + c, h * (xmax - c) / w + c
    root = QNode(float(xmin), float(ymin), float(xmax), float(ymax), 0, n, 0, tree)
    if exact_sift:
        perm = list(range(n))
        pxl, pyl = px.tolist(), py.tolist()
    else:
        perm = np.arange(n)
    stack = [root]
    while stack:
        nd = stack.pop()
        sx, sy = nd.cx, nd.cy  # split = bbox centre (src/quadtree_node.c:237, bbox.c:44-47)
        if exact_sift:
            off = _sift_exact(pxl, pyl, perm, nd.i0, nd.i1, sx, sy)
        else:
            idx = perm[nd.i0:nd.i1]
            x, y = px[idx], py[idx]
            left, low = x <= sx, y <= sy
            quad = np.where(left, np.where(low, 0, 1), np.where(low, 2, 3))
            order = np.argsort(quad, kind="stable")
            perm[nd.i0:nd.i1] = idx[order]
            cnt = np.bincount(quad, minlength=4)
            off = [nd.i0, 0, 0, 0, nd.i1]
            off[1] = off[0] + int(cnt[0])
            off[2] = off[1] + int(cnt[1])
            off[3] = off[2] + int(cnt[2])
        boxes = ((nd.xmin, nd.ymin, sx, sy), (nd.xmin, sy, sx, nd.ymax),
                 (sx, nd.ymin, nd.xmax, sy), (sx, sy, nd.xmax, nd.ymax))
        for q in range(4):
            if off[q + 1] == off[q]:
                continue
            b = boxes[q]
            ch = QNode(b[0], b[1], b[2], b[3], off[q], off[q + 1], nd.depth + 1, tree)
            nd.children.append(ch)
            if ch.npts > LEAF_SIZE_THRESHOLD:
                stack.append(ch)
    return root, np.asarray(perm, dtype=np.int64)


class Level:
    """One BFS level below a node (src/tree_level_iter.c:10-44) with the
    per-node arrays the rank rule needs."""
    __slots__ = ("nodes", "cx", "cy", "r", "npts", "total_pts", "internal")

    def __init__(self, nodes):
        self.nodes = nodes
        self.cx = np.array([n.cx for n in nodes])
        self.cy = np.array([n.cy for n in nodes])
        self.r = np.array([n.r for n in nodes])
        self.npts = np.array([n.npts for n in nodes], dtype=np.int64)
        self.total_pts = int(self.npts.sum())
        self.internal = all(not n.is_leaf for n in nodes)


def levels_below(node: QNode):
    """levels[0] = [node], levels[d] = nodes d levels down in LR order."""
    if node._levels is None:
        levels = [Level([node])]
        while True:
            nxt = [c for n in levels[-1].nodes for c in n.children]
            if not nxt:
                break
            levels.append(Level(nxt))
        node._levels = levels
    return node._levels


def rank_est(k, cx1, cy1, r1, cx2, cy2, r2):
    """bfHelm2RankEstForTwoCircles (src/helm2.c:14-36) with C = 1, eps = 1e-15;
    numpy-broadcasting."""
    R = np.hypot(cx2 - cx1, cy2 - cy1)
    d = R - r1 - r2
    p = k * r1 * r2 / d - 1 * math.log10(1e-15)
    return np.ceil(p).astype(np.int64)


def separated(a: QNode, b: QNode) -> bool:
    """bfQuadtreeNodesAreSeparated, src/quadtree_node.c:393-401."""
    R = float(np.hypot(a.cx - b.cx, a.cy - b.cy))
    return R > a.r + b.r + 1e1 * EPS_MACH


def _ranks_ok(k, tgt: QNode, lvl: Level) -> bool:
    """allRankEstimatesAreOK, src/fac_helm2.c:510-530."""
    rk = rank_est(k, tgt.cx, tgt.cy, tgt.r, lvl.cx, lvl.cy, lvl.r)
    return bool(np.all(rk <= lvl.npts))


def prepare(k, src: QNode, tgt: QNode):
    """bfFacHelm2Prepare (src/fac_helm2.c:551-651).  Returns (numFactors, L)
    where L is the number of levels below `src` at which the source traversal
    starts (src level = src.depth + L)."""
    tl = levels_below(tgt)
    sl = levels_below(src)
    # deepest complete internal level of the target subtree (:583-591)
    max_depth_below = 0
    assert tl[0].internal
    d = 1
    while d < len(tl) and tl[d].internal:
        max_depth_below += 1
        d += 1
    cur = len(sl) - 1  # reverse level order starts at the deepest level
    # depths are relative: src.depth == tgt.depth for every pair the
    # multilevel recursion makes (:829-846)
    assert src.depth == tgt.depth
    while cur > max_depth_below:            # :612-615
        cur -= 1
    while sl[cur].total_pts != src.npts:    # :618-621
        cur -= 1
    while not sl[cur].internal:             # :625-628
        cur -= 1
    while cur > 0 and not _ranks_ok(k, tgt, sl[cur]):   # :634-638
        cur -= 1
    nf = cur + 2 if _ranks_ok(k, tgt, sl[cur]) else 0   # :642-643
    return nf, cur


# --------------------------------------------------------------------------
# descriptor assembly
# --------------------------------------------------------------------------
@dataclass
class Desc:
    """Flat expression (mirrors include/bfhip.h: BfhipDesc)."""
    dtype: int = 0
    kind: list = field(default_factory=list)
    rows: list = field(default_factory=list)
    cols: list = field(default_factory=list)
    children: list = field(default_factory=list)   # per node: list of (child, r0, c0)
    block_kind: list = field(default_factory=list)
    recipe: dict = field(default_factory=dict)     # leaf node -> recipe tuple
    root: int = -1
    top_row_block: list | None = None
    meta: dict = field(default_factory=dict)

    def add(self, kind, rows, cols, children=(), block_kind=0):
        self.kind.append(kind)
        self.rows.append(int(rows))
        self.cols.append(int(cols))
        self.children.append(list(children))
        self.block_kind.append(block_kind)
        return len(self.kind) - 1

    @property
    def num_nodes(self):
        return len(self.kind)

    def arrays(self):
        """numpy arrays in BfhipDesc layout."""
        n = self.num_nodes
        counts = np.fromiter((len(c) for c in self.children), dtype=np.uint64, count=n)
        begin = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(counts, out=begin[1:])
        tot = int(begin[-1])
        flat = np.fromiter((v for ch in self.children for t in ch for v in t), dtype=np.uint64, count=3 * tot).reshape(tot, 3)
        cn, r0, c0 = (np.ascontiguousarray(flat[:, k]) for k in range(3))
        return dict(kind=np.asarray(self.kind, dtype=np.uint8),
                    rows=np.asarray(self.rows, dtype=np.uint64),
                    cols=np.asarray(self.cols, dtype=np.uint64),
                    childBegin=begin, childNode=cn, childRow0=r0, childCol0=c0,
                    blockKind=np.asarray(self.block_kind, dtype=np.uint8))

    # ---- statistics (SURVEY.md section 8(d) "algorithmic bytes / flops") ----
    def leaf_elems(self):
        k = np.asarray(self.kind)
        m = np.asarray(self.rows, dtype=np.int64)
        n = np.asarray(self.cols, dtype=np.int64)
        return int((m * n)[k == NODE_DENSE].sum())


def _node_pts(nd: QNode):
    # "node": a range of the (source) tree's points; "tnode": of a separate target tree's points
    return ("node" if nd.tree == 0 else "tnode", nd.i0, nd.i1)


def _circ_pts(nd: QNode, count):
    return ("circle", nd.cx, nd.cy, nd.r, int(count))


def _make_product(desc: Desc, k, src: QNode, tgt: QNode, nf, L, recipes):
    """bfFacHelm2Make (src/fac_helm2.c:653-704): factors are built first-applied
    first and stored reversed (:692-695) so product order is evaluation-first."""
    sl = levels_below(src)
    tl = levels_below(tgt)
    factors = []  # build order: factor[0] applied first

    # --- makeFirstFactor (:42-160): one block per source node at level L ----
    lvl = sl[L]
    p = rank_est(k, lvl.cx, lvl.cy, lvl.r, tgt.cx, tgt.cy, tgt.r)
    ch, r0, c0 = [], 0, 0
    for i, s in enumerate(lvl.nodes):
        leaf = desc.add(NODE_DENSE, p[i], s.npts)
        if recipes:
            desc.recipe[leaf] = ("reexp", _node_pts(s), _circ_pts(s, p[i]), _circ_pts(tgt, p[i]))
        ch.append((leaf, r0, c0))
        r0 += int(p[i])
        c0 += s.npts
    factors.append(desc.add(NODE_BLOCK, r0, c0, ch, BF_TYPE_BLOCK_DIAG))
    prev_heights = p.astype(np.int64)  # row height of each block row of the previous factor

    # --- makeFactor (:222-401), i = 1 .. nf-2 -------------------------------
    for i in range(1, nf - 1):
        sp = sl[L - i]          # source parents (their children are at level L-i+1)
        tp = tl[i - 1]          # target parents (their children are at level i)
        s_children = [(pi, c) for pi, par in enumerate(sp.nodes) for c in par.children]
        t_children = [(pi, c) for pi, par in enumerate(tp.nodes) for c in par.children]
        n_src_nodes, n_tgt_nodes = len(sp.nodes), len(tp.nodes)
        tot_src_ch, tot_tgt_ch = len(s_children), len(t_children)
        assert tot_src_ch * n_tgt_nodes == len(prev_heights)   # :252
        sc_par = np.array([pi for pi, _ in s_children])
        sc_cx = np.array([c.cx for _, c in s_children]); sc_cy = np.array([c.cy for _, c in s_children]); sc_r = np.array([c.r for _, c in s_children])
        tc_par = np.array([pi for pi, _ in t_children])
        tc_cx = np.array([c.cx for _, c in t_children]); tc_cy = np.array([c.cy for _, c in t_children]); tc_r = np.array([c.r for _, c in t_children])
        # rankOr: (src child circle, tgt parent circle); rankEq: (src parent, tgt child) (:291-296)
        rank_or = rank_est(k, sc_cx[None, :], sc_cy[None, :], sc_r[None, :],
                           tp.cx[tc_par][:, None], tp.cy[tc_par][:, None], tp.r[tc_par][:, None])
        rank_eq = rank_est(k, sp.cx[sc_par][None, :], sp.cy[sc_par][None, :], sp.r[sc_par][None, :],
                           tc_cx[:, None], tc_cy[:, None], tc_r[:, None])
        rank = np.maximum(rank_or, rank_eq)                      # [tgt child, src child]
        # row height of block row (tgt child t, src parent s) = max over s's children (:305-307)
        heights = np.zeros((tot_tgt_ch, n_src_nodes), dtype=np.int64)
        np.maximum.at(heights, (np.arange(tot_tgt_ch)[:, None].repeat(tot_src_ch, 1), sc_par[None, :].repeat(tot_tgt_ch, 0)), rank)
        row_h = heights.reshape(-1)                              # block row i = t*n_src_nodes + s
        row_off = np.concatenate([[0], np.cumsum(row_h)])
        col_off = np.concatenate([[0], np.cumsum(prev_heights)])
        ch = []
        for t in range(tot_tgt_ch):
            tpar = int(tc_par[t])
            tchild = t_children[t][1]
            for sidx in range(tot_src_ch):
                spar = int(sc_par[sidx])
                schild = s_children[sidx][1]
                bi = t * n_src_nodes + spar
                bj = tpar * tot_src_ch + sidx
                m, n = int(row_h[bi]), int(prev_heights[bj])
                leaf = desc.add(NODE_DENSE, m, n)
                if recipes:
                    # :338-358: orig = src child circle (n pts), equiv = src parent circle (m pts),
                    # targets = tgt child circle (m pts)
                    desc.recipe[leaf] = ("reexp", _circ_pts(schild, n), _circ_pts(sp.nodes[spar], m), _circ_pts(tchild, m))
                ch.append((leaf, int(row_off[bi]), int(col_off[bj])))
        factors.append(desc.add(NODE_BLOCK, int(row_off[-1]), int(col_off[-1]), ch, BF_TYPE_BLOCK_COO))
        prev_heights = row_h

    # --- makeLastFactor (:403-509): one block per target node at level nf-2 -
    tlv = tl[nf - 2]
    assert len(tlv.nodes) == len(prev_heights)                  # :419
    ch, r0, c0 = [], 0, 0
    for i, t in enumerate(tlv.nodes):
        n = int(prev_heights[i])
        leaf = desc.add(NODE_DENSE, t.npts, n)
        if recipes:
            desc.recipe[leaf] = ("kernel", _circ_pts(src, n), _node_pts(t))
        ch.append((leaf, r0, c0))
        r0 += t.npts
        c0 += n
    factors.append(desc.add(NODE_BLOCK, r0, c0, ch, BF_TYPE_BLOCK_DIAG))
    assert r0 == tgt.npts

    # product order = reversed build order (:692-695)
    return desc.add(NODE_PRODUCT, tgt.npts, src.npts, [(f, 0, 0) for f in reversed(factors)])


def _multilevel_rec(desc: Desc, k, src_nodes, tgt_nodes, recipes, stats):
    """facHelm2MakeMultilevel_rec (src/fac_helm2.c:860-941): a dense grid of
    blocks; returns (children list, rows, cols) for the enclosing BLOCK."""
    ch = []
    r0 = 0
    for tn in tgt_nodes:
        c0 = 0
        for sn in src_nodes:
            m, n = tn.npts, sn.npts
            if m * n < MAX_DENSE_MATRIX_SIZE:                       # :888
                node = desc.add(NODE_DENSE, m, n)
                if recipes:
                    desc.recipe[node] = ("kernel", _node_pts(sn), _node_pts(tn))
                stats["dense_leaves"] += 1
            elif separated(sn, tn):                                 # :890 -> :777-804
                nf, L = prepare(k, sn, tn)
                if nf == 0:
                    node = desc.add(NODE_DENSE, m, n)
                    if recipes:
                        desc.recipe[node] = ("kernel", _node_pts(sn), _node_pts(tn))
                    stats["dense_leaves"] += 1
                else:
                    node = _make_product(desc, k, sn, tn, nf, L, recipes)
                    stats["products"][nf] = stats["products"].get(nf, 0) + 1
            else:                                                   # :814-858
                sub, rr, cc = _multilevel_rec(desc, k, sn.children, tn.children, recipes, stats)
                node = desc.add(NODE_BLOCK, rr, cc, sub, BF_TYPE_BLOCK_DENSE)
                stats["block_dense"] += 1
            ch.append((node, r0, c0))
            c0 += n
        r0 += tn.npts
    cols = sum(s.npts for s in src_nodes)
    return ch, r0, cols


def helm2_multilevel_structure(points: np.ndarray, k: float, recipes: bool = False,
                               exact_sift: bool = False, tgt_points: np.ndarray | None = None):
    """bfFacHelm2MakeMultilevel (src/fac_helm2.c:943-1002).  With `tgt_points` None the source and
    target trees are the same quadtree (examples/simple/bf_all_blocks.c:130) and (Desc, root QNode,
    perm) is returned; with separate target points (the evaluation butterfly of
    examples/multiple_scattering/multiple_scattering_context.c:998) a second quadtree is built and
    (Desc, (src root, tgt root), (src perm, tgt perm)) is returned -- rows follow the target tree."""
    root, perm = build_quadtree(points, exact_sift=exact_sift)
    troot, tperm = (root, perm) if tgt_points is None else build_quadtree(tgt_points, exact_sift=exact_sift, tree=1)
    lv, tlv = levels_below(root), levels_below(troot)
    if len(lv) < 3 or len(tlv) < 3:
        raise ValueError("quadtree has fewer than 3 levels")
    src2, tgt2 = lv[2].nodes, tlv[2].nodes                          # level-2 nodes (:956-982)
    desc = Desc(dtype=0)
    stats = {"dense_leaves": 0, "products": {}, "block_dense": 0}
    ch, rr, cc = _multilevel_rec(desc, k, src2, tgt2, recipes, stats)
    stats["block_dense"] += 1
    desc.root = desc.add(NODE_BLOCK, rr, cc, ch, BF_TYPE_BLOCK_DENSE)
    # block-row id of each root child, for row sharding
    ncol = len(src2)
    desc.top_row_block = [i // ncol for i in range(len(ch))]
    desc.meta = dict(stats=stats, n=len(points), k=float(k), top_rows=[t.npts for t in tgt2])
    if tgt_points is None:
        return desc, root, perm
    desc.meta["m"] = len(tgt_points)
    return desc, (root, troot), (perm, tperm)


def single_product_structure(points: np.ndarray, k: float, src_path, tgt_path, recipes=True):
    """One butterfly for a chosen (src, tgt) node pair, as
    examples/simple/bf_one_block.c does with bfFacHelm2MakeSingleLevel
    (src/fac_helm2.c:706-729).  Paths are child-index sequences from the root
    (indices into the *non-empty* children list)."""
    root, perm = build_quadtree(points)

    def walk(path):
        nd = root
        for c in path:
            nd = nd.children[c]
        return nd
    sn, tn = walk(src_path), walk(tgt_path)
    nf, L = prepare(k, sn, tn)
    if nf == 0:
        raise ValueError("pair is not butterfliable")
    desc = Desc(dtype=0)
    desc.root = _make_product(desc, k, sn, tn, nf, L, recipes)
    desc.meta = dict(n=len(points), k=float(k), src=(sn.i0, sn.i1), tgt=(tn.i0, tn.i1), num_factors=nf)
    return desc, root, perm, sn, tn


def shard_desc(desc: Desc, row_blocks):
    """Restrict a multilevel descriptor to a subset of top-level block rows:
    returns a new root BLOCK whose children are those rows' blocks re-based to
    contiguous local rows.  The node arrays are shared."""
    keep = set(row_blocks)
    top_rows = desc.meta["top_rows"]
    base, acc = {}, 0
    for rb in sorted(keep):
        base[rb] = acc
        acc += top_rows[rb]
    orig_off = np.concatenate([[0], np.cumsum(top_rows)])
    ch = []
    for (c, r0, c0), rb in zip(desc.children[desc.root], desc.top_row_block):
        if rb in keep:
            ch.append((c, r0 - int(orig_off[rb]) + base[rb], c0))
    new_root = desc.add(NODE_BLOCK, acc, desc.cols[desc.root], ch, BF_TYPE_BLOCK_DENSE)
    return new_root, acc


def shard_desc_blocks(desc: Desc, child_indices):
    """Restrict a multilevel descriptor to a subset of its top-level (row, col)
    blocks, kept at their original offsets: the result maps the full x to a
    full-length *partial* y (rows without blocks are zero); partial results of
    all ranks add up to y (one all-reduce).  Returns the new root id."""
    keep = set(child_indices)
    ch = [c for i, c in enumerate(desc.children[desc.root]) if i in keep]
    return desc.add(NODE_BLOCK, desc.rows[desc.root], desc.cols[desc.root], ch, BF_TYPE_BLOCK_DENSE)


def shard_desc_children(desc: Desc, child_indices):
    """Restrict a multilevel descriptor to a subset of its top-level (row, col) blocks and compact the rows: the block
    rows those blocks touch, in global order, stacked without gaps.  The result maps the full x to this rank's PARTIAL
    rows -- complete for a block row it owns whole, a partial sum for one it shares by columns with other ranks (their
    partials are added after the all-gather: dist.py "rowsum").  Returns (new root id, [touched block rows], rows)."""
    keep = sorted(set(child_indices))
    ch_all = desc.children[desc.root]
    trb = desc.top_row_block
    touched = sorted({trb[i] for i in keep})
    top_rows = desc.meta["top_rows"]
    orig_off = np.concatenate([[0], np.cumsum(top_rows)])
    base, acc = {}, 0
    for rb in touched:
        base[rb] = acc
        acc += top_rows[rb]
    ch = [(ch_all[i][0], ch_all[i][1] - int(orig_off[trb[i]]) + base[trb[i]], ch_all[i][2]) for i in keep]
    return desc.add(NODE_BLOCK, acc, desc.cols[desc.root], ch, BF_TYPE_BLOCK_DENSE), touched, acc


# --------------------------------------------------------------------------
# array-backed descriptor: what the native layout (bfhip_layout.c) returns
# --------------------------------------------------------------------------
class _ChildIndex:
    def __init__(self, d):
        self._d = d

    def __getitem__(self, node):
        a = self._d._a
        b, e = int(a["childBegin"][node]), int(a["childBegin"][node + 1])
        return [(int(a["childNode"][i]), int(a["childRow0"][i]), int(a["childCol0"][i])) for i in range(b, e)]


class ArrayDesc:
    """The same expression as `Desc`, held as the flat BfhipDesc arrays (no per-node Python objects):
    what `native_multilevel_structure` returns.  Supports what compiling, building and sharding need:
    `arrays()`, `children[node]`, `add(...)`, `kind/rows/cols[node]`, `recipe_array`."""

    def __init__(self, arrays, root, dtype, top_row_block, recipe_array, meta):
        self._a = arrays
        self.root, self.dtype = int(root), int(dtype)
        self.top_row_block = list(top_row_block)
        self.recipe_array = recipe_array
        self.meta = meta
        self.children = _ChildIndex(self)

    @property
    def num_nodes(self):
        return len(self._a["kind"])

    kind = property(lambda self: self._a["kind"])
    rows = property(lambda self: self._a["rows"])
    cols = property(lambda self: self._a["cols"])

    def arrays(self):
        return self._a

    def add(self, kind, rows, cols, children=(), block_kind=0):
        a = self._a
        ch = list(children)
        a["kind"] = np.append(a["kind"], np.uint8(kind))
        a["blockKind"] = np.append(a["blockKind"], np.uint8(block_kind))
        a["rows"] = np.append(a["rows"], np.uint64(rows))
        a["cols"] = np.append(a["cols"], np.uint64(cols))
        a["childBegin"] = np.append(a["childBegin"], np.uint64(int(a["childBegin"][-1]) + len(ch)))
        for key, col in (("childNode", 0), ("childRow0", 1), ("childCol0", 2)):
            a[key] = np.concatenate([a[key], np.array([c[col] for c in ch], dtype=np.uint64)])
        return self.num_nodes - 1

    def leaf_elems(self):
        a = self._a
        return int((a["rows"].astype(np.int64) * a["cols"].astype(np.int64))[a["kind"] == NODE_DENSE].sum())

    def subtree_leaf_elems(self):
        """Leaf elements under every node (bfhipDescSubtreeLeafElems)."""
        from . import _capi
        da = _capi.DescArrays(self)
        out = np.zeros(self.num_nodes, dtype=np.uint64)
        _capi.check(_capi.load().bfhipDescSubtreeLeafElems(da.byref(), out.ctypes.data))
        return out.astype(np.int64)


def native_multilevel_structure(points: np.ndarray, k: float, tgt_points: np.ndarray | None = None):
    """helm2_multilevel_structure through the C layout (bfhipHelm2LayoutCreate[2]): same arrays
    (tests/test_layout_cpu.py), ~25x faster.  Returns (ArrayDesc, perm), or (ArrayDesc, (perm,
    tgt perm)) with a separate target tree."""
    from . import _capi
    lay = _capi.Helm2Layout(points, k, tgt_points)
    a = lay.arrays()
    rb, re = int(a["childBegin"][lay.root]), int(a["childBegin"][lay.root + 1])
    ncol = 0
    while ncol < re - rb and lay.top_row_block[ncol] == 0:
        ncol += 1
    top_rows = [int(a["rows"][int(a["childNode"][rb + i * ncol])]) for i in range((re - rb) // ncol)]
    prod = a["kind"] == NODE_PRODUCT
    nfac = (a["childBegin"][1:] - a["childBegin"][:-1])[prod].astype(np.int64)
    cnt = (a["childBegin"][1:] - a["childBegin"][:-1]).astype(np.int64)
    begin = a["childBegin"].astype(np.int64)
    factors = np.concatenate([a["childNode"][begin[p]:begin[p + 1]] for p in np.nonzero(prod)[0]]).astype(np.int64) if prod.any() else np.zeros(0, np.int64)
    stats = {"dense_leaves": int((a["kind"] == NODE_DENSE).sum()) - int(cnt[factors].sum()),   # near-field leaves only, as Desc's stats
             "products": {int(f): int(c) for f, c in zip(*np.unique(nfac, return_counts=True))},
             "block_dense": int(((a["kind"] == NODE_BLOCK) & (a["blockKind"] == BF_TYPE_BLOCK_DENSE)).sum())}
    desc = ArrayDesc(a, lay.root, lay.dtype, lay.top_row_block, lay.recipes,
                     dict(stats=stats, n=len(points), k=float(k), top_rows=top_rows))
    if tgt_points is None:
        return desc, lay.perm
    desc.meta["m"] = len(tgt_points)
    return desc, (lay.perm, lay.tgt_perm)


def native_single_product_structure(points: np.ndarray, k: float, src_path, tgt_path):
    """single_product_structure through the C layout (bfhipHelm2LayoutCreateSingle).  Returns
    (ArrayDesc, perm); the operator maps the source node's points to the target node's."""
    from . import _capi
    lay = _capi.Helm2Layout(points, k, single=(src_path, tgt_path))
    desc = ArrayDesc(lay.arrays(), lay.root, lay.dtype, [], lay.recipes, dict(n=len(points), k=float(k)))
    desc.top_row_block = None
    return desc, lay.perm
