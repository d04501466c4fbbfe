"""Structure of a streamed (`fac_streamer`) real butterfly factorization.

BASELINE config 5 applies the operand `examples/covariance/lbo_cov.c:188-189` gets from
`bfFacSpanGetMat(bfFacStreamerGetFacSpan(fs))`: a 1 x numFacs BlockDense row of products
`[Psi, W0, W1, ...]` (reference src/fac_span.c:126-155, src/fac.c:53-75).  Its block structure is not a
fixed pattern: it is whatever the streamer's merge-and-split recursion leaves behind.  This module
follows that recursion step by step on a small block algebra of its own and emits the flat
descriptor `include/bfhip.h: BfhipDesc` consumes:

  octree, leaf size 1 (row tree)                     src/octree_node.c:152-268, 270-296; src/bbox.c:66-85
  complete binary interval tree, post order (columns) src/interval_tree_node.c:69-101; src/tree_iter_post_order.c
  bfFacStreamerFeed: leaf-band partial factorization  src/fac_streamer.c:386-518; getPsiAndW src/fac.c:717-777
  continueFactorizing / mergeAndSplit                 src/fac_streamer.c:303-363; src/fac.c:1080-1294
  merge cut                                           src/fac.c:509-573
  Psi / W0 blocks of one row node                     src/fac.c:168-371, 575-647
  epsilon-rank cut                                    src/fac.c:867-1049; getLowRankApproximation :779-865
  block algebra the recursion leans on:
    GetRowRangeCopy  BlockDense src/mat_block_dense.c:336-432, BlockDiag src/mat_block_diag.c:273-340,
                     BlockCoo src/mat_block_coo.c:305-380, DenseReal src/mat_dense_real.c:417-459,
                     Identity src/mat_identity.c:123-147
    NumBytes         src/mat_block_dense.c:211-233, src/mat_block_coo.c:238-258, src/mat_block_diag.c:232-237,
                     src/mat_dense_real.c:202-207, src/mat_identity.c:102-105
    constructors     src/mat_block_diag.c:738-776, src/mat_block_dense.c:1150-1280, src/mat_block_coo.c:921-1045
    nonzero columns  src/mat_block_dense.c:903-927, src/mat_block_coo.c:538-613

The one numerical step -- the truncated SVD of a block (src/linalg.c:26-35, 1002-1082) -- is a
callback (`factorizer`).  With values it is the numpy restatement in `oracle/streamer_values.py` (test
infrastructure: parity operands at N <= ~16k).  Without values it is a *rank model*: blocks carry shapes
only, and the same recursion lays out a structure-exact, value-synthetic operand at sizes where the
SVDs themselves are out of reach of a CPU (N = 1M; SURVEY.md section 8(d) "Config 5").

Nothing here is on the timed path; it is host-side operand preparation.
"""
from __future__ import annotations

from bisect import bisect_left, bisect_right

import numpy as np

from .helm2_structure import (BF_TYPE_BLOCK_COO, BF_TYPE_BLOCK_DENSE, BF_TYPE_BLOCK_DIAG, Desc, NODE_BLOCK, NODE_DENSE,
                              NODE_IDENTITY, NODE_PRODUCT)

EPS = 2.220446049250313e-16   # include/bf/def.h:27
SIZEOF_BLOCK_DENSE, SIZEOF_BLOCK_COO = 64, 88   # sizeof(BfMatBlockDense / BfMatBlockCoo), SURVEY section 8(b)


# ---------------------------------------------------------------------------------------------------
# block algebra (shapes, optionally values)
# ---------------------------------------------------------------------------------------------------
class Dense:
    __slots__ = ("m", "n", "a")

    def __init__(self, m, n, a=None):
        self.m, self.n, self.a = int(m), int(n), a

    def num_bytes(self):
        return 8 * self.m * self.n

    def copy(self):
        return Dense(self.m, self.n, self.a)

    def row_range_copy(self, i0, i1):
        if i0 >= i1 or i1 > self.m:
            raise RuntimeError("DenseReal row range out of bounds (mat_dense_real.c:428-432)")
        return Dense(i1 - i0, self.n, None if self.a is None else self.a[i0:i1])

    def col_range_copy(self, j0, j1):
        return Dense(self.m, j1 - j0, None if self.a is None else self.a[:, j0:j1])


class Identity:
    __slots__ = ("m", "n")

    def __init__(self, n):
        self.m = self.n = int(n)

    def num_bytes(self):
        return 0

    def copy(self):
        return Identity(self.n)

    def row_range_copy(self, i0, i1):
        if i0 == 0 and i1 == self.m:
            return Identity(self.m)
        raise NotImplementedError("partial row range of an Identity (mat_identity.c:143-144)")


def _offsets(sizes):
    off = [0]
    for s in sizes:
        off.append(off[-1] + s)
    return off


class BlockDiag:
    __slots__ = ("blocks", "ro", "co", "m", "n")

    def __init__(self, blocks):
        self.blocks = blocks = list(blocks)
        ro, co, r, c = [0], [0], 0, 0
        for b in blocks:
            r += b.m; c += b.n
            ro.append(r); co.append(c)
        self.ro, self.co, self.m, self.n = ro, co, r, c

    def num_bytes(self):
        return sum(b.num_bytes() for b in self.blocks)

    def copy(self):
        return BlockDiag(b.copy() for b in self.blocks)

    def row_range_copy(self, i0, i1):
        if i0 > i1:
            raise ValueError
        indexed = []
        # the reference scans all blocks (mat_block_diag.c:287-294); only those in this window overlap
        k_lo = max(bisect_right(self.ro, i0) - 1, 0)
        k_hi = min(bisect_left(self.ro, i1), len(self.blocks))
        for k in range(k_lo, k_hi):
            block = self.blocks[k]
            i0_, i1_ = self.ro[k], self.ro[k + 1]
            if i1_ <= i0 or i1 <= i0_:
                continue
            m_ = i1_ - i0_
            a = i0 - i0_ if i0_ < i0 else 0
            b = m_ - (i1_ - i1 if i1 < i1_ else 0)
            indexed.append((0 if i0_ < i0 else i0_ - i0, self.co[k], block.row_range_copy(a, b)))
        return BlockCoo(i1 - i0, self.n, indexed)


class BlockDense:
    """nbr x nbc grid, blocks row-major."""
    __slots__ = ("nbr", "nbc", "blocks", "ro", "co", "m", "n")

    def __init__(self, nbr, nbc, blocks):
        self.nbr, self.nbc, self.blocks = nbr, nbc, list(blocks)
        assert len(self.blocks) == nbr * nbc and self.blocks
        rs = [self.blocks[p * nbc].m for p in range(nbr)]
        cs = [self.blocks[q].n for q in range(nbc)]
        for p in range(nbr):
            for q in range(nbc):
                b = self.blocks[p * nbc + q]
                assert b.m == rs[p] and b.n == cs[q], "BlockDense blocks do not tile (mat_block_dense.c:1187-1193)"
        self.ro, self.co = _offsets(rs), _offsets(cs)
        self.m, self.n = self.ro[-1], self.co[-1]

    @classmethod
    def col(cls, blocks):
        blocks = list(blocks)
        return cls(len(blocks), 1, blocks)

    @classmethod
    def row(cls, blocks):
        blocks = list(blocks)
        return cls(1, len(blocks), blocks)

    def num_bytes(self):
        nb = len(self.blocks)
        return SIZEOF_BLOCK_DENSE + sum(b.num_bytes() for b in self.blocks) + (self.nbr + 1) * 8 + (self.nbc + 1) * 8 + 2 * nb * 8

    def copy(self):
        return BlockDense(self.nbr, self.nbc, (b.copy() for b in self.blocks))

    def row_range_copy(self, i0, i1):
        if i0 > i1 or i1 > self.m:
            raise ValueError("BlockDense row range out of bounds")
        ro = self.ro
        # p0: first block row whose offset is >= i0; p1: one past the last block row whose offset is <= i0
        # (sic: compared with i0, mat_block_dense.c:352-363 -- linear scans there, bisection here)
        p0 = bisect_left(ro, i0, 0, self.nbr)
        p1 = max(p0, bisect_right(ro, i0, 0, self.nbr))
        if p0 == p1:
            p0 -= 1
            assert ro[p0] <= i0 and i1 <= ro[p1]
        i0_, i1_ = ro[p0], ro[p1]
        m_ = i1_ - i0_
        a = i0 - i0_ if i0_ < i0 else 0
        b = m_ - (i1_ - i1 if i1 < i1_ else 0)
        assert a <= b <= m_
        out = []
        for p in range(p0, p1):
            for q in range(self.nbc):
                block = self.blocks[p * self.nbc + q]
                assert block.m == m_, "row range spans differently sized block rows (mat_block_dense.c:406)"
                out.append(block.row_range_copy(a, b))
        return BlockDense(p1 - p0, self.nbc, out)

    def nonzero_col_ranges(self):
        return [(0, self.n)]            # "assume there are *no* zero blocks" (mat_block_dense.c:911-919)


class BlockCoo:
    """m x n, blocks placed at (i0, j0); block rows / columns = the distinct offsets."""
    __slots__ = ("m", "n", "blocks", "i0s", "j0s", "ro", "co")

    def __init__(self, m, n, indexed):
        self.m, self.n = int(m), int(n)
        self.blocks = [b for _, _, b in indexed]
        self.i0s = [int(i) for i, _, _ in indexed]
        self.j0s = [int(j) for _, j, _ in indexed]
        ro, co = {0, self.m}, {0, self.n}
        for i, j, b in indexed:
            ro.update((i, i + b.m))
            co.update((j, j + b.n))
        self.ro, self.co = sorted(ro), sorted(co)

    def num_bytes(self):
        nb = len(self.blocks)
        return SIZEOF_BLOCK_COO + sum(b.num_bytes() for b in self.blocks) + len(self.ro) * 8 + len(self.co) * 8 + 2 * nb * 8

    def copy(self):
        return BlockCoo(self.m, self.n, [(i, j, b.copy()) for i, j, b in zip(self.i0s, self.j0s, self.blocks)])

    def row_range_copy(self, i0, i1):
        if i0 > i1:
            raise ValueError
        indexed = []
        for i0_, j0_, block in zip(self.i0s, self.j0s, self.blocks):
            m_ = block.m
            i1_ = i0_ + m_
            if i1_ <= i0 or i1 <= i0_:
                continue
            a = i0 - i0_ if i0_ < i0 else 0
            b = m_ - (i1_ - i1 if i1 < i1_ else 0)
            indexed.append((0 if i0_ < i0 else i0_ - i0, j0_, block.row_range_copy(a, b)))
        return BlockCoo(i1 - i0, self.n, indexed)

    def _mark(self, nonzero, base):
        for j0, block in zip(self.j0s, self.blocks):
            if isinstance(block, BlockCoo):
                block._mark(nonzero, base + j0)
            elif isinstance(block, (Dense, Identity)):
                nonzero[base + j0:base + j0 + block.n] = True
            else:
                raise NotImplementedError("setColumnNonzerosForBlock: block type (mat_block_coo.c:562-563)")

    def nonzero_col_ranges(self):
        nonzero = np.zeros(self.n, dtype=bool)
        self._mark(nonzero, 0)
        edges = np.flatnonzero(np.diff(np.concatenate([[False], nonzero, [False]]).astype(np.int8)))
        return [(int(a), int(b)) for a, b in zip(edges[0::2], edges[1::2])]


class Product:
    __slots__ = ("factors", "m", "n")

    def __init__(self, factors):
        self.factors = list(factors)
        for a, b in zip(self.factors[:-1], self.factors[1:]):
            assert a.n == b.m, "product factors do not chain"
        self.m, self.n = self.factors[0].m, self.factors[-1].n

    def num_bytes(self):
        return sum(f.num_bytes() for f in self.factors)


# ---------------------------------------------------------------------------------------------------
# trees
# ---------------------------------------------------------------------------------------------------
class Octree:
    """bfOctreeInit(points, maxLeafSize): arrays over nodes (node 0 = root).  first/last = the index
    range in tree order (bfTreeNodeGetFirstIndex / GetLastIndex, src/tree_node.c:160-176), child[v] = the 8
    child slots in octant order (-1 = empty), perm = BfTree.perm."""

    def __init__(self, points, max_leaf_size=1):
        pts = np.ascontiguousarray(points, dtype=np.float64)
        n = len(pts)
        lo, hi = pts.min(axis=0), pts.max(axis=0)
        c = (lo + hi) / 2                                   # bfBoundingBox3RescaleToCube, src/bbox.c:66-85
        dmax = float((hi - lo).max())
        lo, hi = c - dmax / 2, c + dmax / 2
        self.bbox = (lo.copy(), hi.copy())
        lo, hi = lo - 1e2 * EPS, hi + 1e2 * EPS                 # src/octree_node.c:283-286
        order = np.arange(n)
        plo = np.tile(lo, (n, 1))
        phi = np.tile(hi, (n, 1))
        first, last, depth, parent, slot = [0], [n], [0], [-1], [-1]
        node_of = np.zeros(n, dtype=np.int64)               # current node of each point (in `order`)
        active = np.full(n, n > max_leaf_size)
        d = 0
        while active.any():
            idx = np.flatnonzero(active)                    # positions (tree order) still being split
            p = pts[order[idx]]
            split = (plo[idx] + phi[idx]) / 2               # bfBoundingBox3GetCenter of the node's box
            gt = p > split                                  # inOctant1..8: <= goes low (src/octree_node.c:105-140)
            q = gt[:, 0] * 4 + gt[:, 1] * 2 + gt[:, 2] * 1
            key = node_of[idx] * 8 + q
            srt = np.argsort(key, kind="stable")
            # positions idx are grouped by node already, so sorting by key only permutes inside nodes
            order[idx] = order[idx][srt]
            key = key[srt]
            gts = gt[srt]
            nlo = np.where(gts, split[srt], plo[idx][srt])
            nhi = np.where(gts, phi[idx][srt], split[srt])
            plo[idx], phi[idx] = nlo, nhi
            uniq, start, counts = np.unique(key, return_index=True, return_counts=True)
            base = len(first)
            new_ids = base + np.arange(len(uniq))
            for u, s, cnt in zip(uniq.tolist(), start.tolist(), counts.tolist()):
                first.append(int(idx[s])); last.append(int(idx[s]) + cnt); depth.append(d + 1)
                parent.append(u // 8); slot.append(u % 8)
            rep = np.repeat(new_ids, counts)
            node_of[idx] = rep
            active[idx] = np.repeat(counts > max_leaf_size, counts)
            d += 1
            if d > 64:
                raise RuntimeError("octree deeper than 64 levels: coincident points?")
        self.perm = order
        self.first = np.asarray(first, dtype=np.int64)
        self.last = np.asarray(last, dtype=np.int64)
        self.depth = np.asarray(depth, dtype=np.int32)
        self.parent = np.asarray(parent, dtype=np.int64)
        m = len(first)
        self.child = np.full((m, 8), -1, dtype=np.int64)
        ids = np.arange(1, m)
        self.child[self.parent[1:], np.asarray(slot[1:], dtype=np.int64)] = ids
        self.max_depth = int(self.depth.max())
        self.num_points = n

    def children(self, v):
        return [int(c) for c in self.child[v] if c >= 0]

    def level(self, d):
        """bfTreeGetLevelPtrArray: the nodes of depth d in tree order."""
        ids = np.flatnonzero(self.depth == d)
        return [int(i) for i in ids[np.argsort(self.first[ids], kind="stable")]]

    def rows(self, v):
        return int(self.last[v] - self.first[v])


class BinaryTree:
    """bfIntervalTreeInitEmpty(tree, a, b, 2, depth): complete binary tree; nodes in heap numbering
    (root 1, children 2v, 2v + 1), leaves at depth `depth`."""

    def __init__(self, depth):
        self.depth = int(depth)

    def is_leaf(self, v):
        return v >= (1 << self.depth)

    def children(self, v):
        return [2 * v, 2 * v + 1]

    def post_order(self):
        out = []

        def rec(v):
            if not self.is_leaf(v):
                rec(2 * v); rec(2 * v + 1)
            out.append(v)
        rec(1)
        return out

    def leaf_interval(self, v, a, b):
        """(a_v, b_v, isLeftmost, isRightmost) of leaf v (src/interval_tree_node.c:72-95)."""
        path = []
        while v > 1:
            path.append(v & 1)
            v >>= 1
        left = right = True
        for bit in reversed(path):
            delta = (b - a) / 2
            if bit == 0:
                a, b = a, a + delta
                right = False
            else:
                a, b = a + delta, b
                left = False
        return a, b, left, right


# ---------------------------------------------------------------------------------------------------
# the streamer
# ---------------------------------------------------------------------------------------------------
class Fac:
    __slots__ = ("col_node", "row_nodes", "Psi", "W")

    def __init__(self, col_node, row_nodes, Psi, W):
        self.col_node, self.row_nodes, self.Psi, self.W = col_node, list(row_nodes), Psi, list(W)


class Streamer:
    """bfFacStreamer over (row octree, binary column tree of `col_depth` levels).

    factorizer.svd(block, row_node, col_node, tree) -> (k, numSingularValues, U, W) where `block` is a
    block-algebra node (m x n), U is Dense(m, k), W is Dense(k, n) = S V^T, and k the number of singular
    values kept by the tolerance (bfTruncSpecGetNumTerms, src/linalg.c:26-35).  `truncated` = k < min(m, n)."""

    def __init__(self, row_tree, col_depth, factorizer, min_num_rows=20, min_num_cols=20, row_tree_init_depth=0):
        self.tree = row_tree
        self.cols = BinaryTree(col_depth)
        self.order = self.cols.post_order()
        self.pos = 0
        self.fz = factorizer
        self.min_rows, self.min_cols = int(min_num_rows), int(min_num_cols)
        self.init_depth = int(row_tree_init_depth)
        self.partial = []
        self.stats = dict(svds=0, merges=0, feeds=0)

    # ---- iteration ------------------------------------------------------------------------------
    def is_done(self):
        return self.pos >= len(self.order)

    def current_col_node(self):
        return self.order[self.pos]

    # ---- bfFacStreamerFeed (src/fac_streamer.c:386-518) ---------------------------------------------
    def feed(self, phi):
        """phi: Dense(N, n_j) holding (or standing for) the columns of the current leaf column node,
        rows in row-tree order."""
        col_node = self.current_col_node()
        assert self.cols.is_leaf(col_node)
        t = self.tree
        if phi.m != t.num_points:
            raise ValueError("Phi has the wrong number of rows")
        Psis, Ws, row_nodes = [], [], []
        stack = list(reversed(t.level(self.init_depth)))
        while stack:
            v = stack.pop()
            i0, i1 = int(t.first[v]), int(t.last[v])
            block = phi.row_range_copy(i0, i1)          # bfMatGetRowRange (a view in the reference)
            ok = True
            if phi.n < self.min_cols:                   # getPsiAndW_skinny (src/fac.c:649-676, 741-742)
                Psi, W = block.copy(), Identity(phi.n)
            elif i1 - i0 < self.min_rows:               # src/fac.c:746-760
                Psi, W = Identity(i1 - i0), block.copy()
            else:                                       # getPsiAndW_normal (src/fac.c:678-715)
                k, ns, U, SVt = self.fz.svd(block, v, col_node, t)
                self.stats["svds"] += 1
                ok = k < ns
                Psi, W = U, SVt
            if ok:
                Psis.append(Psi); Ws.append(W); row_nodes.append(v)
                continue
            kids = t.children(v)
            assert kids, "uncompressed leaf row node (src/fac_streamer.c:447)"
            stack.extend(reversed(kids))
        # makeLeafNodePartialFac (src/fac.c:84-121)
        self.partial.append(Fac(col_node, row_nodes, BlockDiag(Psis), [BlockDense.col(Ws)]))
        self.stats["feeds"] += 1
        self.pos += 1
        self._continue()

    # ---- continueFactorizing (src/fac_streamer.c:303-363) -------------------------------------------
    def _continue(self):
        while not self.is_done():
            cn = self.current_col_node()
            if self.cols.is_leaf(cn):
                break
            kids = self.cols.children(cn)
            facs = [next(f for f in self.partial if f.col_node == c) for c in kids]   # getCurrentPartialFacs
            merged = self._merge_and_split(facs, cn)
            self.partial.append(merged)
            self.partial = [f for f in self.partial if f.col_node not in kids]          # deletePrevFacs
            self.stats["merges"] += 1
            self.pos += 1

    # ---- getMergeCut (src/fac.c:509-573) --------------------------------------------------------------
    def _merge_cut(self, facs):
        t = self.tree
        span = [(int(t.first[f.row_nodes[0]]), int(t.last[f.row_nodes[-1]])) for f in facs]
        if any(s != span[0] for s in span):
            raise ValueError("partial factorizations differ in row span (src/fac.c:519-520)")

        def argmax_last(nodes):
            best = nodes[0]
            for v in nodes[1:]:
                if t.last[v] > t.last[best]:
                    best = v
            return int(t.last[best]), best
        i1, node = argmax_last([f.row_nodes[0] for f in facs])
        cut = [node]
        i1_final, _ = argmax_last([f.row_nodes[-1] for f in facs])
        while i1 != i1_final:
            nodes = []
            for f in facs:
                v = next((v for v in f.row_nodes if t.first[v] == i1), None)    # getNodeByFirstIndex
                if v is None:
                    raise RuntimeError("no row node starts at the merge cut (src/fac.c:495-496)")
                nodes.append(v)
            i1, node = argmax_last(nodes)
            cut.append(node)
        return cut

    # ---- getPsiAndW0BlocksByRowNodeForPartialFac (src/fac.c:227-371) ----------------------------------
    def _psi_w0_of_fac(self, fac, i0, i1):
        subs = []

        def rec(mat, i0p, j0p):                           # getIndexedPsiSubblocksInRowRangeRec (:168-209)
            if isinstance(mat, BlockDiag):
                for k, block in enumerate(mat.blocks):
                    a = i0p + mat.ro[k]
                    b = a + block.m
                    if not (i1 <= a or b <= i0):
                        rec(block, a, j0p + mat.co[k])
            elif isinstance(mat, (Dense, BlockCoo, BlockDense, Identity)):
                subs.append((i0p, j0p, mat))
            else:
                raise NotImplementedError
        rec(fac.Psi, 0, 0)
        assert subs
        W0 = fac.W[0]
        Ps, Ws = [], []
        i1p = j1p = None
        for a, j0, mat in subs:
            b, j1 = a + mat.m, j0 + mat.n
            assert i0 <= a and b <= i1, "Psi subblock straddles the row node (src/fac.c:287)"
            assert i1p is None or a == i1p          # adjacent row spans (:290)
            assert j1p is None or j0 == j1p         # adjacent column spans (:298)
            i1p, j1p = b, j1
            Ps.append(mat.copy())
            Ws.append(W0.row_range_copy(j0, j1))
        if len(Ps) == 1:
            return Ps[0], Ws[0]
        return BlockDiag(Ps), BlockDense.col(Ws)

    # ---- findEpsilonRankCutAndGetNewBlocks (src/fac.c:867-1049) ---------------------------------------
    def _eps_rank_cut(self, root, psi_star, col_node):
        t = self.tree
        i0 = int(t.first[root])
        assert psi_star.m == t.rows(root)
        cut, Ps, Ws = [], [], []
        stack = [root]
        while stack:
            v = stack.pop()
            a, b = int(t.first[v]) - i0, int(t.last[v]) - i0
            sub = psi_star.row_range_copy(a, b)
            m, n = sub.m, sub.n
            if m < self.min_rows:                                   # :944-956
                Psi, W0 = Identity(m), sub
            elif n < self.min_cols:                                 # :963-975
                Psi, W0 = sub, Identity(n)
            else:
                k, ns, U, W0 = self.fz.svd(sub, v, col_node, t)     # getLowRankApproximation (:779-865)
                self.stats["svds"] += 1
                truncated = k < ns
                ranges = sub.nonzero_col_ranges()
                if len(ranges) > 1 or ranges[0][0] > 0 or ranges[0][1] < n:      # shouldFixSparsity (:810-851)
                    W0 = BlockCoo(W0.m, W0.n, [(0, j0, W0.col_range_copy(j0, j1)) for j0, j1 in ranges])
                compressed = W0.num_bytes() < sub.num_bytes()       # :981
                if not (truncated and compressed):
                    stack.extend(reversed(t.children(v)))           # :988-1001
                    continue
                Psi = U
            cut.append(v); Ps.append(Psi); Ws.append(W0)
        return cut, BlockDiag(Ps), BlockDense.col(Ws)

    # ---- mergeAndSplit (src/fac.c:1080-1294) ----------------------------------------------------------
    def _merge_and_split(self, facs, col_node):
        num_w = facs[0].W.__len__()
        if any(len(f.W) != num_w for f in facs):
            raise RuntimeError("partial factorizations differ in depth (src/fac.c:1100-1106)")
        t = self.tree
        row_nodes, Ps, W0s, W1s = [], [], [], []
        for v in self._merge_cut(facs):
            i0, i1 = int(t.first[v]), int(t.last[v])
            pw = [self._psi_w0_of_fac(f, i0, i1) for f in facs]       # getPsiAndW0BlocksByRowNode (:575-647)
            psi_star = BlockDense.row(p for p, _ in pw)
            W1s.append(BlockDiag(w for _, w in pw))
            assert psi_star.n == W1s[-1].m
            cut, Psi, W0 = self._eps_rank_cut(v, psi_star, col_node)
            row_nodes += cut; Ps.append(Psi); W0s.append(W0)
        W = [BlockDiag(W0s), BlockDense.col(W1s)]
        for k in range(1, num_w):                                     # :1227-1252
            W.append(BlockDiag(f.W[k] for f in facs))
        return Fac(col_node, row_nodes, BlockDiag(Ps), W)

    # ---- bfFacSpanGetMat (src/fac_span.c:126-155; bfFacGetMatProduct src/fac.c:53-75) -----------------
    def get_mat(self):
        return BlockDense.row(Product([f.Psi] + f.W) for f in self.partial)


# ---------------------------------------------------------------------------------------------------
# rank model (value-free factorizer for structure-exact synthetic operands)
# ---------------------------------------------------------------------------------------------------
class LboRankModel:
    """Stands in for the truncated SVD where only shapes are wanted (N = 1M: the SVDs themselves are
    out of reach of a CPU, SURVEY.md section 8(d) "Config 5").

    The columns streamed in examples/covariance are Laplace-Beltrami eigenvectors; restricted to a surface
    patch that holds a fraction f of the points, the eigenfunctions with frequency in [w0, w1) span,
    to a relative tolerance, a space whose dimension follows the local Weyl law with a boundary layer:

        rank(f, w0, w1) = (alpha sqrt(f) w1 + delta)^2 - max(alpha sqrt(f) w0 - delta, 0)^2

    in units where the whole surface holds w^2 eigenfunctions below frequency w (unit sphere: l (l + 1) =
    w^2, count (l + 1)^2).  alpha = 1.25, delta = 3.0 are fitted to the ranks the numpy SVDs of
    oracle/streamer_values.py find at tol = 1e-3 on spherical harmonics, N = 4096 ... 16384 (rms log
    error 0.25; tests/test_streamer_structure.py compares the structures the two produce).  The result is
    clipped to the block (min(m, n)): a block the model cannot compress is reported as not truncated and
    the recursion descends, exactly as with a real SVD."""

    def __init__(self, num_points, wmax, col_depth, band_columns, alpha=1.75, delta=3.0):
        self.n, self.wmax, self.depth = int(num_points), float(wmax), int(col_depth)
        self.alpha, self.delta = float(alpha), float(delta)
        # columns under every node of the frequency tree (heap numbering): no block of a band can have
        # a higher rank than the band has columns, whatever the patch
        nb = 1 << self.depth
        self.band_cols = [0] * (2 * nb)
        for j, c in enumerate(band_columns):
            self.band_cols[nb + j] = int(c)
        for v in range(nb - 1, 0, -1):
            self.band_cols[v] = self.band_cols[2 * v] + self.band_cols[2 * v + 1]

    def band(self, col_node):
        lvl = int(col_node).bit_length() - 1
        width = self.wmax / (1 << lvl)
        idx = col_node - (1 << lvl)
        return idx * width, (idx + 1) * width

    def rank(self, rows, cols, col_node):
        w0, w1 = self.band(col_node)
        s = self.alpha * np.sqrt(rows / self.n)
        a = s * w1 + self.delta
        b = max(s * w0 - self.delta, 0.0)
        return max(1, min(int(rows), int(cols), self.band_cols[col_node], int(np.ceil(a * a - b * b))))

    def svd(self, block, row_node, col_node, tree):
        m, n = block.m, block.n
        parts = block.blocks if isinstance(block, BlockDense) and block.nbr == 1 else [block]
        if all(isinstance(b, (Dense, Identity)) for b in parts):
            # the row node lies inside ONE patch of every child factorization: the block's columns
            # are those patches' bases restricted to its rows, i.e. (a basis of) the band on this patch
            k = self.rank(m, n, col_node)
        else:
            # the row node covers several patches of a child: their bases are independent (block
            # diagonal), nothing is shared; only all-zero columns (patches outside these rows) drop out
            k = max(1, min(m, nonzero_cols(block)))
        return k, min(m, n), Dense(m, k), Dense(k, n)


def nonzero_cols(mat):
    """Number of columns of a block-algebra node that hold at least one block."""
    if isinstance(mat, (Dense, Identity)):
        return mat.n
    if isinstance(mat, BlockDiag):
        return sum(nonzero_cols(b) for b in mat.blocks)
    if isinstance(mat, BlockDense):
        return sum(max(nonzero_cols(mat.blocks[p * mat.nbc + q]) for p in range(mat.nbr)) for q in range(mat.nbc))
    if isinstance(mat, BlockCoo):
        if len(mat.blocks) == 1:
            return nonzero_cols(mat.blocks[0])
        mask = np.zeros(mat.n, dtype=bool)
        for j, b in zip(mat.j0s, mat.blocks):
            mask[j:j + b.n] = True        # a nested block's own zero columns are not tracked: upper bound
        return int(mask.sum())
    raise TypeError(type(mat))


def fibonacci_sphere(n):
    """n quasi-uniform points on the unit sphere (the point set of the survey's streamer probe and of
    tests/generate_data_for_test_linalg.py:29-32 in the reference)."""
    i = np.arange(n, dtype=np.float64)
    x = 1 - 2.0 * (i + 0.5) / n
    r = np.sqrt(1 - x * x)
    th = np.pi * (np.sqrt(5.0) - 1) * i
    return np.stack([x, r * np.cos(th), r * np.sin(th)], axis=1)


def sphere_band_columns(wmax, col_depth):
    """Number of Laplace-Beltrami eigenfunctions of the unit sphere per leaf band of the frequency tree
    over [0, wmax]: degree l has 2l + 1 of them at frequency sqrt(l (l + 1)); the leftmost / rightmost
    leaves are open-ended (src/lbo.c:41-68).  Returns ([count per leaf, left to right], lmax)."""
    nb = 1 << col_depth
    counts = [0] * nb
    l = 0
    while True:
        w = np.sqrt(l * (l + 1.0))
        if w >= wmax:
            break
        j = min(int(w / (wmax / nb)), nb - 1)
        counts[j] += 2 * l + 1
        l += 1
    return counts, l - 1


def stream_structure(tree, wmax, col_depth, band_columns, model=None, min_rows=20, min_cols=20, max_cols=None):
    """Value-free run of the streamer: leaf band j of the frequency tree receives band_columns[j] columns
    (bands with no column are fed as in the reference: an empty block is still a feed).  Stops early after
    `max_cols` columns like lbo_cov.c:141.  Returns the Streamer (get_mat() = the operand's graph)."""
    model = model or LboRankModel(tree.num_points, wmax, col_depth, band_columns)
    st = Streamer(tree, col_depth, model, min_rows, min_cols, 0)
    fed = 0
    while not st.is_done():
        leaf = st.current_col_node()
        j = leaf - (1 << col_depth)
        st.feed(Dense(tree.num_points, band_columns[j]))
        fed += band_columns[j]
        if max_cols is not None and fed >= max_cols:
            break
    return st


def octree_depth(points):
    """Depth of the leaf-size-1 octree on `points` (bfhipStreamerOctreeDepth: the C layout's own tree)."""
    import ctypes as C

    from . import _capi
    pts = np.ascontiguousarray(points, dtype=np.float64)
    d = C.c_uint32(0)
    _capi.check(_capi.load().bfhipStreamerOctreeDepth(pts.ctypes.data, len(pts), C.byref(d)))
    return int(d.value)


def native_stream_structure(points, wmax, col_depth, band_columns, min_rows=20, min_cols=20, max_cols=None, alpha=1.75, delta=3.0):
    """stream_structure + get_mat + to_desc under the rank model, through the C layout (bfhipStreamerLayoutCreate,
    butterfly_amd/csrc/bfhip_streamer_layout.c): the same arrays (tests/test_streamer_layout_cpu.py), ~50x faster --
    N = 1M x 65536 columns in seconds instead of minutes.  Returns (ArrayDesc, perm, stats) with stats = the graph
    statistics of graph_stats plus the streamer's counters."""
    from . import _capi
    from .helm2_structure import ArrayDesc
    lay = _capi.StreamerLayout(points, wmax, col_depth, band_columns, min_rows, min_cols, max_cols, alpha, delta)
    desc = ArrayDesc(lay.arrays(), lay.root, lay.dtype, [], None, dict(n=len(points), stats=lay.stats))
    desc.top_row_block = None
    return desc, lay.perm, lay.stats


# ---------------------------------------------------------------------------------------------------
# flat descriptor
# ---------------------------------------------------------------------------------------------------
def to_desc(root, with_values=True):
    """Block-algebra graph -> (Desc, {leaf node id: values}).  BlockDense / BlockDiag / BlockCoo
    children sit at their (row offset, column offset) exactly as bfhip_ir.c reads them from a BfMat."""
    d = Desc(dtype=1)
    vals = {}
    # (Desc.add inlined: millions of nodes; sizes are Python ints already, child lists are fresh)
    kinds, rows, cols, children, bkinds = d.kind, d.rows, d.cols, d.children, d.block_kind

    def add(kind, m, n, ch=(), bk=0):
        kinds.append(kind); rows.append(m); cols.append(n); children.append(ch); bkinds.append(bk)
        return len(kinds) - 1

    def rec(mat):
        t = type(mat)
        if t is Dense:
            node = add(NODE_DENSE, mat.m, mat.n)
            if with_values and mat.a is not None:
                vals[node] = np.ascontiguousarray(mat.a, dtype=np.float64)
            return node
        if t is Identity:
            return add(NODE_IDENTITY, mat.m, mat.n)
        if t is Product:
            return add(NODE_PRODUCT, mat.m, mat.n, [(rec(f), 0, 0) for f in mat.factors])
        if t is BlockDiag:
            ro, co = mat.ro, mat.co
            ch = [(rec(b), ro[k], co[k]) for k, b in enumerate(mat.blocks)]
            return add(NODE_BLOCK, mat.m, mat.n, ch, BF_TYPE_BLOCK_DIAG)
        if t is BlockDense:
            ch = [(rec(mat.blocks[p * mat.nbc + q]), mat.ro[p], mat.co[q]) for p in range(mat.nbr) for q in range(mat.nbc)]
            return add(NODE_BLOCK, mat.m, mat.n, ch, BF_TYPE_BLOCK_DENSE)
        if t is BlockCoo:
            ch = [(rec(b), i, j) for i, j, b in zip(mat.i0s, mat.j0s, mat.blocks)]
            return add(NODE_BLOCK, mat.m, mat.n, ch, BF_TYPE_BLOCK_COO)
        raise TypeError(t)
    d.root = rec(root)
    return d, vals


def graph_stats(root):
    """Node counts the way the survey's probe walked the reference graph."""
    st = dict(product=0, blockCoo=0, blockDense=0, blockDiag=0, denseReal=0, identity=0, maxNest=0, leafBytes=0,
              minM=1 << 62, maxM=0, minN=1 << 62, maxN=0)

    def rec(mat, depth):
        st["maxNest"] = max(st["maxNest"], depth)
        if isinstance(mat, Dense):
            st["denseReal"] += 1; st["leafBytes"] += 8 * mat.m * mat.n
            st["minM"] = min(st["minM"], mat.m); st["maxM"] = max(st["maxM"], mat.m)
            st["minN"] = min(st["minN"], mat.n); st["maxN"] = max(st["maxN"], mat.n)
        elif isinstance(mat, Identity):
            st["identity"] += 1
        elif isinstance(mat, Product):
            st["product"] += 1
            for f in mat.factors:
                rec(f, depth + 1)
        else:
            st[{BlockDiag: "blockDiag", BlockDense: "blockDense", BlockCoo: "blockCoo"}[type(mat)]] += 1
            for b in mat.blocks:
                rec(b, depth + 1)
    rec(root, 0)
    return st
