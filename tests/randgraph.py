"""Test infrastructure: random expression graphs (Dense / Identity / Block /
Product, arbitrarily nested, ragged sizes) in the flat descriptor form, plus a
numpy densifier used as an independent known answer.  Mirrors the zoo of the
reference's real (fac_streamer) operands: Identity leaves inside factors,
blocks nested in products nested in blocks."""
import numpy as np

from butterfly_amd.helm2_structure import (BF_TYPE_BLOCK_COO, BF_TYPE_BLOCK_DENSE, BF_TYPE_BLOCK_DIAG, Desc,
                                           NODE_BLOCK, NODE_DENSE, NODE_IDENTITY, NODE_PRODUCT)


def _split(rng, total, parts):
    """`parts` positive integers summing to `total`."""
    parts = max(1, min(parts, total))
    cuts = np.sort(rng.choice(np.arange(1, total), size=parts - 1, replace=False)) if parts > 1 else np.array([], dtype=int)
    edges = np.concatenate([[0], cuts, [total]])
    return [int(b - a) for a, b in zip(edges[:-1], edges[1:])]


def _gen(rng, d, vals, m, n, depth, cplx, coo=True):
    """A node computing an m x n operator."""
    choice = rng.random()
    if depth == 0 or m < 4 or n < 4 or choice < 0.25:
        if m == n and rng.random() < 0.2:
            return d.add(NODE_IDENTITY, m, n)
        node = d.add(NODE_DENSE, m, n)
        v = rng.standard_normal((m, n)) / np.sqrt(n)
        if cplx:
            v = v + 1j * rng.standard_normal((m, n)) / np.sqrt(n)
        vals[node] = v
        return node
    if choice < 0.5:   # product through random inner dims
        nf = int(rng.integers(2, 4))
        dims = [m] + [int(rng.integers(2, max(3, min(m, n) + 8))) for _ in range(nf - 1)] + [n]
        fs = [_gen(rng, d, vals, dims[i], dims[i + 1], depth - 1, cplx, coo) for i in range(nf)]
        return d.add(NODE_PRODUCT, m, n, [(f, 0, 0) for f in fs])
    rs = _split(rng, m, int(rng.integers(1, 4)))
    cs = _split(rng, n, int(rng.integers(1, 4)))
    ro = np.concatenate([[0], np.cumsum(rs)])
    co = np.concatenate([[0], np.cumsum(cs)])
    if choice < 0.65 and len(rs) == len(cs):   # block diagonal
        ch = [(_gen(rng, d, vals, rs[i], cs[i], depth - 1, cplx, coo), int(ro[i]), int(co[i])) for i in range(len(rs))]
        return d.add(NODE_BLOCK, m, n, ch, BF_TYPE_BLOCK_DIAG)
    if choice < 0.85 and coo:                  # sparse of blocks, possibly with empty block rows
        ch = []
        for i in range(len(rs)):
            for j in range(len(cs)):
                if rng.random() < 0.6:
                    ch.append((_gen(rng, d, vals, rs[i], cs[j], depth - 1, cplx, coo), int(ro[i]), int(co[j])))
        return d.add(NODE_BLOCK, m, n, ch, BF_TYPE_BLOCK_COO)
    ch = [(_gen(rng, d, vals, rs[i], cs[j], depth - 1, cplx, coo), int(ro[i]), int(co[j]))
          for i in range(len(rs)) for j in range(len(cs))]
    return d.add(NODE_BLOCK, m, n, ch, BF_TYPE_BLOCK_DENSE)


def random_operand(rng, depth=3, size_hint=80, cplx=False, m=None, n=None, coo=True):
    """coo=False: no BfMatBlockCoo nodes (the one container the reference cannot transpose: its Transpose slot is NULL)."""
    d = Desc(dtype=0 if cplx else 1)
    vals = {}
    m = m or int(rng.integers(size_hint // 2, size_hint * 2))
    n = n or int(rng.integers(size_hint // 2, size_hint * 2))
    d.root = _gen(rng, d, vals, m, n, depth, cplx, coo)
    return d, vals


def random_real_operand(rng, depth=3, size_hint=80):
    return random_operand(rng, depth, size_hint, cplx=False)


def densify(d, vals, node):
    k = d.kind[node]
    m, n = d.rows[node], d.cols[node]
    if k == NODE_DENSE:
        return np.asarray(vals[node])
    if k == NODE_IDENTITY:
        return np.eye(m)
    if k == NODE_PRODUCT:
        out = None
        for c, _, _ in d.children[node]:
            a = densify(d, vals, c)
            out = a if out is None else out @ a
        return out
    dt = np.complex128 if d.dtype == 0 else np.float64
    out = np.zeros((m, n), dtype=dt)
    for c, r0, c0 in d.children[node]:
        a = densify(d, vals, c)
        out[r0:r0 + a.shape[0], c0:c0 + a.shape[1]] += a
    return out


def long_contraction_operand(rng, dtype):
    """BlockDiag(wide 40 x 40000 leaf, tall 37000 x 21 leaf, block column of 90 leaves over 33 columns): every
    way a row group's contraction gets long, forward or transposed.  Returns (desc, vals, dense-transpose fn)."""
    from butterfly_amd import helm2_structure as hs
    cplx = dtype == 0

    def val(m, n):
        return rng.standard_normal((m, n)) + (1j * rng.standard_normal((m, n)) if cplx else 0)
    d = hs.Desc(dtype=dtype)
    vals = {}
    wide = d.add(hs.NODE_DENSE, 40, 40000); vals[wide] = val(40, 40000)
    tall = d.add(hs.NODE_DENSE, 37000, 21); vals[tall] = val(37000, 21)
    col, ch, r0 = [], [], 0
    for i in range(90):
        b = d.add(hs.NODE_DENSE, 500 + (i % 7), 33); vals[b] = val(500 + (i % 7), 33)
        col.append(b); ch.append((b, r0, 0)); r0 += d.rows[b]
    colnode = d.add(hs.NODE_BLOCK, r0, 33, ch, hs.BF_TYPE_BLOCK_DENSE)
    d.root = d.add(hs.NODE_BLOCK, 40 + 37000 + r0, 40000 + 21 + 33, [(wide, 0, 0), (tall, 40, 40000), (colnode, 37040, 40021)], hs.BF_TYPE_BLOCK_DIAG)

    def apply_t(v):
        out = np.zeros(d.cols[d.root], dtype=np.result_type(v.dtype, vals[wide].dtype))
        out[:40000] = vals[wide].T @ v[:40]
        out[40000:40021] = vals[tall].T @ v[40:37040]
        p = 37040
        for b in col:
            out[40021:] += vals[b].T @ v[p:p + d.rows[b]]; p += d.rows[b]
        return out
    return d, vals, apply_t, val


def few_row_operand(rng):
    """What a streamed butterfly ends in: row nodes of 1 - 8 rows whose leaves are hundreds to thousands of columns wide,
    next to ordinary leaves in the same block column, with Identity terms on the few-row groups.  Real (f64) operand.
    Returns (desc, vals, dense matrix)."""
    from butterfly_amd import helm2_structure as hs
    d = hs.Desc(dtype=1)
    vals = {}
    widths = [700, 130, 2900]                      # three block columns
    co = np.concatenate([[0], np.cumsum(widths)])
    heights = [5, 1, 8, 40, 3, 7, 64, 2, 6, 4]     # block rows: few-row ones (row-major) and ordinary ones
    ro = np.concatenate([[0], np.cumsum(heights)])
    m, n = int(ro[-1]), int(co[-1])
    dense = np.zeros((m, n))
    ch = []
    for i, h in enumerate(heights):
        for j, w in enumerate(widths):
            if (i + j) % 4 == 3:
                continue                            # a hole: BlockCoo-like sparsity
            a = rng.standard_normal((h, w)) / np.sqrt(w)
            leaf = d.add(hs.NODE_DENSE, h, w)
            vals[leaf] = a
            dense[ro[i]:ro[i + 1], co[j]:co[j] + w] += a
            ch.append((leaf, int(ro[i]), int(co[j])))
        if h <= 8:                                   # an Identity term on the same rows (pass-through of a previous level)
            ident = d.add(hs.NODE_IDENTITY, h, h)
            c0 = int(co[1]) + 3 * i
            dense[ro[i]:ro[i + 1], c0:c0 + h] += np.eye(h)
            ch.append((ident, int(ro[i]), c0))
    d.root = d.add(hs.NODE_BLOCK, m, n, ch, hs.BF_TYPE_BLOCK_COO)
    return d, vals, dense


def narrow_items_operand(rng):
    """The inner factors of a streamed butterfly in one BlockCoo: 40 block rows, most of 1 - 8 rows (two leaves of a few
    dozen columns plus an Identity term), every fifth of 20 - 90 rows, and 11 rows nobody writes.  Real (f64) operand.
    Returns (desc, vals, dense matrix)."""
    from butterfly_amd import helm2_structure as hs
    d = hs.Desc(dtype=1)
    vals, ch, r0 = {}, [], 0
    blocks = []
    for i in range(40):
        h = int(rng.integers(1, 9)) if i % 5 else int(rng.integers(20, 90))
        if i == 17:
            r0 += 11; blocks.append(np.zeros((11, 300)))     # rows nobody writes: a zero-fill item
        row = np.zeros((h, 300))
        for c0, w in ((3 * i, int(rng.integers(4, 50))), (150 + i, int(rng.integers(4, 60)))):
            a = rng.standard_normal((h, w)); leaf = d.add(hs.NODE_DENSE, h, w); vals[leaf] = a
            ch.append((leaf, r0, c0)); row[:, c0:c0 + w] += a
        if h <= 8:
            ident = d.add(hs.NODE_IDENTITY, h, h); ch.append((ident, r0, 280)); row[:, 280:280 + h] += np.eye(h)
        blocks.append(row); r0 += h
    dense = np.vstack(blocks)
    d.root = d.add(hs.NODE_BLOCK, dense.shape[0], 300, ch, hs.BF_TYPE_BLOCK_COO)
    return d, vals, dense


def few_row_column_operand(rng, leaves=70, width=900):
    """A block column of a streamed butterfly's W factor: `leaves` few-row leaves (1 - 8 rows, `width` columns) stacked
    on top of one another with two tall leaves among them -- in the transposed plan a chain of that many pieces per
    item (the workgroup-shared path), row-major and column-major pieces over the same outputs.  Real (f64) operand.
    Returns (desc, vals, dense matrix)."""
    from butterfly_amd import helm2_structure as hs
    d = hs.Desc(dtype=1)
    vals, ch, r0 = {}, [], 0
    blocks = []
    for i in range(leaves):
        h = 100 + i if i in (7, 41) else int(rng.integers(1, 9))
        a = rng.standard_normal((h, width)) / np.sqrt(width)
        leaf = d.add(hs.NODE_DENSE, h, width); vals[leaf] = a
        ch.append((leaf, r0, 0)); blocks.append(a); r0 += h
    d.root = d.add(hs.NODE_BLOCK, r0, width, ch, hs.BF_TYPE_BLOCK_DENSE)
    return d, vals, np.vstack(blocks)
