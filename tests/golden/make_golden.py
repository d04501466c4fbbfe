#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

No part of the reference is executed here (it cannot be built in this image
without stand-in BLAS headers).  Known answers come from the *definition* of
the operator -- the dense single-layer Helmholtz kernel matrix (i/4) H0(k r)
applied to the same vector -- which is the acceptance check the reference's
own example uses (examples/simple/bf_all_blocks.c:149-153,
examples/simple/bf_one_block.c:262-280).  Operands are built by
oracle/helm2_build.py (a restatement of the reference's builder).

  helm2_one_block_n2048_k128.npz   one butterfly (bf_one_block-style pair of
      level-2 nodes): descriptor, factor values, x (the reference's seeded
      PRNG stream, src/rand.c), y_dense = K[tgt, src] @ x, y_oracle
  helm2_multilevel_n2048_k128_vectors.npz   whole HODBF operator at N = 2048,
      k = 128 (BASELINE.json configs[0] scale, with butterflies): vectors only
      (x, y_dense, y_oracle); tests rebuild the operand with the builder
  real_nested_small.npz            random real (f64) operand exercising
      Identity leaves, nested products inside blocks, ragged sizes: x, y_dense
      (y_dense from densifying the expression with numpy)

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from butterfly_amd import helm2_structure as hs  # noqa: E402
from oracle import bfref, helm2_build as hb  # noqa: E402
from fixtures import save_fixture  # noqa: E402
import randgraph  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def one_block():
    n, k = 2048, 128
    pts = hs.circle_points(n)
    root, perm = hs.build_quadtree(pts)
    lv2 = hs.levels_below(root)[2].nodes
    # first separated level-2 pair that is butterfliable
    pick = None
    for ti, t in enumerate(lv2):
        for si, s in enumerate(lv2):
            if t.npts * s.npts >= hs.MAX_DENSE_MATRIX_SIZE and hs.separated(s, t) and hs.prepare(k, s, t)[0] >= 3:
                pick = (si, ti)
                break
        if pick:
            break
    # paths from the root to those level-2 nodes
    def path_of(node):
        for a, c1 in enumerate(root.children):
            for b, c2 in enumerate(c1.children):
                if c2 is node:
                    return (a, b)
    sp, tp_ = path_of(lv2[pick[0]]), path_of(lv2[pick[1]])
    desc, root2, perm2, sn, tn = hs.single_product_structure(pts, k, sp, tp_, recipes=True)
    tree_pts = pts[perm2]
    vals = hb.leaf_values(desc, k, tree_pts)
    x = hb.complex_randn(sn.npts, 0)
    kd = hb.kernel_matrix(k, tree_pts[sn.i0:sn.i1], tree_pts[tn.i0:tn.i1])
    y_dense = kd @ x
    y_or = bfref.mat_mul(bfref.from_desc(desc, vals), x)
    err = np.linalg.norm(y_or - y_dense) / np.linalg.norm(y_dense)
    print(f"one_block: src={sp} tgt={tp_} factors={desc.meta['num_factors']} leaves={len(vals)} "
          f"bytes={sum(v.nbytes for v in vals.values())} rel err vs dense {err:.2e}")
    save_fixture(os.path.join(HERE, "helm2_one_block_n2048_k128.npz"), desc, vals, x=x, y_dense=y_dense, y_oracle=y_or,
                 k=np.float64(k), src_path=np.asarray(sp), tgt_path=np.asarray(tp_))


def multilevel():
    n, k = 2048, 128
    pts = hs.circle_points(n)
    desc, root, perm = hs.helm2_multilevel_structure(pts, k, recipes=True)
    tp = pts[perm]
    vals = hb.leaf_values(desc, k, tp)
    x = hb.complex_randn(n, 0)
    y_dense = hb.kernel_matrix(k, tp, tp) @ x
    y_or = bfref.mat_mul(bfref.from_desc(desc, vals), x)
    print(f"multilevel: N={n} k={k} {desc.meta['stats']} leafMB={desc.leaf_elems() * 16 / 1e6:.1f} "
          f"rel err vs dense {np.linalg.norm(y_or - y_dense) / np.linalg.norm(y_dense):.2e}")
    np.savez_compressed(os.path.join(HERE, "helm2_multilevel_n2048_k128_vectors.npz"), n=n, k=np.float64(k), x=x,
                        y_dense=y_dense, y_oracle=y_or, leaf_elems=desc.leaf_elems(),
                        num_products=sum(desc.meta["stats"]["products"].values()))


def real_nested():
    seed = 20261003
    while True:   # first seed whose graph exercises every node kind, nested
        rng = np.random.default_rng(seed)
        desc, vals = randgraph.random_real_operand(rng, depth=4, size_hint=160)
        kinds = np.bincount(desc.kind, minlength=4)
        if desc.num_nodes >= 40 and kinds[1] >= 2 and kinds[3] >= 3:
            break
        seed += 1
    n = desc.cols[desc.root]
    x = rng.standard_normal(n)
    y_dense = randgraph.densify(desc, vals, desc.root) @ x
    y_or = bfref.mat_mul_vec(bfref.from_desc(desc, vals), x)
    kinds = np.bincount(desc.kind, minlength=4)
    print(f"real_nested: {desc.rows[desc.root]}x{n} nodes={desc.num_nodes} kinds(dense,identity,block,product)={kinds.tolist()} rel err {np.linalg.norm(y_or - y_dense) / np.linalg.norm(y_dense):.2e}")
    save_fixture(os.path.join(HERE, "real_nested_small.npz"), desc, vals, x=x, y_dense=y_dense, y_oracle=y_or)


if __name__ == "__main__":
    one_block()
    multilevel()
    real_nested()
