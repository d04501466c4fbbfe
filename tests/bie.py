"""Test infrastructure: a second-kind-like system matrix  A = I + alpha * S  built
from a fac_helm2 operand, the shape the reference's BIE driver solves with GMRES
(examples/simple/helm2_bie.c:109-121 adds 1/2 I to the scaled layer potential).
The identity enters as BfMatIdentity leaves next to the diagonal top-level
blocks (a Block sums the children that share its cell)."""
import numpy as np

from butterfly_amd import helm2_structure as hs


def scale_operator(desc, vals, node, alpha):
    """Multiply the operator under `node` by the scalar alpha, in place."""
    kind = desc.kind[node]
    if kind == hs.NODE_DENSE:
        vals[node] = vals[node] * alpha
    elif kind == hs.NODE_BLOCK:
        for c, _, _ in desc.children[node]:
            scale_operator(desc, vals, c, alpha)
    elif kind == hs.NODE_PRODUCT:
        scale_operator(desc, vals, desc.children[node][0][0], alpha)    # last-applied factor only
    else:
        raise ValueError("cannot scale an identity leaf")


def identity_plus(desc, vals, alpha):
    """Turn the multilevel operator S (root = Block grid) into I + alpha * S.
    Returns the new root node id."""
    scale_operator(desc, vals, desc.root, alpha)
    ch = []
    for (c, r0, c0) in desc.children[desc.root]:
        if r0 == c0 and desc.rows[c] == desc.cols[c]:
            m = desc.rows[c]
            eye = desc.add(hs.NODE_IDENTITY, m, m)
            wrapped = desc.add(hs.NODE_BLOCK, m, m, [(c, 0, 0), (eye, 0, 0)], hs.BF_TYPE_BLOCK_COO)
            ch.append((wrapped, r0, c0))
        else:
            ch.append((c, r0, c0))
    return desc.add(hs.NODE_BLOCK, desc.rows[desc.root], desc.cols[desc.root], ch, hs.BF_TYPE_BLOCK_DENSE)


def second_kind_case(n, k):
    """(desc, root, vals, dense matrix) of A = I + (4 pi / n) * S on the unit circle."""
    from conftest import helm2_case
    from oracle import helm2_build as hb
    desc, tp, vals = helm2_case(n, k)
    vals = {nd: v.copy() for nd, v in vals.items()}
    alpha = 2 * (2 * np.pi / n)
    root = identity_plus(desc, vals, alpha)
    dense = np.eye(n) + alpha * hb.kernel_matrix(k, tp, tp)
    return desc, root, vals, dense
