"""Save / load small operands (flat descriptor + leaf values + vectors) as .npz
fixtures.  A fixture is data only: shapes, values, inputs, expected outputs."""
import numpy as np

from butterfly_amd.helm2_structure import Desc


def save_fixture(path, desc, vals, **vectors):
    a = desc.arrays()
    nodes = sorted(vals)
    offs = np.zeros(len(nodes) + 1, dtype=np.int64)
    for i, nd in enumerate(nodes):
        offs[i + 1] = offs[i] + vals[nd].size
    flat = np.concatenate([np.ravel(vals[nd]) for nd in nodes]) if nodes else np.zeros(0)
    np.savez_compressed(path, dtype=desc.dtype, root=desc.root, leaf_nodes=np.asarray(nodes, dtype=np.int64),
                        leaf_offsets=offs, leaf_values=flat, **a, **vectors)


def load_fixture(path):
    z = np.load(path)
    d = Desc(dtype=int(z["dtype"]))
    n = len(z["kind"])
    cb = z["childBegin"].astype(np.int64)
    for i in range(n):
        ch = [(int(z["childNode"][c]), int(z["childRow0"][c]), int(z["childCol0"][c])) for c in range(cb[i], cb[i + 1])]
        d.add(int(z["kind"][i]), int(z["rows"][i]), int(z["cols"][i]), ch, int(z["blockKind"][i]))
    d.root = int(z["root"])
    vals = {}
    offs = z["leaf_offsets"]
    for i, nd in enumerate(z["leaf_nodes"]):
        nd = int(nd)
        vals[nd] = np.ascontiguousarray(z["leaf_values"][offs[i]:offs[i + 1]].reshape(d.rows[nd], d.cols[nd]))
    extra = {k: z[k] for k in z.files if k not in ("dtype", "root", "leaf_nodes", "leaf_offsets", "leaf_values", "kind", "rows",
                                                    "cols", "childBegin", "childNode", "childRow0", "childCol0", "blockKind")}
    return d, vals, extra
