"""Generator of tests/golden/streamer_lbo_stats.json: node statistics of streamed butterfly
factorizations computed with REAL truncated SVDs (oracle/streamer_values.py) on spherical harmonics, kept so
that the CPU suite can hold the value-free rank model of butterfly_amd/streamer_structure.py to them
without redoing minutes of SVDs.  These numbers come from this repository's own restatement (not from
the reference): they pin the model to the restatement, nothing more.

    python tests/golden/make_streamer_golden.py            # ~3 minutes: the two small cases
    python tests/golden/make_streamer_golden.py --large    # + N = 32768 and 65536 with 4096 columns (lmax 63): ~15 min and ~1 h of SVDs;
                                                           #   also writes every SVD's (rows, cols, row depth, column node, rank) to
                                                           #   streamer_svd_records_n<N>.npz for fitting rank models
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from butterfly_amd import streamer_structure as ss  # noqa: E402
from oracle import streamer_values as sv  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
path = os.path.join(HERE, "streamer_lbo_stats.json")
out = json.load(open(path)) if os.path.exists(path) else {}
cases = [(4096, 15, 3), (16384, 31, 4)]
if "--large" in sys.argv:
    cases += [(32768, 63, None), (65536, 63, None)]
for n, lmax, fd in cases:
    pts, phi, freqs = sv.sphere_lbo_problem(n, lmax)
    wmax = float(np.sqrt(lmax * (lmax + 1.0)) * 1.0001)
    if fd is None:
        fd = ss.Octree(pts, 1).max_depth - 3            # examples/covariance/lbo_cov.c:97-98
    rec = [] if n >= 32768 else None
    st, a_phi = sv.stream_columns(pts, phi, freqs, wmax, fd, record=rec)
    A = st.get_mat()
    gs = ss.graph_stats(A)
    x = np.random.default_rng(0).standard_normal(a_phi.shape[1])
    err = float(np.linalg.norm(sv.apply(A, x) - a_phi @ x) / np.linalg.norm(a_phi @ x))
    out[f"n{n}_lmax{lmax}_fd{fd}"] = dict(n=n, lmax=lmax, freq_depth=fd, wmax=wmax, stats=gs, num_bytes=A.num_bytes(),
                                          row_nodes=[len(f.row_nodes) for f in st.partial], num_w=[len(f.W) for f in st.partial],
                                          rel_err_vs_dense=err,
                                          # leaf bytes of every factor [Psi, W0, W1, ...] of every product: the per-stage byte shares
                                          factor_leaf_bytes=[[ss.graph_stats(f)["leafBytes"] for f in p.factors] for p in A.blocks],
                                          streamer=st.stats)
    if rec is not None:
        np.savez_compressed(os.path.join(HERE, f"streamer_svd_records_n{n}.npz"), records=np.asarray(rec, dtype=np.int64),
                            columns=np.array(["rows", "cols", "row_depth", "col_node", "rank"]))
    print(n, lmax, fd, gs, err)
    json.dump(out, open(path, "w"), indent=1)
