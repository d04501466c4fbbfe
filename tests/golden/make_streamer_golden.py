"""Generator of tests/golden/streamer_lbo_stats.json: node statistics of streamed butterfly
factorizations computed with REAL truncated SVDs (oracle/streamer_values.py) on spherical harmonics, kept so
that the CPU suite can hold the value-free rank model of butterfly_amd/streamer_structure.py to them
without redoing minutes of SVDs.  These numbers come from this repository's own restatement (not from
the reference): they pin the model to the restatement, nothing more.

    python tests/golden/make_streamer_golden.py        # ~3 minutes
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from butterfly_amd import streamer_structure as ss  # noqa: E402
from oracle import streamer_values as sv  # noqa: E402

out = {}
for n, lmax, fd in ((4096, 15, 3), (16384, 31, 4)):
    pts, phi, freqs = sv.sphere_lbo_problem(n, lmax)
    wmax = float(np.sqrt(lmax * (lmax + 1.0)) * 1.0001)
    st, a_phi = sv.stream_columns(pts, phi, freqs, wmax, fd)
    A = st.get_mat()
    gs = ss.graph_stats(A)
    x = np.random.default_rng(0).standard_normal(a_phi.shape[1])
    err = float(np.linalg.norm(sv.apply(A, x) - a_phi @ x) / np.linalg.norm(a_phi @ x))
    out[f"n{n}_lmax{lmax}_fd{fd}"] = dict(n=n, lmax=lmax, freq_depth=fd, wmax=wmax, stats=gs, num_bytes=A.num_bytes(),
                                          row_nodes=[len(f.row_nodes) for f in st.partial], num_w=[len(f.W) for f in st.partial],
                                          rel_err_vs_dense=err)
    print(n, lmax, fd, gs, err)
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "streamer_lbo_stats.json"), "w"), indent=1)
