"""The structure generator (butterfly_amd/helm2_structure.py) must reproduce the
reference builder's block layout exactly: compared with the statistics the
survey recorded from the real compiled reference (tests/golden/survey_probe_stats.json)."""
import json
import os

import numpy as np
import pytest

from butterfly_amd import helm2_structure as hs

HERE = os.path.dirname(os.path.abspath(__file__))
STATS = json.load(open(os.path.join(HERE, "golden", "survey_probe_stats.json")))


def factor_stats(desc):
    kind = np.asarray(desc.kind)
    rows = np.asarray(desc.rows)
    cols = np.asarray(desc.cols)
    agg = {}
    inter = 0
    for node in np.nonzero(kind == hs.NODE_PRODUCT)[0]:
        fs = [c for c, _, _ in desc.children[node]]
        for fi, f in enumerate(fs):
            ch = [c for c, _, _ in desc.children[f]]
            a = agg.setdefault((len(fs), fi), dict(blocks=0, bytes=0, m_min=10**9, m_max=0, n_min=10**9, n_max=0))
            a["blocks"] += len(ch)
            m, n = rows[ch], cols[ch]
            a["bytes"] += int((m * n).sum()) * 16
            a["m_min"] = min(a["m_min"], int(m.min())); a["m_max"] = max(a["m_max"], int(m.max()))
            a["n_min"] = min(a["n_min"], int(n.min())); a["n_max"] = max(a["n_max"], int(n.max()))
            inter += int(rows[f])          # the probe counts every factor's output length
    return agg, inter


@pytest.mark.parametrize("case", [c for c in STATS["cases"] if c["factors"] and c["n"] <= 65536 and (c["n"], c["k"]) != (65536, 100)],
                         ids=lambda c: f"N{c['n']}_k{c['k']}")
def test_block_layout_matches_reference_probe(case):
    desc, root, perm = hs.helm2_multilevel_structure(hs.circle_points(case["n"]), case["k"])
    st = desc.meta["stats"]
    assert sum(st["products"].values()) == case["products"]
    assert {str(k): v for k, v in st["products"].items()} == case["by_num_factors"]
    assert st["dense_leaves"] == case["dense_leaves"]
    assert st["block_dense"] == case["block_dense_nodes"]
    assert round(desc.leaf_elems() * 16 / 1e6, 2) == case["total_leaf_mb"]
    agg, inter = factor_stats(desc)
    assert len(agg) == len(case["factors"])
    for f in case["factors"]:
        a = agg[(f["num_factors"], f["factor"])]
        assert a["blocks"] == f["blocks"]
        assert round(a["bytes"] / 1e6, 2) == f["mb"]
        assert (a["m_min"], a["m_max"], a["n_min"], a["n_max"]) == (f["m_min"], f["m_max"], f["n_min"], f["n_max"])
    assert inter == case["intermediate_elems"]


def test_small_case_is_all_dense():
    """N=1024, k=100: 144 dense blocks, no butterflies below the 128^2 threshold
    (reference src/fac_helm2.c:20,888)."""
    desc, root, perm = hs.helm2_multilevel_structure(hs.circle_points(1024), 100)
    assert desc.meta["stats"] == {"dense_leaves": 144, "products": {}, "block_dense": 1}
    assert desc.leaf_elems() == 1024 * 1024


def test_exact_sift_and_fast_partition_agree_on_structure():
    pts = hs.circle_points(2048)
    d1, r1, p1 = hs.helm2_multilevel_structure(pts, 128, exact_sift=True)
    d2, r2, p2 = hs.helm2_multilevel_structure(pts, 128, exact_sift=False)
    assert d1.rows == d2.rows and d1.cols == d2.cols and d1.kind == d2.kind
    assert sorted(p1.tolist()) == list(range(2048)) and sorted(p2.tolist()) == list(range(2048))


def test_shard_desc_partitions_rows():
    desc, root, perm = hs.helm2_multilevel_structure(hs.circle_points(2048), 128)
    nrb = len(desc.meta["top_rows"])
    r1, m1 = hs.shard_desc(desc, range(0, nrb // 2))
    r2, m2 = hs.shard_desc(desc, range(nrb // 2, nrb))
    assert m1 + m2 == 2048
    assert desc.cols[r1] == desc.cols[r2] == 2048
