import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Both shared libraries are built in-tree; build them if a fresh checkout lacks them."""
    lib = os.path.join(ROOT, "butterfly_amd", "csrc", "libbfhip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "butterfly_amd", "csrc"), "-s"])
    ref = os.path.join(ROOT, "oracle", "libbfref.so")
    if not os.path.exists(ref):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    yield


def helm2_case(n, k, recipes=True, exact_sift=False):
    """(desc, tree_points, leaf values) of a real fac_helm2-style factorization."""
    from butterfly_amd import helm2_structure as hs
    from oracle import helm2_build as hb
    pts = hs.circle_points(n)
    desc, root, perm = hs.helm2_multilevel_structure(pts, k, recipes=recipes, exact_sift=exact_sift)
    tp = pts[perm]
    vals = hb.leaf_values(desc, k, tp) if recipes else None
    return desc, tp, vals


_CASE_CACHE = {}


@pytest.fixture(scope="session")
def helm2_cases():
    def get(n, k):
        key = (n, k)
        if key not in _CASE_CACHE:
            _CASE_CACHE[key] = helm2_case(n, k)
        return _CASE_CACHE[key]
    return get
