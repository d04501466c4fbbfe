"""The C-ABI library loads without a GPU and exports every function that
include/bfhip.h declares (no compute calls here)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "bfhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bfhip[A-Z]\w*)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    from butterfly_amd import _capi
    lib = _capi.load()
    names = declared_functions()
    assert len(names) >= 18
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_error_strings_and_synthetic_stream():
    from butterfly_amd import _capi
    lib = _capi.load()
    assert lib.bfhipErrorString(0) == b"BF_ERROR_NONE"
    assert lib.bfhipErrorString(3) == b"BF_ERROR_NOT_IMPLEMENTED"
    assert lib.bfhipErrorString(8) == b"BF_ERROR_INCOMPATIBLE_SHAPES"
    vals = [lib.bfhipSyntheticValue(7, i, im) for i in range(1000) for im in (0, 1)]
    assert all(-1.0 <= v < 1.0 for v in vals)
    assert abs(sum(vals) / len(vals)) < 0.1
    assert len(set(vals)) == len(vals)
    assert lib.bfhipSyntheticValue(7, 5, 0) == lib.bfhipSyntheticValue(7, 5, 0)
    assert lib.bfhipSyntheticValue(7, 5, 0) != lib.bfhipSyntheticValue(8, 5, 0)


def test_product_does_not_link_the_oracle():
    """libbfhip.so must not depend on, or contain, the CPU oracle."""
    import subprocess
    from butterfly_amd import _capi
    out = subprocess.check_output(["ldd", _capi.LIB_PATH], text=True)
    assert "bfref" not in out
    syms = subprocess.check_output(["nm", "-D", "--defined-only", _capi.LIB_PATH], text=True)
    assert "bfMatMul" not in syms and "bfref" not in syms
