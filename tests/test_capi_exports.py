"""The C-ABI library loads without a GPU and exports every function that
include/*.h declare (no compute calls here)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for hdr in ("bfhip.h", "bfhip_build.h"):
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(bfhip[A-Z]\w*)\s*\(", src))
    return sorted(names)


def test_every_declared_symbol_is_exported():
    from butterfly_amd import _capi
    lib = _capi.load()
    names = declared_functions()
    assert len(names) >= 22 and "bfhipBuildHelm2" in names
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


def test_product_holds_no_experimental_executor():
    """The one-launch ("flow") executor, the persistent ticket launch and the timeline instantiation -- spin-waiting
    kernels and getenv-switched code paths -- are built into libbfhip_exp.so only (make experimental); the product
    library neither defines their symbols nor reads their environment variables, and refuses BFHIP_FLAG_FLOW."""
    import subprocess
    from butterfly_amd import _capi
    prod = os.path.join(ROOT, "butterfly_amd", "csrc", "libbfhip.so")
    syms = subprocess.check_output(["nm", "-D", "--defined-only", prod], text=True)
    for name in ("bfFlowKernelC128", "bfdevLaunchFlow", "bfdevFlowGrid", "bfdevPersistentGrid", "bfdevLaunchPersistC128", "bfStageKernelC128P",
                 "bfStageKernelC128Timeline", "bfdevLaunchStageExperimental"):
        assert name not in syms, name
    blob = open(prod, "rb").read()
    for env in (b"BFHIP_FLOW", b"BFHIP_PERSISTENT", b"BFHIP_TIMELINE_FILE", b"BFHIP_FLOW_SPIN"):
        assert env not in blob, env
    exp = os.path.join(ROOT, "butterfly_amd", "csrc", "libbfhip_exp.so")
    if os.path.exists(exp):
        esyms = subprocess.check_output(["nm", "-D", "--defined-only", exp], text=True)
        assert "bfdevLaunchFlow" in esyms and "bfdevPersistentGrid" in esyms
    if os.path.basename(_capi.LIB_PATH) == "libbfhip.so":
        import numpy as np
        from butterfly_amd import helm2_structure as hs
        from butterfly_amd.operator import HipOperator
        import pytest
        desc, _ = hs.native_multilevel_structure(hs.circle_points(1024), 64.0)
        with pytest.raises(_capi.BfhipError) as ei:
            HipOperator.from_desc(desc, None, flags=_capi.FLAG_PLAN_ONLY | _capi.FLAG_FLOW)
        assert ei.value.code == 3


def test_rhs_block_kernels_have_no_scratch_and_keep_their_occupancy(tmp_path):
    """bfStageKernelC128Mfma issues its fragment loads as asm statements and counts them by hand (s_waitcnt vmcnt(n)): a
    register spill inside its k-loop would be a scratch load the count does not know of -- silently wrong results.  The
    compiled kernel must use no scratch, spill nothing and fit two wavefronts per SIMD (<= 256 VGPRs)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    out = tmp_path / "dev.s"
    subprocess.check_call([hipcc, "-O3", "-g", "-fPIC", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out),
                           os.path.join(ROOT, "butterfly_amd", "csrc", "bfhip_device.hip")], stderr=subprocess.DEVNULL)
    txt = open(out).read()
    # the 4-tile kernel (2 wavefronts per SIMD) and its 2- and 1-tile instantiations (3 and 5)
    for sym, vmax, mfmas in (("_Z21bfStageKernelC128Mfma11StageParams", 256, 96), ("_Z22bfStageKernelC128Mfma211StageParams", 168, 48),
                             ("_Z22bfStageKernelC128Mfma111StageParams", 96, 18)):
        i = txt.index(".name:           " + sym)
        meta = txt[i:i + 800]
        get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", meta).group(1))
        assert get("private_segment_fixed_size") == 0 and get("vgpr_spill_count") == 0 and get("sgpr_spill_count") == 0, sym
        assert get("vgpr_count") <= vmax, (sym, get("vgpr_count"))
        body = txt[txt.index(sym + ":"):]
        body = body[:body.index("s_endpgm")]
        assert "scratch_" not in body and body.count("v_mfma_f64_16x16x4_f64") >= mfmas, sym


def test_hand_counted_waits_of_the_rhs_block_kernels_hold_in_the_compiled_code(tmp_path):
    """The k-loops of the RHS-block kernels issue their fragment loads as asm statements and wait for them by count
    (bfhip_stage_mfma.h).  hipcc models neither: (a) a vector-memory instruction of its own inside a k-loop would make every
    vmcnt(n) there one short; (b) it may read, copy or reuse a fragment register before the wait that retires its load (it did, in
    round 5: a v_mov between a ds_read and its s_waitcnt in some instantiations of the LDS-ring loop).  Audit of the compiled
    code (tests/asm_audit.py replays the in-order counters over the instruction stream): in every innermost MFMA loop of all six
    kernels (three tile counts x Gauss / exact) and of the LDS-ring build the only vector-memory instructions are the expected
    buffer_load_dwordx4, and inside those loops no instruction touches a register a pending load may still write."""
    import shutil
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import asm_audit
    # the audit itself, on the pattern it exists for: a copy of a fragment between its ds_read and the wait
    fake = ("k:\n.LBB0_1:\n s_waitcnt vmcnt(0) lgkmcnt(0)\n.LBB0_2:\n ds_read_b128 v[104:107], v1 offset:0\n v_mov_b64_e32 v[126:127], v[106:107]\n"
            " s_waitcnt lgkmcnt(0)\n v_mfma_f64_16x16x4_f64 v[2:9], v[104:105], v[126:127], v[2:9]\n s_cbranch_scc1 .LBB0_2\n s_endpgm\n")
    assert [p[1] for p in asm_audit.audit(fake, "k")] == ["v_mov_b64_e32 v[126:127], v[106:107]"] * 3
    assert not asm_audit.audit(fake.replace(" v_mov_b64_e32 v[126:127], v[106:107]\n s_waitcnt lgkmcnt(0)\n", " s_waitcnt lgkmcnt(0)\n v_mov_b64_e32 v[126:127], v[106:107]\n"), "k")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    for tag, defs in (("product", []), ("ring", ["-DBF_MF_DMA=1"])):
        out = tmp_path / f"dev_{tag}.s"
        subprocess.check_call([hipcc, "-O3", "-g", "-fPIC", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", *defs, "-o", str(out),
                               os.path.join(ROOT, "butterfly_amd", "csrc", "bfhip_device.hip")], stderr=subprocess.DEVNULL)
        txt = open(out).read()
        syms = re.findall(r"^(_Z\d+bfStageKernelC128Mfma\w*11StageParams):", txt, flags=re.M)
        assert len(syms) == 6, syms
        for sym in syms:
            loops = asm_audit.loop_vmem(txt, sym)
            assert loops, sym
            for mfmas, vm in loops:
                assert mfmas % 2 == 0 and vm and set(vm) <= {"buffer_load_dwordx4", "buffer_load_dwordx4 lds"}, (sym, mfmas, vm)
            bad = asm_audit.audit(txt, sym)
            assert not bad, (tag, sym, bad[:4])


def test_builder_structs_match_the_header_and_arguments_are_checked(tmp_path):
    """ctypes / numpy mirrors of include/bfhip_build.h have the C sizes; bad
    arguments are refused before any device is touched."""
    import subprocess
    import numpy as np
    from butterfly_amd import _capi
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "bfhip_build.h"\nint main(void){printf("%zu %zu %zu %zu\\n", sizeof(BfhipPointSet), '
                   'sizeof(BfhipHelm2Recipe), sizeof(BfhipHelm2Problem), sizeof(BfhipBuildStats));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert sizes == [_capi.POINT_SET_DTYPE.itemsize, _capi.RECIPE_DTYPE.itemsize, C.sizeof(_capi.BfhipHelm2Problem),
                     C.sizeof(_capi.BfhipBuildStats)]
    rec = _capi.recipe_array({5: ("kernel", ("node", 3, 9), ("circle", 0.5, -0.25, 2.0, 17)),
                              2: ("reexp", ("circle", 0, 0, 1, 4), ("circle", 0, 0, 2, 6), ("circle", 3, 0, 1, 6))})
    assert list(rec["node"]) == [2, 5] and list(rec["kind"]) == [_capi.LEAF_REEXP, _capi.LEAF_KERNEL]
    assert rec[1]["src"]["first"] == 3 and rec[1]["src"]["count"] == 6 and rec[1]["tgt"]["r"] == 2.0 and rec[0]["equiv"]["count"] == 6
    lib = _capi.load()
    pts = np.zeros((4, 2))
    out = np.zeros(64, dtype=complex)
    prob = _capi.Helm2Problem(pts, 1.0, {0: ("kernel", ("node", 0, 9), ("node", 0, 2))})          # 9 points of 4
    assert lib.bfhipHelm2BuildLeaf(prob.byref(), 0, -1, out.ctypes.data) == 1
    assert b"exceed numPoints" in lib.bfhipLastErrorMessage()
    prob = _capi.Helm2Problem(pts, 1.0, {0: ("reexp", ("node", 0, 2), ("circle", 0, 0, 1, 8), ("circle", 3, 0, 1, 4))})
    assert lib.bfhipHelm2BuildLeaf(prob.byref(), 0, -1, out.ctypes.data) == 3                      # fewer check points than unknowns
    prob = _capi.Helm2Problem(pts, 0.0, {0: ("kernel", ("node", 0, 2), ("node", 0, 2))})
    assert lib.bfhipHelm2BuildLeaf(prob.byref(), 0, -1, out.ctypes.data) == 1                      # wavenumber
    assert lib.bfhipHelm2DenseApply(None, -1, out.ctypes.data, out.ctypes.data) == 1
    prob = _capi.Helm2Problem(pts, 1.0, {0: ("kernel", ("node", 0, 2), ("node", 0, 2))}, layer_pot="Sp")   # S' without normals
    assert lib.bfhipHelm2BuildLeaf(prob.byref(), 0, -1, out.ctypes.data) == 1 and b"normals" in lib.bfhipLastErrorMessage()
    prob = _capi.Helm2Problem(pts, 1.0, {0: ("kernel", ("node", 0, 2), ("circle", 0, 0, 1, 4))}, layer_pot="Sp", normals=pts)
    assert lib.bfhipHelm2BuildLeaf(prob.byref(), 0, -1, out.ctypes.data) == 1                      # S' leaf with circle targets
    prob = _capi.Helm2Problem(pts, 1.0, {0: ("kernel", ("node", 0, 2), ("node", 0, 2))}, layer_pot=4)         # PV_NORMAL_DERIV_DOUBLE:
    assert lib.bfhipHelm2BuildLeaf(prob.byref(), 0, -1, out.ctypes.data) == 3                      # no kernel in the reference either
    prob = _capi.Helm2Problem(pts, 1.0, {0: ("kernel", ("node", 0, 2), ("node", 0, 2))}, layer_pot="D")       # source normals missing
    assert lib.bfhipHelm2BuildLeaf(prob.byref(), 0, -1, out.ctypes.data) == 1
    prob = _capi.Helm2Problem(pts, 1.0, {0: ("kernel", ("node", 0, 2), ("node", 0, 2))}, kr_order=4, orig_index=np.arange(4))
    assert lib.bfhipHelm2BuildLeaf(prob.byref(), 0, -1, out.ctypes.data) == 1                      # KR order must be 2, 6 or 10
