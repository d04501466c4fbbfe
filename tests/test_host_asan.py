"""The host C code of libbfhip (BfMat graph walker on oracle-built graphs, BfhipDesc walker, planner incl. the transposed plan and row shards,
native layout incl. two trees, plan inspection, error paths) under AddressSanitizer + UBSan +
LeakSanitizer, on the CPU: tests/native/asan_host.c replaces the device layer by stubs that abort,
so the run also proves that plan-only paths never touch a device."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "butterfly_amd", "csrc")
HOST = ["bfhip_ir.c", "bfhip_plan.c", "bfhip_api.c", "bfhip_gmres.c", "bfhip_build.c", "bfhip_layout.c", "bfhip_streamer_layout.c"]


def test_host_code_is_clean_under_sanitizers(tmp_path):
    exe = str(tmp_path / "asan_host")
    cmd = ["gcc", "-std=gnu11", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"), "-I", SRC,
           os.path.join(ROOT, "tests", "native", "asan_host.c"), os.path.join(ROOT, "oracle", "bfref.c")] + \
          [os.path.join(SRC, f) for f in HOST] + ["-lm", "-ldl", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    # (only a missing sanitizer RUNTIME is a reason to skip: a compile error of the harness must fail, and the command line itself holds the word)
    if b.returncode != 0 and re.search(r"cannot find -l(asan|ubsan)|libasan|libubsan", b.stderr) and "error:" not in b.stderr.replace("ld: error", ""):
        pytest.skip("no sanitizer runtime for gcc here")
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "host code clean" in p.stdout, (p.stdout + p.stderr)[-4000:]
    assert "runtime error" not in p.stderr and "ERROR: AddressSanitizer" not in p.stderr and "LeakSanitizer" not in p.stderr
