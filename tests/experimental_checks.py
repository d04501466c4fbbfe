"""Bit-identity checks of the EXPERIMENTAL executors (butterfly_amd/csrc/bfhip_experimental.hip, bfhip_persist.hip): not
collected with the suite (the product library does not contain them) -- tests/test_gpu_parity.py::
test_experimental_executors_in_their_own_build runs this file in a child process with BFHIP_LIB_PATH pointing at
libbfhip_exp.so (`make -C butterfly_amd/csrc experimental`)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-12
pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b)))


def test_this_run_loads_the_experimental_build():
    from butterfly_amd import _capi
    assert os.path.basename(_capi.LIB_PATH) == "libbfhip_exp.so"


def test_stage_profile_of_the_one_launch_executor():
    import torch
    from butterfly_amd import _capi, helm2_structure as hs
    from butterfly_amd.operator import HipOperator
    desc, _ = hs.native_multilevel_structure(hs.circle_points(4096), 100.0)
    x = torch.randn(4096, dtype=torch.complex128, device="cuda")
    op = HipOperator.from_desc(desc, None, seed=3, flags=_capi.FLAG_PROFILE | _capi.FLAG_FLOW)
    assert op.flow_status()[0]
    for _ in range(3):
        op.apply_device(x)
    ms, launches, nbytes = op.stage_profile()
    # the whole plan is one dependency-driven launch: reported under stage 0, with the bytes of all stages
    assert launches[0] == 3 and ms[0] > 0 and not launches[1:].any() and not nbytes[1:].any()
    op.close()


@pytest.mark.parametrize("n,k", [(4096, 100.0), (16384, 1024.0), (65536, 4096.0)])
def test_one_dependency_driven_launch_equals_the_staged_launches(n, k):
    """BFHIP_FLAG_FLOW (experimental; complex128 operators at one right-hand side; two and more run the matrix-core kernel, staged): the whole plan as ONE persistent
    launch whose items wait for the intermediate vectors they read (bfFlowKernelC128) instead of one launch per stage.
    Same items, same arithmetic: bit-identical to the staged launches and to itself over hundreds of applies (the
    counters run on from apply to apply), equal to the oracle, and no wait ever gives up."""
    import torch
    from butterfly_amd import _capi, helm2_structure as hs
    from butterfly_amd.operator import HipOperator
    from oracle import bfref
    desc, perm = hs.native_multilevel_structure(hs.circle_points(n), k)
    flow = HipOperator.from_desc(desc, None, seed=9, max_rhs=2, flags=_capi.FLAG_PROFILE | _capi.FLAG_FLOW)
    staged = HipOperator.from_desc(desc, None, seed=9, max_rhs=2)
    assert flow.flow_status() == (True, False) and staged.flow_status() == (False, False)
    rng = np.random.default_rng(n)
    for nrhs in (1, 2):
        shape = (n,) if nrhs == 1 else (n, nrhs)
        x = torch.from_numpy((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2)).cuda()
        want = staged.apply_device(x).clone()
        got = flow.apply_device(x).clone()
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        if n <= 16384:
            y_ref = bfref.mat_mul(bfref.from_desc(desc, None, seed=9), x.cpu().numpy().reshape(n, nrhs)).reshape(shape)
            assert rel(got.cpu().numpy(), y_ref) <= TOL
        y = torch.empty_like(got)
        for _ in range(300):                       # back to back on one stream: apply e waits for e x writers
            flow.apply_device(x, y)
        torch.cuda.synchronize()
        assert torch.equal(y, want)
        # a different x every apply: an intermediate left over from the apply before (a stale line, a read that overtook
        # its poll) cannot hide behind equal inputs
        ys, yf = torch.empty_like(got), torch.empty_like(got)
        for i in range(40):
            xi = torch.roll(x, i + 1, 0) * (1.0 + 0.125 * i)
            staged.apply_device(xi, ys)
            flow.apply_device(xi, yf)
            assert torch.equal(yf, ys), i
    # three right-hand sides and more go to the matrix-core kernel, stage by stage, on the same operator
    x3 = torch.from_numpy((rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))) / np.sqrt(2)).cuda()
    assert torch.equal(flow.apply_device(x3), staged.apply_device(x3))
    ms, launches, nbytes = flow.stage_profile()
    assert launches[0] > 0 and ms[0] > 0              # profiled applies of the one-launch path are reported under stage 0
    assert flow.flow_status() == (True, False)
    flow.close(); staged.close()


_PERSISTENT_CHILD = r"""
import hashlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["BF_REPO"])
from butterfly_amd import helm2_structure as hs
from butterfly_amd.operator import HipOperator
n, k = 65536, 4096.0
desc, perm = hs.native_multilevel_structure(hs.circle_points(n), k)
op = HipOperator.from_desc(desc, None, seed=9, max_rhs=2)
rng = np.random.default_rng(5)
out = {}
for nrhs in (1, 2):
    shape = (n,) if nrhs == 1 else (n, nrhs)
    x = torch.from_numpy((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2)).cuda()
    y = op.apply_device(x).clone()
    for i in range(50):                                    # the ticket counters must come back to zero after every launch
        assert torch.equal(op.apply_device(x), y), i
    z = op.apply_device(torch.roll(x, 7, 0) * 1.5).clone()
    torch.cuda.synchronize()
    out[str(nrhs)] = [hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest(), hashlib.sha256(z.cpu().numpy().tobytes()).hexdigest()]
tl = os.environ.get("BFHIP_TIMELINE_FILE")
if tl:
    head = [l for l in open(tl).read().split("\n") if l.startswith("launch")]
    out["timeline_launches"] = len(head)
    out["timeline_items"] = sum(int(l.split()[1]) for l in head)
print("RESULT " + json.dumps(out))
"""


def _run_child(env_extra):
    import json
    import subprocess
    import sys
    env = dict(os.environ, BF_REPO=ROOT, **env_extra)
    r = subprocess.run([sys.executable, "-c", _PERSISTENT_CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.split("\n") if l.startswith("RESULT ")][-1][7:])


def test_persistent_ticket_launch_and_item_timeline_are_bit_identical_to_the_plain_launches(tmp_path):
    """BFHIP_PERSISTENT=1 (experimental): stages with more items than wavefront slots run as one persistent grid whose
    wavefronts draw pooled tickets (bfStageKernelC128P); BFHIP_TIMELINE_FILE: the diagnostic launch that records every
    item's start and end.  Both read the environment once per process, hence the child processes.  Same items, same
    arithmetic: the results are bit-identical to the plain launches', apply after apply (the last draw of a launch
    resets its pool)."""
    plain = _run_child({"BFHIP_PERSISTENT": "0"})
    persistent = _run_child({"BFHIP_PERSISTENT": "1"})
    assert persistent["1"] == plain["1"] and persistent["2"] == plain["2"]
    tl = str(tmp_path / "items.timeline")
    for mode in ("0", "1"):
        if os.path.exists(tl):
            os.remove(tl)
        traced = _run_child({"BFHIP_PERSISTENT": mode, "BFHIP_TIMELINE_FILE": tl})
        assert traced["1"] == plain["1"] and traced["2"] == plain["2"]
        assert traced["timeline_launches"] > 0 and traced["timeline_items"] > 100000
