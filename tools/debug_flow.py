#!/usr/bin/env python3
"""Diagnostic: one flow-mode apply on a small operand next to the staged launches (BFHIP_FLOW_DEBUG=1 prints the counters)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the executors this tool drives live in the experimental build only (make -C butterfly_amd/csrc experimental)
os.environ.setdefault("BFHIP_LIB_PATH", os.path.join(ROOT, "butterfly_amd", "csrc", "libbfhip_exp.so"))
os.environ.setdefault("BFHIP_FLOW_DEBUG", "1")
os.environ.setdefault("BFHIP_FLOW_SPIN", "2000")
def say(*a):
    print(f"[{time.time() - T0:7.2f}]", *a, flush=True)
T0 = time.time()
import numpy as np
import torch
say("torch imported")
from butterfly_amd import _capi, helm2_structure as hs
from butterfly_amd.operator import HipOperator
for n, k in ((2048, 128.0), (8192, 512.0)):
    desc, perm = hs.native_multilevel_structure(hs.circle_points(n), k)
    say("layout", n)
    staged = HipOperator.from_desc(desc, None, seed=9)
    say("staged compiled")
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal(n) + 1j * rng.standard_normal(n)).cuda()
    want = staged.apply_device(x)
    torch.cuda.synchronize()
    say("staged applied")
    flow = HipOperator.from_desc(desc, None, seed=9, flags=_capi.FLAG_FLOW)
    say("flow compiled", flow.flow_status())
    for rep in range(3):
        t0 = time.time()
        got = flow.apply_device(x)
        say("flow enqueued")
        torch.cuda.synchronize()
        say(n, "apply", rep, f"{time.time() - t0:.3f}s equal={torch.equal(got, want)} status={flow.flow_status()} stages={flow.stats()['numStages']}")
    flow.close(); staged.close()
