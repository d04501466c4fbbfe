#!/usr/bin/env python3
"""Build a real fac_helm2 operand on the device (bfhipBuildHelm2) and check it the way the
reference's examples do (examples/simple/bf_all_blocks.c:132-153): relative l2 error of the
butterfly apply against the dense kernel matvec, here evaluated matrix-free on the GPU.
Prints one JSON line.   usage: tools/build_fullsize.py [--npoints N] [--wavenumber K]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--npoints", type=int, default=65536)
    ap.add_argument("--wavenumber", type=float, default=None)
    ap.add_argument("--workspace-gb", type=float, default=0.0, help="0: library default (half of free HBM)")
    ap.add_argument("--no-dense-check", action="store_true")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from butterfly_amd import _capi
    from butterfly_amd import helm2_structure as hs
    from butterfly_amd.operator import HipOperator, helm2_dense_apply
    n = args.npoints
    k = args.wavenumber if args.wavenumber is not None else n / 16.0
    t0 = time.time()
    pts = hs.circle_points(n)
    desc, perm = hs.native_multilevel_structure(pts, k)
    tp = pts[perm]
    t_struct = time.time() - t0
    print(f"structure: N={n} k={k:g} leaves={len(desc.recipe_array)} [{t_struct:.1f}s]", file=sys.stderr, flush=True)
    t0 = time.time()
    op, st = HipOperator.build_helm2(desc, tp, k, device=0, flags=_capi.FLAG_PROFILE, workspace_bytes=int(args.workspace_gb * 2**30))
    torch.cuda.synchronize()
    t_build = time.time() - t0
    print(f"device build: {t_build:.1f}s {st}", file=sys.stderr, flush=True)
    rng = np.random.default_rng(0)
    x = torch.from_numpy((rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)).cuda()
    y = op.apply_device(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        y = op.apply_device(x)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / args.steps * 1e3
    out = {"workload": f"fac_helm2 multilevel butterfly built on the device, unit circle, N={n}, k={k:g}",
           "leaf_bytes": op.stats()["leafBytes"], "structure_seconds": t_struct, "build_seconds": t_build,
           "build_stats": st, "apply_ms": ms, "apply_hbm_gbs": op.stats()["leafBytes"] / ms / 1e6}
    if not args.no_dense_check:
        t0 = time.time()
        yd = helm2_dense_apply(tp, k, x)
        torch.cuda.synchronize()
        out["dense_apply_seconds"] = time.time() - t0
        out["rel_l2_error_vs_dense"] = float((torch.linalg.norm(y - yd) / torch.linalg.norm(yd)).item())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
