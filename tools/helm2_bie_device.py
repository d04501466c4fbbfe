#!/usr/bin/env python3
"""The reference's BIE driver (examples/simple/helm2_bie.c:93-171) with every heavy step on the GPU:
S' factorization values (bfhipBuildHelm2, PV_NORMAL_DERIV_SINGLE), trapezoid column weights and
the I/2 term folded into the values, unrestarted GMRES with the Krylov basis in HBM
(bfhipSolveGMRESDevice), and the residual of the solution against the DENSE system evaluated
matrix-free (bfhipHelm2DenseApply).  No KR quadrature correction (helm2_bie.c:100-107).
Prints one JSON line.   usage: tools/helm2_bie_device.py [--npoints N] [--wavenumber K]"""
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
    ap.add_argument("--tol", type=float, default=1e-9)
    ap.add_argument("--max-iter", type=int, default=600)
    args = ap.parse_args()
    import torch
    from butterfly_amd import helm2_structure as hs
    from butterfly_amd.operator import HipOperator, helm2_dense_apply
    n = args.npoints
    k = args.wavenumber if args.wavenumber is not None else n / 16.0
    t0 = time.time()
    pts = hs.circle_points(n)
    desc, _, perm = hs.helm2_multilevel_structure(pts, k, recipes=True)
    tp = pts[perm]
    t_struct = time.time() - t0
    deco = dict(layer_pot="Sp", normals=tp.copy(), col_weights=np.full(n, 2 * np.pi / n), self_value=0.5)
    t0 = time.time()
    op, st = HipOperator.build_helm2(desc, tp, k, device=0, **deco)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    d = np.array([np.cos(0.3), np.sin(0.3)])
    b = torch.from_numpy(1j * k * (tp @ d) * np.exp(1j * k * (tp @ d))).cuda()      # d/dn of a plane wave on the unit circle
    t0 = time.time()
    sigma, iters, res = op.solve_gmres_device(b, tol=args.tol, max_num_iter=args.max_iter)
    torch.cuda.synchronize()
    t_solve = time.time() - t0
    t0 = time.time()
    r = helm2_dense_apply(tp, k, sigma, **deco) - b
    torch.cuda.synchronize()
    t_dense = time.time() - t0
    print(json.dumps({
        "workload": f"second-kind BIE (I/2 + S' w) sigma = dn u_inc, unit circle, N={n}, k={k:g}, butterfly built/applied/solved on one MI355X",
        "leaf_bytes": op.stats()["leafBytes"], "structure_seconds": t_struct, "build_seconds": t_build, "build_stats": st,
        "gmres_iterations": iters, "gmres_reported_residual": res, "gmres_seconds": t_solve,
        "gmres_ms_per_iteration": t_solve / max(iters, 1) * 1e3,
        "dense_residual_rel_l2": float((torch.linalg.norm(r) / torch.linalg.norm(b)).item()), "dense_apply_seconds": t_dense}))


if __name__ == "__main__":
    main()
