#!/usr/bin/env python3
"""The reference's BIE driver (examples/simple/helm2_bie.c:93-171) with every heavy step on the GPU:
S' factorization values (bfhipBuildHelm2, PV_NORMAL_DERIV_SINGLE), trapezoid column weights and
the I/2 term folded into the values, unrestarted GMRES with the Krylov basis in HBM
(bfhipSolveGMRESDevice), the residual of the solution against the DENSE system evaluated
matrix-free (bfhipHelm2DenseApply), and the example's acceptance check (helm2_bie.c:180-214): the
single-layer potential of the solution against the field of the interior point source at exterior
targets.  6th-order Kapur-Rokhlin correction folded into the values (helm2_bie.c:14,113).
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
    desc, perm = hs.native_multilevel_structure(pts, k)
    tp = pts[perm]
    t_struct = time.time() - t0
    w = np.full(n, 2 * np.pi / n)
    deco = dict(layer_pot="Sp", normals=tp.copy(), col_weights=w, self_value=0.5, kr_order=6, orig_index=perm)
    t0 = time.time()
    op, st = HipOperator.build_helm2(desc, tp, k, device=0, **deco)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    from scipy.special import hankel1
    src = np.array([0.1, 0.2])                                                       # interior point source
    dxy = tp - src[None, :]
    r = np.hypot(dxy[:, 0], dxy[:, 1])
    b = torch.from_numpy(0.25j * k * hankel1(1, k * r) / r * np.sum(tp * dxy, axis=1)).cuda()   # S' kernel, helm2_bie.c:76
    t0 = time.time()
    sigma, iters, res = op.solve_gmres_device(b, tol=args.tol, max_num_iter=args.max_iter)
    torch.cuda.synchronize()
    t_solve = time.time() - t0
    t0 = time.time()
    r = helm2_dense_apply(tp, k, sigma, **deco) - b
    torch.cuda.synchronize()
    t_dense = time.time() - t0
    # exterior field through the evaluation butterfly (srcTree != tgtTree; helm2_bie.c:183 builds G_eval densely)
    mt = max(4096, n // 4)
    th = 2 * np.pi * np.arange(mt) / mt
    tgt = 2.0 * np.stack([np.cos(th), np.sin(th)], axis=1)
    t0 = time.time()
    ev, (eps_, ept), est = HipOperator.fac_helm2_make_multilevel(pts, k, col_weights=w, tgt_points=tgt, device=0)
    sig_orig = torch.empty_like(sigma)
    sig_orig[torch.from_numpy(perm).cuda()] = sigma                                   # tree order -> original order
    phi = ev.apply_device(sig_orig[torch.from_numpy(eps_).cuda()]).cpu().numpy()
    t_eval = time.time() - t0
    tt = tgt[ept]
    phi_exact = 0.25j * hankel1(0, k * np.hypot(tt[:, 0] - src[0], tt[:, 1] - src[1]))
    print(json.dumps({
        "workload": f"exterior Neumann BIE (I/2 + S' KR6 w) sigma = dn G(. - x0), unit circle, N={n}, k={k:g}, butterfly built/applied/solved on one MI355X",
        "leaf_bytes": op.stats()["leafBytes"], "structure_seconds": t_struct, "build_seconds": t_build, "build_stats": st,
        "gmres_iterations": iters, "gmres_reported_residual": res, "gmres_seconds": t_solve,
        "gmres_ms_per_iteration": t_solve / max(iters, 1) * 1e3,
        "dense_residual_rel_l2": float((torch.linalg.norm(r) / torch.linalg.norm(b)).item()), "dense_apply_seconds": t_dense,
        "exterior_targets": mt, "evaluation_butterfly_build_and_apply_seconds": t_eval,
        "exterior_field_rel_l2_error": float(np.linalg.norm(phi - phi_exact) / np.linalg.norm(phi_exact))}))


if __name__ == "__main__":
    main()
