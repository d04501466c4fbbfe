#!/usr/bin/env python3
"""bench.py -- butterfly matvecs/sec + achieved HBM GB/s on MI355X.

Workload (BASELINE.json `metric`): 2D Helmholtz single-layer operator on N
equispaced points of the unit circle, k = N/16 (16 points per wavelength),
fac_helm2-style multilevel butterfly (HODBF), complex128, one right-hand side.
The operand is *structure-exact, value-synthetic* (SURVEY.md section 8(d)): every
block shape is what the reference's builder would produce (checked against the
survey's probe statistics in tests/test_structure.py); values are a seeded
counter-based stream generated directly in HBM.  Apply cost does not depend on
the values.

A "step" is one full apply y = A x with x, y resident in HBM.  With N GPUs the
top-level row blocks are dealt to ranks (strong scaling: the operator is fixed);
each step then ends with one RCCL all-gather of y over xGMI.

Prints ONE JSON line on rank 0 (contract in the round prompt): metric, value,
roofline{...} for the stage kernel from hipEvents inside the library, and
cpu_baseline{...} = the CPU oracle (oracle/bfref.c, a port of the reference's
bfMatMul) timed on a bounded sample of the same operand on this box.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable copy)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense FP64 matrix peak = FP64 vector peak = 1/2 of the 157.3 TFLOP/s FP32 rate


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(desc, seed, total_leaf_elems, weights, budget_bytes, nrhs, x, y_gpu_full, row_offsets):
    """Oracle (port of the reference bfMatMul) on a bounded sample: the top-level
    block rows, smallest first, that fit in `budget_bytes` of leaf data."""
    from butterfly_amd import helm2_structure as hs
    from oracle import bfref
    blas = bfref.try_use_openblas()
    if blas:
        try:                                   # "cores": 1 must be true: the reference is serial
            import ctypes
            ctypes.CDLL(blas).scipy_openblas_set_num_threads(1)
        except Exception:
            pass
    order = np.argsort(weights)
    chosen, acc = [], 0
    for rb in order:
        if weights[rb] == 0:
            continue
        if acc + weights[rb] * 16 > budget_bytes and chosen:
            break
        chosen.append(int(rb))
        acc += weights[rb] * 16
    chosen.sort()
    root, nrows = hs.shard_desc(desc, chosen)
    t0 = time.time()
    A = bfref.from_desc(desc, None, seed=seed, root=root)
    t_build = time.time() - t0
    xs = x if nrhs > 1 else x[:, None]
    best = None
    reps = 0
    t_start = time.time()
    y = None
    while reps < 5 and (reps < 2 or time.time() - t_start < 20):
        bfref.reset_counters()
        t0 = time.perf_counter()
        y = bfref.mat_mul(A, xs)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        reps += 1
    cnt = bfref.counters()
    # The reference is single-threaded and its leaf GEMMs are small; for blocks of RHS also time
    # the same port with OpenBLAS on every host core this process may use
    multi = None
    if blas and nrhs >= 3:
        try:
            import ctypes
            ob = ctypes.CDLL(blas)
            ncores = len(os.sched_getaffinity(0))
            ob.scipy_openblas_set_num_threads(ncores)
            bm = None
            for _ in range(2):
                t0 = time.perf_counter()
                bfref.mat_mul(A, xs)
                dt = time.perf_counter() - t0
                bm = dt if bm is None else min(bm, dt)
            ob.scipy_openblas_set_num_threads(1)
            multi = (ncores, bm)
        except Exception:
            multi = None
    # parity of the device result on the sampled rows (full-size operand)
    ref_rows = np.concatenate([np.arange(row_offsets[rb], row_offsets[rb + 1]) for rb in chosen])
    yg = y_gpu_full[ref_rows]
    err = float(np.linalg.norm(yg - y.reshape(yg.shape)) / np.linalg.norm(y))
    sample_elems = sum(weights[rb] for rb in chosen)
    frac = sample_elems / total_leaf_elems
    full_equiv = nrhs / (best / frac)
    extra = {}
    if multi:
        extra["all_cores"] = {"value": nrhs / (multi[1] / frac), "unit": "matvec/s", "cores": multi[0],
                              "note": "same port, OpenBLAS threads = host cores available to this process"}
    return dict(value=full_equiv, unit="matvec/s", cores=1, kind="port", **extra,
                sample=(f"top-level block rows {chosen} of {len(weights)} ({sample_elems * 16 / 1e9:.2f} GB of "
                        f"{total_leaf_elems * 16 / 1e9:.2f} GB leaf data, {frac * 100:.1f}%), best of {reps} in "
                        f"{best * 1e3:.1f} ms, scaled by leaf bytes; blas={os.path.basename(blas) if blas else 'builtin-c'}; "
                        f"{cnt['gemmCalls']} leaf gemm calls, {cnt['mallocs']} mallocs per apply; graph build {t_build:.1f}s"),
                sample_seconds=best, parity_rel_l2=err)


def main():
    # Exactly one line may reach stdout (the JSON).  Native libraries (RCCL prints a
    # version banner at communicator creation) write to fd 1 directly, so fd 1 is
    # pointed at stderr for the whole run and the JSON goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--npoints", "--n", dest="n", type=int, default=262144,
                    help="(use --npoints under torch.distributed.run: its parser treats a bare --n as ambiguous)")
    ap.add_argument("--wavenumber", "--k", dest="k", type=float, default=None, help="wavenumber (default N/16)")
    ap.add_argument("--nrhs", type=int, default=1)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--cpu-budget-gb", type=float, default=4.0, help="leaf bytes of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["c128", "f64", "f32"], default="c128",
                    help="c128: the fac_helm2 operand (headline).  f64 / f32: the SAME block layout with real values, a proxy "
                         "for the real BfMatDenseReal path (BASELINE.json configs[4]); fp32 is the build's extension")
    ap.add_argument("--adjoint", action="store_true", help="also time y = A^T x (RmulVec path) and report it next to the headline")
    ap.add_argument("--pcie", action="store_true", help="also time the host-buffer path (H2D + apply + D2H)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic: on ONE GPU, time the shard that rank --emulate-rank of an N-rank job would own (no collective)")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--python-layout", action="store_true", help="lay the operand out with butterfly_amd/helm2_structure.py instead of the C layout")
    ap.add_argument("--force-collective", action="store_true",
                    help="rehearsal: initialise RCCL and run the step's collective even with one rank")
    ap.add_argument("--shard", choices=["auto", "rows", "blocks"], default="auto",
                    help="multi-GPU: top-level block rows + all-gather, or (row, col) blocks + all-reduce")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from butterfly_amd import _capi, helm2_structure as hs
    from butterfly_amd.dist import (ShardLayout, ShardedApply, assign_row_blocks, block_weights, choose_mode,
                                    row_block_weights)
    from butterfly_amd.operator import HipOperator

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_pg = world > 1 or args.force_collective
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    if world != args.gpus and rank == 0:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")

    n = args.n
    k = args.k if args.k is not None else n / 16.0
    t0 = time.time()
    # block layout of the operand: the native layout (bfhip_layout.c) unless the Python restatement is asked for
    if args.python_layout:
        desc, _, perm = hs.helm2_multilevel_structure(hs.circle_points(n), k)
    else:
        desc, perm = hs.native_multilevel_structure(hs.circle_points(n), k)
    t_struct = time.time() - t0
    real = args.dtype != "c128"
    if real:
        desc.dtype = 1          # BFHIP_F64 leaves; --dtype f32 demotes them on upload
    esz = {"c128": 16, "f64": 8, "f32": 4}[args.dtype]
    tdtype = {"c128": torch.complex128, "f64": torch.float64, "f32": torch.float32}[args.dtype]
    weights = row_block_weights(desc)
    total_leaf = int(sum(weights))
    top_rows = desc.meta["top_rows"]
    row_offsets = np.concatenate([[0], np.cumsum(top_rows)]).astype(np.int64)
    sworld, srank = (args.emulate_world, args.emulate_rank) if args.emulate_world > 1 else (world, rank)
    mode = choose_mode(desc, sworld, args.shard)
    if mode == "rows":
        owner, loads = assign_row_blocks(weights, sworld)
        mine = [rb for rb in range(len(weights)) if owner[rb] == srank]
    else:
        bw = block_weights(desc)
        bowner, loads = assign_row_blocks(bw, sworld)
        mine = [i for i in range(len(bw)) if bowner[i] == srank]
        owner = [0] * len(weights)
    if rank == 0:
        log(f"structure: N={n} k={k:g} nodes={desc.num_nodes} leafGB={total_leaf * 16 / 1e9:.2f} "
            f"products={desc.meta['stats']['products']} [{t_struct:.1f}s]; shard mode={mode}; rank loads GB="
            f"{[round(l * 16 / 1e9, 2) for l in loads]}")

    t0 = time.time()
    if sworld == 1:
        root, local_rows = desc.root, n
    elif mode == "rows":
        root, local_rows = hs.shard_desc(desc, mine)
    else:
        root, local_rows = hs.shard_desc_blocks(desc, mine), n
    op = HipOperator.from_desc(desc, None, root=root, device=local_rank, flags=_capi.FLAG_PROFILE | (_capi.FLAG_ADJOINT if args.adjoint else 0),
                               seed=args.seed, max_rhs=args.nrhs, demote_to_f32=(args.dtype == "f32"))
    torch.cuda.synchronize()
    t_compile = time.time() - t0
    st = op.stats()
    if rank == 0:
        log(f"compile+synthesize: {t_compile:.1f}s stages={st['numStages']} items={st['numItems']} pieces={st['numPieces']} "
            f"arenaGB={st['arenaBytes'] / 1e9:.2f} metaMB={st['metaBytes'] / 1e6:.1f} tempElems={st['tempElems']}")

    # x: seeded complex normal, identical on every rank (replicated input)
    rng = np.random.default_rng(args.seed)
    shape = (n,) if args.nrhs == 1 else (n, args.nrhs)
    if real:
        x_host = rng.standard_normal(shape)
    else:
        x_host = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2)
    x = torch.from_numpy(x_host).to(dev).to(tdtype)
    layout = ShardLayout(top_rows, owner, sworld)
    if args.emulate_world > 1:
        layout.world = 1        # one process: run this shard's local apply only
    assert mode == "blocks" or layout.rows_of[srank] == local_rows
    step = ShardedApply(layout, srank, lambda xin, out: op.apply_device(xin, out), dev, tdtype, nrhs=args.nrhs,
                        mode=mode, force_collective=args.force_collective)

    for _ in range(args.warmup):
        y_full = step(x)
    torch.cuda.synchronize()
    op.stage_profile(reset=True)
    if use_pg:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y_full = step(x)
    torch.cuda.synchronize()
    if use_pg:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_pg:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ms, launches, sbytes = op.stage_profile()
    if rank == 0:
        kern_ms = float(ms.sum())
        n_launch = int(launches.sum())
        bytes_per_apply = int(sbytes.sum())
        avg_launch_ms = kern_ms / max(n_launch, 1)
        achieved = (bytes_per_apply * (n_launch / len(ms))) / 1e9 / (kern_ms / 1e3) if kern_ms > 0 else 0.0
        for s in range(len(ms)):
            log(f"  stage {s}: {ms[s] / max(launches[s], 1):8.3f} ms/launch  {sbytes[s] / 1e9:8.3f} GB  "
                f"{(sbytes[s] / 1e9) / (ms[s] / max(launches[s], 1) / 1e3) if ms[s] > 0 else 0:8.1f} GB/s")
        if args.nrhs >= 3 and not real:
            # block of right-hand sides: 8*nrhs flops per leaf element (32 flop/B at nrhs=64) -> FP64-MFMA bound
            flops_per_apply = 8.0 * args.nrhs * st["leafElems"]
            tf = flops_per_apply * (n_launch / len(ms)) / 1e12 / (kern_ms / 1e3) if kern_ms > 0 else 0.0
            roofline = {"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tf / FP64_MFMA_PEAK_TFLOPS, "traffic": None, "kernel": "bfStageKernelC128Mfma",
                        "launches_per_apply": len(ms), "avg_launch_ms": avg_launch_ms,
                        "algorithmic_flops_per_apply": flops_per_apply, "hbm_gbs_algorithmic": achieved,
                        "kernel_ms_per_apply": kern_ms / max(launches.max(), 1)}
        else:
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                        "kernel": "bfStageKernelC128" if not real else f"bfStageKernelReal<{args.dtype}>", "launches_per_apply": len(ms),
                        "avg_launch_ms": avg_launch_ms, "algorithmic_bytes_per_apply": bytes_per_apply,
                        "kernel_ms_per_apply": kern_ms / max(launches.max(), 1)}
        # HBM traffic per launch from the committed PMC profile of this very command (rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 FETCH correction); PMC counters cannot be
        # read from inside the process, so other configurations report null.
        prof = os.path.join(ROOT, "profiles", "r1_pmc_summary.json")
        if world == 1 and n == 262144 and abs(k - 16384) < 1e-9 and args.nrhs in (1, 64) and not real and os.path.exists(prof):
            try:
                key = "bfStageKernelC128_per_launch" if args.nrhs == 1 else "bfStageKernelC128Mfma_per_launch"
                pm = json.load(open(prof))[key]
                roofline["traffic"] = pm["hbm_bytes"]
                roofline["traffic_source"] = f"profiles/r1_pmc_summary.json [{key}]: HBM bytes per launch of this command under rocprofv3 --pmc"
            except Exception:
                pass
        # context for `frac` (informational; `peak` stays the guide's figure): what a pure stream of
        # non-temporal reads / a bare loop of these MFMAs reaches on an MI355X of this pool
        for key, fn, fld in (("measured_stream_read_gbs", "r1_hbm_peak.json", "read_nt_gbs"), ("measured_mfma_loop_tflops", "r1_mfma_peak.json", "fp64_mfma_16x16x4_tflops")):
            if (roofline["bound"] == "hbm") == (fld == "read_nt_gbs"):
                try:
                    roofline[key] = json.load(open(os.path.join(ROOT, "profiles", fn)))[fld]
                except Exception:
                    pass
        if args.pcie:
            # host-buffer path (bfhipApply): H2D of x, all stages, D2H of y -- never the headline value
            t1 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                op.apply_host(x_host)
            pcie_ms = (time.perf_counter() - t1) / reps * 1e3
        out = {
            "metric": "butterfly matvecs/sec (2D Helmholtz HODBF apply)",
            "value": args.steps * args.nrhs / elapsed,
            "unit": "matvec/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (structure-exact fac_helm2 layout, seeded values generated in HBM)",
            "config": {"workload": f"fac_helm2 multilevel butterfly, unit circle, N={n}, k={k:g} (16 ppw), nrhs={args.nrhs}"
                                   + ("" if not real else f" [real-valued {args.dtype} proxy on the same block layout]"),
                       "n": n, "k": k, "nrhs": args.nrhs, "leaf_bytes": total_leaf * esz,
                       "stages": st["numStages"], "sharding": ("none" if world == 1 else "top-level row blocks (LPT by leaf bytes) + one all-gather" if mode == "rows"
                                    else "top-level (row, col) blocks (LPT by leaf bytes) + one all-reduce")},
            "roofline": roofline,
            "hbm_gbs_whole_step": (total_leaf * esz / 1e9) / (elapsed / args.steps),
        }
        if args.adjoint and world == 1:
            xt = torch.from_numpy(x_host).to(dev).to(tdtype)
            for _ in range(2):
                yt = op.apply_transpose_device(xt)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                yt = op.apply_transpose_device(xt)
            torch.cuda.synchronize()
            adj_ms = (time.perf_counter() - t1) / args.steps * 1e3
            out["adjoint"] = {"ms_per_apply": adj_ms, "matvec_per_s": args.nrhs / (adj_ms / 1e3),
                              "hbm_gbs": total_leaf * esz / 1e9 / (adj_ms / 1e3)}
        if args.pcie:
            out["pcie_inclusive"] = {"ms_per_apply": pcie_ms, "matvec_per_s": args.nrhs / (pcie_ms / 1e3),
                                     "note": "bfhipApply on pageable host buffers: H2D x + apply + D2H y"}
        if args.emulate_world > 1:
            out["emulated_shard"] = {"world": args.emulate_world, "rank": args.emulate_rank, "mode": mode,
                                     "shard_leaf_bytes": st["leafBytes"]}
        if world == 1 and not args.no_cpu_baseline and not real and args.emulate_world <= 1:
            try:
                y_host = y_full.cpu().numpy()
                out["cpu_baseline"] = cpu_baseline(desc, args.seed, total_leaf, weights, args.cpu_budget_gb * 1e9,
                                                   args.nrhs, x_host, y_host, row_offsets)
            except Exception as e:  # the baseline must never take the measurement down
                out["cpu_baseline"] = {"value": None, "unit": "matvec/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    op.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
