#!/usr/bin/env python3
"""bench.py -- butterfly matvecs/sec + achieved HBM GB/s on MI355X.

Headline workload (BASELINE.json `metric`, --workload helm2): 2D Helmholtz single-layer operator on N
equispaced points of the unit circle, k = N/16 (16 points per wavelength), fac_helm2-style multilevel
butterfly (HODBF), complex128, one right-hand side.  The operand is *structure-exact, value-synthetic*
(SURVEY.md section 8(d)): every block shape is what the reference's builder would produce (checked against
the survey's probe statistics in tests/test_structure.py); values are a seeded counter-based stream
generated directly in HBM.  Apply cost does not depend on the values.

--workload streamer (BASELINE configs[4]): the streamed real butterfly of examples/covariance --
Laplace-Beltrami eigenvectors of a sphere with N vertices, octree rows, binary frequency tree, tol 1e-3,
minNumRows = minNumCols = 20 -- laid out by the reference's merge-and-split recursion
(butterfly_amd/streamer_structure.py) under the fitted rank model, values synthetic, fp32 by default.

A "step" is one full apply y = A x with x, y resident in HBM.  With N GPUs (`--gpus N`: this script starts
the N rank processes itself when it was not started by torch.distributed.run) the top-level blocks are
dealt to ranks (strong scaling: the operator is fixed); each step then ends with ONE RCCL collective on y
over xGMI, issued by libbfhip.so on the apply stream.

Prints ONE JSON line on rank 0 (contract in the round prompt): metric, value, roofline{...} for the stage
kernel from hipEvents inside the library, and cpu_baseline{...} = the CPU oracle (oracle/bfref.c, a port of
the reference's bfMatMul / bfMatMulVec) timed on a bounded sample of the same operand on this box.
The default line (N = 262144, one GPU) also carries two extra keys measured after the headline loop, so that the
driver's own run times them: `nrhs64` (BASELINE configs[2], FP64-MFMA kernel) and `configs4_streamer` (the
--workload streamer measurement at N = 1M, in a child process: the operand is laid out in ~6 s by the C layout,
compiled + synthesized in ~10 s, its CPU baseline sample takes ~20 s); --no-extra / --no-streamer skip them.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable copy)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense FP64 matrix peak = FP64 vector peak = 1/2 of the 157.3 TFLOP/s FP32 rate
PROFILE_ROUND = "r5"      # which committed rocprofv3 summaries the `traffic` figure is read from


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_pmc_summary():
    """(summary dict, its round tag) of the newest committed rocprofv3 --pmc summary under profiles/."""
    for rnd in (PROFILE_ROUND, "r4", "r3"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary.json")
        if os.path.exists(path):
            return json.load(open(path)), rnd
    raise FileNotFoundError("no profiles/*_pmc_summary.json")


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of one node (default: the launcher's WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["helm2", "streamer"], default="helm2")
    ap.add_argument("--npoints", "--n", dest="n", type=int, default=None, help="points (default 262144 helm2, 1048576 streamer)")
    ap.add_argument("--wavenumber", "--k", dest="k", type=float, default=None, help="helm2: wavenumber (default N/16)")
    ap.add_argument("--lmax", type=int, default=255, help="streamer: highest spherical-harmonic degree streamed (J = (lmax + 1)^2 columns)")
    ap.add_argument("--freq-depth", type=int, default=None, help="streamer: depth of the frequency tree (default: row-tree depth - 3, lbo_cov.c:97-98)")
    ap.add_argument("--nrhs", type=int, default=1)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--cpu-budget-gb", type=float, default=None,
                    help="leaf bytes of the cpu_baseline sample (default: a quarter of the operand's leaf bytes, at least 4 GB and at most 24 GB)")
    ap.add_argument("--adjoint-both", action="store_true",
                    help="with --adjoint: after the packed-copy measurement compile the operator again with the shared-leaf adjoint plan and time that too (`adjoint_shared` key)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the 64-RHS and configs[4] measurements appended to the default line")
    ap.add_argument("--no-streamer", action="store_true", help="skip only the configs[4] (streamed butterfly) measurement appended to the default line")
    ap.add_argument("--dtype", choices=["c128", "f64", "f32"], default=None,
                    help="helm2: c128 (headline); f64 / f32 = the same block layout with real values (a kernel proxy).  "
                         "streamer: f32 (default; the build's extension, configs[4]) or f64 (the reference's type)")
    ap.add_argument("--adjoint", action="store_true", help="also time y = A^T x (RmulVec path) and report it next to the headline")
    ap.add_argument("--adjoint-shared", action="store_true",
                    help="with --adjoint: the adjoint plan reads the forward plan's packed leaves through the transposed kernels (BFHIP_FLAG_ADJOINT: no extra "
                         "leaf memory) instead of a packed copy of its own on the forward kernels (BFHIP_FLAG_ADJOINT_PACKED, the default here: twice the leaf memory)")
    ap.add_argument("--pcie", action="store_true", help="also time the host-buffer path (H2D + apply + D2H)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic: on ONE GPU, time the shard that rank --emulate-rank of an N-rank job would own (no collective); "
                         "--emulate-rank -1 times every rank's shard one after another")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--python-layout", action="store_true", help="lay the operand out with the Python restatements (helm2_structure.py / streamer_structure.py) instead of the C layouts")
    ap.add_argument("--force-collective", action="store_true", help="rehearsal: run the sharded C-ABI path (RCCL communicator + collective) even with one rank")
    ap.add_argument("--shard", choices=["auto", "rows", "rowblocks", "blocks", "rowsum"], default="auto",
                    help="multi-GPU: rows = balanced contiguous row ranges (bfhipRowPartition) + ONE all-gather, bit-identical to one GPU (default); "
                         "rowblocks = whole top-level block rows by LPT + all-gather; blocks = top-level (row, col) blocks by LPT + ONE all-reduce; "
                         "rowsum = whole block rows + column shares of the rest, ONE all-gather, shared rows' partials added in rank order (no replication)")
    args = ap.parse_args(argv)
    if args.gpus is None:          # started by torch.distributed.run without --gpus: the launcher's world size is the job's
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    return args


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """--gpus N without a launcher: start N rank processes (torch.distributed.run on 127.0.0.1) BEFORE this
    process touches a GPU, relay rank 0's JSON line, and fail if any rank fails."""
    # torch.distributed.run's own parser expands unambiguous prefixes even behind the script name: spell the aliases out
    alias = {"--n": "--npoints", "--k": "--wavenumber"}
    argv = [alias.get(a, a) for a in argv]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    log("bench.py: starting", args.gpus, "ranks:", " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        log(f"bench.py: rank processes failed (exit code {p.returncode}, {len(lines)} JSON line(s))")
        sys.stdout.write(p.stdout)
        sys.exit(p.returncode or 1)
    print(lines[-1], flush=True)
    sys.exit(0)


# ---------------------------------------------------------------------------------------------------
# What a multi-rank run checks about itself (same code on GPUs over RCCL and in the CPU dry run over gloo)
# ---------------------------------------------------------------------------------------------------
def ranks_max_abs_diff(dist, y, use_pg):
    """max over ranks r of max |y_r - y_0|: after the closing collective every rank must hold the same y."""
    import torch
    if not use_pg:
        return 0.0
    ref = y.detach().clone()
    view = torch.view_as_real(ref) if ref.is_complex() else ref
    dist.broadcast(view, src=0)
    d = (y - ref).abs().max().to(torch.float64).reshape(1) if y.numel() else torch.zeros(1, dtype=torch.float64, device=y.device)
    dist.all_reduce(d, op=dist.ReduceOp.MAX)
    return float(d.item())


def all_ranks_ok(dist, ok, text, use_pg):
    """(every rank succeeded, [(rank, error text) of those that did not]): a collective decision, so that either every
    rank keeps libbfhip's own RCCL communicator or every rank falls back to torch.distributed."""
    if not use_pg:
        return bool(ok), ([] if ok else [(0, text)])
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, (bool(ok), str(text)[:300]))
    bad = [(r, t) for r, (o, t) in enumerate(box) if not o]
    return not bad, bad


def pad_to_common_length(dist, vec, use_pg):
    """ranks may have compiled plans of different depth (blocks / rowsum shards): pad a per-stage vector with zeros to the
    longest before it is all-gathered"""
    import torch
    if not use_pg:
        return vec
    m = torch.tensor([vec.numel()], dtype=torch.int64, device=vec.device)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return torch.cat([vec, torch.zeros(int(m.item()) - vec.numel(), dtype=vec.dtype, device=vec.device)])


# ---------------------------------------------------------------------------------------------------
# CPU baselines (the oracle as the checker and as the stated baseline; never the thing shipped)
# ---------------------------------------------------------------------------------------------------
def _blas_one_thread():
    from oracle import bfref
    blas = bfref.try_use_openblas()
    if blas:
        try:                                   # "cores": 1 must be true: the reference is serial
            import ctypes
            ctypes.CDLL(blas).scipy_openblas_set_num_threads(1)
        except Exception:
            pass
    return blas


def cpu_baseline_helm2(desc, seed, total_leaf_elems, weights, budget_bytes, nrhs, x, y_gpu_full, row_offsets, first=()):
    """Oracle (port of the reference bfMatMul) on a bounded sample: the top-level block rows, smallest
    first, that fit in `budget_bytes` of leaf data; extrapolated to the whole operand by leaf bytes.
    `first`: block rows the sample must start with (multi-GPU rowsum mode: a block row whose result is the sum
    of two ranks' partials, so that the parity figure of the line covers the shared path)."""
    import numpy as np
    from butterfly_amd import helm2_structure as hs
    from oracle import bfref
    blas = _blas_one_thread()
    order = [int(rb) for rb in first] + [int(rb) for rb in np.argsort(weights) if int(rb) not in set(first)]
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
    best, reps, y = None, 0, None
    t_start = time.time()
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
                sample=(f"EXTRAPOLATED from a {frac * 100:.1f}% sample: top-level block rows {chosen} of {len(weights)} "
                        f"({sample_elems * 16 / 1e9:.2f} GB of {total_leaf_elems * 16 / 1e9:.2f} GB leaf data), best of {reps} in "
                        f"{best * 1e3:.1f} ms, scaled by leaf bytes; blas={os.path.basename(blas) if blas else 'builtin-c'}; "
                        f"{cnt['gemmCalls']} leaf gemm calls, {cnt['mallocs']} mallocs per sample apply; graph build {t_build:.1f}s"),
                sample_seconds=best, parity_rel_l2=err)


def cpu_baseline_streamer(desc, seed, budget_bytes):
    """Oracle bfMatMulVec on a bounded sample of the streamed operand: of every factor of the (first)
    product a prefix of its top-level blocks, `budget_bytes` of leaves in total, each timed as its own
    sub-operator; extrapolated to the whole operand by leaf bytes (fp64, the reference's only type).
    `desc`: the operand's descriptor (root = the 1 x numFacs BlockDense row of products, src/fac_span.c:126-155)."""
    import numpy as np
    from butterfly_amd import helm2_structure as hs
    from oracle import bfref
    blas = _blas_one_thread()
    a = desc.arrays()
    sub_elems = desc.subtree_leaf_elems()
    kind, bkind = a["kind"], a["blockKind"]
    top = [c for c, _, _ in desc.children[desc.root]] if kind[desc.root] == hs.NODE_BLOCK else [desc.root]
    prods = [c for c in top if kind[c] == hs.NODE_PRODUCT] or top
    factors = [f for p in prods for f, _, _ in desc.children[p]]
    total = int(sum(int(sub_elems[f]) for f in factors)) * 8
    per = budget_bytes / max(len(factors), 1)
    rng = np.random.default_rng(seed)
    t_sum, b_sum, pieces = 0.0, 0, []
    names = {hs.BF_TYPE_BLOCK_DIAG: "BlockDiag", hs.BF_TYPE_BLOCK_DENSE: "BlockDense", hs.BF_TYPE_BLOCK_COO: "BlockCoo"}
    for f in factors:
        ch = desc.children[f] if kind[f] == hs.NODE_BLOCK else []
        one_col = bool(ch) and all(c0 == 0 for _, _, c0 in ch) and bkind[f] == hs.BF_TYPE_BLOCK_DENSE
        sub, taken = f, len(ch)
        if ch and (bkind[f] == hs.BF_TYPE_BLOCK_DIAG or one_col):
            # a prefix of a BlockDiag's blocks / of a one-column BlockDense's block rows is itself such a block matrix
            kids, acc = [], 0
            for c, r0, c0 in ch:
                kids.append((c, r0, c0))
                acc += int(sub_elems[c]) * 8
                if acc >= per:
                    break
            taken = len(kids)
            last = kids[-1]
            m = last[1] + int(a["rows"][last[0]])
            n = (last[2] + int(a["cols"][last[0]])) if bkind[f] == hs.BF_TYPE_BLOCK_DIAG else int(a["cols"][f])
            sub = desc.add(hs.NODE_BLOCK, m, n, kids, int(bkind[f]))
            a = desc.arrays()
            acc_b = acc
        else:
            acc_b = int(sub_elems[f]) * 8
        if acc_b == 0:
            continue
        M = bfref.from_desc(desc, None, seed=seed, root=sub)
        x = rng.standard_normal(int(a["cols"][sub]))
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            bfref.mat_mul_vec(M, x)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        t_sum += best
        b_sum += acc_b
        pieces.append(f"{names.get(int(bkind[f]), 'leaf')}[{taken}/{len(ch)}]")
    frac = b_sum / total
    return dict(value=1.0 / (t_sum / frac), unit="matvec/s", cores=1, kind="port",
                sample=(f"EXTRAPOLATED from a {frac * 100:.2f}% sample: a prefix of the top-level blocks of each of the {len(factors)} factors "
                        f"({', '.join(pieces)}; {b_sum / 1e9:.2f} GB of {total / 1e9:.2f} GB fp64 leaf data), each sub-operator's bfMatMulVec best of 3, "
                        f"{t_sum * 1e3:.1f} ms in total, scaled by leaf bytes; fp64 (the reference has no fp32); blas={os.path.basename(blas) if blas else 'builtin-c'}"),
                sample_seconds=t_sum)


# ---------------------------------------------------------------------------------------------------
def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before HIP starts: dmabuf IPC for RCCL between rank processes
    # BENCH_FORCE_LAUNCH=1 (tests): go through the launcher path with a single rank as well
    if (args.gpus > 1 or os.environ.get("BENCH_FORCE_LAUNCH") == "1") and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, argv)          # never returns

    # Exactly one line may reach stdout (the JSON).  Native libraries (RCCL prints a version banner at
    # communicator creation) write to fd 1 directly, so fd 1 is pointed at stderr for the whole run and the
    # JSON goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from butterfly_amd import _capi, helm2_structure as hs
    from butterfly_amd.dist import (RcclShardedApply, ShardedApply, ShardLayout, assign_row_blocks, block_weights, choose_mode,
                                    row_block_weights, row_partition, rowsum_partition)
    from butterfly_amd.operator import HipOperator

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_DRY_RUN=1 (tests, CPU): everything a rank does before it selects its device -- arguments, rendezvous (gloo),
    # the world-size check, the operand's layout and who owns what -- then one JSON line and out
    dry = os.environ.get("BENCH_DRY_RUN") == "1"
    dev = None
    if not dry:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    # --force-collective: the multi-rank code path with ONE rank (a rehearsal on a one-GPU box): a 1-rank process group too, so
    # that everything a real N > 1 run does -- the collective decision, rank agreement, the torch.distributed fallback
    # (BENCH_FORCE_TORCH_COLLECTIVE=1) and, with BENCH_ALSO_TIME_ROWS=1, the second partition -- runs here as well
    use_pg = world > 1 or (args.force_collective and not dry)
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 1000))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        world = dist.get_world_size()
    if world != args.gpus:
        # a line that says n_gpus = 8 must come from 8 ranks: refuse to measure anything else (no JSON, non-zero exit)
        if rank == 0:
            log(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s): refusing to run")
        if use_pg:
            dist.destroy_process_group()
        sys.exit(2)

    streamer = args.workload == "streamer"
    # the default line (headline operand, one GPU) also times the adjoint apply of the same operand
    if (not streamer and world == 1 and args.n in (None, 262144) and args.k is None and args.nrhs == 1 and args.dtype in (None, "c128")
            and not args.no_extra and args.emulate_world <= 1 and not dry):
        args.adjoint = True
        args.adjoint_both = True  # ... both ways: on a packed copy for A^T (twice the leaf memory) and on the shared leaves
        args.pcie = True          # ... and the host-vector path an unmodified reference caller takes (pcie_inclusive)
    dtype = args.dtype or ("f32" if streamer else "c128")
    if streamer and dtype == "c128":
        raise SystemExit("the streamed operand is real: --dtype f32 or f64")
    if streamer and (world > 1 or args.emulate_world > 1) and args.shard not in ("auto", "rows"):
        raise SystemExit("the streamed operand is one product: it has no top-level blocks to deal out, only row ranges (--shard rows)")
    real = dtype != "c128"
    esz = {"c128": 16, "f64": 8, "f32": 4}[dtype]
    tdtype = {"c128": torch.complex128, "f64": torch.float64, "f32": torch.float32}[dtype]
    # (round 4: the packed copy is the default for the streamed operand too -- with runs of narrow pieces contracted as one block
    #  and >= 32768 items per stage its transposed expression runs at 0.765 of the HBM peak on the forward kernels, the transposed
    #  kernels on the shared leaves at 0.737: DESIGN_EXPERIMENTS.md section 10; --adjoint-shared measures the latter)
    # (a sharded operator's adjoint is the shared-leaf plan: a row shard's is the pruned transposed task list, and a packed copy would double every rank's memory)
    shared_adj = args.adjoint_shared or world > 1 or args.force_collective or args.emulate_world > 1
    flags = _capi.FLAG_PROFILE | ((_capi.FLAG_ADJOINT if shared_adj else _capi.FLAG_ADJOINT_PACKED) if args.adjoint else 0)

    # ---- the operand's block layout -------------------------------------------------------------------
    t0 = time.time()
    graph = None
    if streamer:
        from butterfly_amd import streamer_structure as ss
        n = args.n or 1048576
        pts3 = ss.fibonacci_sphere(n)
        wmax = float(np.sqrt(args.lmax * (args.lmax + 1.0)) * 1.0001)
        if args.python_layout:
            # the tested Python restatement of the recursion (minutes at N = 1M); the C layout below is held to it array for array
            tree = ss.Octree(pts3, 1)
            fd = args.freq_depth if args.freq_depth is not None else tree.max_depth - 3
            counts, _ = ss.sphere_band_columns(wmax, fd)
            st_run = ss.stream_structure(tree, wmax, fd, counts)
            graph = st_run.get_mat()
            gstats = ss.graph_stats(graph)
            pdesc, _ = ss.to_desc(graph, with_values=False)
            desc = hs.ArrayDesc(pdesc.arrays(), pdesc.root, 1, [], None, {})
            desc.top_row_block = None
            num_w = [len(f.W) for f in st_run.partial]
        else:
            fd = args.freq_depth if args.freq_depth is not None else ss.octree_depth(pts3) - 3      # lbo_cov.c:97-98
            counts, _ = ss.sphere_band_columns(wmax, fd)
            desc, _, gstats = ss.native_stream_structure(pts3, wmax, fd, counts)
            num_w = [gstats["numW"]]
        ncols = int(desc.cols[desc.root])
        total_leaf = gstats["leafBytes"] // 8
        workload = (f"fac_streamer butterfly of the N x J Laplace-Beltrami eigenvector matrix of a sphere (examples/covariance): N={n} octree rows, "
                    f"J={ncols} columns (lmax={args.lmax}), frequency tree depth {fd}, tol=1e-3 rank model (rank-model structure, NOT the reference's SVD-driven structure: DESIGN_EXPERIMENTS.md section 12), minNumRows=minNumCols=20, nrhs={args.nrhs}")
        config = {"workload": workload, "n": n, "num_cols": ncols, "lmax": args.lmax, "freq_depth": fd, "nrhs": args.nrhs,
                  "leaf_bytes": total_leaf * esz, "num_w": num_w,
                  "graph": {k: gstats[k] for k in ("denseReal", "identity", "blockCoo", "blockDense", "blockDiag", "maxNest")},
                  "dense_bytes": n * ncols * esz, "sharding": "none",
                  "layout": "python restatement" if args.python_layout else "native (bfhipStreamerLayoutCreate)"}
        data = "synthetic (block structure laid out by the fac_streamer merge-and-split recursion under the fitted rank model -- not what the reference's truncated SVDs would give at this size; seeded values generated in HBM)"
        metric = "butterfly matvecs/sec (streamed real butterfly apply, examples/covariance)"
        top_rows, weights, row_offsets = [n], [total_leaf], np.array([0, n])
        sworld, srank = (args.emulate_world, max(args.emulate_rank, 0)) if args.emulate_world > 1 else (world, rank)
        mode, cuts = "rows", [0, n]
        owner, loads = [0], [total_leaf]
        if sworld > 1:
            cuts, loads = row_partition(desc, sworld)
            top_rows, owner = [cuts[r + 1] - cuts[r] for r in range(sworld)], list(range(sworld))
    else:
        n = args.n or 262144
        ncols = n
        k = args.k if args.k is not None else n / 16.0
        # the native layout (bfhip_layout.c) unless the Python restatement is asked for
        if args.python_layout:
            desc, _, perm = hs.helm2_multilevel_structure(hs.circle_points(n), k)
        else:
            desc, perm = hs.native_multilevel_structure(hs.circle_points(n), k)
        if real:
            desc.dtype = 1          # BFHIP_F64 leaves; --dtype f32 demotes them on upload
        weights = row_block_weights(desc)
        total_leaf = int(sum(weights))
        top_rows = desc.meta["top_rows"]
        row_offsets = np.concatenate([[0], np.cumsum(top_rows)]).astype(np.int64)
        sworld, srank = (args.emulate_world, max(args.emulate_rank, 0)) if args.emulate_world > 1 else (world, rank)
        mode = choose_mode(desc, sworld, args.shard)
        if use_pg:
            # every rank chose on its own copy of the layout: they must have chosen the same way before anything is compiled
            modes = [None] * world
            dist.all_gather_object(modes, mode)
            if any(m_ != modes[0] for m_ in modes):
                if rank == 0:
                    log(f"bench.py: the ranks chose different sharding modes {modes}: refusing to run")
                dist.destroy_process_group()
                sys.exit(3)
        cuts = [0, n]
        seg_rows = top_rows                     # the row segments of y the closing all-gather carries, in global order
        if mode == "rows":
            # contiguous row ranges, cut one level or more below the top-level blocks: one segment per rank
            if sworld > 1:
                cuts, loads = row_partition(desc, sworld)
            else:
                loads = [total_leaf]
            seg_rows, owner = [cuts[r + 1] - cuts[r] for r in range(sworld)], list(range(sworld))
        elif mode == "rowblocks":
            owner, loads = assign_row_blocks(weights, sworld)
        elif mode == "rowsum":
            bowner, loads, rsegs = rowsum_partition(desc, sworld)
            owner = [0] * len(weights)
        else:
            bw = block_weights(desc)
            bowner, loads = assign_row_blocks(bw, sworld)
            owner = [0] * len(weights)
        workload = (f"fac_helm2 multilevel butterfly, unit circle, N={n}, k={k:g} (16 ppw), nrhs={args.nrhs}"
                    + ("" if not real else f" [real-valued {dtype} proxy on the same block layout]"))
        config = {"workload": workload, "n": n, "k": k, "nrhs": args.nrhs, "leaf_bytes": total_leaf * esz}
        data = "synthetic (structure-exact fac_helm2 layout, seeded values generated in HBM)"
        metric = "butterfly matvecs/sec (2D Helmholtz HODBF apply)"
    if args.cpu_budget_gb is None:      # >= 25 % of the leaf bytes (the oracle works in the reference's double precision), bounded so that the line stays within minutes
        # (whole top-level block rows are sampled, 1/12 of the operand each on a circle: a third of the bytes as budget is four of them)
        args.cpu_budget_gb = min(max(0.33 * total_leaf * (16 if not real else 8) / 1e9, 4.0), 24.0 if not streamer else 40.0)
    t_struct = time.time() - t0
    if rank == 0:
        log(f"structure: {workload}; nodes={desc.num_nodes} leafGB={total_leaf * esz / 1e9:.2f} [{t_struct:.1f}s]; "
            f"shard mode={mode}; rank loads GB={[round(l * esz / 1e9, 2) for l in loads]}; "
            f"max/mean = {max(loads) / (sum(loads) / len(loads)):.3f}, replication = {sum(loads) / total_leaf:.3f}")

    if dry:
        # every rank laid the operand out and partitioned it on its own: they must agree before anything is compiled
        mine_sig = [float(c) for c in cuts] + [float(v) for v in loads]
        same = True
        if use_pg:
            sigs = [None] * world
            dist.all_gather_object(sigs, mine_sig)
            same = all(sg == sigs[0] for sg in sigs)
        # the checks a real multi-rank run makes about itself, on CPU tensors over gloo: (a) every rank ends with the same
        # y (BENCH_DRY_PERTURB_RANK=r makes rank r's copy differ), (b) the collective decision between libbfhip's own RCCL
        # communicator and torch.distributed (BENCH_DRY_FAIL_RCCL_RANK=r: creation "fails" on rank r only)
        yd = torch.arange(1000, dtype=torch.float64).to(torch.complex128) * (1 + 2j)
        if os.environ.get("BENCH_DRY_PERTURB_RANK") == str(rank):
            yd[17] += 1e-3
        ydiff = ranks_max_abs_diff(dist, yd, use_pg)
        fail_rank = os.environ.get("BENCH_DRY_FAIL_RCCL_RANK")
        ok_all, bad = all_ranks_ok(dist, fail_rank != str(rank), "simulated bfhipCommInitRank failure", use_pg)
        prof_len = int(pad_to_common_length(dist, torch.zeros(3 * (5 + rank)), use_pg).numel())
        rows_alt = None
        if world > 1 and mode != "rows" and not streamer:
            c2, l2 = row_partition(desc, world)
            rows_alt = {"cuts": [int(c) for c in c2], "replication": sum(l2) / total_leaf, "imbalance": max(l2) / (sum(l2) / len(l2))}
        if rank == 0:
            real_stdout.write(json.dumps({"dry_run": True, "n_gpus": world, "mode": mode, "cuts": [int(c) for c in cuts],
                                          "rank_leaf_gb": [l * esz / 1e9 for l in loads], "imbalance": max(loads) / (sum(loads) / len(loads)),
                                          "replication": sum(loads) / total_leaf, "ranks_agree": same, "workload": workload,
                                          "multi_gpu": {"ranks_max_abs_diff": ydiff, "ranks_agree": ydiff == 0.0,
                                                        "collective_impl": "libbfhip (RCCL)" if ok_all else "torch.distributed",
                                                        "collective_fallback_reason": bad, "padded_profile_len": prof_len,
                                                        "also_timed": ({"mode": "rows", **rows_alt} if rows_alt else None),
                                                        "shared_block_rows": sorted({rb for rb, _ in rsegs if sum(1 for b2, _ in rsegs if b2 == rb) > 1}) if mode == "rowsum" else []}}) + "\n")
            real_stdout.flush()
        if use_pg:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(0 if same and ydiff == 0.0 else 3)

    def compile_shard(r, max_rhs, mode=None, cuts=None):
        """The operator rank r of an `sworld`-rank job holds (mode / cuts: another way to deal it than the line's own)."""
        mode = mode or mode_main
        cuts = cuts or cuts_main
        rr = None
        if sworld == 1:
            root, rows = desc.root, n
        elif mode == "rows":
            root, rows, rr = desc.root, cuts[r + 1] - cuts[r], (cuts[r], cuts[r + 1])
        elif mode == "rowblocks":
            mine_r = [rb for rb in range(len(weights)) if owner[rb] == r]
            root, rows = hs.shard_desc(desc, mine_r)
        elif mode == "rowsum":
            root, _, rows = hs.shard_desc_children(desc, [i for i in range(len(bowner)) if bowner[i] == r])
        else:
            mine_r = [i for i in range(len(bowner)) if bowner[i] == r]
            root, rows = hs.shard_desc_blocks(desc, mine_r), n
        o = HipOperator.from_desc(desc, None, root=root, device=local_rank, flags=flags, seed=args.seed, max_rhs=max_rhs,
                                  demote_to_f32=(dtype == "f32"), row_range=rr)
        return o, rows

    mode_main, cuts_main = mode, cuts
    t0 = time.time()
    op, local_rows = compile_shard(srank, args.nrhs)
    torch.cuda.synchronize()
    t_compile = time.time() - t0
    st = op.stats()
    if rank == 0:
        log(f"compile+synthesize: {t_compile:.1f}s stages={st['numStages']} items={st['numItems']} pieces={st['numPieces']} "
            f"arenaGB={st['arenaBytes'] / 1e9:.2f} metaMB={st['metaBytes'] / 1e6:.1f} tempElems={st['tempElems']}")

    # x: seeded normal, identical on every rank (replicated input)
    rng = np.random.default_rng(args.seed)
    shape = (ncols,) if args.nrhs == 1 else (ncols, args.nrhs)
    if real:
        x_host = rng.standard_normal(shape)
    else:
        x_host = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2)
    x = torch.from_numpy(x_host).to(dev).to(tdtype)

    class TorchStep:
        """The same step with the closing collective issued by torch.distributed (still RCCL over xGMI on GPUs): the
        fallback when libbfhip's own communicator cannot be created on some rank.  Times its two halves with events."""
        def __init__(self, the_op, layout_, mode_):
            self.op = the_op
            self.inner = ShardedApply(layout_, rank, lambda xin, out: the_op.apply_device(xin, out), dev, tdtype, nrhs=args.nrhs, mode="rows" if mode_ == "rowsum" else mode_,
                                      force_collective=args.force_collective)
            self.timing, self.ev = False, [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            inner_apply = self.inner.local_apply

            def timed_local(xin, out):
                r_ = inner_apply(xin, out)
                if self.timing:
                    self.ev[1].record()
                return r_
            self.inner.local_apply = timed_local

        def __call__(self, xin):
            if self.timing:
                self.ev[0].record()
            y_ = self.inner(xin)
            if self.timing:
                self.ev[2].record()
            return y_

        def set_timing(self, on):
            self.timing = bool(on)

        def apply_transpose(self, v):
            return self.inner.apply_transpose(v, lambda vr, out: self.op.apply_transpose_device(vr, out), ncols)

        def last_times(self):
            torch.cuda.synchronize()
            return self.ev[0].elapsed_time(self.ev[1]), self.ev[1].elapsed_time(self.ev[2])

        def close(self):
            pass

    def make_layout(mode_, cuts_):
        if mode_ == "rowsum":
            return ShardLayout(top_rows, owner, world, segments=rsegs)
        if mode_ == "rows" and not streamer:
            return ShardLayout([cuts_[r + 1] - cuts_[r] for r in range(world)], list(range(world)), world)
        return ShardLayout(seg_rows if not streamer else top_rows, owner, world)

    def make_sharded(the_op, mode_, cuts_):
        """(step, which implementation issues the collective, [(rank, why)] if libbfhip's communicator was given up).
        Every rank takes the same branch: the outcome of the attempt is agreed on through torch's process group."""
        layout_ = make_layout(mode_, cuts_)

        def bcast(payload):
            box = [payload]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        sh_, err = None, ""
        if os.environ.get("BENCH_FORCE_TORCH_COLLECTIVE") == "1":
            err = "BENCH_FORCE_TORCH_COLLECTIVE=1"
        else:
            try:
                sh_ = RcclShardedApply(layout_, rank, the_op, local_rank, nrhs=args.nrhs, mode=mode_, bcast=bcast if world > 1 else None)
            except Exception as e:        # a rank that cannot create its communicator must not leave the others waiting in a collective
                err = repr(e)
        ok_all, bad = all_ranks_ok(dist, sh_ is not None, err, use_pg)
        if ok_all:
            return sh_, "libbfhip (RCCL, dlopen)", []
        if sh_ is not None:
            sh_.close()
        if rank == 0:
            log("bench.py: libbfhip's RCCL communicator unavailable on", bad, "-> the collective goes through torch.distributed")
        return TorchStep(the_op, layout_, mode_), "torch.distributed", bad

    sharded, coll_impl, coll_bad = None, None, []
    if (world > 1 or args.force_collective) and args.emulate_world <= 1:
        sharded, coll_impl, coll_bad = make_sharded(op, mode, cuts)
        step = sharded
    else:
        y_buf = torch.empty((local_rows,) + shape[1:], dtype=tdtype, device=dev)

        def step(xin):
            return op.apply_device(xin, y_buf)

    for _ in range(args.warmup):
        y_full = step(x)
    torch.cuda.synchronize()
    op.stage_profile(reset=True)
    # an event pair between two launches costs ~10 us of stream time (kernels without one in between start back to
    # back): bracket four applies of the timed region, the others run as a caller's would
    ev_every = max(4, args.steps // 4) if args.steps >= 8 else 1
    op.set_profile_sampling(ev_every)
    if sharded is not None:
        sharded.set_timing(False)          # the local / collective split is measured after the timed region
    if use_pg:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y_full = step(x)
    enqueue_s = time.perf_counter() - t0          # host time to enqueue all steps (launches are asynchronous)
    torch.cuda.synchronize()
    if use_pg:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_pg:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms, launches, sbytes = op.stage_profile()
    op.set_profile_sampling(1)

    # multi-GPU: per-rank leaf bytes, the slowest rank's local stages, and the collective, separately
    # (a few extra steps outside the timed region: reading the events synchronizes the host)
    multi = None
    prof_rank = rank
    if sharded is not None:
        sharded.set_timing(True)
        loc, coll = [], []
        for _ in range(5):
            step(x)
            a, b = sharded.last_times()
            loc.append(a); coll.append(b)
        # this rank's stage profile (kernel ms per apply, launches, algorithmic bytes per apply) travels too: the line's
        # roofline is the SLOWEST rank's, the one that bounds the step
        per_apply_ms = float(ms.sum()) / max(float(launches.max()), 1.0)
        v = torch.tensor([float(np.median(loc)), float(np.median(coll)), st["leafBytes"] / 1e9, per_apply_ms], dtype=torch.float64, device=dev)
        # (ranks may have compiled plans of different depth: each of the three per-stage vectors is padded to the longest)
        parts = [pad_to_common_length(dist, torch.tensor(np.asarray(a_, dtype=np.float64), dtype=torch.float64, device=dev), use_pg)
                 for a_ in (ms, launches, sbytes)]
        prof = torch.cat(parts)
        if use_pg:
            allv = [torch.zeros_like(v) for _ in range(world)]
            dist.all_gather(allv, v)
            allp = [torch.zeros_like(prof) for _ in range(world)]
            dist.all_gather(allp, prof)
        else:
            allv, allp = [v], [prof]
        allv = torch.stack(allv).cpu().numpy()
        allp = torch.stack(allp).cpu().numpy()
        prof_rank = int(np.argmax(allv[:, 3]))
        S_ = int(parts[0].numel())
        ms, launches, sbytes = allp[prof_rank, :S_], allp[prof_rank, S_:2 * S_].astype(np.uint64), allp[prof_rank, 2 * S_:].astype(np.uint64)
        # every rank must have ended with the same y
        ydiff = ranks_max_abs_diff(dist, step(x), use_pg)
        gb = allv[:, 2]
        multi = {"mode": mode, "collective": "ncclAllReduce (sum)" if mode == "blocks" else "ncclAllGather (in place) + segment reorder",
                 "rank_leaf_gb": [round(float(g), 3) for g in gb], "imbalance": float(gb.max() / gb.mean()),
                 "replication": float(gb.sum() * 1e9 / (total_leaf * esz)),
                 "rank_local_ms": [round(float(r[0]), 4) for r in allv], "rank_kernel_ms": [round(float(r[3]), 4) for r in allv],
                 "max_local_ms": float(allv[:, 0].max()), "collective_ms_per_rank": [round(float(r[1]), 4) for r in allv],
                 "max_collective_ms": float(allv[:, 1].max()), "collective_bytes": int(n * args.nrhs * esz),
                 "roofline_rank": prof_rank, "bit_identical_to_one_gpu": mode in ("rows", "rowblocks"),
                 "ranks_max_abs_diff": ydiff, "ranks_agree": ydiff == 0.0,
                 "collective_impl": coll_impl, "collective_fallback_reason": coll_bad}
        if args.adjoint:
            # the adjoint of the sharded step (bfhipShardedApplyTransposeDevice): every rank applies A_r^T to its entries of v, ONE all-reduce
            try:
                vt = torch.from_numpy(rng.standard_normal((n,) + shape[1:]) if real else x_host).to(dev).to(tdtype) if ncols == n else None
                if vt is None:
                    vt = torch.from_numpy(np.random.default_rng(args.seed + 1).standard_normal((n,) + shape[1:])).to(dev).to(tdtype)
                for _ in range(2):
                    zt = sharded.apply_transpose(vt)
                torch.cuda.synchronize()
                if use_pg:
                    dist.barrier()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    zt = sharded.apply_transpose(vt)
                torch.cuda.synchronize()
                if use_pg:
                    dist.barrier()
                elT = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
                if use_pg:
                    dist.all_reduce(elT, op=dist.ReduceOp.MAX)
                dT = ranks_max_abs_diff(dist, zt, use_pg)
                # <A x, v> = <x, A^T v> ties the sharded adjoint to the sharded forward step on the full-size operand
                yx = step(x)
                lhs = torch.sum(yx.to(torch.complex128 if not real else torch.float64) * vt.to(torch.complex128 if not real else torch.float64))
                rhs = torch.sum(x.to(lhs.dtype) * zt.to(lhs.dtype))
                multi["adjoint"] = {"ms_per_step": float(elT.item()) / args.steps * 1e3, "value": args.steps * args.nrhs / float(elT.item()), "unit": "matvec/s",
                                    "collective": "ncclAllReduce (sum) of the full-length partials A_r^T v_r", "ranks_max_abs_diff": dT, "ranks_agree": dT == 0.0,
                                    "transpose_identity_rel": float(abs(lhs - rhs) / max(abs(lhs), 1e-300)),
                                    "layout": "shared leaves (BFHIP_FLAG_ADJOINT): a shard's adjoint plan is its transposed task list pruned by reachability from its rows"}
            except Exception as e:
                multi["adjoint"] = {"error": repr(e)[:300]}
        # The default at this world size is not the bit-identical row-range shard (8 ranks: rowsum): time that one too
        if ((world > 1 and args.shard == "auto") or os.environ.get("BENCH_ALSO_TIME_ROWS") == "1") and mode != "rows" and not streamer:
            try:
                cuts2, loads2 = row_partition(desc, world)
                sharded.close()
                op.close()
                op, _rows2 = compile_shard(rank, args.nrhs, mode="rows", cuts=cuts2)
                sharded, impl2, bad2 = make_sharded(op, "rows", cuts2)
                step = sharded
                sharded.set_timing(False)
                op.set_profile_sampling(1 << 30)
                for _ in range(max(args.warmup, 2)):
                    step(x)
                torch.cuda.synchronize()
                if use_pg:
                    dist.barrier()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    y2 = step(x)
                torch.cuda.synchronize()
                if use_pg:
                    dist.barrier()
                el2 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
                if use_pg:
                    dist.all_reduce(el2, op=dist.ReduceOp.MAX)
                d2 = ranks_max_abs_diff(dist, y2, use_pg)
                multi["also_timed"] = {"mode": "rows", "bit_identical_to_one_gpu": True, "ms_per_step": float(el2.item()) / args.steps * 1e3,
                                       "value": args.steps * args.nrhs / float(el2.item()), "unit": "matvec/s",
                                       "replication": float(sum(loads2) / total_leaf), "imbalance": float(max(loads2) / (sum(loads2) / len(loads2))),
                                       "ranks_max_abs_diff": d2, "ranks_agree": d2 == 0.0, "collective_impl": impl2}
            except Exception as e:
                multi["also_timed"] = {"mode": "rows", "error": repr(e)[:300]}

    if rank == 0:
        kern_ms = float(ms.sum())
        n_launch = int(launches.sum())
        applies = int(launches.max()) if len(launches) else 0          # sampled applies (every stage that launches does so once per apply)
        per_apply = int((launches > 0).sum())                           # launches per apply: 1 when the plan runs as one dependency-driven launch
        bytes_per_apply = int(sbytes.sum())
        avg_launch_ms = kern_ms / max(n_launch, 1)
        achieved = bytes_per_apply * applies / 1e9 / (kern_ms / 1e3) if kern_ms > 0 else 0.0
        one_launch = (not real) and args.nrhs < 2 and op is not None and op.flow_status()[0]
        for s in range(len(ms)):
            if launches[s]:
                log(f"  stage {s}: {ms[s] / max(launches[s], 1):8.3f} ms/launch  {sbytes[s] / 1e9:8.3f} GB  "
                    f"{(sbytes[s] / 1e9) / (ms[s] / max(launches[s], 1) / 1e3) if ms[s] > 0 else 0:8.1f} GB/s" + ("   [all stages: one launch]" if one_launch else ""))
        if args.nrhs >= 2 and not real:
            # block of right-hand sides: 8*nrhs flops per leaf element (32 flop/B at nrhs=64) -> FP64-MFMA bound
            flops_per_apply = 8.0 * args.nrhs * st["leafElems"]
            tf = flops_per_apply * applies / 1e12 / (kern_ms / 1e3) if kern_ms > 0 else 0.0
            roofline = {"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tf / FP64_MFMA_PEAK_TFLOPS, "traffic": None, "kernel": "bfStageKernelC128Mfma",
                        "launches_per_apply": per_apply, "avg_launch_ms": avg_launch_ms,
                        "algorithmic_flops_per_apply": flops_per_apply, "hbm_gbs_algorithmic": achieved,
                        "algorithmic_bytes_per_apply": bytes_per_apply,
                        "kernel_ms_per_apply": kern_ms / max(launches.max(), 1),
                        "executed_mfma_flops_per_apply": 6.0 * args.nrhs * st["leafElems"],
                        "note": "algorithmic flops = 8 nrhs sum(m n) (SURVEY 8(d)); the kernel forms each complex product with Gauss's 3 real "
                                "multiplications: 6 nrhs sum(m n) MFMA flops issued (+ tile padding) -- the ceiling of `frac` is 4/3, values above 1 are not an error",
                        "event_sampling": f"HIP events around every launch of 1 apply in {ev_every} of the timed region"}
        else:
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                        "kernel": ("bfFlowKernelC128" if one_launch else "bfStageKernelC128") if not real else f"bfStageKernelRealBoth<{dtype}>",
                        "launches_per_apply": per_apply,
                        **({"rank": prof_rank, "note": "the slowest rank's stage kernels (it bounds the step); bytes = that rank's shard"} if multi else {}),
                        "avg_launch_ms": avg_launch_ms, "algorithmic_bytes_per_launch": bytes_per_apply / max(per_apply, 1),
                        "algorithmic_bytes_per_apply": bytes_per_apply, "kernel_ms_per_apply": kern_ms / max(launches.max(), 1),
                        "event_sampling": f"HIP events around every launch of 1 apply in {ev_every} of the timed region"}
        # `traffic` = HBM bytes PER LAUNCH (like `achieved`), from the COMMITTED rocprofv3 --pmc profile of this
        # very command (FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 FETCH correction): counters cannot
        # be read from inside the process, so it is not measured in this run and other configurations report null.
        for rnd in (PROFILE_ROUND, "r4", "r3"):
            prof = os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary.json")
            helm_cfg = (not streamer and n == 262144 and abs(k - 16384) < 1e-9 and args.nrhs in (1, 64) and not real)
            cfg2 = (not streamer and n == 65536 and abs(k - 4096) < 1e-9 and args.nrhs == 1 and not real)
            strm_cfg = (streamer and n == 1048576 and args.lmax == 255 and dtype == "f32" and args.nrhs == 1 and args.freq_depth is None)
            if not (world == 1 and (helm_cfg or strm_cfg or cfg2) and args.emulate_world <= 1 and os.path.exists(prof)) or one_launch:
                continue
            try:
                key = ("bfStageKernelReal_f32_streamer_per_launch" if strm_cfg else "bfStageKernelC128_n65536_per_launch" if cfg2 else
                       "bfStageKernelC128_per_launch" if args.nrhs == 1 else "bfStageKernelC128Mfma_per_launch")
                pm = json.load(open(prof))[key]
                roofline["traffic"] = pm["hbm_bytes"]
                roofline["traffic_per_apply"] = pm["hbm_bytes"] * pm.get("launches_per_apply", len(ms))
                roofline["traffic_source"] = (f"profiles/{rnd}_pmc_summary.json [{key}]: from the committed profile of this command "
                                              "(rocprofv3 --pmc, separate passes), NOT measured in this run; per launch, as `achieved`")
                break
            except Exception:
                pass
        # context for `frac` (informational; `peak` stays the guide's figure): what a pure stream of
        # non-temporal reads / a bare loop of these MFMAs reaches on an MI355X of this pool
        for key, fn, fld in (("measured_stream_read_gbs", "hbm_peak.json", "read_nt_gbs"), ("measured_mfma_loop_tflops", "mfma_peak.json", "fp64_mfma_16x16x4_tflops")):
            if (roofline["bound"] == "hbm") == (fld == "read_nt_gbs"):
                for rnd in (PROFILE_ROUND, "r4", "r3"):
                    try:
                        roofline[key] = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_{fn}")))[fld]
                        break
                    except Exception:
                        pass
        if args.pcie:
            # host-buffer path (bfhipApply, what the vtable shim's Mul runs): never the headline value.  Three kinds of caller
            # memory (include/bfhip.h): pageable (packed through pinned staging), registered (DMA straight from / to it), device
            pcie = {}
            hx = np.ascontiguousarray(x_host.astype(np.complex128 if not real else np.float64))
            hy = np.empty((local_rows,) + shape[1:], dtype=hx.dtype)
            reps = 10

            def timed(fn):
                fn(); fn()
                t1_ = time.perf_counter()
                for _ in range(reps):
                    fn()
                return (time.perf_counter() - t1_) / reps * 1e3
            pcie["pageable_ms"] = timed(lambda: op.apply_host_into(hx, hy, args.nrhs))
            try:
                HipOperator.host_register(hx); HipOperator.host_register(hy)
                pcie["registered_ms"] = timed(lambda: op.apply_host_into(hx, hy, args.nrhs))
                HipOperator.host_unregister(hx); HipOperator.host_unregister(hy)
            except Exception as e:
                pcie["registered_error"] = repr(e)[:200]
            if dtype != "f32":
                yd_ = torch.empty((local_rows,) + shape[1:], dtype=tdtype, device=dev)

                def dev_call():
                    op.apply_pointers(x.data_ptr(), yd_.data_ptr(), args.nrhs)
                pcie["device_pointers_ms"] = timed(dev_call)
            t1 = time.perf_counter()
            for _ in range(reps):
                op.apply_device(x, y_buf if sharded is None else None)
            torch.cuda.synchronize()
            pcie["resident_ms_same_moment"] = (time.perf_counter() - t1) / reps * 1e3
            pcie_ms = pcie["pageable_ms"]
        config["stages"] = st["numStages"]
        # what this configuration keeps in HBM: leaf arena(s) (both packed copies with the packed adjoint), vector arena, index tables
        config["resident_bytes"] = int(st["arenaBytes"] + st["tempElems"] * args.nrhs * esz + st["metaBytes"])
        config["arena_bytes"] = int(st["arenaBytes"])
        config["sharding"] = ("none" if sworld == 1 else
                              "contiguous row ranges below the top-level blocks (bfhipRowPartition: balanced, source-side factors replicated) + one all-gather" if mode == "rows"
                              else "top-level row blocks (LPT by leaf bytes) + one all-gather" if mode == "rowblocks"
                              else "whole top-level block rows + column shares of the rest (no replication) + one all-gather, shared rows' partials added in rank order" if mode == "rowsum"
                              else "top-level (row, col) blocks (LPT by leaf bytes) + one all-reduce")
        out = {
            "metric": metric,
            "value": args.steps * args.nrhs / elapsed,
            "unit": "matvec/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",      # the operator is fixed: more GPUs share it
            "vs_baseline": None,
            "dtype": dtype,
            "data": data,
            "config": config,
            "roofline": roofline,
            # leaf bytes this process streamed per step / step time (one rank's share in a multi-GPU or emulated run)
            "hbm_gbs_whole_step": (st["leafBytes"] / 1e9) / (elapsed / args.steps),
            # what a hipGraph could remove at most: the host's share of a step (it runs ahead of the GPU)
            "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3,
        }
        if multi:
            out["multi_gpu"] = multi
        if args.adjoint and world == 1 and args.emulate_world <= 1:      # (an emulated shard's adjoint partial is timed with the shards, below)
            xt = torch.from_numpy(rng.standard_normal((local_rows,) + shape[1:]) if real else x_host).to(dev).to(tdtype)
            for _ in range(2):
                yt = op.apply_transpose_device(xt)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                yt = op.apply_transpose_device(xt)
            torch.cuda.synchronize()
            adj_ms = (time.perf_counter() - t1) / args.steps * 1e3
            out["adjoint"] = {"ms_per_apply": adj_ms, "matvec_per_s": args.nrhs / (adj_ms / 1e3),
                              "hbm_gbs": st["leafBytes"] / 1e9 / (adj_ms / 1e3), "frac_of_hbm_peak": st["leafBytes"] / 1e9 / (adj_ms / 1e3) / HBM_PEAK_GBS,
                              "traffic_per_apply": None,
                              "layout": ("shared: the forward plan's packed leaves read by the transposed kernels (BFHIP_FLAG_ADJOINT)" if shared_adj else
                                         "packed: a second copy of the leaves laid out for A^T, applied by the forward kernels (BFHIP_FLAG_ADJOINT_PACKED; "
                                         f"{2 * st['leafBytes'] / 1e9:.1f} GB of leaves resident)")}
            try:        # HBM bytes per transposed apply from the committed --pmc passes of this command (all bfStageKernelT launches of one apply)
                pm, pm_round = load_pmc_summary()
                akey = ("bfStageKernelT_f32_streamer_adjoint_per_apply" if (streamer and n == 1048576 and args.lmax == 255 and dtype == "f32") else
                        "bfStageKernelT_c128_adjoint_per_apply" if (not streamer and n == 262144 and not real and abs(k - 16384) < 1e-9) else None)
                if akey and pm.get(akey) and args.nrhs == 1 and args.adjoint_shared:
                    out["adjoint"]["traffic_per_apply"] = pm[akey]["hbm_bytes"]
                    out["adjoint"]["traffic_ratio"] = pm[akey]["ratio"]
                    out["adjoint"]["traffic_source"] = f"profiles/{pm_round}_pmc_summary.json [{akey}]: committed rocprofv3 --pmc passes, not measured in this run"
            except Exception:
                pass
            # <A x, v> = <x, A^T v>: ties the two plans together on the full-size operand
            yx = op.apply_device(x)
            lhs = torch.sum(yx.to(torch.complex128 if not real else torch.float64) * xt.to(torch.complex128 if not real else torch.float64))
            rhs = torch.sum(x.to(lhs.dtype) * yt.to(lhs.dtype))
            out["adjoint"]["transpose_identity_rel"] = float(abs(lhs - rhs) / max(abs(lhs), 1e-300))
            if streamer and args.nrhs == 1:
                # the path's caller: cov_matvec of examples/covariance/lbo_cov.c:48-60 as one device call
                # (permute, A^T, GammaLam twice, A, permute), with a random diagonal and row permutation
                gam = torch.rand(ncols, dtype=tdtype, device=dev) + 0.5
                perm = torch.randperm(n, device=dev)
                rev = torch.empty_like(perm); rev[perm] = torch.arange(n, device=dev)
                for _ in range(2):
                    zc = op.cov_matvec_device(gam, perm, rev, xt)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    zc = op.cov_matvec_device(gam, perm, rev, xt)
                torch.cuda.synchronize()
                cov_ms = (time.perf_counter() - t1) / args.steps * 1e3
                # check against the two applies done separately
                t_ = torch.empty_like(xt); t_[rev] = xt
                u_ = op.apply_transpose_device(t_).clone() * gam * gam
                z_ = op.apply_device(u_)
                zr = torch.empty_like(z_); zr[perm] = z_
                out["cov_matvec"] = {"ms_per_product": cov_ms, "products_per_s": 1e3 / cov_ms,
                                     "hbm_gbs": 2 * st["leafBytes"] / 1e9 / (cov_ms / 1e3),
                                     "rel_vs_separate_applies": float(torch.linalg.norm(zc - zr) / torch.linalg.norm(zr)),
                                     "note": "bfhipCovMatvecDevice: z = P A diag(g)^2 A^T P' v, everything resident"}
            if args.adjoint_both and not shared_adj:
                # the same adjoint on the SHARED leaves (BFHIP_FLAG_ADJOINT: index tables only, no second copy): what a sharded
                # operator, or one that fills most of the HBM, has to use.  The operator is compiled again without the packed copy.
                try:
                    st_packed = st
                    op.close()
                    torch.cuda.empty_cache()
                    op = HipOperator.from_desc(desc, None, root=desc.root, device=local_rank, flags=_capi.FLAG_PROFILE | _capi.FLAG_ADJOINT, seed=args.seed,
                                               max_rhs=args.nrhs, demote_to_f32=(dtype == "f32"))
                    for _ in range(2):
                        yt2 = op.apply_transpose_device(xt)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        yt2 = op.apply_transpose_device(xt)
                    torch.cuda.synchronize()
                    adj2 = (time.perf_counter() - t1) / args.steps * 1e3
                    out["adjoint_shared"] = {"ms_per_apply": adj2, "matvec_per_s": args.nrhs / (adj2 / 1e3), "hbm_gbs": st["leafBytes"] / 1e9 / (adj2 / 1e3),
                                             "frac_of_hbm_peak": st["leafBytes"] / 1e9 / (adj2 / 1e3) / HBM_PEAK_GBS,
                                             "rel_vs_packed": float(torch.linalg.norm(yt2.to(yt.dtype) - yt) / torch.linalg.norm(yt)),
                                             "arena_bytes": int(op.stats()["arenaBytes"]), "arena_bytes_packed": int(st_packed["arenaBytes"]),
                                             "layout": "shared: the forward plan's packed leaves read by the transposed kernels (BFHIP_FLAG_ADJOINT), no extra leaf memory"}
                except Exception as e:
                    out["adjoint_shared"] = {"error": repr(e)[:300]}
        if args.pcie:
            out["pcie_inclusive"] = {"ms_per_apply": pcie_ms, "matvec_per_s": args.nrhs / (pcie_ms / 1e3), **pcie,
                                     "ratio_pageable": pcie_ms / pcie["resident_ms_same_moment"],
                                     "ratio_registered": (pcie["registered_ms"] / pcie["resident_ms_same_moment"]) if "registered_ms" in pcie else None,
                                     "note": "bfhipApply on host vectors, synchronous (what the vtable shim's Mul runs): pageable = pack into pinned staging + H2D + "
                                             "apply + D2H + unpack; registered = DMA straight from / to the caller's registered buffers (bfhipHostRegister); "
                                             "device_pointers = the same entry handed device memory (used in place); resident = bfhipApplyDevice, timed at the same moment"}
        if args.emulate_world > 1:
            out["emulated_shard"] = {"world": args.emulate_world, "rank": srank, "mode": mode, "shard_leaf_bytes": st["leafBytes"]}
        # rank 0 times the CPU baseline whatever the world size (the other ranks wait at the closing barrier)
        if not args.no_cpu_baseline and args.emulate_world <= 1:
            try:
                if streamer:
                    out["cpu_baseline"] = cpu_baseline_streamer(desc, args.seed, args.cpu_budget_gb * 1e9)
                elif not real:
                    # rowsum mode: the sample starts with a block row that two ranks share (its result is a sum of partials)
                    shared_rb = (sorted({rb for rb, _ in rsegs if sum(1 for b2, _ in rsegs if b2 == rb) > 1}, key=lambda rb: weights[rb])[:1]
                                 if (multi and mode == "rowsum") else [])
                    out["cpu_baseline"] = cpu_baseline_helm2(desc, args.seed, total_leaf, weights, args.cpu_budget_gb * 1e9, args.nrhs, x_host,
                                                             y_full.cpu().numpy(), row_offsets, first=shared_rb)
                    if shared_rb:
                        out["cpu_baseline"]["parity_covers_shared_block_row"] = int(shared_rb[0])
            except Exception as e:  # the baseline must never take the measurement down
                out["cpu_baseline"] = {"value": None, "unit": "matvec/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}

    # every rank's shard of an N-rank job, one after another on this GPU (the slowest bounds the job)
    if args.emulate_world > 1 and args.emulate_rank < 0 and rank == 0:
        times = []
        op.close()
        for r in range(args.emulate_world):
            o, rows = compile_shard(r, args.nrhs)
            o.set_profile_sampling(1 << 30)      # no events inside the timed loop (a pair per launch is 10 us: 5 % of a 1.5 ms shard)
            yb = torch.empty((rows,) + shape[1:], dtype=tdtype, device=dev)
            for _ in range(2):
                o.apply_device(x, yb)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                o.apply_device(x, yb)
            torch.cuda.synchronize()
            times.append({"rank": r, "leaf_gb": o.stats()["leafBytes"] / 1e9, "ms_per_apply": (time.perf_counter() - t1) / args.steps * 1e3})
            if args.adjoint:
                # the rank's part of the sharded adjoint step: A_r^T on its rows of v (shared leaves), before the one all-reduce
                vr = torch.randn((rows,) + shape[1:], dtype=tdtype, device=dev)
                zb = torch.empty(shape, dtype=tdtype, device=dev)
                for _ in range(2):
                    o.apply_transpose_device(vr, zb)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    o.apply_transpose_device(vr, zb)
                torch.cuda.synchronize()
                times[-1]["adjoint_ms_per_apply"] = (time.perf_counter() - t1) / args.steps * 1e3
                del vr, zb
            o.close()
            del yb
        out["emulated_shard"]["all_ranks"] = times
        out["emulated_shard"]["slowest_ms"] = max(t["ms_per_apply"] for t in times)
        if all("adjoint_ms_per_apply" in t for t in times):
            out["emulated_shard"]["adjoint_slowest_ms"] = max(t["adjoint_ms_per_apply"] for t in times)
        op = None

    # BASELINE configs[2] rides along on the default line: the same operand applied to 64 right-hand sides
    # (FP64 MFMA kernel), timed after the headline loop so that the driver's run carries it
    if (rank == 0 and world == 1 and not streamer and not real and args.nrhs == 1 and not args.no_extra and args.emulate_world <= 1
            and n == 262144):
        try:
            op.close()
            op = None
            nr = 64
            o64 = HipOperator.from_desc(desc, None, root=desc.root, device=local_rank, flags=_capi.FLAG_PROFILE, seed=args.seed, max_rhs=nr)
            x64_host = (rng.standard_normal((n, nr)) + 1j * rng.standard_normal((n, nr))) / np.sqrt(2)
            x64 = torch.from_numpy(x64_host).to(dev)
            y64 = torch.empty((n, nr), dtype=tdtype, device=dev)
            for _ in range(8):          # (the clock settles over the first few hundred ms of matrix-core load: early applies measure 1 - 3 % low)
                o64.apply_device(x64, y64)
            torch.cuda.synchronize()
            o64.stage_profile(reset=True)
            o64.set_profile_sampling(4)        # events around every launch of one apply in four
            reps = 20
            t1 = time.perf_counter()
            for _ in range(reps):
                o64.apply_device(x64, y64)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / reps
            ms64, l64, _ = o64.stage_profile()
            flops = 8.0 * nr * o64.stats()["leafElems"]
            sampled = max(int(l64.max()), 1)
            tf = flops * sampled / 1e12 / (float(ms64.sum()) / 1e3)
            out["nrhs64"] = {"config": "BASELINE configs[2]: the same operand, 64 right-hand sides (bfStageKernelC128Mfma)", "steps": reps,
                             "matvec_per_s": nr / dt, "ms_per_apply": dt * 1e3,
                             # the whole apply (stage kernels + the reduce launches of the last stage + launch gaps) against the same peak
                             "whole_apply_tflops": flops / 1e12 / dt, "whole_apply_frac": flops / 1e12 / dt / FP64_MFMA_PEAK_TFLOPS,
                             "resident_bytes": int(o64.stats()["arenaBytes"] + o64.stats()["tempElems"] * nr * 16 + o64.stats()["metaBytes"]),
                             "roofline": {"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TFLOPS,
                                          "kernel": "bfStageKernelC128Mfma", "kernel_ms_per_apply": float(ms64.sum()) / sampled,
                                          "algorithmic_flops_per_apply": flops,
                                          "executed_mfma_flops_per_apply": 6.0 * nr * o64.stats()["leafElems"],
                                          "note": "algorithmic flops = 8 nrhs sum(m n) (4 real multiply-adds per complex one, SURVEY 8(d)); the kernel forms each "
                                                  "complex product with Gauss's 3 real multiplications, i.e. issues 6 nrhs sum(m n) MFMA flops (+ tile padding): "
                                                  "the ceiling of this fraction is 4/3, values above 1 are not an error"}}
            try:        # the matrix pipe's share of the cycles and the clock the chip sustained under this kernel (committed --pmc pass)
                pm, pm_round = load_pmc_summary()
                clk = pm["mfma_sustained_clock"]
                out["nrhs64"]["roofline"].update({
                    "sustained_clock_ghz": clk["ghz"], "peak_clock_ghz": clk["peak_ghz"], "mfma_busy_fraction": clk["mfma_busy_fraction"],
                    "traffic": pm["bfStageKernelC128Mfma_per_launch"]["hbm_bytes"],
                    "clock_source": f"profiles/{pm_round}_pmc_summary.json [mfma_sustained_clock]: GRBM_GUI_ACTIVE / kernel time of the committed --pmc pass of "
                                    "`bench.py --nrhs 64`, NOT measured in this run; peak 78.6 TFLOP/s assumes 2.4 GHz"})
            except Exception:
                pass
            # the CPU baseline of THIS configuration, in the same run: the oracle's bfMatMul on 64 right-hand sides (1 thread = the reference's
            # behaviour, and OpenBLAS on every host core: SURVEY 8(d)), on a block-row sample, with the parity of the device result on it
            if not args.no_cpu_baseline:
                try:
                    y64_host = y64.cpu().numpy()
                    out["nrhs64"]["cpu_baseline"] = cpu_baseline_helm2(desc, args.seed, total_leaf, weights, min(args.cpu_budget_gb, 6.0) * 1e9, nr, x64_host,
                                                                       y64_host, row_offsets)
                    del y64_host
                except Exception as e:
                    out["nrhs64"]["cpu_baseline"] = {"value": None, "unit": "matvec/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}
            o64.close()
        except Exception as e:
            out["nrhs64"] = {"error": repr(e)}

        # BASELINE configs[1] (N = 65536, k = 4096, one right-hand side: launches of ~1 GB) in a child process: its own line
        try:
            torch.cuda.empty_cache()
            cmd = [sys.executable, os.path.abspath(__file__), "--npoints", "65536", "--steps", "50", "--warmup", "5", "--no-extra", "--pcie",
                   "--seed", str(args.seed)]
            env = dict(os.environ)
            for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k_, None)
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
            line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
            if pr.returncode != 0 or not line:
                raise RuntimeError(f"child exit {pr.returncode}: {pr.stderr[-300:]}")
            c = json.loads(line[-1])
            out["n65536"] = {"config": "BASELINE configs[1]: " + c["config"]["workload"], "value": c["value"], "unit": c["unit"], "dtype": c["dtype"],
                             "steps": c["steps"], "ms_per_step": c["ms_per_step"], "roofline": c["roofline"], "cpu_baseline": c.get("cpu_baseline"),
                             "pcie_inclusive": c.get("pcie_inclusive")}
        except Exception as e:
            out["n65536"] = {"error": repr(e)[:400]}

        # BASELINE configs[4] rides along too: the streamed real butterfly at N = 1M x 65536 columns, fp32, in a child
        # process of its own (the C layout needs seconds; bounded at 4.5 minutes, after which the key holds the error)
        if not args.no_streamer:
            try:
                torch.cuda.empty_cache()
                cmd = [sys.executable, os.path.abspath(__file__), "--workload", "streamer", "--steps", "10", "--warmup", "2", "--adjoint", "--adjoint-both",
                       "--no-extra", "--seed", str(args.seed)]
                env = dict(os.environ)
                for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                    env.pop(k, None)
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env)
                line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
                if pr.returncode != 0 or not line:
                    raise RuntimeError(f"child exit {pr.returncode}: {pr.stderr[-300:]}")
                c = json.loads(line[-1])
                out["configs4_streamer"] = {"config": c["config"]["workload"], "metric": c["metric"], "value": c["value"], "unit": c["unit"],
                                            "dtype": c["dtype"], "steps": c["steps"], "ms_per_step": c["ms_per_step"], "roofline": c["roofline"],
                                            "adjoint": c.get("adjoint"), "adjoint_shared": c.get("adjoint_shared"), "cov_matvec": c.get("cov_matvec"),
                                            "cpu_baseline": c.get("cpu_baseline"), "resident_bytes": c["config"].get("resident_bytes"),
                                            "layout": c["config"].get("layout")}
            except Exception as e:
                out["configs4_streamer"] = {"error": repr(e)[:400]}

    if rank == 0:
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if use_pg:
        dist.barrier()            # ranks > 0 wait here while rank 0 times the CPU baseline
    if sharded is not None:
        sharded.close()
    if op is not None:
        op.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
