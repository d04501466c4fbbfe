"""bench.py contract on a real GPU: one JSON line on stdout with the keys the
driver reads, on a small operand so it runs in seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def run_bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--npoints", "8192", *extra]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_default_contract_keys():
    d = run_bench("--cpu-budget-gb", "0.05")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "c128" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["parity_rel_l2"] < 1e-12
    assert abs(d["value"] - 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


def test_rhs_block_reports_mfma_roofline():
    d = run_bench("--nrhs", "64", "--no-cpu-baseline")
    r = d["roofline"]
    # (algorithmic flops over the peak: Gauss's three multiplications put the ceiling of this fraction at 4 / 3; the full-size
    #  operand reaches 0.97 - 1.02, this small one a fraction of that)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6 and 0 < r["frac"] < 4 / 3


def test_bench_starts_its_own_ranks(monkeypatch):
    """`bench.py --gpus N` without a launcher starts the rank processes itself (torch.distributed.run on
    127.0.0.1) and relays rank 0's JSON line; here with one rank (a one-GPU box) through the same code path,
    plus the sharded C-ABI step (RCCL communicator + collective) inside it."""
    monkeypatch.setenv("BENCH_FORCE_LAUNCH", "1")
    d = run_bench("--no-cpu-baseline", "--force-collective")
    assert d["n_gpus"] == 1 and d["value"] > 0
    m = d["multi_gpu"]
    assert m["mode"] == "rows" and len(m["rank_local_ms"]) == 1 and m["max_local_ms"] > 0 and m["max_collective_ms"] >= 0


def test_multi_rank_reporting_with_one_rank(monkeypatch):
    """What a real N > 1 line adds -- rank agreement, which library issued the collective, the torch.distributed fallback,
    the bit-identical row shard timed next to a default that is not `rows`, the parity sample on a shared block row -- through
    the same code with ONE rank (--force-collective starts a 1-rank process group)."""
    monkeypatch.setenv("BENCH_ALSO_TIME_ROWS", "1")
    d = run_bench("--force-collective", "--shard", "rowsum", "--cpu-budget-gb", "0.05")
    m = d["multi_gpu"]
    assert m["mode"] == "rowsum" and m["ranks_agree"] and m["ranks_max_abs_diff"] == 0.0 and m["collective_impl"].startswith("libbfhip")
    assert m["also_timed"]["mode"] == "rows" and m["also_timed"]["bit_identical_to_one_gpu"] and m["also_timed"]["ranks_agree"] and m["also_timed"]["ms_per_step"] > 0
    assert d["cpu_baseline"]["parity_rel_l2"] < 1e-12
    monkeypatch.setenv("BENCH_FORCE_TORCH_COLLECTIVE", "1")
    d = run_bench("--force-collective", "--shard", "blocks", "--no-cpu-baseline")
    m = d["multi_gpu"]
    assert m["collective_impl"] == "torch.distributed" and m["collective_fallback_reason"] == [[0, "BENCH_FORCE_TORCH_COLLECTIVE=1"]]
    assert m["ranks_agree"] and m["max_local_ms"] > 0 and m["max_collective_ms"] >= 0 and m["also_timed"]["collective_impl"] == "torch.distributed"


@pytest.mark.parametrize("shard,torch_coll", [("rows", False), ("rowsum", False), ("blocks", True)])
def test_multi_rank_adjoint_reporting_with_one_rank(monkeypatch, shard, torch_coll):
    """`bench.py --gpus N --adjoint`: the adjoint of the sharded step (every rank's A_r^T on its entries of v, ONE all-reduce) is
    timed next to the forward step with its own rank agreement and <A x, v> = <x, A^T v>; here with ONE rank through the same code,
    over libbfhip's RCCL communicator and over the torch.distributed fallback."""
    if torch_coll:
        monkeypatch.setenv("BENCH_FORCE_TORCH_COLLECTIVE", "1")
    d = run_bench("--force-collective", "--adjoint", "--shard", shard, "--no-cpu-baseline", "--pcie")
    a = d["multi_gpu"]["adjoint"]
    assert "error" not in a, a
    assert a["ranks_agree"] and a["ms_per_step"] > 0 and a["transpose_identity_rel"] < 1e-12
    assert d["adjoint"]["layout"].startswith("shared")
    p = d["pcie_inclusive"]
    assert p["pageable_ms"] > 0 and p["registered_ms"] > 0 and p["device_pointers_ms"] > 0 and p["resident_ms_same_moment"] > 0


def test_default_line_keys_of_round_5():
    """The keys the default line gained in round 5, on a small operand: adjoint on the packed copy AND on the shared leaves,
    resident bytes, the host-vector path for three kinds of caller memory."""
    d = run_bench("--adjoint", "--adjoint-both", "--pcie", "--no-cpu-baseline")
    assert d["adjoint"]["layout"].startswith("packed") and d["adjoint_shared"]["layout"].startswith("shared")
    assert d["adjoint_shared"]["rel_vs_packed"] < 1e-12 and d["adjoint_shared"]["arena_bytes"] * 2 == d["adjoint_shared"]["arena_bytes_packed"]
    assert d["config"]["resident_bytes"] > d["config"]["arena_bytes"] > 0
    p = d["pcie_inclusive"]
    assert p["ratio_pageable"] > 0.5 and p["ratio_registered"] > 0.5 and p["device_pointers_ms"] > 0


def test_adjoint_layouts_in_the_bench_line():
    """--adjoint: a packed copy of the leaves for A^T on the forward kernels (default for fac_helm2 operands) or the shared
    leaves through the transposed kernels; both tie <A x, v> to <x, A^T v>."""
    for extra, word in ((("--adjoint",), "packed"), (("--adjoint", "--adjoint-shared"), "shared")):
        d = run_bench(*extra, "--no-cpu-baseline")
        a = d["adjoint"]
        assert a["layout"].startswith(word) and a["transpose_identity_rel"] < 1e-12 and a["ms_per_apply"] > 0


def test_bench_fails_loudly_when_a_rank_fails():
    """More ranks than GPUs on the box: a rank cannot get its device; the parent must exit non-zero, no JSON."""
    import torch
    n = torch.cuda.device_count() + 1
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--npoints", "8192",
           "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_streamer_workload_contract():
    d = run_streamer()
    assert d["dtype"] == "f32" and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["config"]["num_cols"] == 32 * 32 and d["config"]["graph"]["identity"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["adjoint"]["transpose_identity_rel"] < 1e-4
    # the path's caller (cov_matvec of lbo_cov.c:48-60) as one device call, checked against the two applies
    assert d["cov_matvec"]["rel_vs_separate_applies"] < 1e-5 and d["cov_matvec"]["ms_per_product"] > 0


def run_streamer():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "streamer", "--npoints", "16384", "--lmax", "31", "--steps", "3",
           "--warmup", "1", "--adjoint", "--cpu-budget-gb", "0.05"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])
