"""bench.py's rank-side logic on CPU: everything a rank does before it selects its device (BENCH_DRY_RUN=1) -- the
launcher path of `--gpus N`, the gloo rendezvous on 127.0.0.1, the world-size check, the operand's layout and the
partition every rank computes on its own."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra, timeout=600):
    env = dict(os.environ, BENCH_DRY_RUN="1", **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env)


def test_two_ranks_start_agree_and_report():
    p = run(["--gpus", "2", "--n", "16384"], {})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 2 and d["mode"] == "rows" and d["ranks_agree"]
    assert d["cuts"] == [0, 8192, 16384] and abs(d["imbalance"] - 1) < 1e-9 and abs(d["replication"] - 1) < 1e-9


def test_three_ranks_get_unequal_but_balanced_ranges():
    p = run(["--gpus", "3", "--npoints", "16384", "--shard", "rows"], {})          # (auto would pick rowsum here: 12 block rows on 3 ranks divide, but not their bytes)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 3 and len(d["cuts"]) == 4 and d["cuts"][0] == 0 and d["cuts"][-1] == 16384
    assert d["imbalance"] < 1.1 and 1.0 <= d["replication"] < 1.15


def test_world_size_mismatch_is_a_hard_failure():
    """A line that says n_gpus = N must come from N ranks: started by a launcher with another world size, bench.py exits
    non-zero without printing JSON."""
    p = run(["--gpus", "2", "--n", "4096"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "refusing to run" in p.stderr


def _line(p):
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout + p.stderr[-2000:]
    return json.loads(lines[0])


def test_ranks_that_end_with_different_results_fail_the_run():
    """After the closing collective every rank must hold the same y: one all_reduce(MAX) of max|y_r - y_0| says so in
    the line (multi_gpu.ranks_agree); a rank whose copy differs makes the run exit non-zero."""
    ok = run(["--gpus", "2", "--n", "16384"], {})
    d = _line(ok)
    assert ok.returncode == 0 and d["multi_gpu"]["ranks_agree"] and d["multi_gpu"]["ranks_max_abs_diff"] == 0.0
    bad = run(["--gpus", "2", "--n", "16384"], {"BENCH_DRY_PERTURB_RANK": "1"})
    d = _line(bad)
    assert bad.returncode != 0 and not d["multi_gpu"]["ranks_agree"] and abs(d["multi_gpu"]["ranks_max_abs_diff"] - 1e-3) < 1e-12


def test_one_rank_without_its_communicator_moves_every_rank_to_torch_distributed():
    """The choice between libbfhip's own RCCL communicator and torch.distributed is collective: a failure on ONE rank
    is seen by all (no rank is left waiting inside a collective the others never enter), recorded with its reason."""
    d = _line(run(["--gpus", "3", "--n", "16384", "--shard", "rows"], {"BENCH_DRY_FAIL_RCCL_RANK": "2"}))
    m = d["multi_gpu"]
    assert m["collective_impl"] == "torch.distributed" and m["collective_fallback_reason"] == [[2, "simulated bfhipCommInitRank failure"]]
    assert _line(run(["--gpus", "3", "--n", "16384", "--shard", "rows"], {}))["multi_gpu"]["collective_impl"] == "libbfhip (RCCL)"
    # ranks whose plans differ in depth can still gather their per-stage profiles: padded to the longest (3 * (5 + 2))
    assert m["padded_profile_len"] == 21


def test_eight_ranks_time_the_bit_identical_row_shard_next_to_the_default():
    """12 block rows on 8 ranks: the default is rowsum (no replication, equal to rounding); the line also carries the
    row-range shard (bit-identical, replicated first factors), and the parity sample starts with a shared block row."""
    d = _line(run(["--gpus", "8", "--n", "16384"], {}, timeout=900))
    assert d["n_gpus"] == 8 and d["mode"] == "rowsum" and abs(d["replication"] - 1) < 1e-9
    alt = d["multi_gpu"]["also_timed"]
    assert alt["mode"] == "rows" and len(alt["cuts"]) == 9 and alt["cuts"][-1] == 16384 and 1.0 < alt["replication"] < 1.2
    assert len(d["multi_gpu"]["shared_block_rows"]) >= 1


def test_a_launcher_without_the_gpus_flag_sets_the_world_size():
    """torchrun --nproc-per-node=2 bench.py (no --gpus): the launcher's WORLD_SIZE is the job's size."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, BENCH_DRY_RUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--npoints", "16384"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _line(p)["n_gpus"] == 2
