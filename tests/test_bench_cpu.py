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
