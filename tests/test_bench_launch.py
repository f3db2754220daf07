"""bench.py's own launcher: `python bench.py --gpus N` (what the driver runs for the scaling bench) must start N ranks by
itself, and a run started by torch.distributed.run must be one of the ranks.  No GPU here: --dry-run keeps the launcher, the
rendezvous on 127.0.0.1, the barrier / max-over-ranks timing and the JSON line, with gloo and placeholder steps."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout   # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_plain_command_launches_two_ranks():
    d = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run"])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["ranks_reporting"] == 2
    assert d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["paths_counted"] == 2 * 4 * 65536
    assert d["config"]["parallelism"] == "dp2"


def test_total_paths_splits_the_job_over_ranks_and_steps():
    d = _run(["--gpus", "2", "--steps", "5", "--warmup", "0", "--dry-run", "--total-paths", "1000000"])
    assert d["scaling"] == "strong" and d["config"]["paths_per_step_per_gpu"] == 100000
    assert d["paths_counted"] == 1000000


def test_single_rank_needs_no_launcher():
    d = _run(["--steps", "2", "--warmup", "0", "--dry-run"])
    assert d["n_gpus"] == 1 and d["ranks_reporting"] == 1


def test_started_by_torchrun_it_is_one_of_the_ranks():
    """the driver's other form: python -m torch.distributed.run ... bench.py --gpus 2"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0])["n_gpus"] == 2


def test_replica_digests_ride_on_the_line_and_a_diverged_replica_fails_the_run():
    """round 5: after the timed steps every rank digests its replica (parameters + Adam state in the real run) and the digests are all-gathered;
    `dp.replicas_bit_identical` is on the line and a run whose replicas differ exits with code 3 (the line still names the digests)."""
    d = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert d["dp"]["replicas_bit_identical"] is True
    per = d["dp"]["replica_digests"]["per_rank"]
    assert len(per) == 2 and per[0] == per[1]
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e["KPRN_DRYRUN_DIVERGE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    bad = json.loads(lines[0])
    assert bad["dp"]["replicas_bit_identical"] is False and bad["dp"]["replica_digests"]["per_rank"][0] != bad["dp"]["replica_digests"]["per_rank"][1]


def test_a_weak_invocation_also_reports_the_strong_scaling_experiment():
    """round 6: the driver's scaling command carries no --total-paths.  A data-parallel run therefore ALSO times the north_star's experiment --
    1 000 000 paths split over ranks and steps, and the same total on rank 0 alone -- and reports it under dp.strong_1M with speed-up and
    efficiency from this one invocation.  (Partitioning unit = pairs: module/MapReduce.lua:24-47.)"""
    d = _run(["--gpus", "2", "--steps", "20", "--warmup", "1", "--dry-run"])
    assert d["scaling"] == "weak" and d["paths_counted"] == 2 * 20 * 65536       # the weak line is what it was
    s1 = d["dp"]["strong_1M"]
    assert s1["total_paths"] == 1000000 and s1["steps"] == 15 and s1["paths_per_rank_step"] == 1000000 // (2 * 15)   # >= 32 768 paths per rank and step
    assert s1["paths_counted"] == 2 * 15 * (1000000 // 30)
    assert s1["n1"]["steps"] == 20 and s1["n1"]["paths_per_step"] == 50000 and s1["n1"]["paths_counted"] == 1000000
    assert s1["speedup_vs_n1"] > 0 and abs(s1["efficiency"] - s1["speedup_vs_n1"] / 2) < 1e-5
    assert d["dp"]["fallback_reason"] is None and d["dp"]["exchange"].startswith("engine")
    # ... and not with --total-paths (that run IS the strong experiment) or --no-strong-1m
    assert "strong_1M" not in _run(["--gpus", "2", "--steps", "5", "--warmup", "0", "--dry-run", "--total-paths", "1000000"])["dp"]
    assert "strong_1M" not in _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run", "--no-strong-1m"])["dp"]


def test_a_communicator_bootstrap_that_never_returns_falls_back_instead_of_hanging():
    """kprn_amd/dp.py call_with_watchdog around kprn_dp_init: one rank's bootstrap hangs (forced), every rank agrees to fall back to the
    torch.distributed collectives, the reason is on the line, the run completes."""
    d = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--no-strong-1m"], env={"KPRN_DP_TEST_INIT_HANG": "1", "KPRN_DP_INIT_TIMEOUT": "1.5"})
    assert d["n_gpus"] == 2 and d["ranks_reporting"] == 2 and d["dp"]["replicas_bit_identical"] is True
    assert "fallback" in d["dp"]["exchange"] and "did not return within 1.5 s" in d["dp"]["fallback_reason"]


def test_watchdog_reports_exceptions_and_successes():
    from kprn_amd.dp import call_with_watchdog
    assert call_with_watchdog(lambda: None, 5, "noop") == (True, None)
    ok, why = call_with_watchdog(lambda: 1 / 0, 5, "div")
    assert not ok and "ZeroDivisionError" in why
    import time
    t0 = time.perf_counter()
    ok, why = call_with_watchdog(lambda: time.sleep(30), 0.3, "sleeper")
    assert not ok and "did not return within 0.3 s" in why and time.perf_counter() - t0 < 5
