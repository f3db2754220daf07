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
