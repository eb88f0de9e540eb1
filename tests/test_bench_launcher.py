"""`python bench.py --gpus N` without torchrun must start its own N ranks: the parent spawns
`python -m torch.distributed.run ...` as a child before touching any GPU, relays rank 0's JSON line and exits with the
child's return code.  Exercised here on CPU with the gloo backend through bench.py's --dry-run protocol self-test (the
same process-group, shard, barrier and max-over-ranks calls as the real run, no engine)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--blocks", "2",
                           "--warmup", "0"] + extra, env=env, capture_output=True, text=True, timeout=300)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    r = _run(["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert "DRY RUN" in line["metric"] and line["data"].startswith("dry-run")
    assert line["config"]["shard"] == [0, 32]                 # rank 0's block of the 64-image global batch


def test_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_child_failure_is_propagated():
    r = _run(["--gpus", "2", "--batch", "-1"])                # shard_range of a negative batch: the ranks must fail
    assert r.returncode != 0
