"""`python bench.py --gpus N` without torchrun must start its own N ranks: the parent spawns
`python -m torch.distributed.run ...` as a child before touching any GPU, relays rank 0's JSON line and exits with the
child's return code.  Exercised here on CPU with the gloo backend through bench.py's --dry-run protocol self-test (the
same process-group, shard, barrier and max-over-ranks calls as the real run, no engine)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"), env_add=None):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    env.update(env_add or {})
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


def test_eight_ranks_uneven_shards_slow_rank0_and_a_straggler():
    """The driver's N = 8 launch, hardened without a node: eight gloo ranks on this host, a global batch that
    does not divide by 8 (35 per rank x 8 is replaced by an uneven split below), rank 0 busy for seconds before the weight
    broadcast while the others wait in it, every rank bound to cuda:{LOCAL_RANK}, and a straggling last rank that must be visible
    in the line (per-rank min / max) while `ms_per_step` stays the max over ranks."""
    r = _run(["--gpus", "8", "--batch", "3"], env_add={"Y4_DRY_SLOW_RANK0": "3", "Y4_DRY_STRAGGLER": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 8 and line["config"]["shard"] == [0, 3]
    assert line["devices"] == [f"cuda:{i}" for i in range(8)]
    per = line["rank_ms_per_step"]["per_rank"]
    assert len(per) == 8 and per.index(max(per)) == 7 and max(per) > 1.5 * min(per)
    assert line["ms_per_step"] >= 0.95 * line["rank_ms_per_step"]["max"]
