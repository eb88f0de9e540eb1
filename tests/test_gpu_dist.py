"""The RCCL path on hardware, as far as ONE GPU allows: `bench.py` under `torch.distributed.run` with one
rank goes through dist.init_process_group('nccl'), the RCCL broadcast of the packed weight workspace, barrier and the MAX
all_reduce of the timing protocol; with Y4_FORCE_ADOPT=1 the bench runs on a FRESH engine that adopted the broadcast bytes
(y4_adopt_packed_weights: what every rank > 0 does) and must produce bit-identical outputs.  The children are fresh
processes started before this process touches the GPU API in them (never an exec of a process that initialised HIP)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_child(extra_env, size=160, classes=3, batch=4, ranks=1, extra_args=(), autotune=False):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--blocks", "1",
           "--warmup", "1", "--no-cpu-baseline", "--no-latency", "--size", str(size), "--classes", str(classes),
           "--batch", str(batch)] + ([] if autotune else ["--no-autotune"]) + list(extra_args)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_under_torchrun_one_rank_and_adopted_weights():
    plain = _run_child({})
    assert plain["n_gpus"] == 1 and plain["steps"] == 2 and plain["scaling"] == "weak" and plain["value"] > 0
    assert "1 rank(s)" in plain["config"]["sharding"] and len(plain["outputs_sha256"]) == 64
    adopted = _run_child({"Y4_FORCE_ADOPT": "1"})
    assert adopted["n_gpus"] == 1
    assert adopted["outputs_sha256"] == plain["outputs_sha256"], "an engine that adopted the broadcast weights computes other outputs"


def test_headline_shape_under_torchrun():
    """The same at the headline shape (608x608, 80 classes, batch 32, bf16): the 129 MB packed workspace through the RCCL
    broadcast, the default fusions, one block of two steps."""
    line = _run_child({}, size=608, classes=80, batch=32)
    assert line["n_gpus"] == 1 and line["config"]["global_batch"] == 32 and line["dtype"] == "bf16"
    assert 0.05 < line["roofline"]["frac"] < 1.0 and line["roofline"]["backbone_frac"] and line["roofline"]["end_to_end_frac"]


def test_two_ranks_with_real_engines_on_one_gpu():
    """N = 2 with REAL engines, as far as a box with one GPU allows: two processes share cuda:0
    (Y4_SHARE_GPU=1; gloo carries the collectives because RCCL refuses two ranks on one device).  Rank 0 packs the weights, the
    packed workspace is broadcast INTO THE OTHER PROCESS, which adopts it (y4_adopt_packed_weights) and computes images
    [batch, 2 batch) of the global batch while rank 0 computes [0, batch): each rank's outputs must be bit-identical to what a
    single process computes for the same global image indices, `value` counts both shards, every rank reports its own time."""
    share = {"Y4_SHARE_GPU": "1", "Y4_DIST_BACKEND": "gloo"}
    two = _run_child(share, ranks=2)
    assert two["n_gpus"] == 2 and two["config"]["global_batch"] == 8 and "2 rank(s)" in two["config"]["sharding"]
    assert len(two["rank_ms_per_step"]["per_rank"]) == 2 and len(two["rank_outputs_sha256"]) == 2
    assert two["rank_outputs_sha256"][0] == two["outputs_sha256"] != two["rank_outputs_sha256"][1]
    first = _run_child({})                                       # one process, images 0..3
    second = _run_child({}, extra_args=["--first-image", "4"])    # one process, images 4..7
    assert first["outputs_sha256"] == two["rank_outputs_sha256"][0]
    assert second["outputs_sha256"] == two["rank_outputs_sha256"][1], "rank 1 (adopted weights, its own shard) differs from one process"
    assert second["first_image"] == 4


def test_two_ranks_run_rank_zeros_tuned_schedule():
    """a shape nothing ships for is tuned ONCE per job -- rank 0 runs y4_autotune, the tile ids and fusion switches go
    to the other rank by broadcast (dist.share_schedule), both ranks report the same tile set, and (16-bit schedules without
    split-K ids being pure speed choices) the outputs equal those of the un-tuned two-rank run."""
    share = {"Y4_SHARE_GPU": "1", "Y4_DIST_BACKEND": "gloo"}
    tuned = _run_child(share, ranks=2, autotune=True, extra_args=["--in-flight", "1"])
    assert tuned["n_gpus"] == 2 and len(tuned["rank_tiles_sha256"]) == 2
    assert tuned["rank_tiles_sha256"][0] == tuned["rank_tiles_sha256"][1]
    assert "autotuned" in tuned["schedule"]
    plain = _run_child(share, ranks=2, extra_args=["--in-flight", "1"])
    assert plain["rank_tiles_sha256"][0] == plain["rank_tiles_sha256"][1] != tuned["rank_tiles_sha256"][0]
    assert plain["rank_outputs_sha256"] == tuned["rank_outputs_sha256"]
