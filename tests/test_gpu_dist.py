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


def _run_child(extra_env, size=160, classes=3, batch=4):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--blocks", "1",
           "--warmup", "1", "--no-cpu-baseline", "--no-autotune", "--size", str(size), "--classes", str(classes),
           "--batch", str(batch)]
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
