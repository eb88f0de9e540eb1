#!/usr/bin/env python3
"""Headline benchmark: images/sec of the whole predict() hot path (forward + decode + NMS + results to the host) at
608x608, 80 classes, batch 32 per GPU, bf16 storage / fp32 accumulate, synthetic data.

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a torchrun environment: this process -- before it touches any GPU -- starts
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>`
      as a child, relays its output (rank 0's JSON line) and exits with its return code.  Under torchrun
      (RANK/LOCAL_RANK/WORLD_SIZE set) it is one rank of the job, one rank per GPU over RCCL.

A step = one y4_predict over one batch of 32 float32 NHWC images already resident in HBM (the PCIe-inclusive rate is
noted in DESIGN.md, it is never `value`) followed by ONE 77 KB device->host copy of the four NMS outputs (+ kept
indices), as SURVEY.md section 8(d) defines the metric.  Timing: W untimed warm-up steps, then R (--blocks, default 5)
blocks of EXACTLY K steps, each block bracketed by barrier + torch.cuda.synchronize() on both sides and reduced with MAX
over ranks; `ms_per_step` / `value` are the MEDIAN block (one 0.13 s sample on a pool whose boxes differ by +-3 % was too
noisy), every block is listed under `blocks_ms_per_step`.
By default `--in-flight 2`: step i runs on HIP stream / activation workspace i % 2 (two independent batches overlap on the
GPU; every step of a block still completes inside the block's bracket); `--in-flight 1` = every step on one stream, and
`single_stream_value` in the line is that rate measured in the same process.
Schedule (tile per launch, fusion kernels on / off -- every choice gives the same bits): the tuned one that ships for the
shape (yolo4hip/schedules/, the kernel mix profiles/r05 was profiled with) if there is one, else a one-off autotune on this
box; `--retune` forces the autotune, `--load-tiles` a file; `schedule` in the line says which ran.
`roofline` is for the dominant kernel family (the conv kernels: convs 2..109 with the stem fusion, else 1..109):
algorithmic conv FLOPs of one step / its summed device time, measured with HIP events recorded on the launch stream
(y4_timing_begin/end) -- inside the timed blocks with one stream, in a single-stream pass right after them with two
(`roofline.measured_in`); `roofline.traffic` quotes the committed PMC passes when the schedule is the profiled one, else
null with the reason.  `cpu_baseline` times the oracle (a torch-CPU/NumPy restatement; the reference's tf.keras path cannot
run here) on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "yolo-v4-tf.keras_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # dense, /opt/skills/guides/MI355X_MICROARCH.md
SUSTAINED_TFLOPS = 1650.0      # dense bf16 / fp16 MFMA rate the chip sustains on random operands (scripts/ubench/mfma_power.hip)
METRIC = "images/sec end-to-end predict() at 608x608 batch 32; conv MFMA %peak"


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            models = [ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")]
        sockets = len({ln for ln in open("/proc/cpuinfo") if ln.startswith("physical id")}) or 1
        return f"{sockets} x {models[0]}" if models else "unknown", len(models)
    except OSError:
        return "unknown", os.cpu_count() or 0


def cpu_baseline(size, ncls, ws, cfg, batch=32, runs=3):
    """the reference's tf.keras CPU predict() cannot run anywhere we run (no TensorFlow), so the figure
    beside the GPU number is a PROXY -- the oracle (torch-CPU fp32 / oneDNN forward + NumPy decode/NMS, kind 'port') on the
    GPU box's host cores, measured as SURVEY.md section 8(d) prescribes: batch `batch` (32, the configuration the metric is
    quoted on) and batch 1, 1 warm-up + `runs` (3) timed runs each.  `value` is the batch-`batch` rate."""
    import torch
    from yolo4hip import weights as W
    from oracle import forward as OF, decode_nms as OD
    # measured on the GPU box's host (2x EPYC 9575F, 256 hardware threads): oneDNN is fastest around 32
    # threads (3.9 img/s) and collapses when every hardware thread is used (0.02 img/s at 256)
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    model, hw_threads = cpu_model()

    def timed(nimg):
        imgs = W.synth_images(nimg, size, seed=0)
        def once():
            heads = OF.yolo_model_forward(imgs, ws, ncls)
            return OD.inference_from_heads(heads, ncls, cfg["anchors"], cfg["xyscale"], size)
        once()                                # warm-up (oneDNN primitive creation, page-in)
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            once()
            ts.append(time.perf_counter() - t0)
        return ts
    t1 = timed(1)
    tb = timed(batch)
    return {"value": round(batch * runs / sum(tb), 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "label": "CPU restatement (proxy for tf.keras CPU; the reference's own CPU path needs TensorFlow, absent here)",
            "cpu_model": model, "hw_threads": hw_threads,
            "batch": batch, "runs_s": [round(t, 3) for t in tb],
            "batch1_value": round(runs / sum(t1), 4), "batch1_runs_s": [round(t, 3) for t in t1],
            "sample": f"{runs} timed runs x {batch} images and {runs} x 1 image, {size}x{size}x3, {ncls} classes, fp32 "
                      f"torch-CPU(oneDNN) forward + NumPy decode/NMS (oracle/), 1 warm-up run each (SURVEY.md 8(d)); "
                      f"{sum(tb) + sum(t1):.1f} s of timed CPU work on {cores} threads (oneDNN is fastest near 32 threads on "
                      f"this host and collapses at all {hw_threads})"}


def latency_block(classes, flat, class_file, img_path, iters=200):
    """The reference's actual call is ONE image -- `Yolov4.predict(img_path)` (models.py:109-127; BASELINE configs 1 and 2): wall
    time of one `y4_predict` + the copy of its results to the host, synchronised per call, median (p50) and p90 of `iters` calls,
    at 608 and 416, bf16 and fp32, each on the latency schedule that ships for it (split-K ids allowed: engine.ensure_schedule),
    and of the facade's `predict` on img/street.jpeg (imread + device resize + predict + DataFrame).  A latency, not `value`."""
    import numpy as np
    import torch
    from yolo4hip import weights as W
    from yolo4hip.config import make_config
    from yolo4hip.engine import Engine
    from yolo4hip.plan import build_plan
    out = {"what": "p50 / p90 wall ms of one synchronised y4_predict of ONE image incl. its results' copy to the host",
           "iters": iters}
    for size in (608, 416):
        plan = build_plan(size, classes)
        for dtype in ("bf16", "f32"):
            eng = Engine(classes, make_config(size), max_batch=1, dtype=dtype, alias_workspace=True)
            eng.load_weight_blob(flat)
            src, path = eng.ensure_schedule(tune=True, verbose=False)
            imgs = torch.from_numpy(W.synth_images(1, size, seed=0)).to(eng.device)
            fl, outs = eng.alloc_outputs_flat(1)
            host = torch.empty(fl.numel(), dtype=torch.int32).pin_memory()
            ts = []
            for i in range(iters + 20):
                t0 = time.perf_counter()
                eng.predict_device(imgs, outs)
                host.copy_(fl, non_blocking=True)
                torch.cuda.synchronize()
                if i >= 20:
                    ts.append(time.perf_counter() - t0)
            ts.sort()
            p50 = ts[len(ts) // 2] * 1e3
            out[f"{size}_{classes}_b1_{dtype}"] = {
                "p50_ms": round(p50, 4), "p90_ms": round(ts[int(len(ts) * 0.9)] * 1e3, 4),
                "frac_of_mfma_peak": round(plan.flops_per_image / (p50 * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[dtype], 4),
                "schedule": src + (": " + os.path.basename(path) if path else "")}
            eng.close()
    try:
        from yolo4hip.api import Yolov4
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            m = Yolov4(None, class_file, make_config(416), dtype="bf16", max_batch=1)
            m.predict(img_path, plot_img=False)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                m.predict(img_path, plot_img=False)
                ts.append(time.perf_counter() - t0)
        ts.sort()
        out["facade_predict_street_jpeg_416_bf16"] = {"p50_ms": round(ts[len(ts) // 2] * 1e3, 3),
                                                      "what": "Yolov4.predict(img/street.jpeg): imread + uint8 upload + device resize + "
                                                              "y4_predict_u8 + get_detection_data (DataFrame) + draw_bbox, host wall time"}
        m.engine.close()
    except Exception as e:                                    # PIL / pandas missing on some host: the engine numbers stand
        out["facade_predict_street_jpeg_416_bf16"] = {"error": repr(e)}
    return out


def committed_traffic(args, fused_stem, chained, staged, res_mask, tiles):
    """HBM bytes per STEP of the conv kernel family from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE runs of this same command, corrected as MI355X_MICROARCH.md prescribes), or (None, reason) when this
    run's schedule is not the profiled one: same shape, dtype and fusion switches, same residual-block mask, and the same
    per-layer tile ids as the profiled run (`tiles.json` next to the profile).  Counters cannot be read from inside the
    process: the figure is NOT measured by this run, `traffic_source` says where it comes from."""
    reasons = []
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", rnd, "hbm_traffic.json")
        try:
            prof = json.load(open(path))
        except (OSError, ValueError):
            continue
        want = {"size": args.size, "classes": args.classes, "batch": args.batch, "dtype": args.dtype,
                "stem_fusion": bool(fused_stem), "chain_fusion": bool(chained), "stage_fusion": bool(staged)}
        have = dict(prof.get("config", {}))
        if have != want:
            reasons.append(f"profiles/{rnd}: profiled configuration {have} != this run's {want}")
            continue
        try:
            saved = json.load(open(os.path.join(ROOT, "profiles", rnd, "tiles.json")))
        except (OSError, ValueError):
            saved = {}
        if int(saved.get("res_fusion_mask", -1)) != int(res_mask):
            reasons.append(f"profiles/{rnd}: residual-block mask {saved.get('res_fusion_mask')} != this run's {res_mask}")
            continue
        if tiles is None or list(saved.get("tiles", [])) != list(tiles):
            ndiff = sum(1 for a, b in zip(saved.get("tiles", []), tiles or []) if a != b) if tiles else -1
            reasons.append(f"profiles/{rnd}: the autotuner picked another tile / fusion set on this box "
                           f"({ndiff} of {len(tiles or [])} layers differ from the profiled run)")
            continue
        return int(prof["conv_igemm_hbm_bytes_per_step"]), \
            f"profiles/{rnd}/hbm_traffic.json (committed rocprofv3 --pmc passes of this command and tile set; not measured by this run)"
    return None, "null because: " + ("; ".join(reasons) if reasons else "no committed PMC pass exists")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_gpus, argv):
    """Parent of an N > 1 run started without torchrun.  Nothing here touches the GPU (no torch import even): the ranks
    are fresh child processes, never an exec of a process that initialised HIP."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps steps each; the median block is reported")
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--classes", type=int, default=80)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the `latency` block (one-image y4_predict at 608 / 416, bf16 / fp32, and the facade's predict)")
    ap.add_argument("--cpu-batch", type=int, default=32,
                    help="batch of the CPU proxy baseline (SURVEY.md 8(d): 32 and 1; 1 warm-up + 3 timed runs each, ~55 s of host time)")
    ap.add_argument("--stop-after-conv", type=int, default=71,
                    help="last conv of the backbone-only passes (`roofline.backbone_wall`): the forward cut behind this conv "
                         "(y4_forward_until), wall-clock timed with one stream and with --in-flight batches in flight; 71 = "
                         "CSPDarknet53 proper (reference custom_layers.py:100-124); -1 skips the passes")

    ap.add_argument("--no-autotune", action="store_true", help="use the built-in tile heuristic instead of measuring")
    ap.add_argument("--tune-reps", type=int, default=3)
    ap.add_argument("--pair-passes", type=int, default=12,
                    help="with --in-flight > 1: which autotune decisions use two batches in flight as objective (y4_autotune_pair) "
                         "instead of one launch alone -- bit 0 tiles, 1 chains / LDS pairs, 2 stage kernel, 3 residual-block kernels. "
                         "Default 12: the spatially tiled kernels' ragged last rounds are filled by the neighbour stream, so they "
                         "are judged by their work (measured +2 %% img/s); tiles stay tuned per launch (their pair-tuned choice "
                         "measured -1 %% and is 12 %% slower one stream at a time).  0 = plain y4_autotune")
    ap.add_argument("--save-tiles", default=None, help="write the autotuned per-layer tile ids to this JSON file")
    ap.add_argument("--load-tiles", default=None, help="use tile ids from this JSON file instead of autotuning")
    ap.add_argument("--retune", action="store_true",
                    help="autotune on this box even when a tuned schedule ships with the package for this shape "
                         "(yolo4hip/schedules/: the headline shape's is the tile set profiles/r05 was profiled with)")
    ap.add_argument("--subbatch", type=int, default=0,
                    help="images per sub-batch for the early layers (0 = whole batch); sub-batching and workspace aliasing exclude each "
                         "other (y4_set_subbatch refuses), so a run with --subbatch uses the plain, un-aliased workspace")
    ap.add_argument("--sub-last-conv", type=int, default=16)
    ap.add_argument("--no-stem-fusion", action="store_true", help="run convs 0 and 1 as two kernels (c0 through HBM)")
    ap.add_argument("--no-chain-fusion", action="store_true", help="run the 3x3+Add -> 1x1 -> 1x1 runs as separate kernels")
    ap.add_argument("--no-stage-fusion", action="store_true", help="run the 304^2 CSP stage (convs 2..7) as separate kernels")
    ap.add_argument("--no-res-fusion", action="store_true", help="run the 1x1 -> 3x3+Add residual blocks of the 64/128-channel stages as separate kernels")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="batches in flight per GPU (HIP streams with one activation workspace each, shared weights); "
                         "1 = every step on one stream")
    ap.add_argument("--alias-workspace", action=argparse.BooleanOptionalAction, default=True,
                    help="activation buffers with disjoint lifetimes share memory (y4_set_workspace_aliasing): 8.0 -> 2.9 GB "
                         "per batch in flight at the headline shape, same bits, +1 %% single-stream")
    ap.add_argument("--per-op", action="store_true", help="also print the per-op time table to stderr")
    ap.add_argument("--first-image", type=int, default=0,
                    help="global index of this run's first synthetic image (image i depends on (seed, i) only: a one-process run "
                         "with --first-image 32 computes what rank 1 of a two-rank run computes)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / protocol self-test on CPU (gloo, no engine, no GPU work): the line says so")
    return ap.parse_args(argv)


def dry_run(args):
    """The multi-process protocol of the bench without an engine (tests/test_bench_launcher.py): process group, shard
    ranges, barrier-bracketed blocks, MAX over ranks, one JSON line from rank 0."""
    import torch.distributed as dist
    from yolo4hip import dist as D
    rank, local_rank, world = D.init_process_group(backend="gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    lo, hi = D.shard_range(args.batch * world, rank, world)
    # what every rank does before its first library call in the real run: bind its own device (recorded, not executed, here)
    bound = f"cuda:{local_rank}"
    if os.environ.get("Y4_DRY_SLOW_RANK0"):          # rank 0 "synthesises the weights" while the others wait in the broadcast
        import torch
        blob = torch.zeros(1 << 16, dtype=torch.uint8)
        if rank == 0:
            time.sleep(float(os.environ["Y4_DRY_SLOW_RANK0"]))
            blob += 7
        D.broadcast_bytes(blob, 0)
        assert int(blob[123]) == 7
    local, own = [], []
    for _ in range(args.blocks):
        D.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.001 * (1 + (rank == world - 1 and os.environ.get("Y4_DRY_STRAGGLER") == "1")))
        own.append(time.perf_counter() - t0)                 # this rank's own K steps, before it waits for the others
        D.barrier()
        local.append(time.perf_counter() - t0)
    blocks = D.max_over_ranks(local)
    my_ms = statistics.median(own) / args.steps * 1e3
    rank_ms, devices = D.gather_objects(my_ms), D.gather_objects(bound)
    if rank == 0:
        dt = statistics.median(blocks)
        print(json.dumps({"metric": METRIC + " [DRY RUN: no GPU work]", "value": round(args.batch * world * args.steps / dt, 2),
                          "unit": "images/sec", "n_gpus": dist.get_world_size() if dist.is_initialized() else 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
                          "rank_ms_per_step": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4),
                                               "per_rank": [round(v, 4) for v in rank_ms]},
                          "devices": devices,
                          "data": "dry-run (no GPU work)", "config": {"workload": "launcher self-test", "shard": [lo, hi]}}),
              flush=True)
    D.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, argv))
    if args.batch < 1 or args.steps < 1 or args.blocks < 1 or args.gpus < 1:
        raise SystemExit("bench.py: --batch, --steps, --blocks and --gpus must be >= 1")
    if args.subbatch > 0:
        args.alias_workspace = False       # see --subbatch
    if args.dry_run:
        return dry_run(args)

    import torch
    import torch.distributed as dist
    from yolo4hip import dist as D, weights as W
    from yolo4hip.config import make_config
    from yolo4hip.engine import Engine
    from yolo4hip.plan import build_plan

    rank, local_rank, world = D.init_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one process per GPU: rank r of the node drives cuda:{LOCAL_RANK}.  Test hook (tests/test_gpu_dist.py, a box with ONE GPU):
    # Y4_SHARE_GPU=1 lets every rank use cuda:0 (with Y4_DIST_BACKEND=gloo: RCCL refuses two ranks on one device) so that the
    # N > 1 path -- shard ranges, the broadcast of the packed weights into a second process, adopt, the reductions -- runs with
    # real engines; the line then says so
    shared_gpu = os.environ.get("Y4_SHARE_GPU") == "1"
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if shared_gpu else local_rank
    local_rank = dev_index
    torch.cuda.set_device(local_rank)
    cfg = make_config(args.size)
    plan = build_plan(args.size, args.classes)
    eng = Engine(args.classes, cfg, max_batch=args.batch, dtype=args.dtype, device=f"cuda:{local_rank}",
                 alias_workspace=args.alias_workspace)
    ws_holder = {}

    def make_flat():
        ws_holder["ws"] = W.synth_weights(plan, seed=0)
        return W.flatten(ws_holder["ws"])

    D.load_weights_distributed(eng, make_flat, src=0)          # rank 0 packs, RCCL broadcast of the packed blob
    if os.environ.get("Y4_FORCE_ADOPT") == "1":
        # what every rank > 0 does, made testable on ONE GPU (tests/test_gpu_dist.py): a FRESH engine takes the packed bytes
        # that travelled through the collective (here: broadcast in place, then a device copy) and adopts them --
        # y4_adopt_packed_weights instead of y4_pack_weights -- and the bench then runs on that engine
        eng2 = Engine(args.classes, cfg, max_batch=args.batch, dtype=args.dtype, device=f"cuda:{local_rank}",
                      alias_workspace=args.alias_workspace)
        D.broadcast_bytes(eng.wts, 0)
        eng2.wts.copy_(eng.wts)
        torch.cuda.synchronize()
        eng2.adopt_packed()
        eng.close()
        eng = eng2
    lo, hi = D.shard_range(args.batch * world, rank, world)    # this rank's slice of the global batch
    lo, hi = lo + args.first_image, hi + args.first_image
    imgs = torch.from_numpy(W.synth_images(hi - lo, args.size, seed=0, first_index=lo)).to(eng.device)
    flat, outs = eng.alloc_outputs_flat(hi - lo)               # five outputs in one block: ONE D2H copy per step
    host = torch.empty(flat.numel(), dtype=torch.int32).pin_memory()
    if args.subbatch > 0:
        eng.set_subbatch(args.subbatch, args.sub_last_conv)
    fused_stem = args.dtype != "f32" and args.size <= 640 and not args.no_stem_fusion
    if fused_stem:
        eng.set_stem_fusion(True)          # convs 0+1 in one kernel; conv 1 then leaves the conv family below
    first_conv = 2 if fused_stem else 1
    chained = 0
    if args.dtype != "f32" and not args.no_chain_fusion:
        chained = eng.set_chain_fusion(True)      # runs of 2-3 convs (CSP stages) -> one conv_igemm launch each
    staged = False
    if args.dtype != "f32" and not args.no_stage_fusion and hasattr(eng, "set_stage_fusion"):
        staged = bool(eng.set_stage_fusion(True))  # convs 2..7 (304^2 CSP stage) as one spatially tiled kernel
    res_mask = 0
    if args.dtype != "f32" and not args.no_res_fusion and hasattr(eng, "set_res_fusion"):
        eng.set_res_fusion(True)                   # residual blocks of the 64/128-channel stages as one kernel each
        res_mask = eng.res_fusion_mask()
    tune_pair = False
    # the schedule: an explicit file, else the tuned one that ships for this shape (all fusions on, none of the debugging
    # switches), else a one-off autotune on this box.  Every schedule gives the same bits; `schedule` in the line says which
    schedule_src = "built-in heuristic (--no-autotune)"
    saved = None
    if args.load_tiles:
        saved = json.load(open(args.load_tiles))
        schedule_src = f"file {args.load_tiles}"
    elif not args.retune and not args.no_autotune and args.subbatch == 0 and args.dtype != "f32" and fused_stem and chained \
            and not args.no_stage_fusion and not args.no_res_fusion:
        saved = eng.shipped_schedule()
        if saved is not None and (int(saved.get("in_flight", args.in_flight)) != args.in_flight or len(saved.get("tiles", [])) != 110):
            saved = None
        if saved is not None:
            schedule_src = "shipped with the package: yolo4hip/schedules/" + os.path.basename(saved["path"])
            args.load_tiles = saved["path"]
    if saved is not None:
        tiles = saved["tiles"]
        eng.apply_schedule(saved)
        staged = bool(eng.stage_fusion_active()) if staged else False
        res_mask = eng.res_fusion_mask() if res_mask else 0
    elif not args.no_autotune:
        schedule_src = "autotuned in this process (y4_autotune_pair)" if args.in_flight > 1 and args.pair_passes > 0 else \
            "autotuned in this process (y4_autotune)"
        eng.predict_device(imgs, outs)                        # real activations in the workspace
        tune_pair = args.in_flight > 1 and args.pair_passes > 0     # some decisions judged with D batches in flight
        tiles = None
        if not tune_pair and rank == 0:                           # ONE tuning run per job: rank 0's, broadcast below
            tiles = eng.autotune(hi - lo, reps=args.tune_reps)    # untimed, one-off: fastest tile / fusion per layer (bit-identical results)
    else:
        tiles = None
        tune_pair = False

    # ---- the timed region.  `--in-flight D` (default 2): step i runs on slot i % D = engine / workspace / stream / resident
    # batch D of the rank's shard; a step is still ONE pass of the whole hot path over ONE batch of `--batch` images and
    # every step of a block completes inside its barrier + synchronize bracket -- steps of different slots merely overlap
    # on the GPU where one leaves compute units idle (partial last rounds, the 32-workgroup NMS, sub-one-round 19^2 layers).
    from yolo4hip.engine import InFlight
    depth = max(1, args.in_flight)
    fl = InFlight(eng, depth)
    if tune_pair and rank == 0:                                # engine 0 and its sibling tuned TOGETHER on their two streams
        fl.engines[1].predict_device(imgs, outs)
        tiles = fl.autotune(hi - lo, reps=args.tune_reps, passes=args.pair_passes)
    if not args.no_autotune and not args.load_tiles and world > 1:
        mine = {"tiles": tiles, "stage_fusion": bool(eng.stage_fusion_active()), "res_fusion_mask": int(eng.res_fusion_mask())} \
            if rank == 0 else None
        got = D.share_schedule(mine, src=0)                    # every rank runs rank 0's tile set
        if rank != 0:
            for e in fl.engines:
                e.apply_schedule(got)
            tiles = got["tiles"]
    if not args.no_autotune and not args.load_tiles:
        if staged:
            staged = bool(eng.stage_fusion_active())          # the tuner may have turned the stage kernel off
        res_mask = eng.res_fusion_mask() if res_mask else 0
    if args.save_tiles and rank == 0 and tiles:
        json.dump({"size": args.size, "classes": args.classes, "batch": args.batch, "dtype": args.dtype, "tiles": tiles,
                   "stage_fusion": bool(staged), "res_fusion_mask": int(res_mask), "in_flight": depth}, open(args.save_tiles, "w"))
    slot_imgs = [imgs] + [imgs.clone() for _ in range(depth - 1)]
    slot_out = [(flat, outs)] + [eng.alloc_outputs_flat(hi - lo) for _ in range(depth - 1)]
    slot_host = [host] + [torch.empty(flat.numel(), dtype=torch.int32).pin_memory() for _ in range(depth - 1)]
    torch.cuda.synchronize()
    counter = [0]

    def step():
        k = counter[0] % depth
        counter[0] += 1
        fl.submit(slot_imgs[k], slot_out[k][1], slot_host[k], slot_out[k][0])

    for _ in range(max(args.warmup, depth)):
        step()
    torch.cuda.synchronize()
    if depth == 1:                                           # one stream: the events sit in the timed blocks themselves
        eng.timing_begin(min(args.steps * args.blocks, 4096), coarse=not args.per_op)
    local, own = [], []
    for _ in range(args.blocks):
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        own.append(time.perf_counter() - t0)               # this rank's own K steps, before it waits for the others
        D.barrier()
        torch.cuda.synchronize()
        local.append(time.perf_counter() - t0)              # nothing but the K steps between the two brackets
    blocks = D.max_over_ranks(local)                        # one reduction, after every timed region
    same = all(torch.equal(slot_host[0], h) for h in slot_host[1:])     # every slot ran the same batch: same outputs
    # ---- kernel-level attribution: HIP events on the launch stream need kernels that run one after the other, so the
    # per-kernel-family times behind `roofline` / `breakdown_ms_per_step` come from a single-stream pass of the same step
    # (`--steps` steps, right here, same engine, same tiles) when D > 1: kernels of two streams overlap and an event pair
    # around one of them also spans its neighbour's work.  With D = 1 the events sit in the timed blocks themselves.
    if depth == 1:
        ops, nrec = eng.timing_end()
        single_ms = statistics.median(blocks) / args.steps * 1e3
    else:
        attrib_steps = min(args.steps, 4096)
        eng.timing_begin(attrib_steps, coarse=not args.per_op)   # 8 events per step unless --per-op
        t0 = time.perf_counter()
        for _ in range(attrib_steps):
            eng.predict_device(imgs, outs)
            host.copy_(flat, non_blocking=True)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t0) / attrib_steps * 1e3
        ops, nrec = eng.timing_end()
    import hashlib
    digest = hashlib.sha256(host.numpy().tobytes()).hexdigest() if same else "SLOTS DIFFER"   # outputs of the last step
    # ---- the backbone on its own, from WALL time, in both regimes: the forward cut behind conv
    # --stop-after-conv (71: CSPDarknet53 proper), K steps with one stream, then K steps with `depth` batches in flight
    backbone_wall = None
    if args.stop_after_conv >= 0:
        def backbone_pass(nslots):
            for k in range(nslots):                                  # warm-up, one per slot
                with torch.cuda.stream(fl.streams[k]):
                    fl.engines[k].forward_until_device(slot_imgs[k], args.stop_after_conv)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                k = i % nslots
                with torch.cuda.stream(fl.streams[k]):
                    fl.engines[k].forward_until_device(slot_imgs[k], args.stop_after_conv)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / args.steps * 1e3
        bb_flops = sum(c.flops_per_image for c in plan.convs[:args.stop_after_conv + 1]) * (hi - lo)
        peak_tf = MFMA_PEAK_TFLOPS[args.dtype]
        backbone_wall = {"last_conv": args.stop_after_conv, "flops_per_step": bb_flops, "steps": args.steps,
                         "timing": "wall clock around K y4_forward_until calls, synchronize on both sides, rank 0's own GPU"}
        for name, nslots in (("one_stream", 1),) + ((("in_flight_%d" % depth, depth),) if depth > 1 else ()):
            ms = min(backbone_pass(nslots) for _ in range(3))        # best of 3 short passes (each K steps)
            backbone_wall[name] = {"ms_per_step": round(ms, 4), "frac": round(bb_flops / (ms * 1e-3) / 1e12 / peak_tf, 4)}
    # every rank's own median block (a straggler must be visible the first time this runs on a real node)
    my_ms = statistics.median(own) / args.steps * 1e3
    rank_ms = [float(v) for v in D.gather_objects(my_ms)]
    rank_digests = D.gather_objects(digest)                  # every rank's outputs of its last step (its own shard of the batch)
    rank_tiles = D.gather_objects(hashlib.sha256(json.dumps(eng.get_tiles()).encode()).hexdigest()[:16])   # ... and its tile set

    if rank == 0:
        dt = statistics.median(blocks)
        n_img = args.batch * world * args.steps
        is_conv = lambda name: name.startswith("c") and name != "c0"
        conv_ms = sum(ms for name, ms in ops if is_conv(name))
        conv_flops = sum(c.flops_per_image for c in plan.convs[first_conv:]) * (hi - lo)
        other = {name: ms for name, ms in ops if not is_conv(name)}
        total_ms = sum(ms for _, ms in ops)
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        conv_launches = eng.conv_launches_per_step()
        traffic, traffic_source = committed_traffic(args, fused_stem, chained, staged, res_mask, tiles)
        # the other fractions of the same peak: the backbone north_star names (convs 0..71, from the coarse
        # segment that ends behind conv 71, plus the stem), the whole forward (every conv FLOP over stem + convs + SPP), and
        # end to end (every conv FLOP over the wall-clock step incl. decode, NMS and the results' copy to the host)
        op_ms = dict(ops)
        seg_names = [name for name, ms in ops if is_conv(name) and ms > 0]
        backbone_ms = other.get("c0", 0.0) + (op_ms.get(seg_names[0], 0.0) if seg_names else 0.0)
        if args.per_op:
            backbone_ms = sum(ms for name, ms in ops if name.startswith("c") and all(int(i) <= 71 for i in name[1:].split("+")))
        backbone_flops = sum(c.flops_per_image for c in plan.convs[:72]) * (hi - lo)
        fwd_ms = conv_ms + other.get("c0", 0.0) + other.get("spp", 0.0)
        fwd_flops = plan.flops_per_image * (hi - lo)
        frac_of = lambda flops, ms: round(flops / (ms * 1e-3) / 1e12 / peak, 4) if ms > 0 else None
        line = {
            "metric": METRIC,
            "value": round(n_img / dt, 2), "unit": "images/sec",
            "n_gpus": dist.get_world_size() if dist.is_initialized() else 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"yolov4 predict(): {args.size}x{args.size}x3 float32 images resident in HBM -> "
                                   f"CSPDarknet53+SPP+PANet forward ({plan.flops_per_image / 1e9:.3f} GFLOP/image) "
                                   f"-> 3-scale decode ({plan.num_boxes} boxes) -> class-aware NMS (100/image) -> "
                                   f"4 NMS outputs on the host; {args.classes} classes, {args.dtype} storage, "
                                   f"fp32 accumulate, seeded synthetic weights",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "sharding": f"batch split over {world} rank(s), no data-path collective"},
            "outputs_sha256": digest,
            "rank_outputs_sha256": rank_digests if world > 1 else None,
            "rank_tiles_sha256": rank_tiles if world > 1 else None,
            "first_image": lo,
            "blocks_ms_per_step": [round(b / args.steps * 1e3, 4) for b in blocks],
            "rank_ms_per_step": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4),
                                 "per_rank": [round(v, 4) for v in rank_ms],
                                 "what": "each rank's own K steps (median block, stopped BEFORE the closing barrier) / K; ms_per_step above is "
                                         "the barrier-bracketed block, max over ranks"},
            "timing": f"median of {args.blocks} blocks of {args.steps} steps, each barrier+synchronize bracketed, max over ranks; "
                      f"{depth} batch(es) in flight per GPU (step i on HIP stream / workspace i % {depth}, shared weights)",
            "in_flight": depth, "schedule": schedule_src,
            "activation_workspace_bytes": int(eng.act_bytes), "workspace_aliasing": bool(args.alias_workspace),
            "single_stream_ms_per_step": round(single_ms, 4),
            "single_stream_value": round(args.batch * world / (single_ms * 1e-3), 2) if world == 1 else None,
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         "backbone_frac": frac_of(backbone_flops, backbone_ms),
                         "backbone_ms_per_step": round(backbone_ms, 4),
                         "backbone_wall": backbone_wall,
                         "whole_forward_frac": frac_of(fwd_flops, fwd_ms),
                         "end_to_end_frac": round(n_img / dt * plan.flops_per_image / 1e12 / peak / world, 4),
                         "timed_region_frac": round(conv_flops / (dt / args.steps) / 1e12 / peak, 4),
                         "frac_definitions": "frac: conv family FLOPs / its HIP-event time; backbone_frac: convs 0..71 (CSPDarknet53, "
                                             "73.696 GFLOP/image at 608) / stem + conv segment up to conv 71 (HIP events, one stream); "
                                             "backbone_wall: the same FLOPs / WALL time of y4_forward_until(conv 71) alone, one stream and "
                                             "with the timed region's batches in flight; whole_forward_frac: all 110 "
                                             "convs / stem + convs + SPP; end_to_end_frac: all 110 convs x images/s (per GPU) / peak; timed_region_frac: the conv "
                                             "family's FLOPs / the WALL time of a step in the timed blocks (a lower bound on the family inside the "
                                             "timed region: that time also holds stem, SPP, decode, NMS and the copy of the results)",
                         # what the matrix pipes SUSTAIN on random 16-bit operands (round 6: a register-only MFMA loop, no memory):
                         # the power management lowers the clock to ~1.64 GHz under it -- `peak` stays the spec figure
                         "sustained_on_random_data": ({"peak": SUSTAINED_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / SUSTAINED_TFLOPS, 4),
                                                       "source": "scripts/ubench/mfma_power.hip, profiles/r06/mfma_power.txt: back-to-back "
                                                                 "v_mfma_f32_32x32x16_bf16 from registers on random normal operands, all 256 CUs: "
                                                                 "1617-1677 TFLOP/s at 1.64-1.70 GHz (zeros: 2157 at 2.17 GHz); DESIGN.md section 6a"}
                                                      if args.dtype != "f32" else None),
                         "traffic": traffic, "traffic_unit": "HBM bytes per step (all launches of the family)",
                         "traffic_source": traffic_source,
                         "kernel": "conv kernel family (convs %d..109, %d launches/step%s)" %
                                   (first_conv, conv_launches, (", convs 2..7 in csp_stage_kernel" if staged else "") +
                                    (", residual blocks of the %s-channel stages in resblock_kernel" %
                                     "/".join(c for b, c in ((1, "128"), (2, "64")) if res_mask & b) if res_mask else "")),
                         "flops_per_step": conv_flops, "kernel_ms_per_step": round(conv_ms, 4),
                         "timed_steps": nrec,
                         "measured_in": "HIP events on the launch stream over %d single-stream steps run right after the timed "
                                        "blocks in this process (kernels of overlapping streams cannot be attributed to one "
                                        "kernel); end_to_end_frac is from the timed blocks themselves" % nrec
                                        if depth > 1 else "HIP events on the launch stream over the timed blocks"},
            "breakdown_ms_per_step": {"conv_family": round(conv_ms, 4), ("stem_c0+c1_fused" if fused_stem else "stem_c0"): round(other.get("c0", 0.0), 4),
                                      "spp": round(other.get("spp", 0.0), 4),
                                      "decode": round(other.get("decode", 0.0), 4),
                                      "nms": round(other.get("nms", 0.0), 4), "sum_of_ops": round(total_ms, 4),
                                      "decode_nms_frac": round((other.get("decode", 0.0) + other.get("nms", 0.0)) /
                                                               max(total_ms, 1e-9), 4)},
        }
        if args.per_op:
            print("tiles:", tiles, file=sys.stderr)
            for name, ms in ops:
                fl = sum(plan.convs[int(i)].flops_per_image for i in name[1:].split("+")) * (hi - lo) if name.startswith("c") else 0
                print(f"{name:8s} {ms:8.4f} ms  {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.1f} TFLOP/s", file=sys.stderr)
        if world == 1 and not args.no_latency and args.classes == 80:
            pkg = os.path.join(ROOT, "yolo-v4-tf.keras_amd")
            line["latency"] = latency_block(args.classes, W.flatten(ws_holder.get("ws") or W.synth_weights(plan, seed=0)),
                                            os.path.join(pkg, "class_names", "coco_classes.txt"), os.path.join(pkg, "img", "street.jpeg"))
        if world == 1 and not args.no_cpu_baseline:
            ws = ws_holder.get("ws") or W.synth_weights(plan, seed=0)
            line["cpu_baseline"] = cpu_baseline(args.size, args.classes, ws, cfg, batch=args.cpu_batch)
        print(json.dumps(line), flush=True)
    D.barrier()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
