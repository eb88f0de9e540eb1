#!/usr/bin/env python3
"""Headline benchmark: images/sec of the whole predict() hot path (forward + decode + NMS) at
608x608, 80 classes, batch 32 per GPU, bf16 storage / fp32 accumulate, synthetic data.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A step = one y4_predict over one batch of 32 images already resident in HBM (float32 NHWC in [0,1]; the
PCIe-inclusive rate is noted in DESIGN.md, it is never `value`).  Timed region: barrier +
torch.cuda.synchronize() on both sides, MAX over ranks, whole-job images / time.
`roofline` is for the dominant kernel family (conv_igemm_kernel: convs 2..109, or 1..109 with --no-stem-fusion): algorithmic conv FLOPs of
one step / its summed per-launch device time, measured with HIP events recorded on the launch stream
inside the timed region (y4_timing_begin/end).  `cpu_baseline` times the oracle (a torch-CPU/NumPy
restatement; the reference's tf.keras path cannot run here) on a bounded sample on rank 0 at N=1.
With no flags the whole command (weight synthesis and packing, one-off tile / fusion autotune, 3 + 20 steps, ~9 s of
CPU baseline) takes 16 s wall on an MI355X box (measured).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "yolo-v4-tf.keras_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # dense, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(size, ncls, ws, cfg, sample_n=8, runs=3):
    """Oracle (kind 'port') on the host cores: forward + decode + NMS of `sample_n` images."""
    import numpy as np
    import torch
    from yolo4hip import weights as W
    from oracle import forward as OF, decode_nms as OD
    # measured on the GPU box's host (2x EPYC 9575F, 256 hardware threads): oneDNN is fastest around 32
    # threads (3.9 img/s) and collapses when every hardware thread is used (0.02 img/s at 256)
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    imgs = W.synth_images(sample_n, size, seed=0)
    def once():
        heads = OF.yolo_model_forward(imgs, ws, ncls)
        return OD.inference_from_heads(heads, ncls, cfg["anchors"], cfg["xyscale"], size)
    once()                                    # warm-up (oneDNN primitive creation, page-in)
    t0 = time.perf_counter()
    for _ in range(runs):
        once()
    dt = time.perf_counter() - t0
    return {"value": round(sample_n * runs / dt, 4), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{runs} runs x {sample_n} images {size}x{size}x3, {ncls} classes, fp32 torch-CPU(oneDNN) forward "
                      f"+ NumPy decode/NMS (oracle/), after 1 warm-up run; {dt:.1f} s of CPU work"}


def measured_traffic(args, fused_stem, chained, launches):
    """HBM bytes per conv_igemm launch from the committed PMC passes (profiles/r01/hbm_traffic_v3.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, corrected as MI355X_MICROARCH.md prescribes),
    or None when this run's configuration is not the profiled one.  Counters cannot be read from inside the process."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "hbm_traffic_v3.json")
    try:
        prof = json.load(open(path))
    except (OSError, ValueError):
        return None
    want = {"size": args.size, "classes": args.classes, "batch": args.batch, "dtype": args.dtype,
            "stem_fusion": bool(fused_stem), "chain_fusion": bool(chained)}
    if prof.get("config") != want or launches <= 0:
        return None
    return round(prof["conv_igemm_hbm_bytes_per_step"] / launches)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--classes", type=int, default=80)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-autotune", action="store_true", help="use the built-in tile heuristic instead of measuring")
    ap.add_argument("--tune-reps", type=int, default=3)
    ap.add_argument("--save-tiles", default=None, help="write the autotuned per-layer tile ids to this JSON file")
    ap.add_argument("--load-tiles", default=None, help="use tile ids from this JSON file instead of autotuning")
    ap.add_argument("--subbatch", type=int, default=0, help="images per sub-batch for the early layers (0 = whole batch)")
    ap.add_argument("--sub-last-conv", type=int, default=16)
    ap.add_argument("--no-stem-fusion", action="store_true", help="run convs 0 and 1 as two kernels (c0 through HBM)")
    ap.add_argument("--no-chain-fusion", action="store_true", help="run the 3x3+Add -> 1x1 -> 1x1 runs as separate kernels")
    ap.add_argument("--per-op", action="store_true", help="also print the per-op time table to stderr")
    args = ap.parse_args()

    import torch
    from yolo4hip import dist as D, weights as W
    from yolo4hip.config import make_config
    from yolo4hip.engine import Engine
    from yolo4hip.plan import build_plan

    rank, local_rank, world = D.init_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    torch.cuda.set_device(local_rank)
    cfg = make_config(args.size)
    plan = build_plan(args.size, args.classes)
    eng = Engine(args.classes, cfg, max_batch=args.batch, dtype=args.dtype, device=f"cuda:{local_rank}")
    ws_holder = {}

    def make_flat():
        ws_holder["ws"] = W.synth_weights(plan, seed=0)
        return W.flatten(ws_holder["ws"])

    D.load_weights_distributed(eng, make_flat, src=0)          # rank 0 packs, RCCL broadcast of the packed blob
    lo, hi = D.shard_range(args.batch * world, rank, world)    # this rank's slice of the global batch
    imgs = torch.from_numpy(W.synth_images(hi - lo, args.size, seed=0, first_index=lo)).to(eng.device)
    outs = eng.alloc_outputs(hi - lo)
    if args.subbatch > 0:
        eng.set_subbatch(args.subbatch, args.sub_last_conv)
    fused_stem = args.dtype != "f32" and args.size <= 640 and not args.no_stem_fusion
    if fused_stem:
        eng.set_stem_fusion(True)          # convs 0+1 in one kernel; conv 1 then leaves the conv_igemm family below
    first_conv = 2 if fused_stem else 1
    chained = 0
    if args.dtype != "f32" and not args.no_chain_fusion:
        chained = eng.set_chain_fusion(True)      # 25 runs of 2-3 convs (CSP stages) -> one conv_igemm launch each
    if args.load_tiles:
        tiles = json.load(open(args.load_tiles))["tiles"]
        eng.set_tiles(tiles)
    elif not args.no_autotune:
        eng.predict_device(imgs, outs)                        # real activations in the workspace
        tiles = eng.autotune(hi - lo, reps=args.tune_reps)                         # untimed, one-off: fastest tile per layer (bit-identical results)
    else:
        tiles = None
    if args.save_tiles and rank == 0 and tiles:
        json.dump({"size": args.size, "classes": args.classes, "batch": args.batch, "dtype": args.dtype, "tiles": tiles},
                  open(args.save_tiles, "w"))

    for _ in range(args.warmup):
        eng.predict_device(imgs, outs)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    eng.timing_begin(args.steps, coarse=not args.per_op)    # 7 events per step (conv runs timed as a whole) unless --per-op
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.predict_device(imgs, outs)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0)
    ops, nrec = eng.timing_end()

    if rank == 0:
        n_img = args.batch * world * args.steps
        conv_ms = sum(ms for name, ms in ops if name.startswith("c") and name != "c0")
        conv_flops = sum(c.flops_per_image for c in plan.convs[first_conv:]) * (hi - lo)
        launches = sum(1 for name, _ in ops if name.startswith("c") and name != "c0")   # fused CSP pairs count once
        other = {name: ms for name, ms in ops if not name.startswith("c") or name == "c0"}
        total_ms = sum(ms for _, ms in ops)
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        # tails that run inside their head's kernel: heads are reported as -tile by the autotuner / tile file
        tails_of = {2: 1, 5: 2, 8: 1, 12: 1, 14: 2, 17: 1}
        tails_of.update({h: 1 for h in list(range(21, 36, 2)) + list(range(42, 57, 2)) + [88, 90, 92]})   # LDS pairs
        fused_tails = sum(n_t for head, n_t in tails_of.items() if chained and (tiles is None or tiles[head] <= 0))
        conv_launches = launches - (first_conv - 1) - fused_tails
        line = {
            "metric": "images/sec end-to-end predict() at 608x608 batch 32; conv MFMA %peak",
            "value": round(n_img / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"yolov4 predict(): {args.size}x{args.size}x3 float32 images resident in HBM -> "
                                   f"CSPDarknet53+SPP+PANet forward ({plan.flops_per_image / 1e9:.3f} GFLOP/image) "
                                   f"-> 3-scale decode ({plan.num_boxes} boxes) -> class-aware NMS (100/image); "
                                   f"{args.classes} classes, {args.dtype} storage, fp32 accumulate, "
                                   f"seeded synthetic weights",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "sharding": f"batch split over {world} rank(s), no data-path collective"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         "traffic": measured_traffic(args, fused_stem, chained, conv_launches),
                         "kernel": "conv_igemm_kernel (convs %d..109, %d launches/step)" % (first_conv, conv_launches),
                         "flops_per_step": conv_flops, "kernel_ms_per_step": round(conv_ms, 4),
                         "timed_steps": nrec},
            "breakdown_ms_per_step": {"conv_igemm": round(conv_ms, 4), ("stem_c0+c1_fused" if fused_stem else "stem_c0"): round(other.get("c0", 0.0), 4),
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
        if world == 1 and not args.no_cpu_baseline:
            ws = ws_holder.get("ws") or W.synth_weights(plan, seed=0)
            line["cpu_baseline"] = cpu_baseline(args.size, args.classes, ws, cfg)
        print(json.dumps(line), flush=True)
    D.barrier()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
