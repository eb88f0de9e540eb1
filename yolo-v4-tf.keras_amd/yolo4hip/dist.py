"""Multi-GPU inference: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI).

The reference never shards inference; its only parallelism is `tf.distribute.MirroredStrategy` around
training (reference models.py:41-44).  The inference analogue: images are independent
units, so the batch dimension is split across ranks with NO data-path collective.  Collectives:
  * init : one broadcast of the packed weight workspace from rank 0 (rank 0 reads/generates + packs;
           the other ranks adopt the bytes) -- a one-off, so the per-link ring bound does not matter;
  * steady state: none.  `gather_results` (optional) all_gathers the ~77 KB/rank of NMS outputs.
Everything except the engine calls works on CPU tensors with the gloo backend (tests/test_dist_cpu.py).
"""
import os

import numpy as np


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when not launched by it."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world == 1 and "RANK" not in os.environ:
        return rank, local_rank, world          # plain `python bench.py`: no process group at all
    if not dist.is_initialized():               # under torchrun (even with one rank) the RCCL path is exercised
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("Y4_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            assert local_rank < torch.cuda.device_count(), f"LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s)"
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def group_rank_world():
    """(rank, world size) of the initialised default process group; (0, 1) without one."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def share_schedule(saved, src=0):
    """Every rank returns rank `src`'s schedule dict (tile ids + fusion switches: `Engine.ensure_schedule`) -- or None if `src`
    has none.  A COLLECTIVE under a process group (host-side pickle broadcast, a few KB, init time only); the identity without."""
    import torch.distributed as dist
    if group_rank_world()[1] == 1:
        return saved
    box = [saved if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def shard_range(n_total, rank, world):
    """Contiguous block partition of images [0, n_total): rank r gets [lo, hi).  Remainders go to the
    lowest ranks, so shards differ by at most one image."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_bytes(buf, src=0):
    """Broadcast a uint8 tensor in place (the packed weight workspace)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() != "nccl" and buf.is_cuda:       # gloo (tests: two ranks on one GPU): through the host
            tmp = buf.cpu()
            dist.broadcast(tmp, src=src)
            buf.copy_(tmp)
        else:
            dist.broadcast(buf, src=src)
    return buf


def load_weights_distributed(engine, make_flat, src=0):
    """Rank `src` builds the Darknet float stream (`make_flat()`), packs it on its GPU; the packed
    workspace is broadcast over RCCL; the other ranks adopt it without touching the file system."""
    import torch
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if multi else 0
    if rank == src:
        engine.load_weight_blob(make_flat())
    if multi:
        torch.cuda.synchronize()
        broadcast_bytes(engine.wts, src)
        if rank != src:
            engine.adopt_packed()
        torch.cuda.synchronize()


def gather_results(outs, n_total=None):
    """all_gather per-rank NMS outputs (tensors with the image dimension first) and concatenate them in rank order;
    returns numpy arrays on every rank.  Shards made by `shard_range` may differ by one image when the batch does not
    divide by the world size, and all_gather needs equal shapes: every rank pads its rows to the largest shard, the
    shard sizes travel in a first (tiny) all_gather, and each rank's padding is stripped before concatenation.
    `n_total` (optional) is checked against the gathered row count."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        res = [o.cpu().numpy() if hasattr(o, "cpu") else np.asarray(o) for o in outs]
        if n_total is not None and res and res[0].shape[0] != n_total:
            raise ValueError(f"gather_results: {res[0].shape[0]} rows, expected {n_total}")
        return res
    world = dist.get_world_size()
    rows = int(outs[0].shape[0])
    mine = torch.tensor([rows], dtype=torch.int64, device=outs[0].device)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes = [int(t.item()) for t in sizes]
    cap = max(sizes)
    if n_total is not None and sum(sizes) != n_total:
        raise ValueError(f"gather_results: shards {sizes} sum to {sum(sizes)}, expected {n_total}")
    res = []
    for o in outs:
        o = o.contiguous()
        if int(o.shape[0]) != rows:
            raise ValueError("gather_results: outputs of one rank must share their first dimension")
        if rows < cap:
            pad = torch.zeros((cap - rows,) + tuple(o.shape[1:]), dtype=o.dtype, device=o.device)
            o = torch.cat([o, pad], dim=0)
        parts = [torch.empty_like(o) for _ in range(world)]
        dist.all_gather(parts, o)
        res.append(torch.cat([p[:sizes[r]] for r, p in enumerate(parts)], dim=0).cpu().numpy())
    return res


def max_over_ranks(value):
    """MAX over ranks of a float or of a list of floats (element-wise; ONE all_reduce for the whole list, so that a bench
    can time every block locally and reduce once, after its timed region)."""
    import torch
    import torch.distributed as dist
    many = isinstance(value, (list, tuple))
    vals = [float(v) for v in value] if many else [float(value)]
    if dist.is_available() and dist.is_initialized():
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        vals = [float(v) for v in t.cpu().tolist()]
    return vals if many else vals[0]


def gather_objects(obj):
    """[rank 0's obj, rank 1's, ...] on every rank (a small python object per rank: a timing, a device name); [obj] without a
    process group.  Host-side pickling -- never inside a timed region."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
