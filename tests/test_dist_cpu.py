"""N > 1 path on CPU: two processes over the gloo backend exercise the same helpers bench.py and the
multi-GPU host use (weight-workspace broadcast from rank 0, batch sharding, result gather, max-over-ranks)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "yolo-v4-tf.keras_amd"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from yolo4hip import dist as D, weights as W
    r, lr, w = D.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    # 1. packed-weight broadcast: rank 0 holds the bytes, the others start from zeros
    blob = torch.arange(4096, dtype=torch.int32).view(torch.uint8).clone() if rank == 0 else torch.zeros(16384, dtype=torch.uint8)
    D.broadcast_bytes(blob, src=0)
    ok_bcast = bool((blob.view(torch.int32) == torch.arange(4096, dtype=torch.int32)).all())
    # 2. batch sharding: every rank builds ITS slice of the global synthetic batch
    n_total = 3 * world
    lo, hi = D.shard_range(n_total, rank, world)
    mine = W.synth_images(hi - lo, 16, seed=3, first_index=lo)
    # stand-in for per-image results: [n,4] + valid[n]
    res = torch.from_numpy(mine.reshape(hi - lo, -1)[:, :4].copy())
    valid = torch.arange(lo, hi, dtype=torch.int32)
    g = D.gather_results([res, valid], n_total)
    full = W.synth_images(n_total, 16, seed=3).reshape(n_total, -1)[:, :4]
    ok_gather = np.array_equal(g[0], full) and np.array_equal(g[1], np.arange(n_total, dtype=np.int32))
    # 2b. a batch that does not divide by the world size: shards of 3 and 2 images (the short shard is NOT the last
    #     rows of the padded gather, so stripping must be per rank)
    n_odd = 2 * world + 1
    lo2, hi2 = D.shard_range(n_odd, rank, world)
    g2 = D.gather_results([torch.arange(lo2, hi2, dtype=torch.int32),
                           torch.arange(lo2, hi2, dtype=torch.float32).view(-1, 1).repeat(1, 3)], n_odd)
    ok_gather = ok_gather and np.array_equal(g2[0], np.arange(n_odd, dtype=np.int32)) and \
        np.array_equal(g2[1], np.arange(n_odd, dtype=np.float32).reshape(-1, 1).repeat(3, axis=1))
    mx = D.max_over_ranks(10.0 + rank)
    # the bench's per-block times: taken locally, reduced element-wise with ONE all_reduce after the last timed block
    mxs = D.max_over_ranks([1.0 + rank, 5.0 - rank, 3.0])
    objs = D.gather_objects({"rank": rank, "dev": f"cuda:{lr}"})
    assert [o["rank"] for o in objs] == list(range(world)) and objs[rank]["dev"] == f"cuda:{rank}"
    # 3. an uneven global batch over the whole world (world 8: 35 images -> shards of 5,5,5,4,...): every image exactly once
    n_u = 4 * world + 3
    lo3, hi3 = D.shard_range(n_u, rank, world)
    g3 = D.gather_results([torch.arange(lo3, hi3, dtype=torch.int32)], n_u)
    assert np.array_equal(g3[0], np.arange(n_u, dtype=np.int32)) and 0 <= (hi3 - lo3) - n_u // world <= 1
    # 4. fewer images than ranks: the last rank's shard is EMPTY and still takes part in the gather
    lo4, hi4 = D.shard_range(world - 1, rank, world)
    g4 = D.gather_results([torch.arange(lo4, hi4, dtype=torch.int32), torch.zeros((hi4 - lo4, 100, 4))], world - 1)
    assert np.array_equal(g4[0], np.arange(world - 1, dtype=np.int32)) and g4[1].shape == (world - 1, 100, 4)
    # 5. one schedule per job: every rank ends up with rank 0's dict (Engine.ensure_schedule(share=True), bench.py), None included
    assert D.group_rank_world() == (rank, world)
    sched = D.share_schedule({"tiles": list(range(110)), "stage_fusion": True, "res_fusion_mask": 5} if rank == 0 else {"tiles": [rank]})
    assert sched == {"tiles": list(range(110)), "stage_fusion": True, "res_fusion_mask": 5}
    assert D.share_schedule(None if rank == 0 else {"tiles": [1]}) is None
    # 6. Engine.ensure_schedule(share=True) on a stand-in engine (no GPU here): the schedule reaches every rank and is applied there;
    #    a rank 0 that raises while resolving broadcasts the failure (the others raise instead of hanging in the collective); an
    #    engine of another shape than rank 0's refuses rank 0's tile ids
    from yolo4hip.engine import Engine

    class Fake:
        img_size, num_classes, max_batch, dtype = 608, 80, 32, "bf16"
        used = None
        fail = False
        def _resolve_schedule(self, tune):
            if self.fail:
                raise ValueError("tuning ran out of memory")
            return "shipped", "/x.json", {"tiles": [7] * 110}
        def _use_schedule(self, saved): self.used = saved
        def say_schedule(self): pass
    f = Fake()
    src, _ = Engine.ensure_schedule(f, tune=True, verbose=False, share=True)
    assert src == ("shipped" if rank == 0 else "shared") and (rank == 0 or f.used == {"tiles": [7] * 110})
    f = Fake(); f.fail = True
    try:
        Engine.ensure_schedule(f, tune=True, verbose=False, share=True)
        raised = False
    except RuntimeError as e:
        raised = "rank 0 could not resolve" in str(e) and "out of memory" in str(e)
    assert raised
    f = Fake()
    if rank == world - 1:
        f.max_batch = 8
    try:
        Engine.ensure_schedule(f, tune=True, verbose=False, share=True)
        assert rank != world - 1, "a schedule for another batch size was applied"
    except RuntimeError as e:
        assert rank == world - 1 and "every rank must build the same engine" in str(e)
    D.barrier()
    ret[rank] = (ok_bcast, ok_gather, mx, mxs)
    torch.distributed.destroy_process_group()


def _protocol(world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        ok_bcast, ok_gather, mx, mxs = ret[rank]
        assert ok_bcast and ok_gather and mx == 9.0 + world and mxs == [float(world), 5.0, 3.0]


def test_share_schedule_is_the_identity_without_a_process_group():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yolo-v4-tf.keras_amd"))
    from yolo4hip import dist as D
    assert D.group_rank_world() == (0, 1)
    d = {"tiles": [1, 2, 3]}
    assert D.share_schedule(d) is d and D.share_schedule(None) is None


def test_two_rank_gloo_protocol():
    _protocol(2)


def test_eight_rank_gloo_protocol():
    """The shape of the driver's 8-GPU launch (one process per GPU of one node), on CPU."""
    _protocol(8)
