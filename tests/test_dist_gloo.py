"""-m "not gpu": the N>1 path of bench.py (stream sharding, barrier, max-over-ranks) with world_size 2 on gloo."""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from alego_loader import load_package
    load_package()
    from alego_amd import dist as D, synth
    from oracle import oracle_py as O
    dist = D.init("gloo")
    assert D.env() == (rank, rank, world)
    B = 2
    ids = D.stream_ids(rank, B)
    p = synth.default_params(16, 1800)
    poses = []
    for s in ids:  # the per-rank worker of this CPU test is the oracle; on GPUs it is the HIP handle
        o = O.Oracle(p)
        for k in range(3):
            o.process_scan(synth.scan(p, k, stream=s), stages=3)
        poses.append(o.get("odom_pose"))
    dist.barrier()
    dt = D.max_over_ranks(1.0 + rank, dist)  # rank 1 is "slower"
    gathered = [None] * world
    dist.all_gather_object(gathered, (ids, [x.tolist() for x in poses]))
    if rank == 0:
        q.put((dt, gathered, D.aggregate_scans_per_s(world, B, 3, dt)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_stream_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    dt, gathered, value = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert dt == 2.0  # max over ranks
    ids = sum((g[0] for g in gathered), [])
    assert sorted(ids) == [0, 1, 2, 3]  # disjoint shards covering all streams
    poses = np.array(sum((g[1] for g in gathered), []))
    assert len({tuple(np.round(x, 9)) for x in poses}) == 4  # four different streams were really processed
    assert value == 2 * 2 * 3 / 2.0
