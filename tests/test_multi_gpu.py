"""-m gpu: both multi-GPU modes on the HIP path over RCCL, one process per GPU.

world = 1 runs everywhere (one rank: the same spawn / RCCL / all-gather code, a one-rank communicator); world = 2 needs two
visible GPUs and is skipped otherwise — the driver's GPU box has one, whoever has a node runs
`python -m pytest tests/test_multi_gpu.py -m gpu`.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SCANS_SHARDING = 14
N_FRAMES_REGISTRATION = 200      # mapping frames (every 2nd scan)


def _setup(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # RCCL's streams next to the handle's
    import torch
    torch.cuda.set_device(rank)
    from alego_loader import load_package
    load_package()
    from alego_amd import binding, dist as D, synth
    dist = D.init("nccl", torch.device("cuda", rank))
    return torch, dist, D, binding, synth


def _worker_stream_sharding(rank, world, port, q):
    """(i) streams across ranks, no data-path collective: rank r advances streams [r B, (r + 1) B) on its own GPU; rank 0 also runs
    every stream itself and all ranks' poses must equal its own, bit for bit."""
    torch, dist, D, binding, synth = _setup(rank, world, port)
    p = synth.default_params(16, 1800)
    B = 2

    def run(ids, device):
        h = binding.Handle(p, device=device, n_slots=len(ids), ring_len=N_SCANS_SHARDING)
        for j, s in enumerate(ids):
            for k in range(N_SCANS_SHARDING):
                h.batch_load(j, k, synth.scan(p, k, stream=s))
        h.batch_run(0, N_SCANS_SHARDING, stages=7)
        out = []
        for j in range(len(ids)):
            _, o, m = h.batch_get_pose(j)
            out.append(np.concatenate([o["t"], o["q"], o["params"], m["t"], m["q"], m["params"]]))
        h.close()
        return np.array(out)

    ids = D.stream_ids(rank, B)
    mine = run(ids, rank)
    dist.barrier()
    rates = D.gather_floats(float(rank + 1), dist, device="cuda")
    gathered = [None] * world
    dist.all_gather_object(gathered, (ids, mine.tobytes()))
    if rank == 0:
        ok, why = len(rates) == world and rates == [float(r + 1) for r in range(world)], ""
        for ids_r, blob in gathered:
            want = run(ids_r, 0)
            got = np.frombuffer(blob, np.float64).reshape(want.shape)
            if not np.array_equal(got.view(np.uint64), want.view(np.uint64)):
                ok, why = False, f"streams {ids_r}: poses differ from the 1-GPU run"
        q.put((ok, why))
    dist.barrier()
    dist.destroy_process_group()


def _worker_sharded_registration(rank, world, port, q):
    """(ii) ONE registration across the ranks (alego_dist_init): every rank replays the same stream, each solver evaluation sums the
    ranks' partial normal equations with ncclAllReduce.  Every scan starts from the params_ of a fused-solver handle on rank 0
    (teacher forcing, as in the oracle tests); after every mapping frame all ranks must hold IDENTICAL params_, within 1e-9 of
    the fused solver's."""
    torch, dist, D, binding, synth = _setup(rank, world, port)
    p = synth.default_params(16, 1800)
    hs = binding.Handle(p, device=rank)
    D.shard_registration(hs, dist, rank, world)
    hf = binding.Handle(p, device=rank) if rank == 0 else None   # the fused on-chip solver
    worst, frames, ok, why = 0.0, 0, True, ""
    for k in range(2 * N_FRAMES_REGISTRATION + 1):
        pts = synth.scan(p, k)
        teach = torch.zeros(12, dtype=torch.float64, device="cuda")
        if rank == 0:
            _, of, mf = hf.batch_get_pose(0)
            teach = torch.tensor(np.concatenate([of["params"], mf["params"]]), dtype=torch.float64, device="cuda")
        dist.broadcast(teach, src=0)
        t = teach.cpu().numpy()
        hs.set_lo_params(t[:6]); hs.set_lm_params(t[6:])
        _, _, ms = hs.scan_process(pts, stages=7)
        every = [torch.zeros(6, dtype=torch.float64, device="cuda") for _ in range(world)]
        dist.all_gather(every, torch.tensor(ms["params"], dtype=torch.float64, device="cuda"))
        if rank == 0:
            _, _, mf = hf.scan_process(pts, stages=7)
            ran = bool(hf.debug_get("lm_info")[2])
            frames += int(ran)
            e = [x.cpu().numpy() for x in every]
            if any(not np.array_equal(x.view(np.uint64), e[0].view(np.uint64)) for x in e[1:]):
                ok, why = False, f"scan {k}: the ranks end the frame with different params_"
            worst = max(worst, float(np.abs(e[0] - mf["params"]).max()))
    if rank == 0:
        if worst > 1e-9:
            ok, why = False, f"sharded params_ {worst:.3e} away from the fused solver"
        if frames < N_FRAMES_REGISTRATION:
            ok, why = False, f"only {frames} mapping frames ran"
        q.put((ok, why))
        hf.close()
    dist.barrier()
    hs.dist_shutdown()
    hs.close()
    dist.destroy_process_group()


def _run(worker, world):
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, {torch.cuda.device_count()} visible")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300) + 7 * world
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    try:
        ok, why = q.get(timeout=900)
    finally:
        [p.join(120) for p in procs]
        for p in procs:
            if p.is_alive():
                p.kill()     # (the exact process we started)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ok, why


@pytest.mark.parametrize("world", [1, 2])
def test_streams_sharded_across_ranks_rccl(world):
    _run(_worker_stream_sharding, world)


@pytest.mark.parametrize("world", [1, 2])
def test_one_registration_sharded_across_ranks_rccl(world):
    _run(_worker_sharded_registration, world)
