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


def _bench_under_launcher(extra, port_off, timeout=900):
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 ...` exactly as the driver launches the N > 1 points of the scaling curve
    (ALEGO_BENCH_FORCE_DIST=1 makes the one rank create the RCCL process group, barrier and gather through it as N ranks would)."""
    import json
    import subprocess
    env = dict(os.environ, ALEGO_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ALEGO_STREAM_GROUPS", None)
    port = 29900 + (os.getpid() % 200) + port_off
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_launcher_path_self_check():
    """the launcher path of bench.py — what the driver's first multi-GPU run goes through — inside the suite: one rank under
    torch.distributed.run on an RCCL process group, 256 streams in stream groups, `--self-check`: after the timed region the rank's slot 0 must equal,
    bit for bit, a one-slot handle that replays the same bag (exit code 3 and bit_equal = false otherwise)."""
    j = _bench_under_launcher(["--steps", "10", "--warmup", "2", "--streams", "256", "--bags", "2", "--prime", "60", "--self-check",
                               "--no-cpu", "--no-check", "--no-isolated"], 0)
    assert j["n_gpus"] == 1 and j["ranks_seen_by_rccl"] == 1 and j["scaling"] == "weak"
    assert j["self_check"]["bit_equal"] is True and j["self_check"]["ranks_checked"] == 1, j["self_check"]
    assert j["truncated_streams"] == 0 and j["value"] > 0
    assert abs(j["value"] - 256 * 10 / (j["ms_per_step"] * 10 / 1e3)) / j["value"] < 1e-3      # value = streams x steps / timed region
    assert j["roofline"]["frac"] > 0 and j["roofline"]["in_B_scan"] is True   # (with 64 streams per launch the latency-bound lo_solve leads the device time, not ip_fused_h)
    assert len(j["roofline_top3"]) == 3 and all(a["share_of_device_time"] >= b["share_of_device_time"] for a, b in zip(j["roofline_top3"], j["roofline_top3"][1:]))


def test_bench_shard_registration_line_labels_the_expected_slowdown():
    """config 5 as a bench line at world = 1: the registration on the sharded kernel sequence (one RCCL all-reduce per solver evaluation) with the fused
    solver timed next to it on the same streams, so that a reader of the first N > 1 run sees `shard_vs_fused` and its `expected` note instead of
    discovering negative strong scaling."""
    j = _bench_under_launcher(["--steps", "8", "--warmup", "2", "--streams", "8", "--bags", "1", "--prime", "40", "--shard-registration",
                               "--no-cpu", "--no-check", "--no-profile"], 1)
    assert j["scaling"] == "strong" and "sharded" in j["config"]["parallelism"]
    sv = j["shard_vs_fused"]
    assert sv["world"] == 1 and sv["fused_scans_per_s"] > 0 and sv["sharded_scans_per_s"] > 0 and "expected" in sv
    assert abs(sv["sharded_scans_per_s"] - j["value"]) / j["value"] < 1e-3
    assert j["shard_allreduce"]["doubles"] == 32
