#!/usr/bin/env python3
"""bench.py — scans/sec of the full IP -> LO -> LM loop on 16x1800 synthetic LiDAR streams (BASELINE.json config 3 / 4).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload: "bag-equivalent" replay (test_0515.bag is not in the reference repository).  `--bags` recorded streams of one
560-scan T0 lap each (stream s starts 70 s scans further along the lap, own noise seed) are resident in HBM
once; each of the `--streams` independent streams of a rank replays one of them cyclically from its own start scan, so every
stream sees the true trajectory (560 distinct scans per lap, never a scan twice in a row) and no two streams are in the
same place.  A "step" advances every resident stream by one scan (ImageProjection, feature extraction + LaserOdometry,
LaserMapping on every 2nd scan); inputs are in HBM before the timed region.  One process per GPU; streams are independent
(no data-path collective), so N GPUs run N x `--streams` streams: weak scaling.  Rank 0 prints one JSON line.

The CPU legs (rank 0, N = 1) run the oracle on exactly the scan sequence of stream 0 (cpu_seq, cpu_pipe3) and of
streams 0..C-1 (cpu_replicas), BASELINE.md §2.
"""
import argparse
import contextlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# two HIP streams per stream group (front end + LaserMapping) + RCCL's own: the runtime's default of 4 hardware queues would alias them
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from alego_loader import load_package  # noqa: E402

load_package()
from alego_amd import binding, synth  # noqa: E402
from alego_amd import dist as D  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md)
LAP = 560          # scans per T0 lap
PMC_ROUNDS = ("r06", "r05", "r04")   # the newest committed --pmc passes of a workload are used


def _pmc_path(stem):
    for r in PMC_ROUNDS:
        f = os.path.join(ROOT, "profiles", r + stem)
        if os.path.exists(f):
            return f
    return os.path.join(ROOT, "profiles", PMC_ROUNDS[0] + stem)


PMC_FILE = _pmc_path("_pmc_traffic.json")


def algorithmic_bytes(c, NS):
    """SURVEY.md §8(d) compulsory HBM bytes of one scan, from the device counters of that scan."""
    b_ip = 16 * c["P"] + 25 * c["M"] + 16 * c["O"] + 8 * NS + 12
    b_fe = 9 * c["M"] + 16 * (c["Qc"] + c["Fc"] + c["Qs"] + c["Fs"])
    b_lo = 16 * (c["Fc"] + c["Fs"] + c["Qc"] + c["Qs"]) + 104
    kraw, kds = c["Kraw_c"] + c["Kraw_s"], c["Kds_c"] + c["Kds_s"]
    b_lm = 16 * kraw + 32 * kds + 16 * (c["Lc"] + c["Ls"]) + 104
    return dict(B_IP=b_ip, B_FE=b_fe, B_LO=b_lo, B_LM=b_lm, B_scan=b_ip + b_fe + b_lo + b_lm / 2)


# kernels whose byte term is NOT one of SURVEY.md §8(d)'s (their data are "on-chip intermediates" by §8(d)'s definition, so B_scan does not contain
# them): charged with what crosses the kernel's own boundary — the arrays it has to read once plus the arrays it has to write once — so that every
# kernel with a large share of the device time has a roofline row
BOUNDARY_ONLY = ("vox_small", "map_update", "lo_grid_build", "lm_fit", "lm_stage")
LO_GRID_CELLS = 4096   # LO_GC, kernels_lo.hip


def kernel_bytes(name, c, NS, H, rebuilds_per_launch=0.0, K=50):
    """Algorithmic HBM bytes one launch of `name` has to move for ONE stream: the terms of SURVEY.md §8(d) the kernel is
    responsible for (stage inputs it reads, stage outputs it writes).  Intermediates between the kernels of a stage (owner /
    range / flag images, index lists, sort scratch) are on-chip by §8(d)'s definition and are NOT charged, so the figures of a
    stage's kernels sum to that stage's B_*.  The map kernels only work for the streams whose key-frame set changed: their
    bytes are scaled by the measured rebuilds per launch.  The kernels of BOUNDARY_ONLY have no §8(d) term; theirs is the data that
    crosses their own boundary (read once + written once), reported with "in_B_scan": false."""
    name = name.strip("()").split("<")[0]
    P, M, O = c["P"], c["M"], c["O"]
    feats = c["Qc"] + c["Fc"] + c["Qs"] + c["Fs"]
    kraw, kds = c["Kraw_c"] + c["Kraw_s"], c["Kds_c"] + c["Kds_s"]
    L = c["Lc"] + c["Ls"]
    rb = rebuilds_per_launch
    t = {
        # B_IP = 16 P (in) + 25 M + 16 O + 8 NS + 12 (out)
        "ip_fused": 16 * P + 25 * M + 16 * O + 8 * NS + 12,   # = B_IP (the curvature phase left the kernel in round 3; fe_* are charged B_FE)
        "ip_fused_h": 16 * P + 25 * M + 16 * O + 8 * NS + 12, "ip_fused_w": 16 * P + 25 * M + 16 * O + 8 * NS + 12,
        "ip_project": 16 * P, "ip_front": 12, "cc_lds16": 25 * M + 16 * O + 8 * NS, "cc_lds": 25 * M + 16 * O + 8 * NS,
        "ip_compact": 25 * M + 16 * O + 8 * NS,
        # B_FE = 9 M (range, col, ground in) + 16 feats (out)
        "fe_curv": 8 * M, "fe_pick4": M, "fe_pick": M, "fe_gather": 16 * feats, "fe_collect": 16 * feats,
        "fe_cand": 9 * M, "fe_ring_out": 16 * feats,   # (fe_pickc works on the candidate lists: intermediates)
        # B_LO = 16 (F' + Q) + 104
        "lo_assoc": 16 * (c["Fc"] + c["Fs"] + c["Qc"] + c["Qs"]) / 2, "lo_solve": 104 / 2, "lo_solve_t": 104 / 2,
        # B_LM = 16 Kraw + 32 Kds + 16 L + 104 per mapping frame
        "vox_big": (16 * kraw + 16 * kds) * rb, "map_accum": (16 * kraw + 16 * kds) * rb, "lm_grid_build": 16 * kds * rb,
        "lm_knn": 16 * L, "lm_solve": 104,
        # ---- BOUNDARY_ONLY (not part of B_scan) ----
        # two VoxelGrid rounds per mapping frame (laserMapping.cpp:329-342): corner / surf / outlier of the scan in, their filtered clouds out, then
        # surf + outlier -> surf_total; averaged over the two launches (the sort of a new key frame, one mapping frame in six, is not charged)
        "vox_small": (16 * (c["Fc"] + c["Fs"] + O) + 16 * L + 32 * c["Ls"]) / 2,
        # per rebuilt map: the sorted voxel list (8 B key + 4 B count) read and written, the keys of the leaving and the entering run
        "map_update": (24 * kds + 32 * kraw / max(K, 1)) * rb,
        # both target clouds read, written again in cell order, + two cell tables of (LO_GC + 2) 16-bit prefixes
        "lo_grid_build": 32 * (c["Fc"] + c["Fs"]) + 4 * (LO_GRID_CELLS + 2),
        # per query: five neighbour indices + the five map points they name in, one 64-byte residual block out
        "lm_fit": (20 + 80 + 64) * L,
        # every scan: /odom/lidar handed over; mapping frames (every 2nd launch): the three clouds copied into LaserMapping's inputs
        "lm_stage": 64 + 32 * (c["Fc"] + c["Fs"] + O) / 2,
    }
    return t.get(name)


def kernel_table(rep):
    tot = sum(v[0] for v in rep.values())
    return {k: dict(ms_total=round(v[0], 3), launches=v[1], avg_us=round(1e3 * v[0] / max(v[1], 1), 2),
                    share=round(v[0] / tot, 4)) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0])}


def set_pmc_file(geometry, keyframes):
    """the committed --pmc passes of this workload: profiles/rNN_pmc_traffic.json (16x1800, the default) or profiles/rNN_geo_<geometry>[_k<K>]_pmc_traffic.json"""
    global PMC_FILE
    if geometry.lower() != "16x1800":
        PMC_FILE = _pmc_path("_geo_%s%s_pmc_traffic.json" % (geometry.lower(), ("_k%d" % keyframes) if keyframes > 0 else ""))


def pmc_bytes_per_scan():
    """measured HBM bytes per scan of the whole pipeline (tools/pmc_traffic.py: per-kernel bytes per launch x launches per scan)"""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        return dict(corrected=d["hbm_bytes_per_scan"], uncorrected=d["hbm_bytes_per_scan_uncorrected"])
    except (OSError, ValueError, KeyError):
        return None


def pmc_traffic(kernel, streams_per_launch):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes; only valid for the launch shape they were collected on."""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        if d.get("streams_per_launch") != streams_per_launch:
            return None
        return d["kernels"].get(kernel.strip("()").split("<")[0], {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError, KeyError):
        return None


def make_bags(p, n_bags, first_stream, flags=0):
    """One T0 lap (560 scans) of each of `n_bags` streams, on the host."""
    jobs = [(b, k) for b in range(n_bags) for k in range(LAP)]
    bags = [[None] * LAP for _ in range(n_bags)]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for (b, k), a in zip(jobs, ex.map(lambda bk: synth.scan(p, bk[1], stream=first_stream + bk[0], flags=flags), jobs)):
            bags[b][k] = a
    return bags


def slot_source(slot, n_bags):
    """(bag, start scan) of a slot: bags round-robin, start scans spread over the lap."""
    return slot % n_bags, ((slot // n_bags) * 37) % LAP


def setup_replay(h, bags, n_slots):
    h.replay_create(len(bags), LAP)
    for b, bag in enumerate(bags):
        for k, a in enumerate(bag):
            h.replay_load(b, k, a)
    for s in range(n_slots):
        h.replay_assign(s, *slot_source(s, len(bags)))


def single_stream(p, bag, device, prime, steps, lanes=8):
    """configs[2] as written — ONE bag stream at unbounded rate: the plain replay of a one-slot handle (every kernel of a scan
    behind the previous one) and alego_stream_run (IP + FE `lanes` scans ahead in shared launches, LO and LM on their own streams)."""
    res = {}
    h1 = binding.Handle(p, device=device, n_slots=1, ring_len=1)
    setup_replay(h1, [bag], 1)
    st = 7 | binding.REPLAY_BAG
    h1.batch_run(0, prime, st)
    t1 = time.perf_counter()
    h1.batch_run(prime, steps, st)
    res["single_stream_serial_scans_per_s"] = round(steps / (time.perf_counter() - t1), 1)
    _, o1, m1 = h1.batch_get_pose(0)
    h1.close()
    h2 = binding.Handle(p, device=device, n_slots=1 + 2 * lanes, ring_len=1)
    h2.replay_create(1, LAP)
    for k, a in enumerate(bag):
        h2.replay_load(0, k, a)
    h2.stream_setup(0, 0)
    h2.stream_run(0, prime, 7)
    t1 = time.perf_counter()
    h2.stream_run(prime, steps, 7)
    res["single_stream_scans_per_s"] = round(steps / (time.perf_counter() - t1), 1)
    _, o2, m2 = h2.batch_get_pose(0)
    h2.close()
    res["single_stream_note"] = (f"{steps} scans after {prime}; look-ahead {lanes} scans; poses bit-identical to the serial replay: "
                                 f"{bool(np.array_equal(m1['t'], m2['t']) and np.array_equal(m1['params'], m2['params']) and np.array_equal(o1['t'], o2['t']))}")
    return res


CHECK_ARRAYS = ("seg_cloud", "seg_col", "less_sharp", "less_flat", "lm_corner_map_ds", "lm_surf_map_ds")


def timed_handle_check(h, p, bags, B, total_steps, device, oracle_map_t=None, per_group=4):
    """The handle that was TIMED (all its slots, all its stream groups, bag replay) against one-slot handles: `per_group` slots of
    every stream group (first and last slot of the group included) are replayed alone — same bag, same start scan, same number
    of steps — and odometry pose, LaserMapping params_ / pose, the segmented cloud, both feature clouds and both filtered maps
    must agree bit for bit.  The replica of slot 0 also logs its per-scan poses, which are set against the free-running oracle
    (`oracle_map_t`: map translation per scan of the same sequence) up to the first scan beyond 1e-4 m."""
    groups, per = h.stream_groups()
    slots = []
    for g in range(groups):
        lo, hi = g * per, min((g + 1) * per, B) - 1
        for j in range(per_group):
            slots.append(lo + (hi - lo) * j // max(per_group - 1, 1))
    slots = sorted(set(slots))
    st = 7 | binding.REPLAY_BAG
    bad, truncated = [], []
    vs_oracle = None
    for s in slots:
        b, start = slot_source(s, len(bags))
        h1 = binding.Handle(p, device=device, n_slots=1, ring_len=1)
        h1.replay_create(1, LAP)
        for k, a in enumerate(bags[b]):
            h1.replay_load(0, k, a)
        h1.replay_assign(0, 0, start)
        if s == 0 and oracle_map_t is not None:
            h1.trajectory_enable(total_steps)
        h1.batch_run(0, total_steps, st)
        try:
            f1, o1, m1 = h1.batch_get_pose(0)
            fb, ob, mb = h.batch_get_pose(s)
        except binding.AlegoError:
            truncated.append(s); h1.close(); continue
        same = all(np.array_equal(bits(ob[k]), bits(o1[k])) for k in ("t", "q", "params")) and \
            all(np.array_equal(bits(mb[k]), bits(m1[k])) for k in ("t", "q", "params")) and fb == f1
        for name in CHECK_ARRAYS:
            a, c = h.debug_get(name, slot=s), h1.debug_get(name)
            same = same and a.shape == c.shape and np.array_equal(bits(a), bits(c))
        if not same:
            bad.append(s)
        if s == 0 and oracle_map_t is not None:
            tr = h1.trajectory(0)
            n = min(len(oracle_map_t), tr.shape[0])
            err = np.linalg.norm(tr[:n, 7:10] - np.asarray(oracle_map_t)[:n], axis=1)
            beyond = np.nonzero(err > 1e-4)[0]
            hz = int(beyond[0]) if beyond.size else None
            vs_oracle = dict(scans_compared=n, first_scan_beyond_tolerance=hz, max_err_m_before=float(err[:hz].max()) if (hz is None or hz > 0) else 0.0,
                             note="replica of slot 0 (bit-equal to the timed slot 0) free-running vs the oracle on the same scans")
        h1.close()
    return dict(slots=len(slots), slot_ids=slots, stream_groups=groups, slots_per_group=per, steps_replayed=total_steps, arrays=list(CHECK_ARRAYS) + ["odom pose", "LM params_ / pose", "flags"],
                bit_equal=not bad and not truncated, differing_slots=bad, capacity_errors=truncated, stream0_vs_oracle=vs_oracle)


class env_override:
    """set an environment variable for the duration of a `with` block and put back what was there (the one-group handles of the
    checks must not drop a user-supplied ALEGO_STREAM_GROUPS for everything created afterwards)"""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = os.environ.get(self.name)
        os.environ[self.name] = self.value

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.old


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else (a.view(np.uint64) if a.dtype == np.float64 else a)


def quat_angle(q1, q2):
    return 2.0 * float(np.arccos(min(1.0, abs(float(np.dot(q1, q2))))))


def cpu_legs(p, bags, prime, seconds=10.0, device=None, n_streams=4096):
    """BASELINE.md §2 on the host cores of this box, on the scan sequences the GPU streams replay (slot 0 = bag 0 from scan 0):
    cpu_seq (1 thread, timed after `prime` scans), cpu_pipe3 (IP || LO || LM threads), cpu_replicas (one oracle per core on the
    sequences of slots 0..C-1).  While cpu_seq primes, one-stream device handles process the same scans: SURVEY.md 8(d)'s pose
    error and exact-match figures (every scan compared)."""
    from oracle import oracle_py
    t_all = time.perf_counter()

    def seq_of(slot, i):
        b, start = slot_source(slot, len(bags))
        return bags[b][(start + i) % LAP]

    o = oracle_py.Oracle(p)
    hf = binding.Handle(p, device=device) if device is not None else None   # teacher-forced: the 1e-4 contract of the parity tests
    hg = binding.Handle(p, device=device) if device is not None else None   # free-running: reported, not a tolerance claim
    et, er, gt, exact, checked, horizon = [], [], [], 0, 0, None
    oracle_map_t = []   # the (free-running) oracle's map translation per scan of stream 0's sequence: timed_handle_check compares the device against it
    for k in range(prime):
        pts = seq_of(0, k)
        if hf is not None:
            hf.set_lo_params(o.get("lo_params")); hf.set_lm_params(o.get("lm_params"))
        o.process_scan(pts)
        oracle_map_t.append(o.get("map_pose")[:3].copy())
        if hf is not None:
            _, _, mp = hf.scan_process(pts, stages=7)
            _, _, mg = hg.scan_process(pts, stages=7)
            want = o.get("map_pose")
            et.append(float(np.linalg.norm(mp["t"] - want[:3]))); er.append(quat_angle(mp["q"], want[3:]))
            gt.append(float(np.linalg.norm(mg["t"] - want[:3])))
            if horizon is None and k > 0 and (gt[-1] > 1e-4 or quat_angle(mg["q"], want[3:]) > 1e-4):
                horizon = k
            if k > 0:
                checked += 1
                m = o.get("seg_cloud").shape[0]
                same = all(np.array_equal(hf.debug_get(g), o.get(g)) for g in ("seg_col", "seg_ground", "sharp_idx", "less_sharp_idx", "flat_idx"))
                same = same and np.array_equal(hf.debug_get("point_label")[5:m - 5], o.get("point_label")[5:m - 5])
                same = same and np.array_equal(hf.debug_get("seg_cloud").view(np.uint32), o.get("seg_cloud").view(np.uint32))
                same = same and np.array_equal(hf.debug_get("less_flat").view(np.uint32), o.get("less_flat").view(np.uint32))
                same = same and np.array_equal(hf.debug_get("lm_surf_map_ds").view(np.uint32), o.get("lm_surf_map_ds").view(np.uint32))
                exact += bool(same)
    parity = None
    if hf is not None:
        hf.close(); hg.close()
        et, er, gt = np.array(et[1:]), np.array(er[1:]), np.array(gt[1:])
        parity = dict(scans=prime, mode="device vs oracle on identical scans, every scan started from the oracle's LO/LM params_ (teacher forcing)",
                      trans_rmse_m=float(np.sqrt(np.mean(et ** 2))), trans_max_m=float(et.max()),
                      rot_rmse_rad=float(np.sqrt(np.mean(er ** 2))), rot_max_rad=float(er.max()),
                      tolerance="1e-4 m / 1e-4 rad (north_star)", index_outputs_bit_exact=f"{exact}/{checked} scans",
                      free_running=dict(first_scan_beyond_tolerance=horizon, trans_max_m=float(gt.max()), trans_final_m=float(gt[-1]),
                                        note="no teacher forcing; profiles/r02_free_run_flip.json names the decision that flips"))
    # ---- cpu_seq
    n, t0 = 0, time.perf_counter()
    stage = np.zeros(3)
    lo_solve_ms = 0.0
    while time.perf_counter() - t0 < seconds:
        o.process_scan(seq_of(0, prime + n))
        tm = o.get("timing_ms")
        stage += tm[:3]
        lo_solve_ms += float(tm[5])   # lo.t_solve_ms: the two ceres::Solve calls of LaserOdometry (README.md:50,54 time exactly these)
        oracle_map_t.append(o.get("map_pose")[:3].copy())
        n += 1
    dt = time.perf_counter() - t0
    info = o.get("lm_info")
    seq_rate = n / dt
    ms = dict(ip=stage[0] / n, lo=stage[1] / n, lm=stage[2] / n)
    # ---- cpu_pipe3: same sequence, three threads
    o3 = oracle_py.Oracle(p)
    o3.run_pipelined([seq_of(0, k) for k in range(prime)])
    n3 = max(50, int(seconds * seq_rate * 1.2))
    t3 = o3.run_pipelined([seq_of(0, prime + k) for k in range(n3)])
    # ---- cpu_replicas: one oracle per host core on the sequences of slots 0..C-1
    C = max(1, min(os.cpu_count() or 1, n_streams))   # min(n_streams, host_cores)
    reps = [oracle_py.Oracle(p) for _ in range(C)]

    def prime_rep(c):
        for k in range(prime):
            reps[c].process_scan(seq_of(c, k))

    def timed_rep(c):
        m_, t_ = 0, time.perf_counter()
        while time.perf_counter() - t_ < seconds:
            reps[c].process_scan(seq_of(c, prime + m_))
            m_ += 1
        return m_, time.perf_counter() - t_

    with ThreadPoolExecutor(max_workers=C) as ex:
        list(ex.map(prime_rep, range(C)))
        tr0 = time.perf_counter()
        done = list(ex.map(timed_rep, range(C)))
        tr = time.perf_counter() - tr0
    return dict(value=seq_rate, unit="scans/s", cores=1, kind="port",
                sample=f"oracle IP->LO->LM on the scan sequence of GPU stream 0 (bag 0 from scan 0, cyclic): steps {prime}..{prime + n - 1} after "
                       f"priming {prime} scans ({int(info[11])} key frames); {dt:.1f} s timed; all CPU legs {time.perf_counter() - t_all:.0f} s",
                ms_per_scan=ms,
                cpu_pipe3=dict(value=n3 / t3, unit="scans/s", cores=3, sample=f"IP || LO || LM threads (launch/test.launch:7-10), steps {prime}..{prime + n3 - 1} of the same sequence, {t3:.1f} s"),
                cpu_replicas=dict(value=sum(m_ for m_, _ in done) / tr, unit="scans/s", cores=C,
                                  sample=f"{C} independent oracles on the sequences of GPU streams 0..{C - 1}, each primed {prime} scans, {tr:.1f} s"),
                lo_opt_ms_per_frame=round(lo_solve_ms / max(n, 1), 4),
                host_cores=os.cpu_count()), parity, oracle_map_t


def rank_self_check(p, h, args, rank, world, dist, device, steps_done, stages):
    """Every rank's slot 0 against a ONE-slot handle on rank 0 that replays the same (bag, start scan) for the same number of steps:
    odometry + map pose bit for bit (what tests/test_multi_gpu.py asserts).  Streams never interact and the result of a stream does not
    depend on which GPU, which slot or how many neighbours it ran with, so any difference is an error of the sharding, not noise."""
    import torch
    _, odom, mp = h.batch_get_pose(0)
    mine = np.concatenate([odom["t"], odom["q"], mp["t"], mp["q"]]).astype(np.float64)
    if dist is not None:
        t = torch.tensor(mine, dtype=torch.float64, device="cuda")
        allp = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allp, t)
        allp = [a.cpu().numpy() for a in allp]
    else:
        allp = [mine]
    res = None
    if rank == 0:
        bad, worst = [], 0.0
        for r in range(world):
            bag = make_bags(p, 1, first_stream=r * args.bags, flags=args.scan_flags)   # bag 0 of rank r = the bag of its slot 0 (slot_source(0, .) = (0, 0))
            with env_override("ALEGO_STREAM_GROUPS", "1"):
                h1 = binding.Handle(p, device=device, n_slots=1, ring_len=1)
            setup_replay(h1, bag, 1)
            h1.batch_run(0, steps_done, stages)
            _, o1, m1 = h1.batch_get_pose(0)
            h1.close()
            ref = np.concatenate([o1["t"], o1["q"], m1["t"], m1["q"]]).astype(np.float64)
            if not np.array_equal(ref.view(np.uint64), allp[r].view(np.uint64)):
                bad.append(r)
                worst = max(worst, float(np.abs(ref - allp[r]).max()))
        res = dict(ranks_checked=world, steps=steps_done, bit_equal=not bad, ranks_differing=bad, max_abs_diff=worst,
                   what="slot 0 of every rank vs a one-slot handle on rank 0 replaying the same bag: odometry and map pose (t, q), 14 doubles")
        if bad:
            print(f"bench.py: SELF-CHECK FAILED: ranks {bad} differ from rank 0's replay of their stream (max |diff| {worst:g})", file=sys.stderr)
    return res


def roofline_rows(kern, iso, counts, p, per, groups, rebuilds, with_traffic=True):
    """one roofline row per kernel of a HIP-event profile pass (`kern`: kernel_table of the pass; `iso`: the same launch shape alone on the chip, or None):
    algorithmic bytes per launch (kernel_bytes x the `per` streams a launch advances) / the kernel's average launch duration, against the HBM peak"""
    roofs = []
    for name in kern:  # kernels in the order of their share of the device time
        rb = rebuilds / max(kern[name]["launches"], 1) / per  # map rebuilds per launch and stream
        kb = kernel_bytes(name, counts, p.n_scan, p.horizon_scan, rb, p.recent_keyframe_num)
        base = name.strip("()").split("<")[0]
        r = dict(bound="hbm", kernel=name, achieved=None, peak=HBM_PEAK / 1e9, unit="GB/s", frac=None, traffic=pmc_traffic(name, per) if with_traffic else None,   # (the committed --pmc passes belong to the headline workload)
                 algorithmic_bytes_per_launch=None, in_B_scan=base not in BOUNDARY_ONLY, streams_per_launch=per,
                 concurrent_stream_groups=groups, avg_launch_us=kern[name]["avg_us"], share_of_device_time=kern[name]["share"])
        if kb:
            ach = kb * per / (kern[name]["avg_us"] * 1e-6)
            r.update(achieved=round(ach / 1e9, 3), frac=round(ach / HBM_PEAK, 6), algorithmic_bytes_per_launch=int(kb * per))
        else:
            r.update(note="no byte term: the kernel only moves intermediates of its stage (candidate lists, bookkeeping)")
        if iso is not None and name in iso:
            r.update(isolated_launch_us=iso[name]["avg_us"])
            if kb:
                r.update(achieved_isolated=round(kb * per / (iso[name]["avg_us"] * 1e-6) / 1e9, 3), frac_isolated=round(kb * per / (iso[name]["avg_us"] * 1e-6) / HBM_PEAK, 6))
        roofs.append(r)
    return roofs


def dominant_roofline(roofs, kern, rebuilds):
    with_term = [r for r in roofs if r["frac"] is not None and r["in_B_scan"]]
    if not with_term:
        return None
    return dict(with_term[0], map_rebuilds_per_launch=round(rebuilds / max(kern[with_term[0]["kernel"]]["launches"], 1), 2))


def parity_sample(p, device, scans=20, stages=7):
    """`scans` synthetic scans of this configuration through a one-slot device handle and the oracle, every scan started from the oracle's LO / LM params_
    (teacher forcing, the contract of the parity tests): index outputs bit for bit, poses against the 1e-4 tolerance."""
    from oracle import oracle_py
    h, o = binding.Handle(p, device=device), oracle_py.Oracle(p)
    exact, checked, et, er = 0, 0, [0.0], [0.0]
    for k in range(scans):
        pts = synth.scan(p, k)
        h.set_lo_params(o.get("lo_params"))
        if stages & 4:
            h.set_lm_params(o.get("lm_params"))
        o.process_scan(pts, stages)
        _, odom, mp = h.scan_process(pts, stages=stages)
        if k == 0:
            continue
        checked += 1
        m = o.get("seg_cloud").shape[0]
        same = all(np.array_equal(h.debug_get(g), o.get(g)) for g in ("seg_col", "seg_ground", "sharp_idx", "less_sharp_idx", "flat_idx"))
        same = same and np.array_equal(h.debug_get("point_label")[5:m - 5], o.get("point_label")[5:m - 5])
        for g in ("seg_cloud", "less_flat") + (("lm_surf_map_ds",) if stages & 4 else ()):
            same = same and np.array_equal(bits(h.debug_get(g)), bits(o.get(g)))
        exact += bool(same)
        want = o.get("map_pose") if stages & 4 else o.get("odom_pose")
        got = mp if stages & 4 else odom
        et.append(float(np.linalg.norm(got["t"] - want[:3]))); er.append(quat_angle(got["q"], want[3:]))
    h.close()
    return dict(scans=scans, mode="teacher-forced, device vs oracle", index_outputs_bit_exact=f"{exact}/{checked} scans", trans_max_m=max(et), rot_max_rad=max(er),
                within_tolerance=bool(exact == checked and max(et) < 1e-4 and max(er) < 1e-4))


def config_line(p, device, streams, n_bags, prime, warmup, steps, stages=7, bags=None, shard=False, parity_scans=20, extra=None, one_group=False):
    """One more single-GPU configuration of BASELINE.json measured the way the headline is: `streams` resident streams on `n_bags` bags, `prime` untimed
    scans, `warmup` untimed steps, then exactly `steps` timed steps between device synchronisations; a HIP-event pass of another `steps` steps gives the
    per-kernel durations its roofline is computed from; a 20-scan teacher-forced sample against the oracle gives its parity line."""
    import torch
    if bags is None:
        bags = make_bags(p, n_bags, first_stream=0)
    with env_override("ALEGO_STREAM_GROUPS", "1") if (shard or one_group) else contextlib.nullcontext():
        h = binding.Handle(p, device=device, n_slots=streams, ring_len=1)
    if shard:
        D.shard_registration(h, None, 0, 1)   # world 1: the split solver (pack / evaluate / all-reduce / step launches) without partners
    setup_replay(h, bags, streams)
    st = stages | binding.REPLAY_BAG
    h.batch_run(0, prime + warmup, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.batch_run(prime + warmup, steps, st, sync=False)
    h.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step = prime + warmup + steps
    cs = [h.batch_get_counts(s_) for s_ in range(0, streams, max(1, streams // 32))]
    counts = {k: int(round(float(np.mean([c[k] for c in cs])))) for k in cs[0]}
    trunc = 0
    for s_ in range(streams):
        try:
            h.batch_get_pose(s_)
        except binding.AlegoError:
            trunc += 1
    groups, per = h.stream_groups()
    rebuilds0 = sum(h.batch_get_counts(s_)["n_rebuild"] for s_ in range(streams))
    h.profile_enable(True)
    h.batch_run(step, steps, st)
    kern = kernel_table(h.profile_report())
    h.profile_enable(False)
    rebuilds = sum(h.batch_get_counts(s_)["n_rebuild"] for s_ in range(streams)) - rebuilds0
    h.close()
    value = streams * steps / dt
    ab = algorithmic_bytes(counts, p.n_scan)
    b_scan = ab["B_scan"] if stages & 4 else ab["B_IP"] + ab["B_FE"] + ab["B_LO"]
    roofs = roofline_rows(kern, None, counts, p, per, groups, rebuilds, with_traffic=False)
    out = dict(streams=streams, stream_groups=groups, bags=len(bags), primed_scans=prime, warmup=warmup, steps=steps, ms_per_step=round(1e3 * dt / steps, 4), value=round(value, 1), unit="scans/s",
               stages="IP->LO->LM" if stages & 4 else "IP->LO", local_map_keyframes=p.recent_keyframe_num, truncated_streams=trunc,
               pipeline_roofline=dict(B_scan=int(b_scan), achieved_GBps=round(value * b_scan / 1e9, 3), frac_of_hbm_peak=round(value * b_scan / HBM_PEAK, 6)),
               roofline=dominant_roofline(roofs, kern, rebuilds), kernels_top5={k: kern[k] for k in list(kern)[:5]})
    if parity_scans:
        out["parity_sample"] = parity_sample(p, device, parity_scans, stages)
    if extra:
        out.update(extra)
    return out, kern, per


def all_configs(args, p_head, bags_head, device, head_kern, head_per, single, cpu):
    """BASELINE.json's other single-GPU configurations, each as a line of its own: streams, steps, ms_per_step, value = streams x steps / time,
    a 20-scan teacher-forced parity sample, its own roofline.  The headline (config 3 / 4 at `--streams` resident streams) is the top-level line; config 1 is the CPU leg
    (`cpu_baseline`); configs 4 and 5 at N > 1 are the driver's multi-GPU runs."""
    cfg = {}
    t_all = time.perf_counter()
    # ---- config 2: ONE 16x1800 scan through IP -> LaserOdometry two-step; the README's budget (surf 5, corner 10 iterations, README.md:54) and HEAD's 5 / 5
    # (laserOdometry.cpp:415,489).  "ms per frame" three ways: one stream alone (every kernel of a scan behind the previous one), amortised over a resident batch,
    # and the two lo_solve launches alone (what the README's "optimisation" time covers); the oracle's two solves timed on this host beside them.
    from oracle import oracle_py
    c2 = {}
    for tag, (i_s, i_c) in (("budget_5_10", (5, 10)), ("budget_5_5", (5, 5))):
        p = p_head.copy()
        p.lo_iters_surf, p.lo_iters_corner = i_s, i_c
        line, kern, per = config_line(p, device, args.streams, 0, 64, 10, 40, stages=3, bags=bags_head)
        h1 = binding.Handle(p, device=device, n_slots=1, ring_len=1)
        setup_replay(h1, [bags_head[0]], 1)
        h1.batch_run(0, 64, 3 | binding.REPLAY_BAG)
        t1 = time.perf_counter()
        h1.batch_run(64, 400, 3 | binding.REPLAY_BAG)
        one = (time.perf_counter() - t1) / 400
        h1.close()
        o = oracle_py.Oracle(p)
        n_o, t_o, solve_ms = 0, time.perf_counter(), 0.0
        for k in range(150):
            o.process_scan(bags_head[0][k], 3)
            if k >= 10:
                solve_ms += float(o.get("timing_ms")[5]); n_o += 1
        cpu_frame_ms = 1e3 * (time.perf_counter() - t_o) / 150
        los = [k for k in kern if "lo_solve" in k]
        line.update(lo_iterations=dict(surf=i_s, corner=i_c),
                    device_ms_per_frame=dict(one_stream_alone=round(1e3 * one, 4), amortised_in_batch=round(line["ms_per_step"] / line["streams"], 6),
                                             lo_solve_two_launches_amortised=(round(2 * kern[los[0]]["avg_us"] / 1e3 / per, 6) if los else None),
                                             lo_solve_launch_us_under_load=(kern[los[0]]["avg_us"] if los else None)),
                    cpu_ms_per_frame=dict(oracle_ip_fe_lo=round(cpu_frame_ms, 4), oracle_two_solves=round(solve_ms / max(n_o, 1), 4), cores=1),
                    reference_readme_ms_per_frame=(2.13 if (i_s, i_c) == (5, 10) else None),
                    reference_readme_note="README.md:54, robo_0529.bag, Ceres, hardware not stated: context, not a baseline")
        c2[tag] = line
    cfg["config2_ip_lo_two_step_16x1800"] = c2
    # ---- config 3 as written: ONE bag stream at unbounded rate (measured by single_stream() above: lifted into the block)
    if single:
        steps3 = max(args.steps, 200)
        cfg["config3_one_bag_stream_16x1800"] = dict(streams=1, steps=steps3, primed_scans=args.prime, value=single["single_stream_scans_per_s"], unit="scans/s",
                                                     ms_per_step=round(1e3 / single["single_stream_scans_per_s"], 4), serial_replay_scans_per_s=single["single_stream_serial_scans_per_s"],
                                                     local_map_keyframes=p_head.recent_keyframe_num, note=single["single_stream_note"],
                                                     parity_sample="the headline's `parity` block: the same one-stream scan sequence, every scan compared",
                                                     roofline=dict(bound="latency", note="one stream = two sequential kernel chains (LaserOdometry ~150 us, LaserMapping ~150 us per scan); no kernel of it is "
                                                                                         "bandwidth-bound: B_scan x value = %.4f of the HBM peak" % ((single["single_stream_scans_per_s"] * 2.59e6) / HBM_PEAK)))
    # ---- the reference's own compiled geometry, 16 x 4000 (utility.h:50-55)
    p = synth.default_params(16, 4000)
    p.kf_cap_surf, p.kf_cap_outlier = 16384, 4096   # (the lap's key frames hold <= 9 k surf points; worst case = 64 000: 768 streams where 2 304 fit — 258 / 277 / 287 k scans/s at 768 / 1 536 / 2 304 in one call)
    cfg["geometry_16x4000_reference_compiled"], _, _ = config_line(p, device, 2304, 8, LAP, 10, 60)
    # ---- config 5's shape at N = 1: 64 x 2048 with a 200-key-frame local map; fused solver, then the registration split as for N ranks (world 1)
    p = synth.default_params(64, 2048)
    p.recent_keyframe_num = 200
    p.kf_cap_surf, p.kf_cap_outlier = 8192, 2048
    bags5 = make_bags(p, 4, first_stream=0)
    S5 = 768   # (512 / 768 streams: 122.5 / 127.4 k scans/s in one call; 1 024 do not fit)
    fused, _, _ = config_line(p, device, S5, 4, 2400, 10, 40, bags=bags5)
    cfg["config5_shape_64x2048_k200_fused"] = fused
    fused1, _, _ = config_line(p, device, S5, 4, 2400, 10, 40, bags=bags5, one_group=True, parity_scans=0)
    shard, _, _ = config_line(p, device, S5, 4, 2400, 10, 40, bags=bags5, shard=True, parity_scans=0)
    frames = max(40 // max(p.lm_every, 1), 1)
    shard["shard_vs_fused"] = dict(fused_scans_per_s=fused1["value"], sharded_scans_per_s=shard["value"], fused_ms_per_step=fused1["ms_per_step"], sharded_ms_per_step=shard["ms_per_step"],
                                   extra_ms_per_mapping_frame=round((shard["ms_per_step"] - fused1["ms_per_step"]) * 40 / frames, 4), world=1,
                                   note="both on ONE stream group (the collectives of a communicator must not overlap); the fused line above runs 4 groups: %.1f scans/s" % fused["value"])
    cfg["config5_shape_64x2048_k200_sharded_world1"] = shard
    cfg["_seconds"] = round(time.perf_counter() - t_all, 1)
    return cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=4096, help="independent streams resident per GPU (round 6: 4096 — 2048 / 3072 / 4096 measured 467 / 475 / 480 k scans/s in one call; "
                                                               "they fit because --kf-cap sizes the key-frame rings to the data instead of the worst case)")
    ap.add_argument("--bags", type=int, default=8, help="recorded 560-scan streams resident per GPU (shared by the streams)")
    ap.add_argument("--prime", type=int, default=LAP, help="untimed scans per stream to fill the local map (one lap fills 50 key frames)")
    ap.add_argument("--geometry", default="16x1800", help="n_scan x horizon_scan: 16x1800 (BASELINE metric), 16x4000, 64x2048")
    ap.add_argument("--keyframes", type=int, default=0, help="local-map window (0 = reference default 50; config 5 uses 200)")
    ap.add_argument("--kf-cap", type=int, default=-1, help="points per key-frame surf cloud (alego_params.kf_cap_surf; outliers a quarter of it); 0 = worst case (n_scan x horizon_scan: "
                                                            "90 MB per stream at 16x1800 / K = 50); default: 8192 at 16x1800 (the lap's key frames hold <= 4.4 k: 16 MB per stream), "
                                                            "worst case elsewhere.  A key frame that does not fit is truncated and REPORTED (truncated_streams, ALEGO_ERR_CAPACITY)")
    ap.add_argument("--sort-mode", type=int, default=0, help="2 = feature picks in libstdc++ std::sort tie order (alego_params.sort_mode)")
    ap.add_argument("--shard-registration", action="store_true",
                    help="BASELINE config 5: every rank replays the SAME streams and each scan-to-map registration is split over the ranks "
                         "(alego_dist_init: query slices + one ncclAllReduce of the normal equations per solver evaluation); strong scaling")
    ap.add_argument("--scan-flags", type=int, default=0, help="synthetic scan generator flags (synth.cpp): 1 azimuth jitter +-0.45 column, 2 NaN returns, 4 azimuth uniform over the column")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip timed_handle_check (slots of the timed handle against one-slot replicas)")
    ap.add_argument("--no-isolated", action="store_true", help="skip the one-group pass that gives the kernels' isolated durations")
    ap.add_argument("--self-check", action="store_true", help="after the timed region rank 0 replays, on its own GPU, the stream every rank's slot 0 ran and compares the poses "
                                                              "bit for bit; always on at N > 1 (--no-self-check turns it off)")
    ap.add_argument("--no-self-check", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (BASELINE.json's other single-GPU configurations, each timed as a line of its own)")
    args = ap.parse_args()

    rank, local, world = D.env()
    use_dist = world > 1 or bool(os.environ.get("ALEGO_BENCH_FORCE_DIST") and "RANK" in os.environ)   # (second form: the launcher path at N = 1, for testing)
    if use_dist:
        # An RCCL communicator brings HIP streams of its own.  With the runtime's default of 4 hardware queues they alias with the
        # handle's 4 stream groups, two of which then share a queue and serialise: 293 k instead of 329 k scans/s with the communicator
        # merely alive (measured at N = 1 under the launcher; 8 queues: 329.7 k).  Must be set before the HIP runtime initialises.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local)
    dist = None
    if use_dist:
        dist = D.init("nccl", torch.device("cuda", local))  # RCCL: only the barrier + max-over-ranks use it

    ns, hs = (int(v) for v in args.geometry.lower().split("x"))
    if args.kf_cap < 0:
        args.kf_cap = 8192 if (ns, hs) == (16, 1800) and args.keyframes == 0 else 0
    set_pmc_file(args.geometry, args.keyframes)
    p = synth.default_params(ns, hs)
    if args.keyframes > 0:
        p.recent_keyframe_num = args.keyframes
    p.sort_mode = args.sort_mode
    if args.kf_cap > 0:
        p.kf_cap_surf, p.kf_cap_outlier = args.kf_cap, max(256, args.kf_cap // 4)
    B = args.streams
    shard = args.shard_registration
    bags = make_bags(p, args.bags, first_stream=0 if shard else rank * args.bags, flags=args.scan_flags)   # config 5: every rank holds the same streams
    if shard:
        os.environ["ALEGO_STREAM_GROUPS"] = "1"   # the collectives of one communicator must not overlap: one stream group
    h = binding.Handle(p, device=local, n_slots=B, ring_len=1)
    if shard:
        D.shard_registration(h, dist, rank, world)
    setup_replay(h, bags, B)
    stages = 7 | binding.REPLAY_BAG
    step = 0
    h.batch_run(step, args.prime, stages); step += args.prime          # state priming (untimed, like loading a map)
    h.batch_run(step, args.warmup, stages); step += args.warmup        # W warmup steps

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    h.batch_run(step, args.steps, stages, sync=False)                   # exactly K timed steps
    h.synchronize()
    fence()
    dt = time.perf_counter() - t0
    step += args.steps
    dt_local = dt
    dt = D.max_over_ranks(dt, dist, device="cuda")
    sample = range(0, B, max(1, B // 64))
    cs = [h.batch_get_counts(s) for s in sample]
    counts = {k: int(round(float(np.mean([c[k] for c in cs])))) for k in cs[0]}      # mean over a sample of streams (they are at different places)
    flags, odom, mp = h.batch_get_pose(0)
    self_check = None
    if (args.self_check or world > 1) and not args.no_self_check and not shard:
        self_check = rank_self_check(p, h, args, rank, world, dist, local, step, stages)
    allreduce_us = None
    if shard:
        allreduce_us = D.gather_floats(h.dist_allreduce_probe(200), dist, device="cuda")   # (a collective: every rank takes part)
    fused_ref = None
    if shard and rank == 0 and not args.no_isolated:
        # the same streams on an UNSHARDED handle (the fused on-chip solver, one launch per mapping frame) on rank 0's GPU alone: what the split is up
        # against.  At these sizes the ~42 evaluation launches + all-reduces per mapping frame cost more than the rows they spread,
        # so a sharded value BELOW this one — negative strong scaling — is the expected reading, not a defect of the run.
        with env_override("ALEGO_STREAM_GROUPS", "1"):
            hf = binding.Handle(p, device=local, n_slots=B, ring_len=1)
        setup_replay(hf, bags, B)
        hf.batch_run(0, args.prime + args.warmup, stages)
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        hf.batch_run(args.prime + args.warmup, args.steps, stages, sync=False)
        hf.synchronize()
        tf = time.perf_counter() - tf0
        hf.close()
        frames = max(args.steps // max(p.lm_every, 1), 1)
        fused_ref = dict(fused_scans_per_s=round(B * args.steps / tf, 1), sharded_scans_per_s=round(B * args.steps / dt, 1),
                         fused_ms_per_step=round(1e3 * tf / args.steps, 4), sharded_ms_per_step=round(1e3 * dt / args.steps, 4),
                         extra_ms_per_mapping_frame=round(1e3 * (dt - tf) / frames, 4), world=world,
                         expected="sharded <= fused at this size: ~42 evaluation launches + 32-double all-reduces per mapping frame against one fused launch (DESIGN.md section 6)")
    rebuilds0 = sum(h.batch_get_counts(s)["n_rebuild"] for s in range(B))

    roof, kern, rebuilds, per, groups = None, None, 0, B, 1
    if rank == 0 and not args.no_profile:
        # per-kernel durations with HIP events on the handle's streams, over another K steps.  One launch covers one
        # stream group (`per` streams); the groups run concurrently, exactly as in the timed region.
        groups, per = h.stream_groups()
        h.profile_enable(True)
        h.batch_run(step, args.steps, stages); step += args.steps
        rep = h.profile_report()
        h.profile_enable(False)
        kern = kernel_table(rep)
        rebuilds = sum(h.batch_get_counts(s)["n_rebuild"] for s in range(B)) - rebuilds0
    value = D.aggregate_scans_per_s(1 if shard else world, B, args.steps, dt)   # config 5: the ranks share the streams
    out = {
        "metric": f"scans/sec ({ns}x{hs} LiDAR) full IP->LO->LM loop", "value": round(value, 1), "unit": "scans/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "f32/f64", "data": "synthetic",
        "config": {"workload": f"{ns}x{hs} S0/T0 bag-equivalent replay (one 560-scan lap per bag, replayed cyclically), IP->LO->LM with a "
                               f"{p.recent_keyframe_num}-key-frame local map, {B} independent streams per GPU on {args.bags} resident bags "
                               f"(one scan per stream per step)",
                   "streams_per_gpu": B, "bags_per_gpu": args.bags, "bag_scans": LAP, "primed_scans": args.prime, "parallelism": (f"registration sharded x{world} (RCCL all-reduce of the normal equations)" if shard else f"streams x{world}"),
                   "scan_flags": args.scan_flags},
    }
    # every rank reports its own rate; rank 0 prints them (the driver computes scaling from `value`, this is for the reader)
    per_rank = D.gather_floats(B * args.steps / dt_local, dist, device="cuda")
    if rank == 0:
        out["per_rank_scans_per_s"] = [round(v, 1) for v in per_rank]
        out["ranks_seen_by_rccl"] = len(per_rank) if dist is not None else 1
        # a key frame that does not fit kf_cap_surf / kf_cap_outlier is truncated and reported by the slot's next pose query:
        # every slot is asked, not just slot 0 (a truncated stream is cheaper to process)
        trunc = 0
        for s in range(B):
            try:
                h.batch_get_pose(s)
            except binding.AlegoError:
                trunc += 1
        out["truncated_streams"] = trunc
        if self_check is not None:
            out["self_check"] = self_check
        if allreduce_us is not None:
            # config 5's only data-path collective: one in-place ncclAllReduce(sum, 32 f64) per solver evaluation (~42 per mapping frame)
            if fused_ref is not None:
                out["shard_vs_fused"] = fused_ref
            out["shard_allreduce"] = dict(usec_per_allreduce_by_rank=[round(v, 2) for v in allreduce_us], doubles=32, per_mapping_frame=42,
                                          note="200 back-to-back all-reduces on the registration's stream, enqueue + completion included")
        ab = algorithmic_bytes(counts, p.n_scan)
        # the device rebuilds a local map only when the stream's key-frame set changed; SURVEY 8(d)'s B_LM charges the map terms on
        # every mapping frame (the reference re-concatenates every time).  Both figures are reported.
        launches_lm = max(kern["lm_solve"]["launches"], 1) if kern and "lm_solve" in kern else 0
        rb_frac = (rebuilds / (launches_lm * per)) if launches_lm else None      # rebuilds per mapping frame and stream
        kraw, kds = counts["Kraw_c"] + counts["Kraw_s"], counts["Kds_c"] + counts["Kds_s"]
        pr = dict(B_scan=int(ab["B_scan"]), achieved_GBps=round(value / world * ab["B_scan"] / 1e9, 3),
                  frac_of_hbm_peak=round(value / world * ab["B_scan"] / HBM_PEAK, 6), **{k: int(v) for k, v in ab.items() if k != "B_scan"})
        if rb_frac is not None:
            b_lm_dev = (16 * kraw + 16 * kds) * rb_frac + 16 * kds + 16 * (counts["Lc"] + counts["Ls"]) + 104
            b_dev = ab["B_IP"] + ab["B_FE"] + ab["B_LO"] + b_lm_dev / 2
            pr.update(map_rebuild_fraction=round(rb_frac, 4), B_scan_device=int(b_dev), achieved_GBps_device=round(value / world * b_dev / 1e9, 3),
                      frac_of_hbm_peak_device=round(value / world * b_dev / HBM_PEAK, 6))
        pm = pmc_bytes_per_scan()
        if pm is not None:
            pr.update(measured_hbm_bytes_per_scan=int(pm["corrected"]), measured_hbm_bytes_per_scan_uncorrected=int(pm["uncorrected"]),
                      measured_over_algorithmic=round(pm["corrected"] / ab["B_scan"], 3), measured_source=os.path.relpath(PMC_FILE, ROOT))
        out["pipeline_roofline"] = pr
        out["counts"] = counts
        if kern is not None:
            out["kernels"] = kern
        omt = None
        if world == 1 and not args.no_cpu:
            out.update(single_stream(p, bags[0], local, args.prime, max(args.steps, 200)))
            out["cpu_baseline"], out["parity"], omt = cpu_legs(p, bags, args.prime, device=local, n_streams=B)
        if not args.no_check and not shard:
            out["timed_handle_check"] = timed_handle_check(h, p, bags, B, step, local, omt)
    h.close()
    if rank == 0 and world == 1 and not shard and not args.no_configs and not args.no_cpu and args.geometry.lower() == "16x1800" and args.keyframes == 0:
        single = {k: out[k] for k in ("single_stream_scans_per_s", "single_stream_serial_scans_per_s", "single_stream_note") if k in out}
        out["configs"] = all_configs(args, p, bags, local, kern, per, single if len(single) == 3 else None, out.get("cpu_baseline"))
    if rank == 0 and kern is not None:
        # the same launch shape (`per` streams per launch) ALONE on the chip: one stream group, nothing else resident
        iso = None
        if not args.no_isolated and groups > 1:
            with env_override("ALEGO_STREAM_GROUPS", "1"):
                hi = binding.Handle(p, device=local, n_slots=per, ring_len=1)
            setup_replay(hi, bags, per)
            hi.batch_run(0, args.prime + args.warmup, stages)
            hi.profile_enable(True)
            hi.batch_run(args.prime + args.warmup, args.steps, stages)
            iso = kernel_table(hi.profile_report())
            hi.close()
            out["kernels_isolated"] = {k: v["avg_us"] for k, v in iso.items()}   # one stream group alone on the chip: every launch by itself
        roofs = roofline_rows(kern, iso, counts, p, per, groups, rebuilds)
        dom = dominant_roofline(roofs, kern, rebuilds)
        if dom:   # the dominant kernel: largest share of the device time among those with a §8(d) term of their own
            out["roofline"] = dom
        out["roofline_top3"] = roofs[:3]        # the three largest kernels by device time, whatever their term
        out["roofline_all"] = {r["kernel"]: dict(frac=r["frac"], frac_isolated=r.get("frac_isolated"), traffic_over_algorithmic=(round(r["traffic"] / r["algorithmic_bytes_per_launch"], 2) if r["traffic"] and (r["algorithmic_bytes_per_launch"] or 0) > 1024 * per else None),   # (the solvers' 104 bytes of pose are not a traffic yardstick)
                                                 share=r["share_of_device_time"], in_B_scan=r["in_B_scan"]) for r in roofs if r["frac"] is not None}
        # the device's counterpart of cpu_baseline.lo_opt_ms_per_frame (the reference README's "optimisation" time per frame, README.md:50,54):
        # LaserOdometry's two ceres::Solve calls = two lo_solve launches per scan, each advancing `per` streams
        los = [k for k in kern if k.startswith("(lo_solve_t") or k.startswith("lo_solve_t")]
        if los:
            name = los[0]
            lo = {"kernel": name, "launches_per_frame": 2, "streams_per_launch": per,
                  "ms_per_launch_under_load": round(kern[name]["avg_us"] / 1e3, 4),
                  "ms_per_frame_amortised": round(2 * kern[name]["avg_us"] / 1e3 / per, 6)}
            if iso is not None and name in iso:
                lo["ms_per_launch_isolated"] = round(iso[name]["avg_us"] / 1e3, 4)
            out["lo_opt_device"] = lo
    # RCCL prints its version banner through C stdio, which is buffered when stdout is a pipe / file: every rank pushes it out,
    # then rank 0 prints the JSON line after the barrier, so that it is the last line of the job's stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and (out.get("self_check") or {}).get("bit_equal") is False:
        sys.exit(3)   # the line is printed (with self_check.bit_equal = false) and the job fails


if __name__ == "__main__":
    main()
