"""-m "not gpu": host-side pieces of the C ABI that need no device (PointCloud2 parsing) and the oracle's
pose-graph pass-through operations."""
import struct

import numpy as np
import pytest

from alego_amd import binding, synth
from oracle import oracle_py as O
from util import assert_bit_equal

F32, U16 = 7, 4   # sensor_msgs/PointField datatypes


def _pack(pts, step, offs, extra=b"", big=False):
    fmt = ">f" if big else "<f"
    out = bytearray()
    for p in pts:
        rec = bytearray(step)
        for k, o in enumerate(offs):
            if o is not None:
                rec[o:o + 4] = struct.pack(fmt, float(p[k]))
        out += rec
    return bytes(out) + extra


def test_pointcloud2_parser_layouts():
    """alego_pc2_to_points = pcl::fromROSMsg<PointXYZI> (imageProjection.cpp:54-55): fields matched by name, FLOAT32 only."""
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(257, 4)).astype(np.float32)
    # PCL's own 32-byte layout: x y z @0/4/8, intensity @16
    pcl = [("x", 0, F32, 1), ("y", 4, F32, 1), ("z", 8, F32, 1), ("intensity", 16, F32, 1)]
    got = binding.pc2_to_points(_pack(pts, 32, (0, 4, 8, 16)), 257, 1, 32, 257 * 32, pcl)
    assert_bit_equal(got, pts, "PCL layout")
    # a driver layout: intensity first, a uint16 ring field in between, fields listed in another order
    drv = [("ring", 4, U16, 1), ("z", 16, F32, 1), ("intensity", 0, F32, 1), ("x", 8, F32, 1), ("y", 12, F32, 1)]
    got = binding.pc2_to_points(_pack(pts, 22, (8, 12, 16, 0)), 257, 1, 22, 257 * 22, drv)
    assert_bit_equal(got, pts, "driver layout")
    # organised cloud with padded rows (row_step > width * point_step)
    w, hgt, step = 64, 4, 16
    rows = b"".join(_pack(pts[r * w:(r + 1) * w], step, (0, 4, 8, 12)) + b"\xee" * 40 for r in range(hgt))
    got = binding.pc2_to_points(rows, w, hgt, step, w * step + 40, [("x", 0, F32, 1), ("y", 4, F32, 1), ("z", 8, F32, 1), ("intensity", 12, F32, 1)])
    assert_bit_equal(got, pts[:w * hgt], "padded rows")
    # big-endian payload
    got = binding.pc2_to_points(_pack(pts, 16, (0, 4, 8, 12), big=True), 257, 1, 16, 257 * 16,
                                [("x", 0, F32, 1), ("y", 4, F32, 1), ("z", 8, F32, 1), ("intensity", 12, F32, 1)], is_bigendian=True)
    assert_bit_equal(got, pts, "big endian")
    # no intensity field (or one of another type): PCL leaves the member at 0
    want = pts.copy(); want[:, 3] = 0
    got = binding.pc2_to_points(_pack(pts, 12, (0, 4, 8, None)), 257, 1, 12, 257 * 12, [("x", 0, F32, 1), ("y", 4, F32, 1), ("z", 8, F32, 1)])
    assert_bit_equal(got, want, "missing intensity")
    got = binding.pc2_to_points(_pack(pts, 16, (0, 4, 8, 12)), 257, 1, 16, 257 * 16,
                                [("x", 0, F32, 1), ("y", 4, F32, 1), ("z", 8, F32, 1), ("intensity", 12, 8, 1)])  # FLOAT64: no match
    assert_bit_equal(got, want, "intensity of another datatype")
    # empty message
    assert binding.pc2_to_points(b"", 0, 1, 16, 0, pcl[:3]).shape == (0, 4)


def test_pointcloud2_parser_rejects_bad_messages():
    f = [("x", 0, F32, 1), ("y", 4, F32, 1), ("z", 8, F32, 1)]
    data = bytes(16 * 10)
    with pytest.raises(binding.AlegoError):
        binding.pc2_to_points(data, 10, 1, 16, 160, f[:2])                       # no z
    with pytest.raises(binding.AlegoError):
        binding.pc2_to_points(data, 11, 1, 16, 176, f)                           # data shorter than width * point_step
    with pytest.raises(binding.AlegoError):
        binding.pc2_to_points(data, 10, 1, 16, 160, [("x", 0, F32, 1), ("y", 4, F32, 1), ("z", 14, F32, 1)])  # field beyond point_step
    with pytest.raises(binding.AlegoError):
        binding.pc2_to_points(data, 10, 1, 16, 160, f, cap=5)                    # caller buffer too small


def test_pointcloud2_of_a_synthetic_scan_feeds_the_oracle(params_a):
    """A scan serialised the way a driver publishes it and parsed back is the scan: same segmentation."""
    pts = synth.scan(params_a, 1)
    raw = bytearray(32 * len(pts))
    v = np.frombuffer(raw, np.uint8).reshape(-1, 32)
    v[:, 0:12] = pts[:, :3].copy().view(np.uint8).reshape(-1, 12)
    v[:, 16:20] = pts[:, 3:4].copy().view(np.uint8).reshape(-1, 4)
    got = binding.pc2_to_points(bytes(raw), len(pts), 1, 32, 32 * len(pts),
                                [("x", 0, F32, 1), ("y", 4, F32, 1), ("z", 8, F32, 1), ("intensity", 16, F32, 1)])
    assert_bit_equal(got, pts, "round trip")
    a, b = O.Oracle(params_a), O.Oracle(params_a)
    a.ip(pts), b.ip(got)
    assert_bit_equal(a.get("label_img"), b.get("label_img"), "labels")


def test_oracle_keyframe_pass_through(params_a):
    """correctPoses on the oracle: rewriting the key poses + clearing the window re-assembles the map from the corrected poses;
    rewriting them with their own values changes nothing beyond what clearing the window alone does (a full window holding
    the duplicate of laserMapping.cpp:227-236 refills without it)."""
    p = params_a.copy()
    p.recent_keyframe_num, p.min_keyframe_dist = 5, 0.09
    a, b = O.Oracle(p), O.Oracle(p)
    for k in range(30):
        pts = synth.scan(p, k)
        a.process_scan(pts), b.process_scan(pts)
    a.lm_reset_window()
    poses = a.get("lm_keyposes").reshape(-1, 6)
    assert len(poses) >= 6
    c, s, o = a.lm_keyframe(len(poses) - 1)
    assert len(c) and len(s)
    for i in range(len(poses)):
        b.lm_set_keypose(i, poses[i])
    b.lm_reset_window()
    for k in range(30, 34):
        pts = synth.scan(p, k)
        a.process_scan(pts), b.process_scan(pts)
        assert_bit_equal(a.get("lm_surf_map_ds"), b.get("lm_surf_map_ds"), f"scan {k}: identity correction changes nothing")
    poses = a.get("lm_keyposes").reshape(-1, 6)
    shifted = poses.copy(); shifted[:, 0] += np.float32(0.25)
    for i in range(len(poses)):
        b.lm_set_keypose(i, shifted[i])
    b.lm_reset_window()
    a.lm_reset_window()   # same window content on both sides: only the poses differ
    b.lm_apply_correction([1, 0, 0, 0.25, 0, 1, 0, 0, 0, 0, 1, 0])
    pts = synth.scan(p, 34)
    a.process_scan(pts), b.process_scan(pts)
    pts = synth.scan(p, 35)
    a.process_scan(pts), b.process_scan(pts)
    ma, mb = a.get("lm_surf_map"), b.get("lm_surf_map")   # the window's transformed clouds before the VoxelGrid
    assert ma.shape == mb.shape
    assert np.abs(mb[:, 0] - ma[:, 0] - 0.25).max() < 1e-4 and np.abs(mb[:, 1:] - ma[:, 1:]).max() < 1e-4, "the corrected map is the old one shifted"
    assert np.abs(b.get("map_pose")[:3] - a.get("map_pose")[:3] - [0.25, 0, 0]).max() < 0.02


def _loop_frames(p, o, closest):
    poses = o.get("lm_keyposes").reshape(-1, 6)
    n = len(poses)
    frames = [(poses[n - 1],) + tuple(o.lm_keyframe(n - 1))]
    for j in range(closest - p.lc_search_num, closest + p.lc_search_num + 1):
        if 0 <= j < n - 1:                                   # laserMapping.cpp:802: j < 0 || j >= latest_history_frame_id_ skipped
            frames.append((poses[j],) + tuple(o.lm_keyframe(j)))
    return frames


def test_loop_detect_and_oracle_icp(params_a):
    """detectLoopClosure's selection (host code of the library) against the oracle's, and the oracle's ICP restatement on a real
    revisit: three quarters round the T0 lap the sensor sees the start area again; the ICP of the newest key frame against the
    sub-map around the oldest ones converges, and it recovers a pose error put into the newest key pose."""
    p = params_a
    o = O.Oracle(p)
    for k in range(420):
        o.process_scan(synth.scan(p, k))
    poses = o.get("lm_keyposes").reshape(-1, 6)
    n = len(poses)
    stamps = np.arange(n) * 1.05                              # ~1 key frame per second at 10 Hz
    cur = o.get("map_pose")[:3]
    rng = np.random.default_rng(4)
    for trial in range(40):                                    # the library's host helper == the oracle on random queries
        q = cur + rng.normal(size=3) * [8, 8, 0.5]
        st = np.sort(rng.uniform(0, 60, n))
        assert binding.loop_detect(p, poses, st, q) == O.loop_detect(p, poses, st, q)
    closest = O.loop_detect(p, poses, stamps, cur)
    assert closest == binding.loop_detect(p, poses, stamps, cur) and 0 <= closest < 8, closest
    assert O.loop_detect(p, poses, np.arange(n) * 0.5, cur) == -1, "nothing is 30 s old yet"
    frames = _loop_frames(p, o, closest)
    r, tgt = O.loop_icp(p, frames)
    assert r["converged"] == 1 and 2 <= r["iterations"] < p.icp_max_iters and r["fitness"] < p.lc_fitness_max
    assert r["n_target"] == len(tgt) > 1000 and r["n_source"] == sum(len(c) for c in frames[0][1:])
    assert np.abs(r["T"][:3, 3]).max() < 0.1, "an un-drifted trajectory needs (almost) no correction"
    # a wrong newest key pose: the correction composed with it must land where the clean alignment landed
    bad = poses[n - 1].copy()
    bad[0] += 0.4; bad[1] -= 0.25; bad[5] += 0.03
    r2, _ = O.loop_icp(p, [(bad,) + frames[0][1:]] + frames[1:])
    assert r2["converged"] == 1 and r2["fitness"] < p.lc_fitness_max
    pts = frames[0][2][:, :3].astype(np.float64)               # surf points of the newest frame, sensor frame
    def world(pose, T):
        w = O.transform_cloud(pose, frames[0][2])[:, :3].astype(np.float64)
        return w @ T[:3, :3].astype(np.float64).T + T[:3, 3].astype(np.float64)
    assert np.abs(world(bad, r2["T"]) - world(poses[n - 1], r["T"])).max() < 0.05


def test_pose_o2b_standalone_frame_convention():
    """alego_pose_o2b = LO.cpp:588-608: tf_o2b = tf_o2l * tf_b2l^-1, quaternion of its rotation block — against numpy (4 x 4 inverse, matrix product) and
    scipy's rotation-matrix -> quaternion, for random poses and random (also non-rigid: the reference inverts a general Matrix4d) mounts."""
    from scipy.spatial.transform import Rotation as R
    rng = np.random.default_rng(5)
    for trial in range(50):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        t = rng.normal(size=3) * 10
        B = np.eye(4)
        if trial:
            B[:3, :3] = R.from_rotvec(rng.normal(size=3)).as_matrix()
            B[:3, 3] = rng.normal(size=3)
        T = np.eye(4)
        T[:3, :3] = R.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()
        T[:3, 3] = t
        want = T @ np.linalg.inv(B)
        tb, qb = binding.pose_o2b(t, q, B)
        np.testing.assert_allclose(tb, want[:3, 3], rtol=0, atol=1e-12)
        wq = R.from_matrix(want[:3, :3]).as_quat()      # x y z w
        wq = np.array([wq[3], wq[0], wq[1], wq[2]])
        assert min(np.abs(qb - wq).max(), np.abs(qb + wq).max()) < 1e-12
    with pytest.raises(binding.AlegoError):
        binding.pose_o2b([0, 0, 0], [1, 0, 0, 0], np.zeros((4, 4)))
