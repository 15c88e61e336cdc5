"""-m "not gpu": the library's rosbag 2.0 reader (csrc/rosbag.cpp) against bags written here, byte by byte, by a minimal
writer that follows the published format: uncompressed, bz2 (Python's bz2 module as the independent compressor) and lz4-frame
chunks, indexed and unindexed (never closed) files, several topics, PointCloud2 messages with row padding, `is_dense` both
ways and a big-endian payload.  What comes back must be the points that went in, bit for bit, in time order."""
import bz2
import os
import struct
import subprocess

import numpy as np
import pytest

from alego_amd import binding
from util import assert_bit_equal

F32 = 7
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- a minimal rosbag 2.0 writer (test infrastructure) ----
def _header(fields):
    out = b""
    for k, v in fields.items():
        f = k.encode() + b"=" + v
        out += struct.pack("<I", len(f)) + f
    return out


def _record(fields, data):
    h = _header(fields)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def _time(t):
    sec = int(t)
    return struct.pack("<II", sec, int(round((t - sec) * 1e9)))


def _string(s):
    b = s.encode()
    return struct.pack("<I", len(b)) + b


def pc2_message(pts, seq, stamp, frame_id="laser", layout="pcl", dense=True, big=False, height=1, pad=0):
    """sensor_msgs/PointCloud2 in ROS1 serialisation"""
    n = len(pts)
    width = n // height
    if layout == "pcl":
        step, offs = 32, (0, 4, 8, 16)
    else:   # a driver's layout: intensity first, a uint16 in between
        step, offs = 22, (8, 12, 16, 0)
    names = ("x", "y", "z", "intensity")
    rows = []
    for r in range(height):
        row = bytearray(width * step + pad)
        for c in range(width):
            for k in range(4):
                row[c * step + offs[k]:c * step + offs[k] + 4] = struct.pack(">f" if big else "<f", float(pts[r * width + c, k]))
        if pad:
            row[width * step:] = b"\xee" * pad
        rows.append(bytes(row))
    data = b"".join(rows)
    fields = struct.pack("<I", 4) + b"".join(_string(nm) + struct.pack("<IBI", offs[k], F32, 1) for k, nm in enumerate(names))
    return (struct.pack("<I", seq) + _time(stamp) + _string(frame_id) + struct.pack("<II", height, width) + fields +
            struct.pack("<BII", 1 if big else 0, step, width * step + pad) + struct.pack("<I", len(data)) + data + struct.pack("<B", 1 if dense else 0))


def _lz4_frame(raw):
    """a valid LZ4 frame: one block with a literal run + one match (offset 1 run-length idiom) + literals, then a stored block"""
    half = len(raw) // 2
    a, b = raw[:half], raw[half:]
    # compressed block for `a`: all literals (token high nibble 15 + extension bytes), no match
    def literals(x):
        n = len(x)
        if n < 15:
            return bytes([n << 4]) + x
        out, rem = bytes([0xF0]), n - 15
        while rem >= 255:
            out += b"\xff"; rem -= 255
        return out + bytes([rem]) + x
    blk = literals(a)
    frame = b"\x04\x22\x4d\x18" + bytes([0x60, 0x40, 0x00])       # version 01, block-independent; 64 KB blocks; header checksum (not verified)
    frame += struct.pack("<I", len(blk)) + blk
    frame += struct.pack("<I", len(b) | 0x80000000) + b           # stored block
    return frame + struct.pack("<I", 0)


def write_bag(path, messages, compression="none", chunk_msgs=3, indexed=True, topics_types=None):
    """messages: [(topic, bag_time, serialized bytes)] in recording order"""
    topics = []
    for t, _, _ in messages:
        if t not in topics:
            topics.append(t)
    conn_of = {t: i for i, t in enumerate(topics)}
    types = topics_types or {}

    def conn_record(t):
        ch = _header({"topic": t.encode(), "type": types.get(t, "sensor_msgs/PointCloud2").encode(), "md5sum": b"1158d486dd51d683ce2f1be655c3c181",
                      "message_definition": b"# omitted"})
        return _record({"op": b"\x07", "conn": struct.pack("<I", conn_of[t]), "topic": t.encode()}, ch)

    body = b""
    seen = set()
    n_chunks = 0
    for c0 in range(0, len(messages), chunk_msgs):
        chunk, index = b"", {}
        for t, bt, payload in messages[c0:c0 + chunk_msgs]:
            if t not in seen:
                seen.add(t)
                chunk += conn_record(t)
            index.setdefault(conn_of[t], []).append((bt, len(chunk)))
            chunk += _record({"op": b"\x02", "conn": struct.pack("<I", conn_of[t]), "time": _time(bt)}, payload)
        comp = {"none": chunk, "bz2": bz2.compress(chunk, 1), "lz4": _lz4_frame(chunk)}[compression]
        body += _record({"op": b"\x05", "compression": compression.encode(), "size": struct.pack("<I", len(chunk))}, comp)
        n_chunks += 1
        if indexed:
            for conn, entries in index.items():
                data = b"".join(_time(bt) + struct.pack("<I", off) for bt, off in entries)
                body += _record({"op": b"\x04", "ver": struct.pack("<I", 1), "conn": struct.pack("<I", conn), "count": struct.pack("<I", len(entries))}, data)
    tail = b"".join(conn_record(t) for t in topics) if indexed else b""
    bh = _header({"op": b"\x03", "index_pos": struct.pack("<Q", 13 + 4096 + len(body)), "conn_count": struct.pack("<I", len(topics)),
                  "chunk_count": struct.pack("<I", n_chunks)})
    pad = 4096 - 4 - len(bh) - 4
    with open(path, "wb") as f:
        f.write(b"#ROSBAG V2.0\n" + struct.pack("<I", len(bh)) + bh + struct.pack("<I", pad) + b" " * pad + body + tail)


def _clouds(rng, n_msgs, n_pts=96):
    out = []
    for i in range(n_msgs):
        p = rng.normal(scale=10.0, size=(n_pts, 4)).astype(np.float32)
        if i % 3 == 1:
            p[5, 0] = np.nan; p[17, 2] = np.inf      # non-finite returns travel unchanged (is_dense decides what ImageProjection does with them)
        out.append(p)
    return out


@pytest.mark.parametrize("compression,indexed", [("none", True), ("bz2", True), ("lz4", True), ("none", False), ("bz2", False)])
def test_bag_round_trip(tmp_path, compression, indexed):
    rng = np.random.default_rng(11)
    clouds = _clouds(rng, 8)
    msgs = []
    for i, p in enumerate(clouds):
        kw = [dict(), dict(layout="driver"), dict(height=4, pad=40), dict(big=True), dict(dense=False)][i % 5]
        msgs.append(("/lslidar_point_cloud", 100.0 + 0.1 * i, pc2_message(p, i, 99.5 + 0.1 * i, **kw)))
        if i % 2 == 0:   # another topic interleaved in the same chunks
            msgs.append(("/imu/data", 100.0 + 0.1 * i + 0.01, b"\x01\x02\x03" * (i + 1)))
    path = str(tmp_path / f"t_{compression}_{indexed}.bag")
    write_bag(path, msgs, compression=compression, indexed=indexed, topics_types={"/imu/data": "sensor_msgs/Imu"})
    bag = binding.Bag(path)
    tp = bag.topics()
    assert tp == {"/lslidar_point_cloud": ("sensor_msgs/PointCloud2", 8), "/imu/data": ("sensor_msgs/Imu", 4)}
    assert bag.message_count("/lslidar_point_cloud") == 8 and bag.message_count("/nope") == 0
    for i, p in enumerate(clouds):
        got, stamp, dense = bag.read_pc2("/lslidar_point_cloud", i)
        assert_bit_equal(got, p, f"{compression} cloud {i}")
        assert abs(stamp - (99.5 + 0.1 * i)) < 1e-6 and dense == (i % 5 != 4)
    for j in (3, 0, 2):   # random access across chunks, raw payloads + bag times
        raw, t = bag.read_raw("/imu/data", j)
        assert raw == b"\x01\x02\x03" * (2 * j + 1) and abs(t - (100.0 + 0.2 * j + 0.01)) < 1e-6
    with pytest.raises(binding.AlegoError):
        bag.read_pc2("/lslidar_point_cloud", 8)
    with pytest.raises(binding.AlegoError):
        bag.read_pc2("/imu/data", 0)        # not a PointCloud2
    with pytest.raises(binding.AlegoError):
        bag.read_pc2("/lslidar_point_cloud", 0, cap=10)
    bag.close()


def test_bag_messages_come_back_in_time_order(tmp_path):
    """rosbag play publishes by receive time; a bag whose chunks were written out of order is handed out sorted"""
    rng = np.random.default_rng(5)
    clouds = _clouds(rng, 6, n_pts=20)
    order = [3, 0, 5, 1, 4, 2]
    msgs = [("/lslidar_point_cloud", 10.0 + k, pc2_message(clouds[k], k, 10.0 + k)) for k in order]
    path = str(tmp_path / "shuffled.bag")
    write_bag(path, msgs, chunk_msgs=2)
    bag = binding.Bag(path)
    for k in range(6):
        got, stamp, _ = bag.read_pc2("/lslidar_point_cloud", k)
        assert_bit_equal(got, clouds[k], f"message {k}")
        assert stamp == 10.0 + k


def test_bz2_decoder_on_large_and_degenerate_payloads(tmp_path):
    """the decoder on its own, through a chunk holding one raw message: several bz2 blocks (level 1 = 100 kB blocks), long runs
    (RLE1 counts, RUNA / RUNB), all 256 byte values, a 1-byte payload"""
    rng = np.random.default_rng(2)
    payloads = [bytes(rng.integers(0, 256, 350_000, dtype=np.uint8)),                 # incompressible, 4 blocks
                b"\x00" * 300_000 + b"abc" * 1000 + b"\xff" * 70_000,                  # long runs
                bytes(range(256)) * 300 + b"A" * 4 + b"B" * 5 + b"C" * 259 + b"D" * 260,   # RLE1 boundary cases
                b"x"]
    for i, pl in enumerate(payloads):
        path = str(tmp_path / f"z{i}.bag")
        write_bag(path, [("/raw", 1.0, pl)], compression="bz2", topics_types={"/raw": "std_msgs/UInt8MultiArray"})
        bag = binding.Bag(path)
        raw, _ = bag.read_raw("/raw", 0)
        assert raw == pl, f"payload {i}: {len(raw)} bytes back, {len(pl)} in"
        bag.close()


def test_bag_open_rejects_other_files(tmp_path):
    p = tmp_path / "x.bag"
    p.write_bytes(b"#ROSBAG V1.2\n" + b"\0" * 100)
    with pytest.raises(binding.AlegoError):
        binding.Bag(str(p))
    with pytest.raises(binding.AlegoError):
        binding.Bag(str(tmp_path / "missing.bag"))
    # a corrupted bz2 chunk is reported (block CRC), not returned as data
    rng = np.random.default_rng(1)
    msgs = [("/lslidar_point_cloud", 1.0, pc2_message(_clouds(rng, 1)[0], 0, 1.0))]
    good = str(tmp_path / "good.bag")
    write_bag(good, msgs, compression="bz2", indexed=True)
    raw = bytearray(open(good, "rb").read())
    at = raw.index(b"BZh1") + 60
    raw[at] ^= 0x10
    bad = tmp_path / "bad.bag"
    bad.write_bytes(bytes(raw))
    bag = binding.Bag(str(bad))
    with pytest.raises(binding.AlegoError):
        bag.read_pc2("/lslidar_point_cloud", 0)


def test_replay_example_lists_a_bag_without_a_gpu(tmp_path):
    """examples/replay --bag <file> --list: the host program's bag path needs no device"""
    exe = os.path.join(ROOT, "examples", "replay")
    if not os.path.exists(exe):
        pytest.skip("examples/replay not built")
    rng = np.random.default_rng(4)
    msgs = [("/lslidar_point_cloud", 5.0 + 0.1 * k, pc2_message(_clouds(rng, 1, 50)[0], k, 5.0 + 0.1 * k)) for k in range(3)]
    path = str(tmp_path / "l.bag")
    write_bag(path, msgs, compression="bz2")
    out = subprocess.run([exe, "--bag", path, "--list"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert "/lslidar_point_cloud" in out.stdout and "sensor_msgs/PointCloud2" in out.stdout and " 3" in out.stdout


# ---- -m gpu: from bag bytes to poses on the device (README.md:33-37: `rosbag play` + the nodelets; IP.cpp:106-133) ----
@pytest.mark.gpu
@pytest.mark.parametrize("compression,dense", [("bz2", True), ("lz4", False)])
def test_bag_bytes_to_device_poses_vs_oracle(tmp_path, compression, dense):
    """A bag written here (bz2 / lz4 chunks; PointCloud2 messages in PCL's padded 32-byte point_step, a driver's 22-byte layout and rows
    with padding; is_dense both ways, the non-dense bag carrying NaN returns) -> alego_bag_read_pc2 -> alego_scan_process on the MI355X,
    against the oracle fed the same parsed points: segmentation and feature outputs bit-exact, poses within 1e-4.  Then the plain C++
    host program (`examples/replay --bag`) on the same file: its final poses must be the binding's, digit for digit."""
    import json
    from alego_amd import synth
    from oracle import oracle_py as O
    p = synth.default_params(16, 1800)
    p.input_is_dense = 1 if dense else 0
    n_scans = 8
    msgs, sent = [], []
    for k in range(n_scans):
        pts = synth.scan(p, k)
        pts = pts[: (len(pts) // 4) * 4].copy()       # (height = 4 rows in one of the layouts)
        if not dense:
            pts[37 + k, 0] = np.nan; pts[1000 + 3 * k, 2] = np.nan
        kw = [dict(), dict(layout="driver"), dict(height=4, pad=24)][k % 3]
        msgs.append(("/lslidar_point_cloud", 50.0 + 0.1 * k, pc2_message(pts, k, 49.9 + 0.1 * k, dense=dense, **kw)))
        if k % 3 == 0:
            msgs.append(("/imu/data", 50.0 + 0.1 * k + 0.02, b"\x00" * 40))
        sent.append(pts)
    path = str(tmp_path / f"dev_{compression}.bag")
    write_bag(path, msgs, compression=compression, chunk_msgs=3, topics_types={"/imu/data": "sensor_msgs/Imu"})
    bag = binding.Bag(path)
    assert bag.message_count("/lslidar_point_cloud") == n_scans
    h, o = binding.Handle(p), O.Oracle(p)
    last = None
    for k in range(n_scans):
        pts, stamp, is_dense = bag.read_pc2("/lslidar_point_cloud", k, cap=p.n_scan * p.horizon_scan)
        assert is_dense == dense and abs(stamp - (49.9 + 0.1 * k)) < 1e-6
        assert_bit_equal(pts, sent[k], f"message {k} as parsed")
        o.process_scan(pts)
        flags, odom, mp, seg, feat = h.scan_process(pts, stages=7, want_outputs=True)
        assert_bit_equal(seg["label_image"], o.get("label_img"), f"scan {k} label image")
        for name, key in (("seg", "seg_cloud"), ("col", "seg_col"), ("ground", "seg_ground"), ("range", "seg_range")):
            assert_bit_equal(seg[name], o.get(key), f"scan {k} {key}")
        for name in ("sharp", "less_sharp", "flat", "less_flat"):
            assert_bit_equal(feat[name], o.get(name), f"scan {k} {name}")
        for name in ("sharp_idx", "less_sharp_idx", "flat_idx"):
            assert_bit_equal(h.debug_get(name), o.get(name), f"scan {k} {name}")
        if k > 0:
            assert np.abs(odom["t"] - o.get("odom_pose")[:3]).max() < 1e-4
            assert np.abs(mp["t"] - o.get("map_pose")[:3]).max() < 1e-4
        last = (odom, mp)
    h.close(); bag.close()
    exe = os.path.join(ROOT, "examples", "replay")
    if not os.path.exists(exe):
        pytest.skip("examples/replay not built")
    out = subprocess.run([exe, "--bag", path, "--n-scan", "16", "--horizon", "1800"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["scans"] == n_scans and j["dropped"] == 0
    assert j["odom_t"] == [float(v) for v in last[0]["t"]], (j["odom_t"], last[0]["t"])
    assert j["map_t"] == [float(v) for v in last[1]["t"]], (j["map_t"], last[1]["t"])
