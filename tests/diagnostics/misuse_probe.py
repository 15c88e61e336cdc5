"""Diagnostic: API misuse must come back as an error code, never as a fault (raw ctypes calls with bad arguments)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
L = binding.lib()
p = synth.default_params(16, 1800)
h = binding.Handle(p, n_slots=3, ring_len=2)
H = h._h
N = 16 * 1800
pts = synth.scan(p, 0)
big = np.zeros((N + 10, 4), np.float32)
odom, mp = binding.Pose(), binding.Pose()
def show(name, rc): print(f"{name:44s} rc={rc}", flush=True)
L.alego_batch_load.restype = C.c_int
show("batch_load slot=-1", L.alego_batch_load(H, -1, 0, pts.ctypes.data, len(pts)))
show("batch_load slot=3", L.alego_batch_load(H, 3, 0, pts.ctypes.data, len(pts)))
show("batch_load pos=2 (ring 2)", L.alego_batch_load(H, 0, 2, pts.ctypes.data, len(pts)))
show("batch_load pos=-1", L.alego_batch_load(H, 0, -1, pts.ctypes.data, len(pts)))
show("batch_load n > capacity", L.alego_batch_load(H, 0, 0, big.ctypes.data, len(big)))
show("batch_load n = -5", L.alego_batch_load(H, 0, 0, pts.ctypes.data, -5))
show("batch_load null pts n>0", L.alego_batch_load(H, 0, 0, None, 10))
show("batch_load null pts n=0", L.alego_batch_load(H, 0, 0, None, 0))
show("batch_run n_scans=-1", L.alego_batch_run(H, 0, -1, 7, 1))
show("batch_run stages=0", L.alego_batch_run(H, 0, 1, 0, 1))
show("batch_run REPLAY_BAG without bags", L.alego_batch_run(H, 0, 1, 7 | binding.REPLAY_BAG, 1))
show("batch_get_pose slot=7", L.alego_batch_get_pose(H, 7, C.byref(odom), C.byref(mp)))
show("batch_get_pose null outputs", L.alego_batch_get_pose(H, 0, None, None))
show("scan_process null in", L.alego_scan_process(H, 0, None, 7, None, None, None, None))
sin = binding.ScanIn(); sin.pts, sin.n, sin.stamp = pts.ctypes.data, len(pts), 0.0
show("scan_process no outputs", L.alego_scan_process(H, 0, C.byref(sin), 7, None, None, None, None))
show("lm_keyframe_count slot=9", L.alego_lm_keyframe_count(H, 9))
kf = binding.KeyFrame()
show("lm_get_keyframe none yet / null bufs", L.alego_lm_get_keyframe(H, 0, -1, C.byref(kf)))
show("lm_set_keypose bad id", L.alego_lm_set_keypose(H, 0, 99, (C.c_float * 6)()))
show("lm_apply_correction null", L.alego_lm_apply_correction(H, 0, None))
show("replay_create 0 bags", L.alego_replay_create(H, 0, 10))
show("replay_assign before create", L.alego_replay_assign(H, 0, 0, 0))
show("stream_setup without bags", L.alego_stream_setup(H, 0, 0))
show("trajectory_get before enable", L.alego_trajectory_get(H, 0, 0, 1, None))
show("debug_get unknown name", L.alego_debug_get(H, 0, b"no_such_thing", None, 0, None, None))
show("set_option unknown", L.alego_debug_set_option(H, b"NOPE", 1))
show("lo_push_imu n=-1", L.alego_lo_push_imu(H, 0, None, -1))
show("null handle batch_run", L.alego_batch_run(None, 0, 1, 7, 1))
show("last_error", L.alego_last_error(H))
# the handle still works
for k in range(3):
    fl, od, m = h.scan_process(synth.scan(p, k), stages=7)
print("handle still fine:", od["t"], flush=True)
h.close()
