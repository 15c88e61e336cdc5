"""Diagnostic: replays tests/test_gpu_parity.py::test_random_keyframe_operations for one seed and, at the first map mismatch, compares the
window's frame ids and every resident key frame (pose, clouds) between the device and the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
from oracle import oracle_py as O
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(100 + seed)
p = synth.default_params(16, 1800)
p.recent_keyframe_num = int(rng.integers(3, 8)); p.min_keyframe_dist = 0.09
print("K", p.recent_keyframe_num)
h, o = binding.Handle(p), O.Oracle(p)
for k in range(70):
    pts = synth.scan(p, k)
    h.set_lo_params(o.get("lo_params")); h.set_lm_params(o.get("lm_params"))
    o.process_scan(pts)
    h.scan_process(pts, stages=7)
    if k == 0: continue
    gi = h.debug_get("lm_info"); oi = o.get("lm_info")
    a, b = h.debug_get("lm_corner_map_ds"), o.get("lm_corner_map_ds")
    ran = bool(oi[0])
    print("scan", k, "ran", ran, "nkf", gi[0], "window", h.debug_get("lm_window") if ran else "", "kraw dev", gi[12], gi[13], flush=True)
    if ran and (a.shape != b.shape or not np.array_equal(a.view(np.uint8), b.view(np.uint8))):
        print("MISMATCH at scan", k, a.shape, b.shape, "oracle raw sizes", o.get("lm_corner_map").shape, o.get("lm_surf_map").shape)
        nkf = h.lm_keyframe_count(); poses = o.get("lm_keyposes").reshape(-1, 6)
        for i in range(max(0, nkf - p.recent_keyframe_num - 1), nkf):
            kf = h.lm_get_keyframe(i); oc, os_, oo = o.lm_keyframe(i)
            same = [np.array_equal(kf["pose"], poses[i]), kf["corner"].shape == oc.shape and np.array_equal(kf["corner"], oc), kf["surf"].shape == os_.shape and np.array_equal(kf["surf"], os_), kf["outlier"].shape == oo.shape and np.array_equal(kf["outlier"], oo)]
            print("  frame", i, "pose/corner/surf/outlier equal:", same, kf["corner"].shape, oc.shape)
        break
    nkf = h.lm_keyframe_count()
    if nkf < 2 or rng.random() > 0.35: continue
    op = int(rng.integers(0, 4))
    poses = o.get("lm_keyposes").reshape(-1, 6).copy()
    print("   op", ("set_all+reset", "reset", "correction", "add")[op], "at nkf", nkf, flush=True)
    if op == 0:
        d = rng.normal(0, 0.05, 6).astype(np.float32) * np.array([1, 1, 0.2, 0.1, 0.1, 0.3], np.float32)
        for i in range(nkf):
            q = (poses[i] + d).astype(np.float32); o.lm_set_keypose(i, q)
            if i >= nkf - p.recent_keyframe_num: h.lm_set_keypose(i, q)
        o.lm_reset_window(); h.lm_reset_window()
    elif op == 1:
        o.lm_reset_window(); h.lm_reset_window()
    elif op == 2:
        a_ = float(rng.normal(0, 0.01))
        rc = np.array([[np.cos(a_), -np.sin(a_), 0, rng.normal(0, 0.05)], [np.sin(a_), np.cos(a_), 0, rng.normal(0, 0.05)], [0, 0, 1, rng.normal(0, 0.01)]])
        o.lm_apply_correction(rc); h.lm_apply_correction(rc)
    else:
        src = int(rng.integers(max(0, nkf - p.recent_keyframe_num), nkf))
        c, s_, ol = o.lm_keyframe(src)
        q = poses[src].copy(); q[:2] += rng.normal(0, 0.3, 2).astype(np.float32)
        try:
            h.lm_add_keyframe(q, c, s_, ol)
        except binding.AlegoError as e:
            print("   refused", flush=True); continue
        o.lm_add_keyframe(q, c, s_, ol)
