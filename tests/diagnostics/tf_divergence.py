"""Teacher-forced full loop: per scan compare LO correspondences / params / solver summaries and LM outputs (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
from oracle import oracle_py as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 560
geom = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16, 1800)
p = synth.default_params(*geom)
h, o = binding.Handle(p), O.Oracle(p)
for k in range(n):
    pts = synth.scan(p, k)
    h.set_lo_params(o.get("lo_params")); h.set_lm_params(o.get("lm_params"))
    o.process_scan(pts)
    flags, odom, mp = h.scan_process(pts, stages=7)
    if k == 0: continue
    msg = []
    oc = o.get("lo_surf_corr").reshape(-1, 4); gc = h.debug_get("lo_surf_corr").reshape(-1, 4); gc = gc[gc[:, 1] >= 0]
    if gc.shape != oc.shape or not np.array_equal(gc, oc): msg.append(f"surf corr differ {gc.shape} {oc.shape}")
    oc = o.get("lo_corner_corr").reshape(-1, 3); gc = h.debug_get("lo_corner_corr").reshape(-1, 4); gc = gc[gc[:, 1] >= 0][:, :3]
    if gc.shape != oc.shape or not np.array_equal(gc, oc): msg.append(f"corner corr differ {gc.shape} {oc.shape}")
    st = h.debug_get("lo_state")
    d1 = np.abs(st[18:24] - o.get("lo_params_after_surf")).max(); d2 = np.abs(odom["params"] - o.get("lo_params")).max()
    info = o.get("lo_solve_info"); sc = h.debug_get("scal")
    s1 = (sc[10] & 0xFF, (sc[10] >> 8) & 0xFF, sc[10] >> 16); s2 = (sc[11] & 0xFF, (sc[11] >> 8) & 0xFF, sc[11] >> 16)
    if s1 != tuple(info[0:3]) or s2 != tuple(info[3:6]): msg.append(f"LO summaries dev {s1} {s2} orc {tuple(info[0:6])}")
    wo = o.get("odom_pose"); want = o.get("map_pose")
    eo = np.abs(odom["t"] - wo[:3]).max(); em = np.abs(mp["t"] - want[:3]).max()
    gi, oi = h.debug_get("lm_info"), o.get("lm_info")
    if bool(gi[2]) and bool(oi[0]):
        if (gi[6], gi[7]) != (oi[3], oi[4]): msg.append(f"LMcorr dev {gi[6]},{gi[7]} orc {oi[3]},{oi[4]}")
        g8 = ((gi[8] & 0xFF, (gi[8] >> 8) & 0xFF, gi[8] >> 16), (gi[9] & 0xFF, (gi[9] >> 8) & 0xFF, gi[9] >> 16))
        if g8 != (tuple(oi[5:8]), tuple(oi[8:11])): msg.append(f"LM summaries dev {g8} orc {tuple(oi[5:11])}")
        if bool(gi[10]) != bool(oi[2]): msg.append("KEYFRAME flag differs")
        dl = np.abs(h.debug_get("lm_state")[0:6] - o.get("lm_params")).max()
    else:
        dl = 0.0
    if msg or max(d1, d2) > 1e-7 or eo > 1e-5 or em > 1e-5:
        print(f"scan {k}: dLOsurf {d1:.2e} dLO {d2:.2e} odom {eo:.2e} map {em:.2e} dLM {dl:.2e} | " + "; ".join(msg))
print("done")
