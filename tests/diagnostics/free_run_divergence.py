"""Where does a free-running device stream start to differ from the oracle?  (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
from oracle import oracle_py as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 560
p = synth.default_params(16, 1800)
h, o = binding.Handle(p), O.Oracle(p)
thr = [1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3]
ti = 0
prev = 0.0
for k in range(n):
    pts = synth.scan(p, k)
    o.process_scan(pts)
    flags, odom, mp = h.scan_process(pts, stages=7)
    if k == 0:
        continue
    want = o.get("map_pose"); wo = o.get("odom_pose")
    e = float(np.abs(mp["t"] - want[:3]).max()); eo = float(np.abs(odom["t"] - wo[:3]).max())
    gi, oi = h.debug_get("lm_info"), o.get("lm_info")
    sc = h.debug_get("scal")
    los = o.get("lo_solve_info")
    note = ""
    if bool(gi[2]) and bool(oi[0]):
        if (gi[6], gi[7]) != (oi[3], oi[4]): note += f" LMcorr dev {gi[6]},{gi[7]} orc {oi[3]},{oi[4]}"
        if bool(gi[10]) != bool(oi[2]): note += f" KEYFRAME dev {gi[10]} orc {oi[2]}"
        if gi[0] != oi[11]: note += f" nkf dev {gi[0]} orc {oi[11]}"
    while ti < len(thr) and max(e, eo) > thr[ti]:
        print(f"scan {k}: first above {thr[ti]:g}: map err {e:.3e} odom err {eo:.3e}"); ti += 1
    if note or e > 3 * prev and e > 1e-7:
        print(f"scan {k}: map err {e:.3e} odom err {eo:.3e}{note}")
    prev = max(e, 1e-12)
print("final", e, eo)
