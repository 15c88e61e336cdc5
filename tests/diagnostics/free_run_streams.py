"""Diagnostic: first scan at which a FREE-running device stream leaves the oracle by more than 1e-4 m, for several streams of the T0 lap."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
from oracle import oracle_py as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 800
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 6
p = synth.default_params(16, 1800)
res = []
for s in range(ns):
    h, o = binding.Handle(p), O.Oracle(p)
    first5 = first4 = None
    emax = 0.0
    for k in range(n):
        pts = synth.scan(p, k, stream=s)
        o.process_scan(pts)
        _, odom, mp = h.scan_process(pts, stages=7)
        if k == 0: continue
        e = max(float(np.abs(mp["t"] - o.get("map_pose")[:3]).max()), float(np.abs(odom["t"] - o.get("odom_pose")[:3]).max()))
        emax = max(emax, e)
        if first5 is None and e > 1e-5: first5 = k
        if first4 is None and e > 1e-4: first4 = k
    res.append({"stream": s, "first_scan_beyond_1e-5": first5, "first_scan_beyond_1e-4": first4, "max_err_m": emax})
    h.close()
    print(json.dumps(res[-1]), flush=True)
