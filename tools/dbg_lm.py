import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
from oracle import oracle_py as O
p = synth.default_params(16, 1800)
h, o = binding.Handle(p), O.Oracle(p)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    pts = synth.scan(p, k)
    o.process_scan(pts)
    print("scan", k, flush=True)
    flags, odom, mp = h.scan_process(pts, stages=7)
    print(" flags", flags, "gpu map", np.round(mp["t"], 4), "oracle", np.round(o.get("map_pose")[:3], 4), "lm_info", h.debug_get("lm_info")[:24], flush=True)
