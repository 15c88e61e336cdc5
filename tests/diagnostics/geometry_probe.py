import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
from oracle import oracle_py as O
"""Diagnostic: unusual sensor geometries through the whole loop against the oracle, one geometry per process (usage: geometry_probe.py NS H [NS H ...]; several pairs run in one process)."""
for geom in [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]:
    p = synth.default_params(*geom)
    try:
        h = binding.Handle(p)
    except Exception as e:
        print(geom, "create refused:", str(e)[:100], flush=True); continue
    o = O.Oracle(p)
    ok = True
    try:
        for k in range(3):
            pts = synth.scan(p, k)
            o.process_scan(pts)
            fl, odom, mp, seg, feat = h.scan_process(pts, stages=7, want_outputs=True)
            for name in ("seg_cloud","outlier","less_sharp_idx","flat_idx","less_flat"):
                a, b = h.debug_get(name), o.get(name)
                if a.shape != b.shape or not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
                    ok = False; print(geom, "scan", k, name, "differs", a.shape, b.shape); break
        print(geom, "ok" if ok else "MISMATCH", flush=True)
    except Exception as e:
        print(geom, "run error:", str(e)[:120])
    h.close()
