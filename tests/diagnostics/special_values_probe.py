"""Diagnostic: scans with non-finite / extreme / degenerate points through ImageProjection (and the whole loop) against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import time
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
from oracle import oracle_py as O
rng = np.random.default_rng(3)
for dense in (0, 1):
    p = synth.default_params(16, 1800)
    p.input_is_dense = dense
    h, o = binding.Handle(p), O.Oracle(p)
    for k in range(4):
        pts = synth.scan(p, k).copy()
        n = len(pts)
        idx = rng.choice(n, 400, replace=False)
        specials = np.array({'nonfinite': [np.inf, -np.inf, np.nan, 0.0, -0.0, 1e-40, 1e-30], 'huge': [1e30, -1e30, 1e19, 3.4e38], 'large': [1e6, -1e6, 3e5, 1e7], 'km': [5000.0, -3000.0, 8000.0, 20000.0]}[sys.argv[1]], np.float32)
        for j, i in enumerate(idx):
            c = j % 7
            if c < 3:
                pts[i, c] = specials[rng.integers(len(specials))]
            elif c == 3:
                pts[i, :3] = 0.0
            elif c == 4:
                pts[i, :3] = specials[rng.integers(len(specials))]
            elif c == 5:
                pts[i, 2] = specials[rng.integers(len(specials))]; pts[i, 0] = 0.0; pts[i, 1] = 0.0
            else:
                pts[i, :3] *= np.float32(1e-20)
        if k == 0:   # first and last point special (orientation block)
            pts[0, :3] = 0.0; pts[-1, 0] = np.inf
        t0 = time.perf_counter(); o.process_scan(pts); t1 = time.perf_counter()
        fl, odom, mp, seg, feat = h.scan_process(pts, stages=7, want_outputs=True)
        bad = []
        for name, got in (("seg_cloud", seg["seg"]), ("outlier", seg["outlier"]), ("orientation", seg["orientation"]), ("less_sharp", feat["less_sharp"]), ("less_flat", feat["less_flat"])):
            want = o.get(name)
            if got.shape != want.shape or not np.array_equal(got.view(np.uint8), want.view(np.uint8)):
                bad.append((name, got.shape, want.shape))
        print("dense", dense, "scan", k, "MISMATCH " + str(bad) if bad else "ok", f"oracle {t1 - t0:.2f} s device {time.perf_counter() - t1:.2f} s", flush=True)
    h.close()
