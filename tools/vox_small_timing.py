"""Phase timing of vox_small on the current-scan clouds (build kernels_voxel with -DALEGO_TIMING)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
p = synth.default_params(16, 1800)
h = binding.Handle(p, n_slots=1, ring_len=1)
for k in range(4):
    flags, feat, odom, = None, None, None
    h.scan_process(synth.scan(p, k), stages=7)
for name, leaf in (("less_flat", 0.8), ("less_sharp", 0.4)):
    pts = h.debug_get(name).reshape(-1, 4).copy()
    for rep in range(3):
        out = h.voxel_grid(pts, leaf)
    t = (C.c_longlong * 16)(); binding.lib().alego_vg_times(t)
    t = np.array(list(t)[:8], dtype=np.float64); t = (t - t[0]) / 100.0
    names = ["bbox", "geom+keys+hist", "scan", "scatter", "bucket sort", "vox scan", "centroids"]
    print(name, len(pts), "->", len(out), " ".join(f"{n} {t[i+1]-t[i]:.1f}" for i, n in enumerate(names)), f"total {t[7]:.1f} us")
