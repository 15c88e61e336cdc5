"""Phase timing of vox_big on a realistic local map (build with CXXFLAGS_EXTRA=-DALEGO_TIMING)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
p = synth.default_params(16, 1800)
h = binding.Handle(p, n_slots=1, ring_len=48)
for k in range(48):
    h.batch_load(0, k, synth.scan(p, k))
h.batch_run(0, 560, 7 | binding.REPLAY_PINGPONG)
for name, leaf in (("lm_surf_map", 0.4), ("lm_corner_map", 0.2)):
    pts = h.debug_get(name).reshape(-1, 4).copy()
    for rep in range(3):
        out = h.voxel_grid(pts, leaf)
    t = (C.c_longlong * 16)()
    binding.lib().alego_vg_times(t)
    t = np.array(list(t), dtype=np.int64)
    us = lambda a, b: (t[b] - t[a]) / 100.0
    print(name, len(pts), "->", len(out), "bbox %.0f keys %.0f prefix0 %.0f pass0 %.0f prefix1 %.0f pass1 %.0f" % (us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6)),
          "p2 %.0f/%.0f" % (us(6, 7), us(7, 8)) if t[8] > t[7] > 0 else "", "heads %.0f centroid %.0f total %.0f us" % (us(11, 12), us(12, 13), us(0, 13)))
