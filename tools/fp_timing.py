"""Phase timing inside fe_pick4 (library built with -DALEGO_TIMING): ticks of wavefront 0 of one workgroup, 384 streams."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
p = synth.default_params(16, 1800)
h = binding.Handle(p, n_slots=B, ring_len=8)
sc = [synth.scan(p, k) for k in range(8)]
for s in range(B):
    for k in range(8):
        h.batch_load(s, k, sc[k])
h.batch_run(0, 30, 3 | binding.REPLAY_PINGPONG)
t = (C.c_longlong * 8)()
binding.lib().alego_fp_times(t)
t = np.array(list(t), dtype=np.int64) / 100.0
print("us: stage %.1f keys %.1f sharp %.1f flat %.1f (sync) %.1f lists %.1f" % tuple(t[:6]), "total %.1f" % t[:6].sum())
