"""Phase timing of cc_lds16 (slot 0 workgroup, last launch) under load; build kernels_ip with -DALEGO_TIMING."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = synth.default_params(16, 1800)
h = binding.Handle(p, n_slots=B, ring_len=8)
sc = [synth.scan(p, k) for k in range(8)]
for s in range(B):
    for k in range(8):
        h.batch_load(s, k, sc[k])
st = 7 | binding.REPLAY_PINGPONG
h.batch_run(0, 100, st)
acc = np.zeros(9)
n = 0
for it in range(20):
    h.batch_run(100 + it * 3, 3, st)
    t = (C.c_longlong * 16)(); binding.lib().alego_cc_times(t)
    t = np.array(list(t)[:9], dtype=np.float64)
    acc += (t - t[0]) / 100.0; n += 1
names = ["start", "flags+init", "unions", "roots", "sizes", "rows+parent", "classify", "scan", "output"]
prev = 0.0
for nm, v in zip(names, acc / n):
    print(f"{nm:12s} +{v - prev:7.1f} us  (t={v:7.1f})"); prev = v
