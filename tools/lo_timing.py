"""Where lo_solve spends its time (one stream; build kernels_lo with -DALEGO_TIMING)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
p = synth.default_params(16, 1800)
h = binding.Handle(p, n_slots=1, ring_len=24)
for k in range(24): h.batch_load(0, k, synth.scan(p, k))
h.batch_run(0, 200, 3 | binding.REPLAY_PINGPONG)
t = (C.c_longlong * 8)(); binding.lib().alego_lo_times(t); t = np.array(list(t), dtype=np.float64)
calls = t[7]
names = ["trig(coop)", "rows", "reduce", "propose+sync", "consume+sync", "store rotation", "kernel total"]
for n, v in zip(names, t[:7]): print(f"{n:16s} {v / 100.0 / calls:8.2f} us per lo_solve call")
print("calls", int(calls))
