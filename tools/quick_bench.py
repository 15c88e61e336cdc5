import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
B = int(sys.argv[1]); R = 8; K = int(sys.argv[2])
p = synth.default_params(16, 1800)
h = binding.Handle(p, n_slots=B, ring_len=R)
scans = [synth.scan(p, k) for k in range(R)]
for s in range(B):
    for k in range(R):
        h.batch_load(s, k, scans[k])
h.batch_run(0, R, stages=3)
t = time.time(); h.batch_run(0, K, stages=3); dt = time.time() - t
print(f"B={B} K={K}: {dt*1e3:.2f} ms total, {dt/K*1e6:.1f} us/step, {B*K/dt:.0f} scans/s", h.batch_get_counts(0))
