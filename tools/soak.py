"""Long run: many steps on many streams, then every stream's flags / poses are checked for errors and non-finite values."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
R = 24
p = synth.default_params(16, 1800)
h = binding.Handle(p, n_slots=B, ring_len=R)
for s in range(B):
    for k in range(R):
        h.batch_load(s, k, synth.scan(p, k, stream=s))
t = time.perf_counter()
h.batch_run(0, steps, 7 | binding.REPLAY_PINGPONG)
dt = time.perf_counter() - t
bad = 0
tmax = 0.0
for s in range(B):
    flags, odom, mp = h.batch_get_pose(s)       # raises on a device capacity error
    if not (np.isfinite(odom["t"]).all() and np.isfinite(mp["t"]).all() and np.isfinite(mp["q"]).all()):
        bad += 1
    tmax = max(tmax, float(np.abs(mp["t"]).max()))
c = h.batch_get_counts(0)
print(f"{B} streams x {steps} steps in {dt:.1f} s ({B * steps / dt:.0f} scans/s); non-finite poses: {bad}; max |map t| {tmax:.2f} m; counts of stream 0: {c}")
