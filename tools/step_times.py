import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R = 48
p = synth.default_params(16, 1800)
h = binding.Handle(p, n_slots=B, ring_len=R)
for s in range(B):
    for k in range(R):
        h.batch_load(s, k, synth.scan(p, k, stream=s))
st = 7 | binding.REPLAY_PINGPONG
h.batch_run(0, 560, st)
step = 560
ts = []
for i in range(40):
    t = time.perf_counter(); h.batch_run(step, 1, st); dt = time.perf_counter() - t; step += 1
    li = h.debug_get("lm_info")
    ts.append((round(dt * 1e6), int(li[0]), int(li[2]), int(li[3]), int(li[10])))
print("per-step (us, NKF, RUN, REBUILD, KF_ADDED):", ts)
