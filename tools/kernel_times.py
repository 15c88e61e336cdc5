import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
stages = int(sys.argv[2]) if len(sys.argv) > 2 else 7
K = int(sys.argv[3]) if len(sys.argv) > 3 else 40
prime = int(sys.argv[4]) if len(sys.argv) > 4 else 100
p = synth.default_params(16, 1800)
R = 24
h = binding.Handle(p, n_slots=B, ring_len=R)
sc = [synth.scan(p, k) for k in range(R)]
for s in range(B):
    for k in range(R):
        h.batch_load(s, k, sc[k])
st = stages | binding.REPLAY_PINGPONG
h.batch_run(0, prime, st)
t = time.perf_counter(); h.batch_run(prime, K, st); dt = time.perf_counter() - t
h.profile_enable(True); h.batch_run(prime + K, K, st); rep = h.profile_report(); h.profile_enable(False)
print(f"B={B} stages={stages}: {dt/K*1e6:.0f} us/step, {B*K/dt:.0f} scans/s")
for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0]):
    print("  %-26s %8.1f us x%4d  %7.1f us/step" % (k, 1e3 * v[0] / max(v[1], 1), v[1], 1e3 * v[0] / K))
