import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p = synth.default_params(16, 1800)
h = binding.Handle(p, n_slots=B, ring_len=48)
for s in range(B):
    for k in range(48):
        h.batch_load(s, k, synth.scan(p, k, stream=s))
st = 7 | binding.REPLAY_PINGPONG
h.batch_run(0, 560, st)
step = 560
for i in range(30):
    h.profile_enable(True)
    t = time.perf_counter(); h.batch_run(step, 1, st); dt = time.perf_counter() - t; step += 1
    rep = h.profile_report(); h.profile_enable(False)
    li = h.debug_get("lm_info")
    if li[3] or dt > 2e-3:
        top = [kv for kv in sorted(rep.items(), key=lambda kv: -kv[1][0]) if kv[0].startswith(('vox', 'lm_', 'memset'))]
        print(f"step {step} {dt*1e6:.0f}us REBUILD={li[3]} Kraw={li[12]},{li[13]} Kds={li[14]},{li[15]}:", [(k, round(v[0]*1e3), v[1]) for k, v in top])
