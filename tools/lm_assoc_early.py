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
for step in range(60):
    h.profile_enable(True); h.batch_run(step, 1, st); rep = h.profile_report(); h.profile_enable(False)
    li = h.debug_get("lm_info")
    if li[2]:
        print(step, "assoc us", round(rep.get("lm_assoc", (0, 0))[0] * 1e3), "solve", round(rep.get("lm_solve", (0, 0))[0] * 1e3), "NKF", li[0], "Kds", li[14], li[15], "Lc/Ls", li[19], li[23], "ncc/nsc", li[6], li[7], "rebuild", li[3])
