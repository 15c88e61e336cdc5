import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
p = synth.default_params(16, 1800)
h = binding.Handle(p, n_slots=1, ring_len=24)
for k in range(24):
    h.batch_load(0, k, synth.scan(p, k))
st = 7 | binding.REPLAY_PINGPONG
h.batch_run(0, 561, st)
for i in range(4):
    h.batch_run(561 + i, 1, st)
    li = h.debug_get("lm_info"); ld = h.debug_get("lm_state")
    print("run", li[2], "sum", hex(li[8]), hex(li[9]), "rows", li[6], li[7], "cycles (100 MHz wall clock ticks x ?; clock64) pose/rows/reduce/control/pack", ld[43:48].astype(np.int64))
