import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
p = synth.default_params(16, 1800)
pts = synth.scan(p, 0)
for stages in (1, 3, 7):
    h = binding.Handle(p)
    h.scan_process(pts, stages=stages)
    print("stages", stages, "scal", h.debug_get("scal")[:16], flush=True)
    if stages == 7:
        print(" lm_info", h.debug_get("lm_info")[:24])
    h.close()
