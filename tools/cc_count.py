import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
p = synth.default_params(16, 1800)
h = binding.Handle(p)
for k in (0, 5):
    seg = h.ip_process(synth.scan(p, k))
    sc = h.debug_get("scal"); fl = h.debug_get("flag_img")
    print("scan", k, "unions", sc[20], "active", int(((fl & 2) > 0).sum()), "right edges", int(((fl & 4) > 0).sum()), "down edges", int(((fl & 8) > 0).sum()), "labels", seg["label_image"].max())
