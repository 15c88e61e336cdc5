"""Phase timing of ipb_band / ipb_merge (workgroup 0 of slot 0, last launch) at 64x2048; library built with -DALEGO_TIMING (ALEGO_LIB=.../libalego_timing.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
stages = int(sys.argv[2]) if len(sys.argv) > 2 else 1
p = synth.default_params(64, 2048)
h = binding.Handle(p, n_slots=B, ring_len=4)
sc = [synth.scan(p, k) for k in range(4)]
for s in range(B):
    for k in range(4):
        h.batch_load(s, k, sc[k])
st = stages | binding.REPLAY_PINGPONG
h.batch_run(0, 20, st)
acc = np.zeros(48); n = 0
for it in range(20):
    h.batch_run(20 + it * 3, 3, st)
    t = (C.c_longlong * 48)(); binding.lib().alego_ipb_times(t)
    acc += np.array(list(t), dtype=np.float64); n += 1
t = acc / n / 100.0
for name, ks in (("ipb_band", [(0, "start"), (1, "walk"), (2, "halo+barrier"), (3, "edges"), (4, "heads+barrier"), (5, "unions+barrier"), (6, "runs+masks")]),
                 ("ipb_merge", [(10, "start"), (11, "strip+seams"), (12, "merge stats"), (13, "flags"), (14, "classify+count"), (15, "prefix"), (16, "offsets"), (17, "zero")])):
    print(name)
    for (k, nm), (k0, _) in zip(ks[1:], ks[:-1]):
        print(f"  {nm:16s} +{t[k] - t[k0]:7.1f} us")
    print(f"  total            {t[ks[-1][0]] - t[ks[0][0]]:7.1f} us")
