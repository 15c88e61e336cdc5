"""Phase timing of ip_fused_t (ip_fused_h / ip_fused_w: workgroup 0 of the last launch); needs a library built with -DALEGO_TIMING (see ipf_timing.py).
usage: ALEGO_LIB=a-lego-loam_amd/libalego_timing.so python tools/iph_timing.py [streams] [groups] — alone on the chip with groups = 1, under the other groups' load otherwise"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
if len(sys.argv) > 2:
    os.environ["ALEGO_STREAM_GROUPS"] = sys.argv[2]
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
p = synth.default_params(16, 1800)
h = binding.Handle(p, n_slots=B, ring_len=8)
sc = [synth.scan(p, k) for k in range(8)]
for s in range(B):
    for k in range(8):
        h.batch_load(s, k, sc[k])
st = int(os.environ.get("STAGES", "1")) | binding.REPLAY_PINGPONG
h.batch_run(0, 40, st)
idx = [0, 1, 3, 8, 9, 10]
names = ["A projection", "B ranges / ground / edges", "C labelling", "D counts + scan", "D emit"]
acc = np.zeros(len(idx)); n = 0
for it in range(20):
    h.batch_run(40 + it * 3, 3, st)
    t = (C.c_longlong * 16)(); binding.lib().alego_ipf_times(t)
    t = np.array([list(t)[i] for i in idx], dtype=np.float64)
    acc += (t - t[0]) / 100.0; n += 1
v = acc / n
for nm, a, b in zip(names, v[:-1], v[1:]):
    print(f"{nm:28s} +{b - a:7.1f} us  (t = {b:7.1f})")
