"""Phase timing of ip_fused (workgroup 0 of the last launch); needs a library built with -DALEGO_TIMING:
   ALEGO_EXTRA_FLAGS=-DALEGO_TIMING ALEGO_BUILD_DIR=build_t ALEGO_SO=libalego_timing.so bash a-lego-loam_amd/build.sh
   ALEGO_LIB=a-lego-loam_amd/libalego_timing.so python tools/ipf_timing.py [streams] [groups]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
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
names = ["zero", "project", "reduce+ori", "gather+ground", "exchange+edges", "own16+par init", "unions", "flatten", "stats+feas", "counts+scan", "emit"]
acc = np.zeros(len(names)); n = 0
for it in range(20):
    h.batch_run(40 + it * 3, 3, st)
    t = (C.c_longlong * 16)(); binding.lib().alego_ipf_times(t)
    t = np.array(list(t)[:len(names)], dtype=np.float64)
    acc += (t - t[0]) / 100.0; n += 1
prev = 0.0
for nm, v in zip(names[1:] , (acc / n)[1:]):
    print(f"{nm:18s} +{v - prev:7.1f} us  (t={v:7.1f})"); prev = v
import time
t0 = time.perf_counter(); h.batch_run(100, 40, st); dt = time.perf_counter() - t0
print(f"B={B} stages={st & 7}: {dt / 40 * 1e6:.0f} us/step, {B * 40 / dt:.0f} scans/s")
