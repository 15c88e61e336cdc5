"""Phase timing of fe_ring_out (ring 8 of the launch's first slot, last launch); build with -DALEGO_TIMING (tools/README.md)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p = synth.default_params(16, 1800)
h = binding.Handle(p, n_slots=B, ring_len=8)
sc = [synth.scan(p, k) for k in range(8)]
for s in range(B):
    for k in range(8): h.batch_load(s, k, sc[k])
st = 3 | binding.REPLAY_PINGPONG
h.batch_run(0, 20, st)
acc = np.zeros(9); n = 0; sub = np.zeros(5); stats = None
for it in range(20):
    h.batch_run(20 + it, 1, st)
    t = (C.c_longlong * 24)(); binding.lib().alego_fo_times(t)
    full = np.array(list(t), dtype=np.float64)
    t = full[:9]; acc += (t - t[0]) / 100.0; n += 1
    sub += (full[9:14] - np.array([full[3], full[9], full[10], full[11], full[6]])) / 100.0; stats = full[16:21]
names = ["prologue", "bbox", "keys+runs", "order", "count", "lookback", "centroids", "boxes"]
a = acc / n
print("  order: zero %.1f hist %.1f scan %.1f scatter %.1f | centroids: box init %.1f | n_all %d runs %d valid %d buckets %d largest %d" % (*(sub / n), *stats))
print(f"B={B}: " + " ".join(f"{nm} {a[i+1]-a[i]:.1f}" for i, nm in enumerate(names)) + f" total {a[8]:.1f} us")
