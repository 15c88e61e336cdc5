"""lo_assoc walk statistics (timing build): wavefront sweeps, walk_eval turns of the pass, surviving boxes and window boxes per query row."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
geo = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "16x1800").split("x"))
p = synth.default_params(*geo)
h = binding.Handle(p, n_slots=B, ring_len=8)
for s in range(B):
    for k in range(8): h.batch_load(s, k, synth.scan(p, k, stream=s))
h.batch_run(0, 40, 3 | binding.REPLAY_PINGPONG)
t = (C.c_ulonglong * 8)(); binding.lib().alego_la_counts(t); t = list(t)
for kind, name in ((0, "surf"), (1, "corner")):
    sw, turns, surv, win = t[kind * 4: kind * 4 + 4]
    print(f"{geo} {name}: {sw} wavefront sweeps (4 query rows each), {turns / max(sw, 1):.2f} walk_eval turns per sweep in the pass, {surv / max(4 * sw, 1):.2f} surviving boxes per row, {win / max(4 * sw, 1):.1f} window boxes per row")
