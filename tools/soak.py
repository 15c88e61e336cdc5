"""Long run on the bench workload (bag replay): many steps on many streams, then every stream's flags / poses are checked for device
errors (capacity, internal "voxel list out of sync") and non-finite values.  usage: soak.py [streams] [steps] [geometry, e.g. 64x2048] [keyframes] [kf_cap]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
geo = tuple(int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else '16x1800').split('x'))
p = synth.default_params(*geo)
if len(sys.argv) > 4 and int(sys.argv[4]) > 0: p.recent_keyframe_num = int(sys.argv[4])
if len(sys.argv) > 5 and int(sys.argv[5]) > 0: p.kf_cap_surf, p.kf_cap_outlier = int(sys.argv[5]), max(256, int(sys.argv[5]) // 4)
bags = bench.make_bags(p, 4, 0)
h = binding.Handle(p, n_slots=B, ring_len=1)
bench.setup_replay(h, bags, B)
t = time.perf_counter()
h.batch_run(0, steps, 7 | binding.REPLAY_BAG)
dt = time.perf_counter() - t
bad = errs = 0
tmax = 0.0
for s in range(B):
    try:
        flags, odom, mp = h.batch_get_pose(s)       # raises on a device capacity / internal error
    except binding.AlegoError as e:
        errs += 1
        print("slot", s, e)
        continue
    if not (np.isfinite(odom["t"]).all() and np.isfinite(mp["t"]).all() and np.isfinite(mp["q"]).all()):
        bad += 1
    tmax = max(tmax, float(np.abs(mp["t"]).max()))
c = h.batch_get_counts(0)
print(f"{B} streams x {steps} steps in {dt:.1f} s ({B * steps / dt:.0f} scans/s); device errors: {errs}; non-finite poses: {bad}; max |map t| {tmax:.2f} m; counts of stream 0: {c}")
