"""Is alego_stream_run bound by the host's launch rate?  Enqueue time (sync=False) vs completion time, and the per-kernel HIP-event table."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
import bench
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 800
p = synth.default_params(16, 1800)
bag = bench.make_bags(p, 1, 0)[0]
h = binding.Handle(p, n_slots=1 + 2 * lanes, ring_len=1)
h.replay_create(1, bench.LAP)
for k, a in enumerate(bag):
    h.replay_load(0, k, a)
h.stream_setup(0, 0)
h.stream_run(0, 560, 7)
t0 = time.perf_counter()
h.stream_run(560, steps, 7, sync=False)
t1 = time.perf_counter()
h.synchronize()
t2 = time.perf_counter()
print(json.dumps(dict(lanes=lanes, steps=steps, enqueue_us_per_scan=round(1e6 * (t1 - t0) / steps, 1), total_us_per_scan=round(1e6 * (t2 - t0) / steps, 1))))
h.profile_enable(True)
h.stream_run(560 + steps, 400, 7)
rep = h.profile_report()
h.profile_enable(False)
tot = sum(v[0] for v in rep.values())
for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:28s} {1e3 * v[0] / 400:8.1f} us/scan  {v[1] / 400:5.2f} launches/scan  {1e3 * v[0] / max(v[1], 1):7.1f} us each")
print("sum", round(1e3 * tot / 400, 1), "us/scan over all three streams")
