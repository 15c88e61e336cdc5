"""Per-kernel HIP-event table for B streams replaying one 560-scan lap (bench.py's workload, one bag).
usage: kernel_times.py B [stages=7] [K=40] [prime=600] [geometry=16x1800] [keyframes=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
stages = int(sys.argv[2]) if len(sys.argv) > 2 else 7
K = int(sys.argv[3]) if len(sys.argv) > 3 else 40
prime = int(sys.argv[4]) if len(sys.argv) > 4 else 600
ns, hs = (int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "16x1800").split("x"))
p = synth.default_params(ns, hs)
if len(sys.argv) > 6 and int(sys.argv[6]) > 0:
    p.recent_keyframe_num = int(sys.argv[6])
bags = bench.make_bags(p, int(os.environ.get("BAGS", "1")), 0)
h = binding.Handle(p, n_slots=B, ring_len=1)
bench.setup_replay(h, bags, B)
st = stages | binding.REPLAY_BAG
h.batch_run(0, prime, st)
t = time.perf_counter(); h.batch_run(prime, K, st); dt = time.perf_counter() - t
h.profile_enable(True); h.batch_run(prime + K, K, st); rep = h.profile_report(); h.profile_enable(False)
print(f"B={B} stages={stages} {ns}x{hs} K={p.recent_keyframe_num}: {dt/K*1e6:.0f} us/step, {B*K/dt:.0f} scans/s", h.batch_get_counts(0))
tot = sum(v[0] for v in rep.values())
for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0]):
    print("  %-26s %8.1f us x%4d  %7.1f us/step %5.1f%%" % (k, 1e3 * v[0] / max(v[1], 1), v[1], 1e3 * v[0] / K, 100 * v[0] / tot))
