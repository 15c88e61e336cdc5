"""What the sharded registration (BASELINE config 5, alego_dist_init) costs on ONE GPU: the same single stream with the fused
on-chip solver (lm_solve) and with the registration split into lm_shard_* launches + one 32-double ncclAllReduce per evaluation
(world = 1: the collective is a device-local copy, so this is the launch + collective-call overhead without any xGMI time).
usage: shard_cost.py [geometry] [keyframes] [scans]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
geom = sys.argv[1] if len(sys.argv) > 1 else "16x1800"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nscan = int(sys.argv[3]) if len(sys.argv) > 3 else 300
ns, hs = (int(v) for v in geom.split("x"))
p = synth.default_params(ns, hs)
if K:
    p.recent_keyframe_num = K
scans = [synth.scan(p, k) for k in range(nscan)]
out = {}
for mode in ("fused", "sharded"):
    h = binding.Handle(p, n_slots=1, ring_len=nscan)
    if mode == "sharded":
        h.dist_init(0, 1, binding.dist_unique_id())
    for k in range(nscan):
        h.batch_load(0, k, scans[k])
    warm = nscan // 3
    h.batch_run(0, warm, stages=7)
    h.profile_enable(True)
    t0 = time.perf_counter()
    h.batch_run(warm, nscan - warm, stages=7)
    dt = time.perf_counter() - t0
    rep = h.profile_report()
    h.profile_enable(False)
    lm = {k: dict(ms=round(v[0], 3), launches=v[1]) for k, v in rep.items() if k.startswith("lm_solve") or k.startswith("lm_shard") or "allreduce" in k.lower()}
    _, _, mp = h.batch_get_pose(0)
    out[mode] = dict(scans=nscan - warm, wall_ms_per_scan=round(1e3 * dt / (nscan - warm), 4), solver_kernels=lm,
                     solver_ms_per_mapping_frame=round(sum(v["ms"] for v in lm.values()) / max(1, (nscan - warm) // 2), 4), map_t=[float(x) for x in mp["t"]])
    if mode == "sharded":
        h.dist_shutdown()
    h.close()
out["poses_bit_equal"] = out["fused"]["map_t"] == out["sharded"]["map_t"]
print(json.dumps(out))
