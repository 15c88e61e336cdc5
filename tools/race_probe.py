"""Diagnostic: run the multi-group batch twice and compare LaserMapping state slot by slot (looks for timing-dependent results)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
p = synth.default_params(16, 1800)
nslot, nscan = int(os.environ.get("PROBE_SLOTS", "5")), int(sys.argv[1]) if len(sys.argv) > 1 else 80
os.environ["ALEGO_STREAM_GROUPS"] = sys.argv[2] if len(sys.argv) > 2 else "3"
scans = [[synth.scan(p, k, stream=s) for k in range(nscan)] for s in range(nslot)]
def run():
    h = binding.Handle(p, n_slots=nslot, ring_len=nscan)
    for s in range(nslot):
        for k in range(nscan):
            h.batch_load(s, k, scans[s][k])
    h.batch_run(0, nscan, stages=7)
    out = []
    for s in range(nslot):
        _, od, mp = h.batch_get_pose(s)
        d = dict(odom=od["t"], mapt=mp["t"], params=mp["params"])
        for name in ("lm_info", "lm_surf_map_ds", "lm_corner_map_ds", "lm_voxel_keys_c", "lm_voxel_keys_s", "lm_kf_surf_map", "lm_kf_corner_map", "lm_keyposes", "lm_surf_total_ds", "lm_corner_ds"):
            d[name] = h.debug_get(name, slot=s, cap_bytes=1 << 24)
        out.append(d)
    h.close()
    return out
ref = run()
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 6):
    cur = run()
    for s in range(nslot):
        bad = [k for k in ref[s] if ref[s][k].shape != cur[s][k].shape or not np.array_equal(ref[s][k].view(np.uint8) if ref[s][k].dtype != object else ref[s][k], cur[s][k].view(np.uint8))]
        if bad:
            print(f"rep {rep} slot {s}: differ {bad}")
            for k in bad:
                a, b = ref[s][k], cur[s][k]
                if a.shape == b.shape:
                    idx = np.nonzero(a.reshape(-1) != b.reshape(-1))[0]
                    print("   ", k, a.shape, "first diffs at", idx[:6], a.reshape(-1)[idx[:4]], b.reshape(-1)[idx[:4]])
                else:
                    print("   ", k, a.shape, b.shape)
print("done")
# single-slot handles on stream 0, host entry point per scan (the reference side of test_batch_full_loop_equals_single_stream)
def run1(s=0):
    h = binding.Handle(p)
    for k in range(nscan):
        _, od, mp = h.scan_process(scans[s][k], stages=7)
    d = dict(odom=od["t"], mapt=mp["t"], params=mp["params"])
    for name in ("lm_info", "lm_surf_map_ds", "lm_corner_map_ds", "lm_voxel_keys_c", "lm_voxel_keys_s", "lm_keyposes", "lm_surf_total_ds", "lm_corner_ds"):
        d[name] = h.debug_get(name, cap_bytes=1 << 24)
    h.close()
    return d
for rep in range(8):
    c = run1()
    bad = [k for k in c if ref[0][k].shape != c[k].shape or not np.array_equal(ref[0][k].view(np.uint8), c[k].view(np.uint8))]
    print("single rep", rep, "differs from batch slot 0 in", bad)
    for k in bad[:4]:
        a, b = ref[0][k], c[k]
        if a.shape == b.shape:
            idx = np.nonzero(a.reshape(-1) != b.reshape(-1))[0]
            print("   ", k, a.shape, "n diff", idx.size, "first", idx[:6], a.reshape(-1)[idx[:3]], b.reshape(-1)[idx[:3]])
        else:
            print("   ", k, a.shape, b.shape)
