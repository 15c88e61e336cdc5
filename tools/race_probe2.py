"""Diagnostic: the body of tests/test_gpu_parity.py::test_batch_full_loop_equals_single_stream, repeated, with the LaserMapping
state compared array by array when the poses differ (which side moved, and where)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("PROBE_TORCH"):
    import torch
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
p = synth.default_params(16, 1800)
NAMES = ("lm_info", "lm_surf_map_ds", "lm_corner_map_ds", "lm_voxel_keys_c", "lm_voxel_keys_s", "lm_keyposes", "lm_surf_total_ds", "lm_corner_ds")

def state(h, slot, batch):
    _, od, mp = h.batch_get_pose(slot) if batch else h.last
    d = dict(odom=od["t"], mapt=mp["t"], params=mp["params"])
    for name in NAMES:
        d[name] = h.debug_get(name, slot=slot, cap_bytes=1 << 24)
    return d

def diff(a, b):
    bad = []
    for k in a:
        if a[k].shape != b[k].shape or not np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)):
            bad.append(k)
    return bad

def body(nslot, groups, nscan, tag):
    os.environ["ALEGO_STREAM_GROUPS"] = str(groups)
    hb = binding.Handle(p, n_slots=nslot, ring_len=nscan)
    del os.environ["ALEGO_STREAM_GROUPS"]
    for s in range(nslot):
        for k in range(nscan):
            hb.batch_load(s, k, synth.scan(p, k, stream=s))
    hb.batch_run(0, nscan, stages=7)
    res = []
    for s in range(nslot):
        h1 = binding.Handle(p)
        for k in range(nscan):
            h1.last = h1.scan_process(synth.scan(p, k, stream=s), stages=7)
        sb, s1 = state(hb, s, True), state(h1, 0, False)
        bad = diff(sb, s1)
        print(tag, "slot", s, "batch vs single differ in", bad, flush=True)
        for k in bad[:5]:
            a, b = sb[k], s1[k]
            if a.shape == b.shape:
                idx = np.nonzero(a.reshape(-1) != b.reshape(-1))[0]
                print("    ", k, a.shape, "n diff", idx.size, "first", idx[:6], a.reshape(-1)[idx[:3]], b.reshape(-1)[idx[:3]])
            else:
                print("    ", k, a.shape, b.shape)
        res.append((sb, s1))
        h1.close()
    hb.close()
    return res

prev = None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    body(3, 1, 30, f"rep{rep} [3-1]")
    cur = body(5, 3, 80, f"rep{rep} [5-3]")
    if prev is not None:
        for s in range(5):
            print(f"rep{rep} slot {s}: batch vs previous batch", diff(prev[s][0], cur[s][0]), " single vs previous single", diff(prev[s][1], cur[s][1]), flush=True)
    prev = cur
print([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "librccl" in l or "libhsa-runtime" in l][::4])
print("done")
