"""Diagnostic: the multi-group batch advanced one scan per call, every slot compared with a single-slot handle's record after
every scan (stage outputs of ImageProjection, feature extraction, LaserOdometry, LaserMapping): names the first scan / slot /
array a timing-dependent result shows up in."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
p = synth.default_params(16, 1800)
nslot, nscan = 5, 80
NAMES = ("scal", "seg_cloud", "outlier", "less_sharp", "less_flat", "sharp", "flat", "lo_state", "poses", "lm_info", "lm_surf_map_ds", "lm_corner_map_ds",
         "lm_surf_total_ds", "lm_corner_ds", "lm_keyposes")
scans = [[synth.scan(p, k, stream=s) for k in range(nscan)] for s in range(nslot)]
def snap(h, slot):
    return {n: h.debug_get(n, slot=slot, cap_bytes=1 << 24).copy() for n in NAMES}
ref = []
for s in range(nslot):
    h1 = binding.Handle(p)
    rec = []
    for k in range(nscan):
        h1.scan_process(scans[s][k], stages=7)
        rec.append(snap(h1, 0))
    ref.append(rec)
    h1.close()
print("reference recorded", flush=True)
per_scan = os.environ.get("PROBE_PER_SCAN", "1") == "1"
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    os.environ["ALEGO_STREAM_GROUPS"] = os.environ.get("PROBE_GROUPS", "3")
    hb = binding.Handle(p, n_slots=nslot, ring_len=nscan)
    for s in range(nslot):
        for k in range(nscan):
            hb.batch_load(s, k, scans[s][k])
    found = False
    if per_scan:
        for k in range(nscan):
            hb.batch_run(k, 1, stages=7)
            for s in range(nslot):
                cur = snap(hb, s)
                bad = [n for n in NAMES if cur[n].shape != ref[s][k][n].shape or not np.array_equal(cur[n].view(np.uint8), ref[s][k][n].view(np.uint8))]
                if bad:
                    found = True
                    print(f"rep {rep} scan {k} slot {s}: differ {bad}", flush=True)
                    for n in bad[:6]:
                        a, b = cur[n], ref[s][k][n]
                        if a.shape == b.shape:
                            idx = np.nonzero(a.reshape(-1) != b.reshape(-1))[0]
                            print("    ", n, a.shape, "n diff", idx.size, "first", idx[:8], a.reshape(-1)[idx[:4]], b.reshape(-1)[idx[:4]])
                        else:
                            print("    ", n, a.shape, b.shape)
            if found:
                break
    else:
        hb.batch_run(0, nscan, stages=7)
        for s in range(nslot):
            cur = snap(hb, s)
            bad = [n for n in NAMES if cur[n].shape != ref[s][-1][n].shape or not np.array_equal(cur[n].view(np.uint8), ref[s][-1][n].view(np.uint8))]
            if bad:
                print(f"rep {rep} slot {s}: differ {bad}", flush=True)
    hb.close()
print("done")
