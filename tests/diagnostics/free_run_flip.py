"""Names the discrete decision at which a free-running device stream leaves the oracle.

Both sides process the same scans with NO teacher forcing.  Per scan every discrete output is compared (segmentation,
feature index lists, LO correspondences, LO / LM solver summaries, LM accepted query lists, key-frame decision) next to
the continuous state entering the scan (LO params_, LM params_).  The first scan at which a discrete output differs is
reported with the margin of the decision that flipped, evaluated under BOTH sides' entering poses:
  * LO correspondence: squared distances (f32 1-NN / f64 walk) of the two competing target points from the query
    transformed by the device's and by the oracle's params_
  * solver summary: the costs and the step-acceptance ratio's inputs of both sides
Writes gpurun_out/free_run_flip.json (copied to profiles/ by hand)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from alego_loader import load_package; load_package()
from alego_amd import binding, synth
from oracle import oracle_py as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 620
p = synth.default_params(16, 1800)
h, o = binding.Handle(p), O.Oracle(p)
report = dict(scans=n, events=[], err_track=[])


def unpack(v):
    return (int(v) & 0xFF, (int(v) >> 8) & 0xFF, int(v) >> 16)


def d2_f32(a, q):
    r = np.float32(0)
    for k in range(3):
        df = np.float32(a[k]) - np.float32(q[k])
        r = np.float32(r + df * df)
    return float(r)


def d2_f64(a, q):
    return float(sum(float(np.float32(a[k]) - np.float32(q[k])) ** 2 for k in range(3)))


first = None
prev_lo = (np.zeros(6), np.zeros(6))
prev_lm = (np.zeros(6), np.zeros(6))
for k in range(n):
    pts = synth.scan(p, k)
    lo_in = (h.debug_get("lo_state")[0:6].copy(), o.get("lo_params").copy())
    lm_in = (h.debug_get("lm_state")[0:6].copy(), o.get("lm_params").copy())
    surf_last_o = o.get("surf_last") if k else None
    corner_last_o = o.get("corner_last") if k else None
    o.process_scan(pts)
    flags, odom, mp = h.scan_process(pts, stages=7)
    if k == 0:
        continue
    want = o.get("map_pose")
    e_map = float(np.abs(mp["t"] - want[:3]).max())
    e_lo = float(np.abs(lo_in[0] - lo_in[1]).max())
    e_lm = float(np.abs(lm_in[0] - lm_in[1]).max())
    report["err_track"].append([k, e_map, e_lo, e_lm])
    ev = []
    m = o.get("seg_cloud").shape[0]
    for name in ("seg_col", "seg_ground", "sharp_idx", "less_sharp_idx", "flat_idx"):
        g, w = h.debug_get(name), o.get(name)
        if g.shape != w.shape or not np.array_equal(g, w):
            ev.append(dict(kind="index list", name=name))
    for nm, width, flat_name, last in (("lo_surf_corr", 4, "flat", surf_last_o), ("lo_corner_corr", 3, "sharp", corner_last_o)):
        oc = o.get(nm).reshape(-1, width)
        gc = h.debug_get(nm).reshape(-1, 4)
        gc = gc[gc[:, 1] >= 0][:, :width]
        if gc.shape != oc.shape or not np.array_equal(gc, oc):
            d = dict(kind="LO correspondence", name=nm, device_rows=int(gc.shape[0]), oracle_rows=int(oc.shape[0]), entering_lo_params_diff=e_lo)
            # first differing query: margins under both poses
            gq, oq = {int(r[0]): r for r in gc}, {int(r[0]): r for r in oc}
            for q in sorted(set(gq) | set(oq)):
                a, b = gq.get(q), oq.get(q)
                if a is None or b is None or not np.array_equal(a, b):
                    feat = o.get(flat_name)[q]
                    detail = dict(query=q, device=None if a is None else a.tolist(), oracle=None if b is None else b.tolist())
                    for side, prm in (("device_pose", lo_in[0]), ("oracle_pose", lo_in[1])):
                        sel = O.transform_to_start(prm, feat[None, :])[0]
                        cand = sorted({int(x) for r in (a, b) if r is not None for x in r[1:] if x >= 0})
                        detail[side] = {str(c): dict(d2_f32=d2_f32(last[c], sel), d2_f64=d2_f64(last[c], sel)) for c in cand}
                    d["first_differing_query"] = detail
                    break
            ev.append(d)
    sc = h.debug_get("scal")
    info = o.get("lo_solve_info")
    if (unpack(sc[10]), unpack(sc[11])) != (tuple(int(x) for x in info[0:3]), tuple(int(x) for x in info[3:6])):
        ev.append(dict(kind="LO solver summary (iterations, successful, termination)", device=[unpack(sc[10]), unpack(sc[11])],
                       oracle=[[int(x) for x in info[0:3]], [int(x) for x in info[3:6]]],
                       device_costs=h.debug_get("lo_state")[24:28].tolist(), oracle_costs=o.get("lo_costs").tolist(), entering_lo_params_diff=e_lo))
    gi, oi = h.debug_get("lm_info"), o.get("lm_info")
    if bool(gi[2]) != bool(oi[0]):
        ev.append(dict(kind="LM ran", device=int(gi[2]), oracle=int(oi[0])))
    elif bool(gi[2]):
        if bool(gi[11]) and bool(oi[1]):
            blocks = h.debug_get("lm_blocks").reshape(-1, 8)
            qc = np.nonzero(blocks[:gi[19], 7] != 0)[0]
            kf_cap_c = 120 * p.n_scan
            qs = np.nonzero(blocks[kf_cap_c:kf_cap_c + gi[23], 7] != 0)[0]
            for nm, g, w in (("lm_corner_corr_q", qc, o.get("lm_corner_corr_q")), ("lm_surf_corr_q", qs, o.get("lm_surf_corr_q"))):
                if g.shape != w.shape or not np.array_equal(g, w):
                    sd = sorted(set(g.tolist()) ^ set(w.tolist()))
                    ev.append(dict(kind="LM accepted queries", name=nm, device_n=int(g.size), oracle_n=int(w.size), symmetric_difference=sd[:8],
                                   entering_map_pose_diff=e_map))
            g8 = (unpack(gi[8]), unpack(gi[9]))
            o8 = (tuple(int(x) for x in oi[5:8]), tuple(int(x) for x in oi[8:11]))
            if g8 != o8:
                ev.append(dict(kind="LM solver summary (iterations, successful, termination)", device=g8, oracle=o8,
                               device_costs=h.debug_get("lm_state")[39:43].tolist(), entering_lm_params_diff=e_lm))
        if bool(gi[10]) != bool(oi[2]):
            ev.append(dict(kind="key frame decision", device=int(gi[10]), oracle=int(oi[2])))
        for name in ("lm_corner_map_ds", "lm_surf_map_ds"):
            g, w = h.debug_get(name), o.get(name)
            if g.shape != w.shape:
                ev.append(dict(kind="map size", name=name, device=int(g.shape[0]), oracle=int(w.shape[0])))
    if ev and len(report["events"]) < 12:
        report["events"].append(dict(scan=k, map_err_m=e_map, entering_lo_params_diff=e_lo, entering_lm_params_diff=e_lm, differences=ev))
        if first is None:
            first = k
            print(f"first discrete difference at scan {k}: {json.dumps(ev)[:1500]}")
report["first_discrete_difference_scan"] = first
tr = np.array(report["err_track"])
report["max_map_err_before_first_difference"] = float(tr[tr[:, 0] < (first or n), 1].max()) if len(tr) else None
report["max_map_err"] = float(tr[:, 1].max())
report["err_track"] = [r for r in report["err_track"] if r[0] % 20 == 0 or (first and abs(r[0] - first) < 6)]
os.makedirs("gpurun_out", exist_ok=True)
json.dump(report, open("gpurun_out/free_run_flip.json", "w"), indent=1)
print("first discrete difference:", first, "max map err before:", report["max_map_err_before_first_difference"], "overall:", report["max_map_err"])
