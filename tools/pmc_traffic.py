"""Build profiles/r01_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: pmc_traffic.py <dir_fetch> <dir_write> <streams_per_launch> [tail] [scan_launches]
HBM bytes per launch = mean over the last `tail` dispatches of each kernel (steady state: the local map is full).
Units as prescribed by MI355X_MICROARCH.md (HBM / rocprofv3 section): on gfx950 FETCH_SIZE counts 64 B per request
where the requests are 128 B wide, so it is doubled; WRITE_SIZE is reported in KB like FETCH_SIZE and left uncorrected.
"""
import csv, glob, json, re, sys, collections

def load(d, counter):
    vals = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                full = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
                k = full.split("<")[0]
                if k == "ip_fused_t":   # the two instantiations under the names bench.py's HIP-event table uses
                    k = "ip_fused_h" if full.startswith("ip_fused_t<512") else "ip_fused_w"
                vals[k].append(float(r["Counter_Value"]))
    return vals

fd, wd, spl = sys.argv[1], sys.argv[2], int(sys.argv[3])
tail = int(sys.argv[4]) if len(sys.argv) > 4 else 16
scan_launches = int(sys.argv[5]) if len(sys.argv) > 5 else 0   # (steps of the run) x (stream groups): launches of a once-per-scan kernel
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
out = {"note": "steady-state mean per launch; FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md HBM section (gfx950 counts 64 B per 128 B request), "
               "WRITE_SIZE (KB) uncorrected; kernels serialised by the counter collection",
       "streams_per_launch": spl, "kernels": {}}
for k in sorted(set(F) | set(W)):
    f = F.get(k, [0.0])[-tail:]
    w = W.get(k, [0.0])[-tail:]
    fb, wb = 2.0 * 1024.0 * sum(f) / len(f), 1024.0 * sum(w) / len(w)
    out["kernels"][k] = {"fetch_bytes": round(fb), "write_bytes": round(wb), "hbm_bytes_per_launch": round(fb + wb), "dispatches": len(F.get(k, []))}
if not scan_launches:   # launches of a once-per-scan kernel: the projection kernel (lo_assoc / lo_solve_t are launched twice per scan)
    once = [v["dispatches"] for k, v in out["kernels"].items() if k in ("ip_fused", "ip_fused_h", "ip_fused_w", "ip_project")]
    scan_launches = max(once) if once else max(v["dispatches"] for k, v in out["kernels"].items() if not k.startswith("__amd"))
# whole pipeline per scan of one stream: every kernel's steady-state bytes per launch x its launches per scan / streams per launch
cor = unc = 0.0
for k, v in out["kernels"].items():
    if k.startswith("__amd"):
        continue
    w = v["dispatches"] / scan_launches / spl
    cor += v["hbm_bytes_per_launch"] * w
    unc += (v["fetch_bytes"] / 2.0 + v["write_bytes"]) * w
    v["bytes_per_scan"] = round(v["hbm_bytes_per_launch"] * w)
out["scan_launches"] = scan_launches
out["hbm_bytes_per_scan"] = round(cor)
out["hbm_bytes_per_scan_uncorrected"] = round(unc)
json.dump(out, sys.stdout, indent=1)
