"""Aggregate a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel over its last `tail` dispatches.
usage: pmc_agg.py <dir> [tail=40] [streams per launch: adds VALU / SALU k-instructions per stream]"""
import csv, glob, sys, collections, json, re
d = sys.argv[1]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 40
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
XCDS = 8   # GRBM_GUI_ACTIVE comes back summed over the eight XCDs of an MI355X; the SQ counters are chip totals in units of 4 cycles (one quad-cycle per
           # wavefront instruction: SQ_ACTIVE_INST_VALU = SQ_INSTS_VALU for plain f32 / f64 / integer code, more where quarter-rate instructions are)
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for k, cs in vals.items():
    out[k] = {c: sum(v[-tail:]) / len(v[-tail:]) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(v) for v in cs.values())
    o = out[k]
    if o.get("GRBM_GUI_ACTIVE") and o.get("SQ_ACTIVE_INST_VALU") is not None:
        # the kernel alone on the chip: share of the 1024 SIMDs' issue cycles that carried a VALU instruction (from the counter, not count x 4)
        o["valu_busy_alone"] = round(o["SQ_ACTIVE_INST_VALU"] * 4.0 / (o["GRBM_GUI_ACTIVE"] / XCDS * 1024.0), 4)
    if streams and o.get("SQ_INSTS_VALU") is not None:
        o["valu_kinst_per_stream"] = round(o["SQ_INSTS_VALU"] / streams / 1e3, 2)
        if o.get("SQ_INSTS_SALU") is not None:
            o["salu_kinst_per_stream"] = round(o["SQ_INSTS_SALU"] / streams / 1e3, 2)
json.dump(out, sys.stdout, indent=1)
