"""Aggregate a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel over its last `tail` dispatches."""
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
for k, cs in vals.items():
    out[k] = {c: sum(v[-tail:]) / len(v[-tail:]) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(v) for v in cs.values())
json.dump(out, sys.stdout, indent=1)
