import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "single", d.get("single_stream_scans_per_s"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
if "roofline" in d: print("roofline", d["roofline"])
print("pipeline", d.get("pipeline_roofline"))
for k, v in list(d.get("kernels", {}).items())[:int(sys.argv[1]) if len(sys.argv) > 1 else 24]:
    print("%-28s %9.1f us x%4d %5.1f%%" % (k, v["avg_us"], v["launches"], 100 * v["share"]))
