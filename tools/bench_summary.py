import json, sys
src = open(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].isdigit() else sys.stdin
d = json.loads(src.read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "single", d.get("single_stream_scans_per_s"), "serial", d.get("single_stream_serial_scans_per_s"), d.get("single_stream_note"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
if "roofline" in d: print("roofline", d["roofline"])
print("pipeline", d.get("pipeline_roofline"))
print("counts", d.get("counts"))
cb = d.get("cpu_baseline") or {}
print("cpu_pipe3", cb.get("cpu_pipe3"), "cpu_replicas", cb.get("cpu_replicas"))
print("parity", d.get("parity"))
iso = d.get("kernels_isolated", {})
for k, v in list(d.get("kernels", {}).items())[:40]:
    print("%-28s %9.1f us x%4d %5.1f%%   alone %8.1f us" % (k, v["avg_us"], v["launches"], 100 * v["share"], iso.get(k, float("nan"))))
