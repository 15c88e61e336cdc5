"""Steady-state per-kernel averages from a rocprofv3 --kernel-trace CSV: for every kernel the LAST n dispatches, n = the
launches bench.py's HIP-event pass counted for it (that pass is the last GPU work of `bench.py --no-cpu`).
usage: trace_tail_stats.py <dir with *_kernel_trace.csv> <bench json under rocprof>  -> CSV on stdout"""
import csv, glob, json, re, sys, collections
d, bj = sys.argv[1], sys.argv[2]
bench = json.loads(open(bj).read().strip().splitlines()[-1])
launches = {k.split("<")[0].strip("()"): v["launches"] for k, v in bench["kernels"].items()}
ev = {k.split("<")[0].strip("()"): v["avg_us"] for k, v in bench["kernels"].items()}
rows = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip().split("<")[0]
            rows[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches_in_tail", "rocprof_avg_us_tail", "hip_event_avg_us", "ratio"])
out = []
for k, n in launches.items():
    if k not in rows:
        continue
    t = sorted(rows[k])[-n:]
    avg = sum(e - s for s, e in t) / len(t) / 1e3
    out.append((avg * len(t), k, len(t), avg))
for _, k, n, avg in sorted(out, reverse=True):
    w.writerow([k, n, round(avg, 2), ev[k], round(avg / ev[k], 3) if ev[k] else ""])
