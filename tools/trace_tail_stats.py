"""Steady-state per-kernel averages from a rocprofv3 --kernel-trace CSV: for every kernel the LAST n dispatches, n = the
launches bench.py's HIP-event pass counted for it (that pass is the last GPU work of `bench.py --no-cpu`).
usage: trace_tail_stats.py <dir with *_kernel_trace.csv> <bench json under rocprof>  -> CSV on stdout

bench.py keys its kernel table by the label the launch site gives (`ALEGO_LAUNCH(ip_fused_h, ...)`), rocprofv3 by the symbol
(`void ip_fused_t<512, 2>(DevCtx, int, int)`): `resolve` maps one to the other — an alias table for the labels that are
constexpr names of template instances, then the label's own template arguments as a prefix of the symbol's (`lo_assoc<0>` ->
`lo_assoc<0, 160>`), then the bare name when only one instance of it was dispatched (`lo_solve_t<LO_SOLVE_BLOCK, true>` ->
`lo_solve_t<64, true>`).  Round 4's table lost the dominant kernel because it compared bare names only."""
import collections
import csv
import glob
import json
import re
import sys

ALIASES = {"ip_fused_h": "ip_fused_t<512, 2>", "ip_fused_w": "ip_fused_t<1024, 2>", "ip_fused": "ip_fused<1024>"}


def canonical(symbol):
    """`void lo_assoc<1, 160>(DevCtx, int)` -> `lo_assoc<1, 160>` (the trailing parameter list goes, template arguments stay)"""
    s = symbol.strip().strip('"')
    s = re.sub(r"^void\s+", "", s)
    depth, cut = 0, len(s)
    for i in range(len(s) - 1, -1, -1):          # the last top-level "(...)"
        if s[i] == ")":
            depth += 1
        elif s[i] == "(":
            depth -= 1
            if depth == 0:
                cut = i
                break
    return s[:cut].strip() if s.endswith(")") else s


def base(name):
    return name.split("<")[0]


def resolve(label, symbols):
    lab = label.strip("()").strip()
    if lab in symbols:
        return lab
    if lab in ALIASES:
        a = ALIASES[lab]
        if a in symbols:
            return a
        same = [s for s in symbols if base(s) == base(a)]
        if len(same) == 1:
            return same[0]
    if "<" in lab:
        args = lab[lab.index("<") + 1:lab.rindex(">")].replace(" ", "")
        pre = [s for s in symbols if base(s) == base(lab) and "<" in s and
               (s[s.index("<") + 1:s.rindex(">")].replace(" ", "") + ",").startswith(args + ",")]
        if len(pre) == 1:
            return pre[0]
    same = [s for s in symbols if base(s) == base(lab)]
    return same[0] if len(same) == 1 else None


def main():
    d, bj = sys.argv[1], sys.argv[2]
    bench = json.loads(open(bj).read().strip().splitlines()[-1])
    rows = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows[canonical(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    w = csv.writer(sys.stdout)
    w.writerow(["bench_label", "rocprof_kernel", "dispatches_in_tail", "rocprof_avg_us_tail", "hip_event_avg_us", "ratio"])
    out, lost = [], []
    for label, v in bench["kernels"].items():
        sym = resolve(label, rows)
        if sym is None:
            lost.append(label)
            continue
        n = v["launches"]
        t = sorted(rows[sym])[-n:]
        avg = sum(e - s for s, e in t) / len(t) / 1e3
        out.append((avg * len(t), label.strip("()"), sym, len(t), avg, v["avg_us"]))
    for _, label, sym, n, avg, ev in sorted(out, reverse=True):
        w.writerow([label, sym, n, round(avg, 2), ev, round(avg / ev, 3) if ev else ""])
    for label in lost:
        w.writerow([label.strip("()"), "NOT FOUND IN TRACE", "", "", bench["kernels"][label]["avg_us"], ""])


if __name__ == "__main__":
    main()
