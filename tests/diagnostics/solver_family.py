"""does the arithmetic FAMILY of the 6-DoF linear solve decide where a free run separates from the oracle?

The oracle solves every trust-region step by Householder QR of the stacked (R + 6) x 6 system (Ceres DENSE_QR, laserOdometry.cpp:413-418);
the device forms the normal equations and solves them by Cholesky.  Before building a TSQR solver on the chip, the question is asked on
the CPU, where it costs a flag: ORACLE_SOLVER=normal makes the oracle take the SAME step through the normal equations + Cholesky.
  * oracle(QR) vs oracle(normal): two CPU runs that differ ONLY in the solver family (same summation orders everywhere else);
  * device vs oracle(QR) and device vs oracle(normal) (needs a GPU): does sharing the family with the device lengthen the horizon?
usage: solver_family.py [streams=6] [scans=800] [--device]   -> JSON on stdout (profiles/r03_solver_family.json)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

def worker(mode, stream, scans, device):
    """per-scan map pose (t, q: 7 numbers) of one free run: mode 'qr' | 'normal' | ... (oracle, ORACLE_SOLVER) or 'device'"""
    from alego_loader import load_package; load_package()
    from alego_amd import synth
    p = synth.default_params(16, 1800)
    out = []
    if mode == "device":
        from alego_amd import binding
        h = binding.Handle(p)
        for k in range(scans):
            _, _, mp = h.scan_process(synth.scan(p, k, stream=stream), stages=7)
            out.append(mp["t"].tolist() + mp["q"].tolist())
        h.close()
    else:
        from oracle import oracle_py as O
        o = O.Oracle(p)
        for k in range(scans):
            o.process_scan(synth.scan(p, k, stream=stream))
            out.append(o.get("map_pose")[:7].tolist())
    print(json.dumps(out))

def run(mode, stream, scans):
    env = dict(os.environ)
    if mode not in ("qr", "device"):
        env["ORACLE_SOLVER"] = mode
    else:
        env.pop("ORACLE_SOLVER", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", mode, str(stream), str(scans)], env=env, capture_output=True, text=True, check=True)
    return np.array(json.loads(r.stdout.strip().splitlines()[-1]))

def horizon(a, b, tol=1e-4):
    e = np.linalg.norm(a[:, :3] - b[:, :3], axis=1)
    bad = np.nonzero(e > tol)[0]
    return (int(bad[0]) if bad.size else None), float(e.max()), float(e[-1])

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), False)
        sys.exit(0)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    streams = int(args[0]) if args else 6
    scans = int(args[1]) if len(args) > 1 else 800
    dev = "--device" in sys.argv
    from concurrent.futures import ThreadPoolExecutor
    extra = [a[len("--also="):] for a in sys.argv[1:] if a.startswith("--also=")]   # e.g. --also=normal_ld: further oracle variants against oracle(QR)
    jobs = [(m, s) for s in range(streams) for m in (["qr", "normal"] + extra + (["device"] if dev else []))]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1) if not dev else 4) as ex:
        res = dict(zip(jobs, ex.map(lambda j: run(j[0], j[1], scans), jobs)))
    rows = []
    for s in range(streams):
        row = dict(stream=s)
        h, mx, fin = horizon(res[("qr", s)], res[("normal", s)])
        row["oracle_qr_vs_oracle_normal"] = dict(first_scan_beyond_tol=h, max_m=mx, final_m=fin)
        for m in extra:
            h, mx, fin = horizon(res[("qr", s)], res[(m, s)])
            row[f"oracle_qr_vs_oracle_{m}"] = dict(first_scan_beyond_tol=h, max_m=mx, final_m=fin)
        if dev:
            for m in ("qr", "normal"):
                h, mx, fin = horizon(res[("device", s)], res[(m, s)])
                row[f"device_vs_oracle_{m}"] = dict(first_scan_beyond_tol=h, max_m=mx, final_m=fin)
        rows.append(row)
    print(json.dumps(dict(scans=scans, tolerance_m=1e-4, note="free runs (no teacher forcing), map translation per scan; horizon = first scan whose poses differ by more than 1e-4 m", streams=rows), indent=1))
