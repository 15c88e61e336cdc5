"""Marginal cost of each (idempotent) kernel inside the concurrent pipeline: bench throughput with the kernel launched twice.
Needs a library built with -DALEGO_DUP_HOOK (every .hip file).  usage: python tools/marginal_cost.py [streams] [steps]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
streams = sys.argv[1] if len(sys.argv) > 1 else "1536"
steps = sys.argv[2] if len(sys.argv) > 2 else "60"
# (round 5: the current kernel set.  lm_solve runs a second time from the optimum the first run found — one evaluation, a lower bound.  lo_solve_t cannot be
#  doubled: its second phase-1 run integrates the pose twice, the trajectories leave the map and LaserMapping reports a capacity error; map_update neither:
#  its second run would take "one run out, one run in" from a list that has already changed.)
kernels = (os.environ.get("ALEGO_DUP_LIST") or ",ip_fused_h,fe_cand,fe_pickc,fe_ring_out,lo_grid_build,lo_assoc<0>,lo_assoc<1>,vox_small,"
           "lm_grid_build,lm_knn,lm_fit,map_accum,lm_solve,lm_stage").split(",")
base = None
for k in kernels:
    env = dict(os.environ, ALEGO_DUP=k)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--streams", streams, "--steps", steps, "--warmup", "10",
                          "--prime", "560", "--no-cpu", "--no-profile", "--no-check", "--no-isolated"], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(k, "FAILED", out.stderr[-400:]); continue
    j = json.loads(line[-1])
    ms = j["ms_per_step"]
    if base is None: base = ms
    print(f"{k or 'baseline':14s} {j['value']:10.0f} scans/s  {ms:8.3f} ms/step  marginal {100 * (ms - base) / base:6.2f} %", flush=True)
