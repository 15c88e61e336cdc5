"""-m gpu: the free-run contract.

Per scan — every scan started from the oracle's state — the device is within 1e-4 m / 1e-4 rad of the oracle (the teacher-forced tests).  A FREE
run is only exact within the solver family: the oracle takes every trust-region step by Householder QR of the stacked system (Ceres DENSE_QR), the
device through the normal equations + Cholesky; the two round differently, and on one of six streams that flips one discrete decision at scan 425,
after which the trajectories are 7 mm apart (profiles/r03_solver_family.json).  Until round 4 that explanation lived in a diagnostic script; this is
the test:

  * device vs ORACLE_SOLVER=normal (the oracle with the device's solver family): every scan of 800 on six streams within 1e-8 m / 1e-8 rad;
  * device vs the default (QR) oracle: within 1e-4 m / 1e-4 rad up to the recorded first flip of the stream, and the flip is where it was recorded —
    a regression that separates the device from the reference EARLIER turns this red.

ORACLE_SOLVER is read once per process (a static in oracle.cpp), so the oracle runs are subprocesses of tests/diagnostics/solver_family.py."""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from alego_amd import binding, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "diagnostics", "solver_family.py")
STREAMS, SCANS = 6, 800
FIRST_FLIP = {0: 425}      # stream -> first scan beyond 1e-4 m against the QR oracle (profiles/r03_solver_family.json); the other streams have none in 800 scans


def _oracle_run(mode, stream):
    env = dict(os.environ)
    env.pop("ORACLE_SOLVER", None)
    if mode != "qr":
        env["ORACLE_SOLVER"] = mode
    r = subprocess.run([sys.executable, SCRIPT, "--worker", mode, str(stream), str(SCANS)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    return np.array(json.loads(r.stdout.strip().splitlines()[-1]))


def _rot_angle(q1, q2):
    """angle of q1^-1 q2 (w x y z) from the VECTOR part of the relative quaternion: 2 atan2(|v|, |w|) resolves 1e-16 rad, where the usual
    2 acos(|q1 . q2|) has a floor of ~1e-8 (acos near 1)"""
    w1, v1, w2, v2 = q1[0], q1[1:4], q2[0], q2[1:4]
    w = w1 * w2 + float(np.dot(v1, v2))
    v = w1 * v2 - w2 * v1 - np.cross(v1, v2)
    return 2.0 * float(np.arctan2(np.linalg.norm(v), abs(w)))


def _errors(a, b):
    et = np.linalg.norm(a[:, :3] - b[:, :3], axis=1)
    er = np.array([_rot_angle(x[3:7], y[3:7]) for x, y in zip(a, b)])
    return et, er


def test_free_run_tracks_the_oracle_of_its_solver_family():
    p = synth.default_params(16, 1800)
    jobs = [(m, s) for s in range(STREAMS) for m in ("qr", "normal")]
    with ThreadPoolExecutor(max_workers=min(12, os.cpu_count() or 1)) as ex:
        fut = {j: ex.submit(_oracle_run, *j) for j in jobs}
        dev = {}
        for s in range(STREAMS):      # the device runs while the oracles do
            h = binding.Handle(p)
            out = []
            for k in range(SCANS):
                _, _, mp = h.scan_process(synth.scan(p, k, stream=s), stages=7)
                out.append(np.concatenate([mp["t"], mp["q"]]))
            h.close()
            dev[s] = np.array(out)
        ora = {j: f.result() for j, f in fut.items()}
    report = {}
    for s in range(STREAMS):
        et, er = _errors(dev[s], ora[("normal", s)])
        assert et.max() <= 1e-8 and er.max() <= 1e-8, f"stream {s}: device vs oracle(normal equations) {et.max():.3e} m / {er.max():.3e} rad at scan {int(et.argmax())}"
        eq, rq = _errors(dev[s], ora[("qr", s)])
        beyond = np.nonzero((eq > 1e-4) | (rq > 1e-4))[0]
        first = int(beyond[0]) if beyond.size else None
        assert first == FIRST_FLIP.get(s), f"stream {s}: first scan beyond 1e-4 against the QR oracle is {first}, recorded {FIRST_FLIP.get(s)}"
        upto = first if first is not None else SCANS
        assert eq[:upto].max() <= 1e-4 and rq[:upto].max() <= 1e-4
        # the two oracles separate where the device separates from the QR one: the flip belongs to the solver family, not to the device
        eo, _ = _errors(ora[("qr", s)], ora[("normal", s)])
        bo = np.nonzero(eo > 1e-4)[0]
        assert (int(bo[0]) if bo.size else None) == first
        report[s] = dict(vs_normal_max_m=float(et.max()), vs_normal_max_rad=float(er.max()), vs_qr_first_beyond=first, vs_qr_max_m_before=float(eq[:upto].max()))
    print(json.dumps(report))
