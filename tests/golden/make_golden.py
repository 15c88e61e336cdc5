"""Generates tests/golden/oracle_cfgA.npz from the oracle (oracle/liboracle.so) in this container.

The reference repository has no tests, fixtures or golden vectors and cannot be built
here, so these vectors pin the ORACLE against regressions (parity unpinned w.r.t. the reference itself).
Inputs are the deterministic synthetic scans of a-lego-loam_amd/csrc/synth.cpp; their digest is stored too.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from alego_loader import load_package  # noqa: E402

load_package()
from alego_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

NSCAN = 12


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    p = synth.default_params(16, 1800)
    o = O.Oracle(p)
    out = {}
    for k in range(NSCAN):
        pts = synth.scan(p, k)
        r = o.process_scan(pts)
        out[f"s{k}_in_digest"] = digest(pts)
        out[f"s{k}_ret"] = np.int32(r)
        for name in ("label_img", "seg_cloud", "seg_col", "seg_range", "seg_ground", "outlier", "less_sharp", "less_flat"):
            out[f"s{k}_{name}_digest"] = digest(o.get(name))
        for name in ("ring_start", "ring_end", "orientation", "sharp_idx", "flat_idx", "lo_params", "odom_pose", "map_pose",
                     "lo_solve_info", "lm_info", "lm_params"):
            out[f"s{k}_{name}"] = o.get(name)
        if k in (0, 1):  # two full small arrays so that a digest mismatch can be localised
            out[f"s{k}_less_sharp_idx"] = o.get("less_sharp_idx")
            out[f"s{k}_lo_surf_corr"] = o.get("lo_surf_corr")
    # a tiny hand-checkable scan: 3 rings x 64 columns of a plane wall + ground, reference geometry parameters
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_cfgA.npz"), **out)
    print("wrote oracle_cfgA.npz with", len(out), "entries")


if __name__ == "__main__":
    main()
