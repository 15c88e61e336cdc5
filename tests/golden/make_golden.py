"""Generates tests/golden/oracle_cfgA.npz (16 x 1800, 12 scans) and oracle_geo_16x4000.npz / oracle_geo_64x2048.npz (3 scans each) from the oracle (oracle/liboracle.so) in this container.

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
# (file, n_scan, horizon_scan, scans): the bench geometry and, with a few scans each, the reference's own 16 x 4000 (utility.h:50-55) and config 5's 64 x 2048
SETS = (("oracle_cfgA.npz", 16, 1800, NSCAN), ("oracle_geo_16x4000.npz", 16, 4000, 3), ("oracle_geo_64x2048.npz", 64, 2048, 3))
DIGESTS = ("label_img", "seg_cloud", "seg_col", "seg_range", "seg_ground", "outlier", "less_sharp", "less_flat")
ARRAYS = ("ring_start", "ring_end", "orientation", "sharp_idx", "flat_idx", "lo_params", "odom_pose", "map_pose", "lo_solve_info", "lm_info", "lm_params")


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make(n_scan, horizon, scans):
    p = synth.default_params(n_scan, horizon)
    o = O.Oracle(p)
    out = {}
    for k in range(scans):
        pts = synth.scan(p, k)
        r = o.process_scan(pts)
        out[f"s{k}_in_digest"] = digest(pts)
        out[f"s{k}_ret"] = np.int32(r)
        for name in DIGESTS:
            out[f"s{k}_{name}_digest"] = digest(o.get(name))
        for name in ARRAYS:
            out[f"s{k}_{name}"] = o.get(name)
        if k in (0, 1):  # two full small arrays so that a digest mismatch can be localised
            out[f"s{k}_less_sharp_idx"] = o.get("less_sharp_idx")
            out[f"s{k}_lo_surf_corr"] = o.get("lo_surf_corr")
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for name, n_scan, horizon, scans in SETS:
        out = make(n_scan, horizon, scans)
        np.savez_compressed(os.path.join(here, name), **out)
        print("wrote", name, "with", len(out), "entries")


if __name__ == "__main__":
    main()
