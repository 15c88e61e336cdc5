"""Shared helpers for the parity tests."""
import numpy as np

from alego_amd import synth


def bits(a):
    """View float arrays as integer bit patterns for exact comparison (NaN-safe)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a.view(np.uint32)
    if a.dtype == np.float64:
        return a.view(np.uint64)
    return a


def assert_bit_equal(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    bad = np.nonzero(bits(got).reshape(-1) != bits(want).reshape(-1))[0]
    assert bad.size == 0, f"{what}: {bad.size} of {got.size} elements differ, first at {bad[:5]}: got {got.reshape(-1)[bad[:5]]} want {want.reshape(-1)[bad[:5]]}"


def quat_angle(q1, q2):
    """Rotation angle (rad) between two unit quaternions (w,x,y,z)."""
    d = abs(float(np.dot(q1, q2)))
    return 2.0 * np.arccos(min(1.0, d))


def scans(params, n, stream=0, flags=0, start=0):
    return [synth.scan(params, start + k, stream, flags) for k in range(n)]
