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


def imu_stream(t0, t1, rate=100.0, yaw_rate=0.3, acc=(0.4, -0.2, 0.0), tilt=(0.02, -0.015)):
    """sensor_msgs/Imu samples [n, 11] (stamp, orientation w x y z, linear_acceleration, angular_velocity) of a platform that turns
    at `yaw_rate` rad/s with a small constant roll / pitch and accelerates by `acc` (body frame, gravity added as an IMU reports it)."""
    n = int(round((t1 - t0) * rate)) + 1
    out = np.zeros((n, 11))
    for i in range(n):
        t = t0 + i / rate
        r, p, y = tilt[0], tilt[1], yaw_rate * t
        cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
        q = (cy * cp * cr + sy * sp * sr, cy * cp * sr - sy * sp * cr, cy * sp * cr + sy * cp * sr, sy * cp * cr - cy * sp * sr)   # Rz Ry Rx
        g = 9.81
        out[i, 0] = t
        out[i, 1:5] = q
        out[i, 5:8] = (acc[0] - g * np.sin(p), acc[1] + g * np.cos(p) * np.sin(r), acc[2] + g * np.cos(p) * np.cos(r))
        out[i, 8:11] = (0.0, 0.0, yaw_rate)
    return out
