"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_DT = {0: np.float32, 1: np.float64, 2: np.int32, 3: np.uint8}


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing — build with `make -C oracle`")
        L = C.CDLL(path)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_void_p]
        L.oracle_destroy.argtypes = [C.c_void_p]
        for f in ("oracle_lo", "oracle_fe", "oracle_lm"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p]
        L.oracle_ip.restype = C.c_int
        L.oracle_ip.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_process_scan.restype = C.c_int
        L.oracle_process_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.oracle_set_lo_params.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_set_lm_params.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_get.restype = C.c_int
        L.oracle_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_push_imu.restype = None
        L.oracle_push_imu.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_set_scan_time.restype = None
        L.oracle_set_scan_time.argtypes = [C.c_void_p, C.c_double]
        L.oracle_std_sort_order.restype = None
        L.oracle_std_sort_order.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_voxel_grid.restype = C.c_int
        L.oracle_voxel_grid.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int]
        L.oracle_eval_block.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_solve.restype = C.c_int
        L.oracle_solve.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p]
        L.oracle_knn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_eig3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_colpiv_qr_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_colpiv_qr_solve.restype = C.c_int
        for f in ("oracle_atan2f_array", "oracle_hypotf_array", "oracle_libm_atan2f_array", "oracle_libm_hypotf_array"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_lm_set_keypose.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_lm_reset_window.argtypes = [C.c_void_p]
        L.oracle_lm_apply_correction.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_lm_add_keyframe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.oracle_lm_process.restype = C.c_int
        L.oracle_lm_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_lm_keyframe.restype = C.c_int
        L.oracle_lm_keyframe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        L.oracle_transform_to_start.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_transform_cloud.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_run_pipelined.restype = C.c_double
        L.oracle_run_pipelined.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_loop_detect.restype = C.c_int
        L.oracle_loop_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_loop_icp.restype = C.c_int
        L.oracle_loop_icp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_normal_eq.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p]
        L.oracle_sincosf_array.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _lib = L
    return _lib


class Oracle:
    """One IP -> LO -> LM chain of the CPU restatement (stateful like the three nodelets)."""

    def __init__(self, params):
        self._p = params
        self._h = lib().oracle_create(C.addressof(params))

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @staticmethod
    def _pts(pts):
        a = np.ascontiguousarray(pts, dtype=np.float32)
        assert a.ndim == 2 and a.shape[1] == 4
        return a

    def ip(self, pts):
        a = self._pts(pts)
        return lib().oracle_ip(self._h, a.ctypes.data, a.shape[0])

    def fe(self):
        return lib().oracle_fe(self._h)

    def push_imu(self, samples):
        """samples[n, 11]: stamp, orientation w x y z, linear_acceleration xyz, angular_velocity xyz (imuHandler)"""
        a = np.ascontiguousarray(samples, np.float64).reshape(-1, 11)
        lib().oracle_push_imu(self._h, a.ctypes.data, a.shape[0])

    def set_scan_time(self, t):
        """stamp of the next segmented cloud (t1 of LaserOdometry::mainLoop), read by adjustDistortion"""
        lib().oracle_set_scan_time(self._h, float(t))

    def lo(self):
        return lib().oracle_lo(self._h)

    def lm(self):
        return lib().oracle_lm(self._h)

    def lm_process(self, corner_last, surf_last, outlier, odom7):
        """LaserMapping on host clouds + an /odom/lidar pose (t xyz, q wxyz): the checker's counterpart of alego_lm_process."""
        c, s, o = self._pts(corner_last), self._pts(surf_last), self._pts(outlier)
        od = np.ascontiguousarray(odom7, dtype=np.float64)
        return lib().oracle_lm_process(self._h, c.ctypes.data, c.shape[0], s.ctypes.data, s.shape[0], o.ctypes.data, o.shape[0], od.ctypes.data)

    def process_scan(self, pts, stages=7):
        a = self._pts(pts)
        return lib().oracle_process_scan(self._h, a.ctypes.data, a.shape[0], stages)

    def run_pipelined(self, scans):
        """cpu_pipe3: IP, LO, LM as three threads over the list of scans; returns wall-clock seconds."""
        arrs = [self._pts(a) for a in scans]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        cnts = (C.c_int * len(arrs))(*[a.shape[0] for a in arrs])
        return float(lib().oracle_run_pipelined(self._h, ptrs, cnts, len(arrs)))

    def set_lo_params(self, p6):
        a = np.ascontiguousarray(p6, dtype=np.float64)
        lib().oracle_set_lo_params(self._h, a.ctypes.data)

    def set_lm_params(self, p6):
        a = np.ascontiguousarray(p6, dtype=np.float64)
        lib().oracle_set_lm_params(self._h, a.ctypes.data)

    # ---- host pose-graph pass-through (correctPoses, laserMapping.cpp:561-584) ----
    def lm_set_keypose(self, kf_id, pose6):
        a = np.ascontiguousarray(pose6, dtype=np.float32)
        lib().oracle_lm_set_keypose(self._h, kf_id, a.ctypes.data)

    def lm_reset_window(self):
        lib().oracle_lm_reset_window(self._h)

    def lm_apply_correction(self, rc12):
        a = np.ascontiguousarray(rc12, dtype=np.float64).reshape(12)
        lib().oracle_lm_apply_correction(self._h, a.ctypes.data)

    def lm_add_keyframe(self, pose6, corner, surf, outlier):
        a = np.ascontiguousarray(pose6, dtype=np.float32)
        c, s, o = self._pts(corner), self._pts(surf), self._pts(outlier)
        lib().oracle_lm_add_keyframe(self._h, a.ctypes.data, c.ctypes.data, c.shape[0], s.ctypes.data, s.shape[0], o.ctypes.data, o.shape[0])

    def lm_keyframe(self, kf_id):
        """(corner, surf, outlier) clouds of key frame kf_id as saveKeyFramesAndFactor stored them."""
        out = []
        for kind in range(3):
            ptr, n = C.c_void_p(), C.c_int()
            if lib().oracle_lm_keyframe(self._h, kf_id, kind, C.byref(ptr), C.byref(n)) != 0:
                raise KeyError(kf_id)
            if n.value == 0 or not ptr.value:
                out.append(np.zeros((0, 4), np.float32))
            else:
                buf = (C.c_char * (n.value * 16)).from_address(ptr.value)
                out.append(np.frombuffer(buf, dtype=np.float32, count=n.value * 4).copy().reshape(-1, 4))
        return out

    def get(self, name, cloud=None):
        ptr, cnt, dt = C.c_void_p(), C.c_int(), C.c_int()
        r = lib().oracle_get(self._h, name.encode(), C.byref(ptr), C.byref(cnt), C.byref(dt))
        if r != 0:
            raise KeyError(name)
        n = cnt.value
        dtype = np.dtype(_DT[dt.value])
        if n == 0 or not ptr.value:
            out = np.zeros(0, dtype=dtype)
        else:
            buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr.value)
            out = np.frombuffer(buf, dtype=dtype, count=n).copy()
        if cloud is None:
            cloud = name in _CLOUDS
        return out.reshape(-1, 4) if cloud else out


_CLOUDS = {"seg_cloud", "undistorted", "outlier", "sharp", "less_sharp", "flat", "less_flat", "surf_last", "corner_last",
           "lm_corner_map_ds", "lm_surf_map_ds", "lm_corner_map", "lm_surf_map", "lm_corner_ds", "lm_surf_ds",
           "lm_outlier_ds", "lm_surf_total_ds"}


def std_sort_order(keys, depth_limit=-1):
    """libstdc++ std::sort of 0..n-1 by `keys[a] < keys[b]` (laserOdometry.cpp:185's kind of comparator)"""
    k = np.ascontiguousarray(keys, np.uint32)
    out = np.empty(k.size, np.int32)
    lib().oracle_std_sort_order(k.ctypes.data, k.size, depth_limit, out.ctypes.data)
    return out


def voxel_grid(pts, leaf, sort_mode=0):
    a = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.empty_like(a)
    n = lib().oracle_voxel_grid(a.ctypes.data, a.shape[0], leaf, sort_mode, out.ctypes.data, a.shape[0])
    return out[:n].copy()


def eval_block(btype, geom13, params6):
    g = np.ascontiguousarray(geom13, dtype=np.float64)
    p = np.ascontiguousarray(params6, dtype=np.float64)
    r = np.zeros(1)
    J = np.zeros(6)
    lib().oracle_eval_block(btype, g.ctypes.data, p.ctypes.data, r.ctypes.data, J.ctypes.data)
    return r[0], J


def solve(blocks14, params6, max_iter, huber=0.1):
    b = np.ascontiguousarray(blocks14, dtype=np.float64)
    p = np.array(params6, dtype=np.float64)
    costs = np.zeros(2)
    code = lib().oracle_solve(b.ctypes.data, b.shape[0], p.ctypes.data, max_iter, huber, costs.ctypes.data)
    return p, dict(iterations=code & 0xFF, termination=(code >> 8) & 0xFF, successful=code >> 16,
                   initial_cost=costs[0], final_cost=costs[1])


def knn(cloud, queries, k):
    c = np.ascontiguousarray(cloud, dtype=np.float32)
    q = np.ascontiguousarray(queries, dtype=np.float32)
    idx = np.empty((q.shape[0], k), dtype=np.int32)
    dist = np.empty((q.shape[0], k), dtype=np.float32)
    lib().oracle_knn(c.ctypes.data, c.shape[0], q.ctypes.data, q.shape[0], k, idx.ctypes.data, dist.ctypes.data)
    return idx, dist


def transform_to_start(params6, pts):
    p = np.ascontiguousarray(params6, dtype=np.float64)
    a = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.empty_like(a)
    lib().oracle_transform_to_start(p.ctypes.data, a.ctypes.data, a.shape[0], out.ctypes.data)
    return out


def transform_cloud(pose6, pts):
    p = np.ascontiguousarray(pose6, dtype=np.float32)
    a = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.empty_like(a)
    lib().oracle_transform_cloud(p.ctypes.data, a.ctypes.data, a.shape[0], out.ctypes.data)
    return out


def sincosf(x, mode):
    """mode 0: restated sinf, 1: restated cosf, 2 / 3: this host's libm"""
    a = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(a)
    lib().oracle_sincosf_array(a.ctypes.data, out.ctypes.data, a.size, mode)
    return out


def normal_eq(blocks14, params6, huber=0.1):
    """28 scalars (upper triangle of J^T J, J^T r, cost) of the given residual blocks at params6."""
    b = np.ascontiguousarray(blocks14, dtype=np.float64).reshape(-1, 14)
    p = np.ascontiguousarray(params6, dtype=np.float64)
    out = np.zeros(28)
    lib().oracle_normal_eq(b.ctypes.data, b.shape[0], p.ctypes.data, huber, out.ctypes.data)
    return out


def loop_detect(params, keyposes6, times, cur_xyz):
    """detectLoopClosure's choice of closest_history_frame_id_ (or -1)."""
    kp = np.ascontiguousarray(keyposes6, dtype=np.float32).reshape(-1, 6)
    t = np.ascontiguousarray(times, dtype=np.float64)
    c = np.ascontiguousarray(cur_xyz, dtype=np.float64)
    return int(lib().oracle_loop_detect(C.addressof(params), kp.ctypes.data, t.ctypes.data, kp.shape[0], c.ctypes.data))


def pack_frames(frames):
    """frames = [(pose6, corner, surf, outlier), ...] -> (poses6, points, offsets) as the loop-closure entry points take them"""
    poses = np.ascontiguousarray([f[0] for f in frames], dtype=np.float32).reshape(-1, 6)
    clouds, offs = [], [0]
    for f in frames:
        for c in f[1:4]:
            c = np.ascontiguousarray(c, dtype=np.float32).reshape(-1, 4)
            clouds.append(c)
            offs.append(offs[-1] + c.shape[0])
    pts = np.concatenate(clouds) if clouds else np.zeros((0, 4), np.float32)
    return poses, np.ascontiguousarray(pts), np.array(offs, np.int32)


def loop_icp(params, frames):
    """performLoopClosure's ICP: frames[0] = newest key frame, frames[1:] = history frames.  Returns (result dict, target cloud)."""
    poses, pts, offs = pack_frames(frames)
    out = np.zeros(21)
    tgt = np.empty((max(pts.shape[0], 1), 4), np.float32)
    lib().oracle_loop_icp(C.addressof(params), poses.ctypes.data, pts.ctypes.data, offs.ctypes.data, len(frames) - 1, out.ctypes.data, tgt.ctypes.data, tgt.shape[0])
    return dict(converged=int(out[0]), iterations=int(out[1]), n_source=int(out[2]), n_target=int(out[3]), fitness=float(out[4]),
                T=out[5:21].astype(np.float32).reshape(4, 4)), tgt[:int(out[3])].copy()
