"""ctypes binding of the C ABI in include/alego_mi355x.h (libalego_mi355x.so).

Host-side mirror of the reference's node interface: `Handle.ip_process` ~ ImageProjection::pcCB,
`Handle.lo_process` ~ LaserOdometry::mainLoop body, `Handle.lm_process` ~ LaserMapping::mainLoop
body, `Handle.scan_process` = the three chained on the device.  There is no CPU fallback: loading
fails loudly if the HIP library is missing and `Handle()` raises if no gfx950 device is visible.
"""
import ctypes as C
import os

import numpy as np

from .params import AlegoParams, AlegoPoint

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_DT = {0: np.float32, 1: np.float64, 2: np.int32, 3: np.uint8}

# every symbol include/alego_mi355x.h declares
EXPORTS = [
    "alego_create", "alego_destroy", "alego_last_error", "alego_device_count", "alego_params_sizeof",
    "alego_ip_process", "alego_lo_process", "alego_lm_process", "alego_scan_process",
    "alego_batch_load", "alego_batch_run", "alego_synchronize", "alego_batch_get_pose", "alego_batch_get_counts",
    "alego_stream", "alego_stream_groups", "alego_profile_enable", "alego_profile_report", "alego_set_lo_params", "alego_set_lm_params", "alego_debug_get", "alego_debug_voxel", "alego_debug_atan2f",
    "alego_lo_push_imu", "alego_lo_get_undistorted", "alego_pose_o2b", "alego_trajectory_enable", "alego_trajectory_get", "alego_debug_check_guards", "alego_debug_math", "alego_debug_std_sort", "alego_debug_eval_blocks", "alego_debug_transform_to_start", "alego_debug_set_option",
    "alego_lm_keyframe_count", "alego_lm_get_keyframe", "alego_lm_set_keypose", "alego_lm_reset_window", "alego_lm_apply_correction",
    "alego_lm_add_keyframe", "alego_pc2_to_points", "alego_replay_create", "alego_replay_load", "alego_replay_assign",
    "alego_dist_unique_id", "alego_dist_init", "alego_dist_shutdown", "alego_dist_allreduce_probe", "alego_stream_setup", "alego_stream_run",
    "alego_loop_detect", "alego_loop_closure_icp",
    "alego_bag_open", "alego_bag_close", "alego_bag_last_error", "alego_bag_topic_count", "alego_bag_topic_info", "alego_bag_message_count",
    "alego_bag_read_raw", "alego_bag_read_pc2", "alego_handle_lock", "alego_handle_unlock",
]

REPLAY_PINGPONG = 0x100
REPLAY_BAG = 0x200
FLAG_LO_INIT, FLAG_FEW_SURF, FLAG_FEW_CORNER, FLAG_LM_SKIPPED, FLAG_LM_FEW_FEATURES, FLAG_LM_KEYFRAME = 1, 2, 4, 8, 16, 32


class ScanIn(C.Structure):
    _fields_ = [("pts", C.c_void_p), ("n", C.c_int32), ("stamp", C.c_double)]


class SegOut(C.Structure):
    _fields_ = [("seg", C.c_void_p), ("seg_cap", C.c_int32), ("m", C.c_int32),
                ("ground", C.c_void_p), ("col", C.c_void_p), ("range", C.c_void_p),
                ("ring_start", C.c_void_p), ("ring_end", C.c_void_p), ("orientation", C.c_float * 3),
                ("outlier", C.c_void_p), ("outlier_cap", C.c_int32), ("n_outlier", C.c_int32),
                ("label_image", C.c_void_p), ("stamp", C.c_double)]


class FeatOut(C.Structure):
    _fields_ = [("sharp", C.c_void_p), ("sharp_cap", C.c_int32), ("n_sharp", C.c_int32),
                ("less_sharp", C.c_void_p), ("less_sharp_cap", C.c_int32), ("n_less_sharp", C.c_int32),
                ("flat", C.c_void_p), ("flat_cap", C.c_int32), ("n_flat", C.c_int32),
                ("less_flat", C.c_void_p), ("less_flat_cap", C.c_int32), ("n_less_flat", C.c_int32),
                ("point_label", C.c_void_p)]


class Pose(C.Structure):
    _fields_ = [("t", C.c_double * 3), ("q", C.c_double * 4), ("params", C.c_double * 6), ("valid", C.c_int32)]

    def as_dict(self):
        return dict(t=np.array(self.t[:]), q=np.array(self.q[:]), params=np.array(self.params[:]), valid=int(self.valid))


def pose_o2b(t, q, tf_b2l):
    """alego_pose_o2b: (t, q = w x y z) of /odom -> /laser and the 4 x 4 base_link -> laser mount -> (t, q) of /odom -> /base_link (LO.cpp:588-608)"""
    a, b = Pose(), Pose()
    a.t[:] = list(t); a.q[:] = list(q)
    m = np.ascontiguousarray(tf_b2l, np.float64).reshape(16)
    rc = lib().alego_pose_o2b(C.byref(a), m.ctypes.data, C.byref(b))
    if rc != 0:
        raise AlegoError(f"alego_pose_o2b failed ({rc})")
    return np.array(b.t[:]), np.array(b.q[:])


class KeyFrame(C.Structure):
    _fields_ = [("id", C.c_int32), ("pose", C.c_float * 6),
                ("corner", C.c_void_p), ("corner_cap", C.c_int32), ("n_corner", C.c_int32),
                ("surf", C.c_void_p), ("surf_cap", C.c_int32), ("n_surf", C.c_int32),
                ("outlier", C.c_void_p), ("outlier_cap", C.c_int32), ("n_outlier", C.c_int32)]


class KfIn(C.Structure):
    _fields_ = [("pose", C.c_float * 6), ("corner", C.c_void_p), ("n_corner", C.c_int32), ("surf", C.c_void_p), ("n_surf", C.c_int32),
                ("outlier", C.c_void_p), ("n_outlier", C.c_int32)]


class IcpResult(C.Structure):
    _fields_ = [("converged", C.c_int32), ("iterations", C.c_int32), ("n_source", C.c_int32), ("n_target", C.c_int32),
                ("fitness", C.c_double), ("correction", C.c_float * 16)]


class Pc2Field(C.Structure):
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint32), ("datatype", C.c_uint8), ("count", C.c_uint32)]


def lib_path():
    # ALEGO_LIB: a development build of the same library (e.g. with -DALEGO_TIMING for tools/*_timing.py)
    return os.environ.get("ALEGO_LIB") or os.path.join(_HERE, "libalego_mi355x.so")


def lib():
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: the HIP extension is not built (run __graft_entry__.build()); "
                               "there is no CPU fallback")
        # a handle drives two HIP streams per stream group (+ three for alego_stream_run): with the runtime's default of 4 hardware queues
        # pairs of them share a queue and serialise (measured: one stream 5.7 k -> 3.0 k scans/s).  Read by the HIP runtime when it initialises.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        L = C.CDLL(path)
        L.alego_create.restype = C.c_int
        L.alego_create.argtypes = [C.POINTER(AlegoParams), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.alego_destroy.argtypes = [C.c_void_p]
        L.alego_last_error.restype = C.c_char_p
        L.alego_last_error.argtypes = [C.c_void_p]
        L.alego_device_count.restype = C.c_int
        L.alego_params_sizeof.restype = C.c_int
        L.alego_ip_process.restype = C.c_int
        L.alego_ip_process.argtypes = [C.c_void_p, C.POINTER(ScanIn), C.POINTER(SegOut)]
        L.alego_lo_process.restype = C.c_int
        L.alego_lo_process.argtypes = [C.c_void_p, C.POINTER(SegOut), C.POINTER(FeatOut), C.POINTER(Pose)]
        L.alego_lm_process.restype = C.c_int
        L.alego_lm_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                       C.POINTER(Pose), C.POINTER(Pose)]
        L.alego_scan_process.restype = C.c_int
        L.alego_scan_process.argtypes = [C.c_void_p, C.c_int, C.POINTER(ScanIn), C.c_int, C.POINTER(SegOut),
                                         C.POINTER(FeatOut), C.POINTER(Pose), C.POINTER(Pose)]
        L.alego_batch_load.restype = C.c_int
        L.alego_batch_load.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int32]
        L.alego_batch_run.restype = C.c_int
        L.alego_batch_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.alego_synchronize.restype = C.c_int
        L.alego_synchronize.argtypes = [C.c_void_p]
        L.alego_batch_get_pose.restype = C.c_int
        L.alego_batch_get_pose.argtypes = [C.c_void_p, C.c_int, C.POINTER(Pose), C.POINTER(Pose)]
        L.alego_batch_get_counts.restype = C.c_int
        L.alego_batch_get_counts.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.alego_profile_enable.restype = C.c_int
        L.alego_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.alego_profile_report.restype = C.c_int
        L.alego_profile_report.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.alego_stream.restype = C.c_void_p
        L.alego_stream.argtypes = [C.c_void_p]
        L.alego_stream_groups.restype = C.c_int
        L.alego_stream_groups.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.alego_set_lo_params.restype = C.c_int
        L.alego_set_lo_params.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.alego_set_lm_params.restype = C.c_int
        L.alego_set_lm_params.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.alego_debug_get.restype = C.c_int
        L.alego_debug_get.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.alego_debug_voxel.restype = C.c_int
        L.alego_debug_voxel.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int]
        L.alego_debug_atan2f.restype = C.c_int
        L.alego_debug_atan2f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.alego_trajectory_enable.restype = C.c_int
        L.alego_trajectory_enable.argtypes = [C.c_void_p, C.c_int32]
        L.alego_trajectory_get.restype = C.c_int
        L.alego_trajectory_get.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_void_p]
        L.alego_lo_push_imu.restype = C.c_int
        L.alego_lo_push_imu.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int32]
        L.alego_lo_get_undistorted.restype = C.c_int
        L.alego_lo_get_undistorted.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int32]
        L.alego_pose_o2b.restype = C.c_int
        L.alego_pose_o2b.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.alego_debug_check_guards.restype = C.c_int
        L.alego_debug_check_guards.argtypes = [C.c_char_p, C.c_int]
        L.alego_debug_std_sort.restype = C.c_int
        L.alego_debug_std_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.alego_debug_math.restype = C.c_int
        L.alego_debug_math.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.alego_debug_eval_blocks.restype = C.c_int
        L.alego_debug_eval_blocks.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.alego_debug_transform_to_start.restype = C.c_int
        L.alego_debug_transform_to_start.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.alego_debug_set_option.restype = C.c_int
        L.alego_debug_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.alego_lm_keyframe_count.restype = C.c_int
        L.alego_lm_keyframe_count.argtypes = [C.c_void_p, C.c_int]
        L.alego_lm_get_keyframe.restype = C.c_int
        L.alego_lm_get_keyframe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(KeyFrame)]
        L.alego_lm_set_keypose.restype = C.c_int
        L.alego_lm_set_keypose.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.alego_lm_reset_window.restype = C.c_int
        L.alego_lm_reset_window.argtypes = [C.c_void_p, C.c_int]
        L.alego_lm_apply_correction.restype = C.c_int
        L.alego_lm_apply_correction.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.alego_lm_add_keyframe.restype = C.c_int
        L.alego_lm_add_keyframe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.alego_pc2_to_points.restype = C.c_int
        L.alego_pc2_to_points.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                          C.POINTER(Pc2Field), C.c_int, C.c_void_p, C.c_int32]
        L.alego_replay_create.restype = C.c_int
        L.alego_replay_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.alego_replay_load.restype = C.c_int
        L.alego_replay_load.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int32]
        L.alego_replay_assign.restype = C.c_int
        L.alego_replay_assign.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.alego_dist_unique_id.restype = C.c_int
        L.alego_dist_unique_id.argtypes = [C.c_char_p]
        L.alego_dist_init.restype = C.c_int
        L.alego_dist_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
        L.alego_dist_shutdown.restype = C.c_int
        L.alego_dist_shutdown.argtypes = [C.c_void_p]
        L.alego_dist_allreduce_probe.restype = C.c_int
        L.alego_dist_allreduce_probe.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        L.alego_stream_setup.restype = C.c_int
        L.alego_stream_setup.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.alego_stream_run.restype = C.c_int
        L.alego_stream_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.alego_loop_detect.restype = C.c_int
        L.alego_loop_detect.argtypes = [C.POINTER(AlegoParams), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.alego_loop_closure_icp.restype = C.c_int
        L.alego_loop_closure_icp.argtypes = [C.c_void_p, C.POINTER(KfIn), C.POINTER(KfIn), C.c_int32, C.POINTER(IcpResult), C.c_void_p, C.c_int32]
        L.alego_handle_lock.restype = C.c_int
        L.alego_handle_lock.argtypes = [C.c_void_p]
        L.alego_handle_unlock.restype = C.c_int
        L.alego_handle_unlock.argtypes = [C.c_void_p]
        L.alego_bag_open.restype = C.c_int
        L.alego_bag_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.alego_bag_close.restype = None
        L.alego_bag_close.argtypes = [C.c_void_p]
        L.alego_bag_last_error.restype = C.c_char_p
        L.alego_bag_last_error.argtypes = [C.c_void_p]
        L.alego_bag_topic_count.restype = C.c_int
        L.alego_bag_topic_count.argtypes = [C.c_void_p]
        L.alego_bag_topic_info.restype = C.c_int
        L.alego_bag_topic_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]
        L.alego_bag_message_count.restype = C.c_int64
        L.alego_bag_message_count.argtypes = [C.c_void_p, C.c_char_p]
        L.alego_bag_read_raw.restype = C.c_int
        L.alego_bag_read_raw.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.alego_bag_read_pc2.restype = C.c_int
        L.alego_bag_read_pc2.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        if L.alego_params_sizeof() != C.sizeof(AlegoParams):
            raise RuntimeError("alego_params layout mismatch between params.py and include/alego_params.h")
        _lib = L
    return _lib


class AlegoError(RuntimeError):
    pass


DIST_ID_BYTES = 128


def dist_unique_id() -> bytes:
    """rank 0: the RCCL unique id every rank passes to Handle.dist_init (ncclGetUniqueId)"""
    buf = C.create_string_buffer(DIST_ID_BYTES)
    rc = lib().alego_dist_unique_id(buf)
    if rc != 0:
        raise AlegoError(f"alego_dist_unique_id failed ({rc})")
    return buf.raw


def check_guards():
    """(count, report) of damaged allocation guards; count = -1 unless ALEGO_DEBUG_CANARY is set"""
    buf = C.create_string_buffer(4096)
    return lib().alego_debug_check_guards(buf, 4096), buf.value.decode(errors="replace")


def loop_detect(params, keyposes6, stamps, cur_xyz):
    """detectLoopClosure's closest_history_frame_id_ (host code of the library; -1 = no candidate)"""
    kp = np.ascontiguousarray(keyposes6, np.float32).reshape(-1, 6)
    t = np.ascontiguousarray(stamps, np.float64)
    c = np.ascontiguousarray(cur_xyz, np.float64)
    return int(lib().alego_loop_detect(C.byref(params), kp.ctypes.data, t.ctypes.data, kp.shape[0], c.ctypes.data))


def pc2_to_points(data: bytes, width, height, point_step, row_step, fields, is_bigendian=False, cap=None):
    """sensor_msgs/PointCloud2 payload -> (n, 4) float32 (x, y, z, intensity); fields = [(name, offset, datatype, count), ...].
    Host-side (no GPU needed): the ROS-free equivalent of pcl::fromROSMsg<PointXYZI> in front of alego_ip_process."""
    n = width * height
    cap = n if cap is None else cap
    out = np.empty((max(cap, 1), 4), np.float32)
    arr = (Pc2Field * len(fields))(*[Pc2Field(nm.encode(), off, dt, cnt) for nm, off, dt, cnt in fields])
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if len(data) else None
    rc = lib().alego_pc2_to_points(buf, len(data), width, height, point_step, row_step, 1 if is_bigendian else 0, arr, len(fields),
                                   out.ctypes.data, cap)
    if rc < 0:
        raise AlegoError(f"alego_pc2_to_points failed ({rc})")
    return out[:rc].copy()


class Bag:
    """A rosbag format 2.0 file opened by the library's own reader (no ROS): topics, and the PointCloud2 messages of a topic in
    time order — what `rosbag play` + the /lslidar_point_cloud subscriber + pcl::fromROSMsg deliver to ImageProjection."""

    def __init__(self, path):
        h = C.c_void_p()
        rc = lib().alego_bag_open(os.fsencode(path), C.byref(h))
        if rc != 0:
            raise AlegoError(f"alego_bag_open({path}) failed ({rc}): not a readable rosbag 2.0 file (see stderr)")
        self._b = h

    def close(self):
        if getattr(self, "_b", None):
            lib().alego_bag_close(self._b)
            self._b = None

    def __del__(self):
        self.close()

    def _err(self):
        return lib().alego_bag_last_error(self._b).decode(errors="replace")

    def topics(self):
        """{topic: (datatype, number of messages)}"""
        out = {}
        for i in range(lib().alego_bag_topic_count(self._b)):
            t, d, n = C.c_char_p(), C.c_char_p(), C.c_int64()
            lib().alego_bag_topic_info(self._b, i, C.byref(t), C.byref(d), C.byref(n))
            out[t.value.decode()] = (d.value.decode(), int(n.value))
        return out

    def message_count(self, topic):
        return int(lib().alego_bag_message_count(self._b, topic.encode()))

    def read_raw(self, topic, index):
        """(serialized message bytes, bag time)"""
        p, n, t = C.c_void_p(), C.c_uint64(), C.c_double()
        rc = lib().alego_bag_read_raw(self._b, topic.encode(), index, C.byref(p), C.byref(n), C.byref(t))
        if rc != 0:
            raise AlegoError(f"alego_bag_read_raw failed ({rc}): {self._err()}")
        return C.string_at(p.value, n.value), float(t.value)

    def read_pc2(self, topic, index, cap=1 << 20):
        """(points[n, 4] float32, header stamp, is_dense)"""
        out = np.empty((cap, 4), np.float32)
        st, dn = C.c_double(), C.c_int32()
        n = lib().alego_bag_read_pc2(self._b, topic.encode(), index, out.ctypes.data, cap, C.byref(st), C.byref(dn))
        if n < 0:
            raise AlegoError(f"alego_bag_read_pc2 failed ({n}): {self._err()}")
        return out[:n].copy(), float(st.value), bool(dn.value)


_CLOUDS = {"seg_cloud", "undistorted", "outlier", "sharp", "less_sharp", "flat", "less_flat"}


class Handle:
    """One alego_handle: `n_slots` independent streams advanced in lock-step on one GPU."""

    def __init__(self, params: AlegoParams, device: int = 0, n_slots: int = 1, ring_len: int = 1):
        L = lib()
        self.params = params
        self.N = params.n_scan * params.horizon_scan
        self.n_slots, self.ring_len = n_slots, ring_len
        h = C.c_void_p()
        rc = L.alego_create(C.byref(params), device, n_slots, ring_len, C.byref(h))
        if rc != 0:
            why = {-1: "no gfx950 device (there is no CPU fallback)", -2: "a HIP call failed", -3: "capacity", -4: "a parameter is outside the supported range (see stderr)"}.get(rc, "?")
            raise AlegoError(f"alego_create failed ({rc}): {why}")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            bad, rep = check_guards()   # (ALEGO_DEBUG_CANARY=1: did a kernel write outside a buffer?  -1 = not enabled)
            lib().alego_destroy(self._h)
            self._h = None
            if bad > 0:
                raise AlegoError(f"device memory guards damaged: {rep}")

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc < 0:
            raise AlegoError(f"{what} failed ({rc}): {lib().alego_last_error(self._h).decode()}")
        return rc

    # ---- buffers ----
    def _seg_bufs(self, want_labels):
        N, NS = self.N, self.params.n_scan
        b = dict(seg=np.empty((N, 4), np.float32), ground=np.empty(N, np.uint8), col=np.empty(N, np.int32),
                 range=np.empty(N, np.float32), ring_start=np.empty(NS, np.int32), ring_end=np.empty(NS, np.int32),
                 outlier=np.empty((N, 4), np.float32), label_image=np.empty(N, np.int32) if want_labels else None)
        s = SegOut()
        s.seg, s.seg_cap = b["seg"].ctypes.data, N
        s.ground, s.col, s.range = b["ground"].ctypes.data, b["col"].ctypes.data, b["range"].ctypes.data
        s.ring_start, s.ring_end = b["ring_start"].ctypes.data, b["ring_end"].ctypes.data
        s.outlier, s.outlier_cap = b["outlier"].ctypes.data, N
        s.label_image = b["label_image"].ctypes.data if want_labels else None
        return s, b

    @staticmethod
    def _seg_result(s, b):
        m, no = s.m, s.n_outlier
        return dict(seg=b["seg"][:m].copy(), ground=b["ground"][:m].copy(), col=b["col"][:m].copy(), range=b["range"][:m].copy(),
                    ring_start=b["ring_start"].copy(), ring_end=b["ring_end"].copy(), orientation=np.array(s.orientation[:], np.float32),
                    outlier=b["outlier"][:no].copy(), label_image=None if b["label_image"] is None else b["label_image"].copy())

    def _feat_bufs(self, want_labels=True):
        N = self.N
        caps = (N, N, N, N)
        b = dict(sharp=np.empty((caps[0], 4), np.float32), less_sharp=np.empty((caps[1], 4), np.float32),
                 flat=np.empty((caps[2], 4), np.float32), less_flat=np.empty((caps[3], 4), np.float32),
                 point_label=np.empty(N, np.int32) if want_labels else None)
        f = FeatOut()
        f.sharp, f.sharp_cap = b["sharp"].ctypes.data, caps[0]
        f.less_sharp, f.less_sharp_cap = b["less_sharp"].ctypes.data, caps[1]
        f.flat, f.flat_cap = b["flat"].ctypes.data, caps[2]
        f.less_flat, f.less_flat_cap = b["less_flat"].ctypes.data, caps[3]
        f.point_label = b["point_label"].ctypes.data if want_labels else None
        return f, b

    @staticmethod
    def _feat_result(f, b, m=None):
        return dict(sharp=b["sharp"][:f.n_sharp].copy(), less_sharp=b["less_sharp"][:f.n_less_sharp].copy(),
                    flat=b["flat"][:f.n_flat].copy(), less_flat=b["less_flat"][:f.n_less_flat].copy(),
                    point_label=None if b["point_label"] is None else (b["point_label"][:m].copy() if m is not None else b["point_label"].copy()))

    @staticmethod
    def _scan(pts, stamp=0.0):
        a = np.ascontiguousarray(pts, dtype=np.float32)
        assert a.ndim == 2 and a.shape[1] == 4
        s = ScanIn()
        s.pts, s.n, s.stamp = a.ctypes.data, a.shape[0], float(stamp)
        return s, a

    # ---- nodelet-shaped entry points ----
    def ip_process(self, pts, want_labels=True, stamp=0.0):
        sin, keep = self._scan(pts, stamp)
        s, b = self._seg_bufs(want_labels)
        self._check(lib().alego_ip_process(self._h, C.byref(sin), C.byref(s)), "alego_ip_process")
        r = self._seg_result(s, b)
        r["stamp"] = s.stamp
        return r

    def lo_process(self, seg):
        """seg: dict as returned by ip_process (what LO receives on /segmented_cloud + /seg_info)."""
        m = seg["seg"].shape[0]
        s = SegOut()
        arrs = [np.ascontiguousarray(seg["seg"], np.float32), np.ascontiguousarray(seg["ground"], np.uint8),
                np.ascontiguousarray(seg["col"], np.int32), np.ascontiguousarray(seg["range"], np.float32),
                np.ascontiguousarray(seg["ring_start"], np.int32), np.ascontiguousarray(seg["ring_end"], np.int32)]
        s.seg, s.seg_cap, s.m = arrs[0].ctypes.data, m, m
        s.ground, s.col, s.range = arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data
        s.ring_start, s.ring_end = arrs[4].ctypes.data, arrs[5].ctypes.data
        if "orientation" in seg:
            s.orientation[:] = [float(v) for v in seg["orientation"]]
        s.stamp = float(seg.get("stamp", 0.0))
        f, fb = self._feat_bufs()
        odom = Pose()
        flags = self._check(lib().alego_lo_process(self._h, C.byref(s), C.byref(f), C.byref(odom)), "alego_lo_process")
        return flags, self._feat_result(f, fb, m), odom.as_dict()

    def lm_process(self, corner_last, surf_last, outlier, odom):
        c = np.ascontiguousarray(corner_last, np.float32)
        s = np.ascontiguousarray(surf_last, np.float32)
        o = np.ascontiguousarray(outlier, np.float32)
        po, pm = Pose(), Pose()
        po.t[:] = list(odom["t"]); po.q[:] = list(odom["q"]); po.valid = 1
        flags = self._check(lib().alego_lm_process(self._h, c.ctypes.data, c.shape[0], s.ctypes.data, s.shape[0],
                                                    o.ctypes.data, o.shape[0], C.byref(po), C.byref(pm)), "alego_lm_process")
        return flags, pm.as_dict()

    def scan_process(self, pts, stages=7, slot=0, want_outputs=False, stamp=0.0):
        sin, keep = self._scan(pts, stamp)
        odom, mp = Pose(), Pose()
        if want_outputs:
            s, b = self._seg_bufs(True)
            f, fb = self._feat_bufs()
            flags = self._check(lib().alego_scan_process(self._h, slot, C.byref(sin), stages, C.byref(s), C.byref(f),
                                                          C.byref(odom), C.byref(mp)), "alego_scan_process")
            return flags, odom.as_dict(), mp.as_dict(), self._seg_result(s, b), self._feat_result(f, fb, s.m)
        flags = self._check(lib().alego_scan_process(self._h, slot, C.byref(sin), stages, None, None, C.byref(odom), C.byref(mp)),
                            "alego_scan_process")
        return flags, odom.as_dict(), mp.as_dict()

    # ---- batch path ----
    def batch_load(self, slot, ring_pos, pts):
        a = np.ascontiguousarray(pts, dtype=np.float32)
        self._check(lib().alego_batch_load(self._h, slot, ring_pos, a.ctypes.data, a.shape[0]), "alego_batch_load")

    def replay_create(self, n_bags, bag_len):
        self._check(lib().alego_replay_create(self._h, n_bags, bag_len), "alego_replay_create")

    def replay_load(self, bag, scan, pts):
        a = np.ascontiguousarray(pts, dtype=np.float32)
        self._check(lib().alego_replay_load(self._h, bag, scan, a.ctypes.data, a.shape[0]), "alego_replay_load")

    def replay_assign(self, slot, bag, start_scan):
        self._check(lib().alego_replay_assign(self._h, slot, bag, start_scan), "alego_replay_assign")

    def stream_setup(self, bag, start_scan=0):
        """slot 0 = one stream replaying `bag`; the other slots become its look-ahead lanes (alego_stream_run)"""
        self._check(lib().alego_stream_setup(self._h, bag, start_scan), "alego_stream_setup")

    def stream_run(self, first_step, n_scans, stages=7, sync=True):
        self._check(lib().alego_stream_run(self._h, first_step, n_scans, stages, 1 if sync else 0), "alego_stream_run")

    def batch_run(self, first_pos, n_scans, stages=7, sync=True):
        self._check(lib().alego_batch_run(self._h, first_pos, n_scans, stages, 1 if sync else 0), "alego_batch_run")

    def synchronize(self):
        self._check(lib().alego_synchronize(self._h), "alego_synchronize")

    def batch_get_pose(self, slot=0):
        odom, mp = Pose(), Pose()
        flags = self._check(lib().alego_batch_get_pose(self._h, slot, C.byref(odom), C.byref(mp)), "alego_batch_get_pose")
        return flags, odom.as_dict(), mp.as_dict()

    def batch_get_counts(self, slot=0):
        out = np.zeros(16, np.int32)
        self._check(lib().alego_batch_get_counts(self._h, slot, out.ctypes.data, 16), "alego_batch_get_counts")
        keys = ["P", "M", "O", "Qc", "Fc", "Qs", "Fs", "n_surf_corr", "n_corner_corr", "Kraw_c", "Kraw_s", "Kds_c", "Kds_s", "Lc", "Ls", "n_rebuild"]
        return dict(zip(keys, out.tolist()))

    def profile_enable(self, on=True):
        self._check(lib().alego_profile_enable(self._h, 1 if on else 0), "alego_profile_enable")

    def profile_report(self):
        """{kernel name: (total ms, launches)} measured with HIP events on the handle's stream."""
        names = C.create_string_buffer(8192)
        tot = np.zeros(256, np.float64)
        cnt = np.zeros(256, np.int32)
        n = self._check(lib().alego_profile_report(self._h, names, 8192, tot.ctypes.data, cnt.ctypes.data, 256), "alego_profile_report")
        ks = names.value.decode().split(";") if n else []
        return {k: (float(tot[i]), int(cnt[i])) for i, k in enumerate(ks)}

    def stream(self):
        return lib().alego_stream(self._h)

    def stream_groups(self):
        """(number of HIP stream groups, slots covered by one kernel launch)"""
        per = C.c_int(0)
        g = lib().alego_stream_groups(self._h, C.byref(per))
        return g, per.value

    # ---- test access ----
    def set_lo_params(self, p6, slot=0):
        a = np.ascontiguousarray(p6, np.float64)
        self._check(lib().alego_set_lo_params(self._h, slot, a.ctypes.data), "alego_set_lo_params")

    def set_lm_params(self, p6, slot=0):
        a = np.ascontiguousarray(p6, np.float64)
        self._check(lib().alego_set_lm_params(self._h, slot, a.ctypes.data), "alego_set_lm_params")

    def debug_get(self, name, slot=0, cap_bytes=None):
        cap = cap_bytes or max(self.N * 64 + 4096, 8 << 20)
        buf = np.empty(cap, np.uint8)
        cnt, dt = C.c_int(), C.c_int()
        self._check(lib().alego_debug_get(self._h, slot, name.encode(), buf.ctypes.data, cap, C.byref(cnt), C.byref(dt)),
                    f"alego_debug_get({name})")
        dtype = np.dtype(_DT[dt.value])
        out = np.frombuffer(buf.tobytes()[:cnt.value * dtype.itemsize], dtype=dtype).copy()
        return out.reshape(-1, 4) if (name in _CLOUDS or name.startswith("lm_") and name.endswith(("_ds", "_map"))) else out

    def voxel_grid(self, pts, leaf):
        """The device VoxelGrid on a host cloud (debug entry; both the LDS path and the bucket-sort path)."""
        a = np.ascontiguousarray(pts, np.float32)
        out = np.empty((max(a.shape[0], 1), 4), np.float32)
        n = self._check(lib().alego_debug_voxel(self._h, a.ctypes.data, a.shape[0], leaf, out.ctypes.data, out.shape[0]), "alego_debug_voxel")
        return out[:n].copy()

    def math(self, mode, a, b=None):
        """device single-precision functions: mode 0 atan2f(a, b), 1 hypotf(a, b), 2 sinf(a), 3 cosf(a)"""
        a = np.ascontiguousarray(a, np.float32)
        b = None if b is None else np.ascontiguousarray(b, np.float32)
        out = np.empty_like(a)
        self._check(lib().alego_debug_math(self._h, mode, a.ctypes.data, None if b is None else b.ctypes.data, out.ctypes.data, a.size), "alego_debug_math")
        return out

    def trajectory_enable(self, capacity):
        self._check(lib().alego_trajectory_enable(self._h, capacity), "alego_trajectory_enable")
        self._traj_cap = capacity

    def trajectory(self, slot=0, first=0, n=None):
        """poses[n, 14] logged for `slot`: odom t(3) q(4), map t(3) q(4) per processed scan"""
        if n is None:
            n = min(self._check(lib().alego_trajectory_get(self._h, slot, 0, 0, None), "alego_trajectory_get"), self._traj_cap) - first
        out = np.empty((max(n, 0), 14), np.float64)
        self._check(lib().alego_trajectory_get(self._h, slot, first, max(n, 0), out.ctypes.data), "alego_trajectory_get")
        return out

    def undistorted(self, slot=0):
        """/undistorted: the de-skewed segmented cloud of the slot's last scan (deskew_mode = 1), [M, 4] f32"""
        out = np.zeros((self.N, 4), np.float32)
        n = self._check(lib().alego_lo_get_undistorted(self._h, slot, out.ctypes.data, out.shape[0]), "alego_lo_get_undistorted")
        return out[:n].copy()

    def push_imu(self, samples, slot=0):
        """samples[n, 11]: stamp, orientation w x y z, linear_acceleration xyz, angular_velocity xyz (sensor_msgs/Imu)"""
        a = np.ascontiguousarray(samples, np.float64).reshape(-1, 11)
        self._check(lib().alego_lo_push_imu(self._h, slot, a.ctypes.data, a.shape[0]), "alego_lo_push_imu")

    def std_sort(self, keys, depth_limit=-1):
        """index order libstdc++'s std::sort gives 0..n-1 under `keys[a] < keys[b]`, as the device reproduces it (sort_mode 2)"""
        k = np.ascontiguousarray(keys, np.uint32)
        out = np.empty(k.size, np.int32)
        self._check(lib().alego_debug_std_sort(self._h, k.ctypes.data, k.size, depth_limit, out.ctypes.data), "alego_debug_std_sort")
        return out

    def eval_blocks(self, btype, geom13, params6):
        """(residuals[n], jacobians[n, 6]) of cost functor `btype` evaluated on the device as the solvers do"""
        g = np.ascontiguousarray(geom13, np.float64).reshape(-1, 13)
        p = np.ascontiguousarray(params6, np.float64)
        r, J = np.empty(g.shape[0]), np.empty((g.shape[0], 6))
        self._check(lib().alego_debug_eval_blocks(self._h, btype, g.shape[0], g.ctypes.data, p.ctypes.data, r.ctypes.data, J.ctypes.data), "alego_debug_eval_blocks")
        return r, J

    def transform_to_start(self, params6, pts):
        p = np.ascontiguousarray(params6, np.float64)
        a = np.ascontiguousarray(pts, np.float32)
        out = np.empty_like(a)
        self._check(lib().alego_debug_transform_to_start(self._h, p.ctypes.data, a.ctypes.data, a.shape[0], out.ctypes.data), "alego_debug_transform_to_start")
        return out

    def set_option(self, name, value):
        self._check(lib().alego_debug_set_option(self._h, name.encode(), int(value)), f"alego_debug_set_option({name})")

    # ---- loop closure ----
    def loop_closure_icp(self, frames):
        """frames = [(pose6, corner, surf, outlier), ...]: the newest key frame, then the history frames.  Returns (result dict, target cloud)."""
        keep, kfs = [], (KfIn * len(frames))()
        total = 0
        for i, f in enumerate(frames):
            kfs[i].pose[:] = [float(v) for v in f[0]]
            arrs = [np.ascontiguousarray(c, np.float32).reshape(-1, 4) for c in f[1:4]]
            keep.append(arrs)
            kfs[i].corner, kfs[i].n_corner = arrs[0].ctypes.data, arrs[0].shape[0]
            kfs[i].surf, kfs[i].n_surf = arrs[1].ctypes.data, arrs[1].shape[0]
            kfs[i].outlier, kfs[i].n_outlier = arrs[2].ctypes.data, arrs[2].shape[0]
            total += sum(a.shape[0] for a in arrs)
        out = IcpResult()
        tgt = np.empty((max(total, 1), 4), np.float32)
        hist = C.cast(C.byref(kfs, C.sizeof(KfIn)), C.POINTER(KfIn)) if len(frames) > 1 else None
        self._check(lib().alego_loop_closure_icp(self._h, C.byref(kfs[0]), hist, len(frames) - 1, C.byref(out), tgt.ctypes.data, tgt.shape[0]),
                    "alego_loop_closure_icp")
        return dict(converged=int(out.converged), iterations=int(out.iterations), n_source=int(out.n_source), n_target=int(out.n_target),
                    fitness=float(out.fitness), T=np.array(out.correction[:], np.float32).reshape(4, 4)), tgt[:out.n_target].copy()

    # ---- one registration sharded over the ranks of an RCCL communicator (BASELINE config 5) ----
    def dist_init(self, rank, world, unique_id: bytes):
        self._check(lib().alego_dist_init(self._h, rank, world, C.create_string_buffer(unique_id, DIST_ID_BYTES)), "alego_dist_init")

    def dist_allreduce_probe(self, iters=200):
        """microseconds per ncclAllReduce of one solver evaluation's 32 doubles (a collective: every rank calls it)"""
        us = C.c_double()
        self._check(lib().alego_dist_allreduce_probe(self._h, iters, C.byref(us)), "alego_dist_allreduce_probe")
        return float(us.value)

    def dist_shutdown(self):
        self._check(lib().alego_dist_shutdown(self._h), "alego_dist_shutdown")

    # ---- key-frame pass-through (host pose graph) ----
    def lm_keyframe_count(self, slot=0):
        return self._check(lib().alego_lm_keyframe_count(self._h, slot), "alego_lm_keyframe_count")

    def lm_get_keyframe(self, kf_id=-1, slot=0):
        """dict(id, pose[6], corner, surf, outlier): a resident key frame as saveKeyFramesAndFactor stored it"""
        N = self.N
        bufs = [np.empty((N, 4), np.float32) for _ in range(3)]
        k = KeyFrame()
        k.corner, k.corner_cap = bufs[0].ctypes.data, N
        k.surf, k.surf_cap = bufs[1].ctypes.data, N
        k.outlier, k.outlier_cap = bufs[2].ctypes.data, N
        self._check(lib().alego_lm_get_keyframe(self._h, slot, kf_id, C.byref(k)), "alego_lm_get_keyframe")
        return dict(id=int(k.id), pose=np.array(k.pose[:], np.float32), corner=bufs[0][:k.n_corner].copy(), surf=bufs[1][:k.n_surf].copy(),
                    outlier=bufs[2][:k.n_outlier].copy())

    def lm_set_keypose(self, kf_id, pose6, slot=0):
        a = np.ascontiguousarray(pose6, np.float32)
        self._check(lib().alego_lm_set_keypose(self._h, slot, kf_id, a.ctypes.data), "alego_lm_set_keypose")

    def lm_reset_window(self, slot=0):
        self._check(lib().alego_lm_reset_window(self._h, slot), "alego_lm_reset_window")

    def lm_apply_correction(self, rc12, slot=0):
        a = np.ascontiguousarray(rc12, np.float64).reshape(12)
        self._check(lib().alego_lm_apply_correction(self._h, slot, a.ctypes.data), "alego_lm_apply_correction")

    def lm_add_keyframe(self, pose6, corner, surf, outlier, slot=0):
        a = np.ascontiguousarray(pose6, np.float32)
        c, s, o = (np.ascontiguousarray(v, np.float32).reshape(-1, 4) for v in (corner, surf, outlier))
        self._check(lib().alego_lm_add_keyframe(self._h, slot, a.ctypes.data, c.ctypes.data, c.shape[0], s.ctypes.data, s.shape[0],
                                                o.ctypes.data, o.shape[0]), "alego_lm_add_keyframe")

    def atan2f(self, y, x):
        y = np.ascontiguousarray(y, np.float32)
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(y)
        self._check(lib().alego_debug_atan2f(self._h, y.ctypes.data, x.ctypes.data, out.ctypes.data, y.size), "alego_debug_atan2f")
        return out
