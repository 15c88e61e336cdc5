"""ctypes mirror of include/alego_params.h (field order and types must match exactly)."""
import ctypes as C

_I, _D, _F = C.c_int32, C.c_double, C.c_float


class AlegoParams(C.Structure):
    _fields_ = [
        ("n_scan", _I), ("horizon_scan", _I), ("ang_res_x", _D), ("ang_res_y", _D), ("ang_bottom", _D),
        ("ground_scan_id", _I), ("laser_type", _I), ("sensor_mount_ang", _D), ("ground_angle_thres", _D),
        ("seg_alpha_x", _D), ("seg_alpha_y", _D), ("seg_theta", _D),
        ("seg_valid_point_num", _I), ("seg_valid_line_num", _I), ("seg_big_num", _I),
        ("near_filter", _I), ("near_thres", _D),
        ("occl_col_diff", _I), ("occl_depth", _D), ("parallel_ratio", _D), ("occl_f32", _I),
        ("n_sectors", _I), ("sector_formula", _I), ("edge_thres", _D), ("surf_thres", _D),
        ("n_sharp", _I), ("n_less_sharp", _I), ("n_flat", _I), ("suppress_radius", _I), ("suppress_col_diff", _I),
        ("less_flat_leaf", _F), ("sort_mode", _I),
        ("nearest_feature_dist", _D), ("ring_window", _I), ("huber_delta", _D),
        ("lo_min_corr", _I), ("lo_iters_surf", _I), ("lo_iters_corner", _I),
        ("lm_leaf_corner", _F), ("lm_leaf_surf", _F), ("lm_leaf_outlier", _F),
        ("min_keyframe_dist", _D), ("recent_keyframe_num", _I), ("lm_every", _I),
        ("lm_outer_iters", _I), ("lm_max_iters", _I),
        ("knn_max_dist", _D), ("line_ratio", _D), ("line_half_len", _D), ("plane_tol", _D),
        ("lm_min_corner", _I), ("lm_min_surf", _I), ("lm_min_map_corner", _I),
        ("lc_search_radius", _D), ("lc_search_num", _I), ("lc_fitness_max", _D), ("lc_leaf", _F), ("lc_min_time_gap", _D),
        ("icp_max_corr_dist", _D), ("icp_max_iters", _I), ("icp_trans_eps", _D), ("icp_fitness_eps", _D),
        ("input_is_dense", _I),
        ("deskew_mode", _I), ("scan_period", _D),
        ("kf_cap_surf", _I), ("kf_cap_outlier", _I),
    ]

    def copy(self):
        q = AlegoParams()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(self))
        return q


class AlegoPoint(C.Structure):
    _fields_ = [("x", _F), ("y", _F), ("z", _F), ("intensity", _F)]
