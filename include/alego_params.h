/*
 * alego_params.h — every tunable of the A-LeGO-LOAM per-scan hot path as one POD.
 *
 * The reference has no runtime configuration: all of these are compile-time
 * constants (include/alego/utility.h:50-73) or literals inside the node bodies
 * (listed per field below).  Defaults written by alego_default_params() are the
 * reference values; the sensor geometry (n_scan, horizon_scan, ang_res_*) is a
 * runtime parameter because BASELINE.json's configs use 16x1800 and 64x2048
 * while the reference compiles in 16x4000.
 *
 * This header is the data contract shared by the C-ABI library
 * (include/alego_mi355x.h) and by the parity oracle (oracle/); it contains no
 * code of either.
 */
#ifndef ALEGO_PARAMS_H_
#define ALEGO_PARAMS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum alego_laser_type { ALEGO_LASER_UNIFORM = 0, /* LSLIDAR_C16 formula, imageProjection.cpp:80 */
                        ALEGO_LASER_RFANS_16M = 1 /* piece-wise ring table, IP.cpp:142-172 */ };

typedef struct alego_params {
  /* ---- sensor geometry: utility.h:50-58 ---- */
  int32_t n_scan;            /* N_SCAN = 16                          utility.h:50 */
  int32_t horizon_scan;      /* Horizon_SCAN = int(360/ang_res_x+.5) utility.h:55 */
  double ang_res_x;          /* 0.09 deg                             utility.h:51 */
  double ang_res_y;          /* 2.0 deg                              utility.h:52 */
  double ang_bottom;         /* 15.0 deg                             utility.h:56 */
  int32_t ground_scan_id;    /* 10                                   utility.h:57 */
  int32_t laser_type;        /* alego_laser_type; nodelet = UNIFORM  utility.h:81 */
  double sensor_mount_ang;   /* 0.0                                  utility.h:58 */
  double ground_angle_thres; /* 10.0 deg        imageProjection.cpp:127 */
  /* ---- segmentation: utility.h:60-65 ---- */
  double seg_alpha_x;        /* rad(ang_res_x)                       utility.h:60 */
  double seg_alpha_y;        /* rad(ang_res_y)                       utility.h:61 */
  double seg_theta;          /* 1.047                                utility.h:63 */
  int32_t seg_valid_point_num; /* 5                                  utility.h:64 */
  int32_t seg_valid_line_num;  /* 3                                  utility.h:65 */
  int32_t seg_big_num;       /* 30              imageProjection.cpp:283 */
  /* ---- input filter (standalone IP.cpp only) ---- */
  int32_t near_filter;       /* 0 nodelet / 1 IP.cpp:117 */
  double near_thres;         /* 1.0 m           IP.cpp:117 */
  /* ---- feature extraction: laserOdometry.cpp:122-293 ---- */
  int32_t occl_col_diff;     /* 10              laserOdometry.cpp:137 */
  double occl_depth;         /* 0.5             laserOdometry.cpp:140 */
  double parallel_ratio;     /* 0.02            laserOdometry.cpp:154 */
  int32_t occl_f32;          /* 0 = nodelet (double depth1/2, laserOdometry.cpp:134), 1 = LO.cpp:203 (float) */
  int32_t n_sectors;         /* 6               laserOdometry.cpp:175 */
  int32_t sector_formula;    /* 0 = laserOdometry.cpp:177-178, 1 = LO.cpp:245-249 */
  double edge_thres;         /* 0.1             laserOdometry.cpp:192 */
  double surf_thres;         /* 0.1             laserOdometry.cpp:242 */
  int32_t n_sharp;           /* 2               laserOdometry.cpp:196 */
  int32_t n_less_sharp;      /* 20              laserOdometry.cpp:202 */
  int32_t n_flat;            /* 4               laserOdometry.cpp:248 */
  int32_t suppress_radius;   /* 5               laserOdometry.cpp:211; 0..5 (the 5-point ring margin): alego_create rejects more */
  int32_t suppress_col_diff; /* 10              laserOdometry.cpp:214 */
  float less_flat_leaf;      /* 0.4 m           laserOdometry.cpp:290 */
  int32_t sort_mode;         /* tie order of equal curvatures in the sector sort (laserOdometry.cpp:185, std::sort with a
                                comparator on the curvature alone) and of the points of one voxel in pcl::VoxelGrid's sort:
                                0 = total order (curvature, index) / input order inside a voxel (the default build rule);
                                2 = sector sort in libstdc++'s std::sort order (what the reference binary does), VoxelGrid as 0;
                                1 = both in std::sort order (oracle only: alego_create rejects it) */
  /* ---- scan-to-scan odometry: laserOdometry.cpp:328-508 ---- */
  double nearest_feature_dist; /* 25.0 (squared) utility.h:73 */
  int32_t ring_window;       /* 2               laserOdometry.cpp:350,441 */
  double huber_delta;        /* 0.1             laserOdometry.cpp:331, laserMapping.cpp:363 */
  int32_t lo_min_corr;       /* 10              laserOdometry.cpp:410,484 */
  int32_t lo_iters_surf;     /* 5               laserOdometry.cpp:415 */
  int32_t lo_iters_corner;   /* 5 (README says 10) laserOdometry.cpp:489 */
  /* ---- scan-to-map registration: laserMapping.cpp:37-49,348-479 ---- */
  float lm_leaf_corner;      /* 0.4             laserMapping.cpp:37 */
  float lm_leaf_surf;        /* 0.8             laserMapping.cpp:38 */
  float lm_leaf_outlier;     /* 1.0             laserMapping.cpp:39 */
  double min_keyframe_dist;  /* 1.0 (squared compare) laserMapping.cpp:43,501-504 */
  int32_t recent_keyframe_num; /* 50            laserMapping.cpp:48 */
  int32_t lm_every;          /* 2               laserMapping.cpp:112 */
  int32_t lm_outer_iters;    /* 2               laserMapping.cpp:360 */
  int32_t lm_max_iters;      /* 20              laserMapping.cpp:470 */
  double knn_max_dist;       /* 1.0 (squared)   laserMapping.cpp:376,426 */
  double line_ratio;         /* 3.0             laserMapping.cpp:403 */
  double line_half_len;      /* 0.1             laserMapping.cpp:406-407 */
  double plane_tol;          /* 0.2             laserMapping.cpp:446 */
  int32_t lm_min_corner;     /* 10              laserMapping.cpp:350 */
  int32_t lm_min_surf;       /* 100             laserMapping.cpp:350 */
  int32_t lm_min_map_corner; /* 10              laserMapping.cpp:350 */
  /* ---- loop closure (laserMapping.cpp:633-824; nodelet values, LM.cpp's in brackets) ---- */
  double lc_search_radius;   /* 20.0 [10.0]     history_search_radius_   laserMapping.cpp:76 */
  int32_t lc_search_num;     /* 25              history_search_num_      laserMapping.cpp:77 */
  double lc_fitness_max;     /* 0.4 [0.3]       history_fitness_score_   laserMapping.cpp:78 */
  float lc_leaf;             /* 1.0 [0.4]       ds_history_keyframes_    laserMapping.cpp:41 */
  double lc_min_time_gap;    /* 30.0 s          laserMapping.cpp:782 */
  double icp_max_corr_dist;  /* 100.0           laserMapping.cpp:671 */
  int32_t icp_max_iters;     /* 100             laserMapping.cpp:672 */
  double icp_trans_eps;      /* 1e-6            laserMapping.cpp:673 */
  double icp_fitness_eps;    /* 1e-6            laserMapping.cpp:674 */
  /* ---- input message property ---- */
  int32_t input_is_dense;    /* sensor_msgs/PointCloud2.is_dense of the driver.  0 (default): pcl::removeNaNFromPointCloud
                                drops non-finite points (imageProjection.cpp:58-59).  1: PCL copies the cloud unfiltered;
                                non-finite points then still count as first / last point of the orientation block (:62-63,
                                giving NaN orientations) and are rejected by the row test (:81-85; int(NaN) is INT_MIN on x86-64) */
  /* ---- motion de-skew: LaserOdometry::adjustDistortion, laserOdometry.cpp:557-726 (its call at :115 is commented out in the
   *      reference, so 0 is the reference as shipped) ---- */
  int32_t deskew_mode;       /* 0 = off; 1 = adjustDistortion with the IMU branch (use_imu = true, utility.h:68) on the samples given to
                                alego_lo_push_imu, before feature extraction */
  double scan_period;        /* 0.2 s           utility.h:53 */
  /* ---- capacities (no counterpart in the reference, whose clouds are std::vectors) ---- */
  int32_t kf_cap_surf;       /* points a key frame's laser_surf_ds_ / laser_outlier_ds_ may hold (laserMapping.cpp:547-555); 0 = the worst */
  int32_t kf_cap_outlier;    /* case n_scan*horizon_scan / 2 resp. / 4.  The local map of K key frames is sized K x (surf + outlier): with 64 x 2048
                                and K = 200 the worst case is 1.5 GB per stream while real frames hold ~2 k points.  A frame that does not fit
                                is truncated and reported (ALEGO_ERR_CAPACITY), never written past */
} alego_params;

/* Fill `p` with the reference defaults for an n_scan x horizon_scan sensor.
 * horizon_scan <= 0 selects the reference geometry 16 x 4000 (ang_res_x 0.09).
 * For n_scan == 64 the HDL-64E-shaped constants of SURVEY.md §8d config 5 are
 * used (ang_res_y 26.8/63, ang_bottom 24.8, ground_scan_id 50). */
static inline void alego_default_params(alego_params* p, int n_scan, int horizon_scan) {
  const double kPi = 3.14159265358979323846;
  if (n_scan <= 0) n_scan = 16;
  p->n_scan = n_scan;
  if (horizon_scan <= 0) {
    p->ang_res_x = 0.09;
    p->horizon_scan = (int32_t)(360.0 / p->ang_res_x + 0.5);
  } else {
    p->horizon_scan = horizon_scan;
    p->ang_res_x = 360.0 / (double)horizon_scan;
  }
  if (n_scan == 64) {
    p->ang_res_y = 26.8 / 63.0;
    p->ang_bottom = 24.8;
    p->ground_scan_id = 50;
  } else {
    p->ang_res_y = 2.0;
    p->ang_bottom = 15.0;
    p->ground_scan_id = 10;
  }
  p->laser_type = ALEGO_LASER_UNIFORM;
  p->sensor_mount_ang = 0.0;
  p->ground_angle_thres = 10.0;
  p->seg_alpha_x = p->ang_res_x / 180.0 * kPi; /* ANGLE2RAD, utility.h:48 */
  p->seg_alpha_y = p->ang_res_y / 180.0 * kPi;
  p->seg_theta = 1.047;
  p->seg_valid_point_num = 5;
  p->seg_valid_line_num = 3;
  p->seg_big_num = 30;
  p->near_filter = 0;
  p->near_thres = 1.0;
  p->occl_col_diff = 10;
  p->occl_depth = 0.5;
  p->parallel_ratio = 0.02;
  p->occl_f32 = 0;
  p->n_sectors = 6;
  p->sector_formula = 0;
  p->edge_thres = 0.1;
  p->surf_thres = 0.1;
  p->n_sharp = 2;
  p->n_less_sharp = 20;
  p->n_flat = 4;
  p->suppress_radius = 5;
  p->suppress_col_diff = 10;
  p->less_flat_leaf = 0.4f;
  p->sort_mode = 0;
  p->deskew_mode = 0;
  p->scan_period = 0.2;
  p->kf_cap_surf = 0;
  p->kf_cap_outlier = 0;
  p->nearest_feature_dist = 25.0;
  p->ring_window = 2;
  p->huber_delta = 0.1;
  p->lo_min_corr = 10;
  p->lo_iters_surf = 5;
  p->lo_iters_corner = 5;
  p->lm_leaf_corner = 0.4f;
  p->lm_leaf_surf = 0.8f;
  p->lm_leaf_outlier = 1.0f;
  p->min_keyframe_dist = 1.0;
  p->recent_keyframe_num = 50;
  p->lm_every = 2;
  p->lm_outer_iters = 2;
  p->lm_max_iters = 20;
  p->knn_max_dist = 1.0;
  p->line_ratio = 3.0;
  p->line_half_len = 0.1;
  p->plane_tol = 0.2;
  p->lm_min_corner = 10;
  p->lm_min_surf = 100;
  p->lm_min_map_corner = 10;
  p->lc_search_radius = 20.0;
  p->lc_search_num = 25;
  p->lc_fitness_max = 0.4;
  p->lc_leaf = 1.0f;
  p->lc_min_time_gap = 30.0;
  p->icp_max_corr_dist = 100.0;
  p->icp_max_iters = 100;
  p->icp_trans_eps = 1e-6;
  p->icp_fitness_eps = 1e-6;
  p->input_is_dense = 0;
}

/* PointXYZI as it crosses the boundary: the first 16 bytes-worth of the PCL
 * point (x,y,z at 0/4/8, intensity at 16 in PCL's 32-byte layout) packed to 16 B. */
typedef struct alego_point {
  float x, y, z, intensity;
} alego_point;

#ifdef __cplusplus
}
#endif
#endif /* ALEGO_PARAMS_H_ */
