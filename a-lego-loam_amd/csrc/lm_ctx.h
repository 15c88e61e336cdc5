// lm_ctx.h — device-side view of the LaserMapping state of one handle.
#ifndef ALEGO_LM_CTX_H_
#define ALEGO_LM_CTX_H_
#include "dev_common.h"

enum {
  LI_NKF = 0,      // key frames saved so far (cloud_keyposes_3d_->size())
  LI_DIRTY,        // key-frame set changed since the map was last voxel-filtered
  LI_RUN,          // mapping body runs for this scan (laserMapping.cpp:107-112)
  LI_REBUILD,      // map concat + VoxelGrid + grid rebuild this frame
  LI_FRAME,        // frame_cnt (laserMapping.cpp:111)
  LI_FLAGS,        // ALEGO_FLAG_LM_* of the last call
  LI_NCC, LI_NSC,  // corner / surf correspondences
  LI_SUM0, LI_SUM1,  // packed summaries of the two ceres::Solve calls
  LI_KF_ADDED, LI_OPTIMIZED,
  LI_KRAW_C, LI_KRAW_S, LI_KDS_C, LI_KDS_S,
  LI_NIN_C, LI_NIN_S, LI_NIN_O,          // staged inputs (/corner_last, /surf_last, /outlier)
  LI_NCUR_C, LI_NCUR_S, LI_NCUR_O,       // laser_corner_ds_, laser_surf_ds_, laser_outlier_ds_
  LI_NTOTAL, LI_NTOTAL_DS,               // laser_surf_total_, laser_surf_total_ds_
  LI_NREBUILD,     // map rebuilds so far (bench: expected work of the map VoxelGrid kernels)
  LI_OVERFLOW,     // a capacity was exceeded (clouds truncated): reported as an error by the host
  LI_REC_CNT,      // recent_*_keyframes_.size() (laserMapping.cpp:208)
  LI_LATEST,       // latest_frame_id_ (laserMapping.cpp:226), -1 until the deque has been full once
  LI_COUNT = 32
};
enum {
  LD_PARAMS = 0,     // params_[6] (absolute map pose)
  LD_T_M2O = 6, LD_Q_M2O = 9,    // map -> odom
  LD_T_O2L = 13, LD_Q_O2L = 16,  // odom -> laser (last /odom/lidar)
  LD_T_M2L = 20, LD_Q_M2L = 23,  // map -> laser
  LD_PARAMS_IT = 27,             // params_ after each outer iteration [2][6]
  LD_COSTS = 39,                 // initial/final cost of both solves
  LD_COUNT = 48
};

struct GridGeom { float ox, oy, oz, inv; int gx, gy, gz, ncell; };

struct LmCtx {
  int K;                         // recent_keyframe_num
  int kf_cap_c, kf_cap_s, kf_cap_o;
  int in_cap_c, in_cap_s, in_cap_o;
  int map_cap_c, map_cap_s;      // K*kf_cap_c, K*(kf_cap_s+kf_cap_o)
  int total_cap;                 // kf_cap_s + kf_cap_o
  int gcap;                      // grid cells capacity per map
  int qcap;                      // residual rows capacity: kf_cap_c + total_cap
  int* li;                       // [slot][LI_COUNT]
  double* ld;                    // [slot][LD_COUNT]
  // staged inputs
  float4 *in_corner, *in_surf, *in_outl;          // [slot][in_cap_*]
  // key-frame ring (clouds already transformed into the map frame, laserMapping.cpp:216-218)
  float4 *kf_corner, *kf_surf, *kf_outl;          // [slot][K][kf_cap_*]
  // the same key frames as saveKeyFramesAndFactor stores them (sensor frame, :553-555): host read-back + pose correction
  float4 *kf_raw_c, *kf_raw_s, *kf_raw_o;         // [slot][K][kf_cap_*]
  int* rec;                                       // [slot][K] frame ids held by recent_*_keyframes_, front first
  int* kf_cnt;                                    // [slot][K][4]
  float* kf_pose;                                 // [slot][K][8]  x y z roll pitch yaw (PointXYZIRPYT f32)
  // local map
  float4 *map_corner_raw, *map_surf_raw;          // [slot][map_cap_*]
  float4 *map_corner_ds, *map_surf_ds;
  // current scan
  float4 *cur_corner_ds, *cur_surf_ds, *cur_outl_ds, *cur_total, *cur_total_ds;
  // uniform grid over the down-sampled maps (index 0 corner, 1 surf)
  GridGeom* grid;                                 // [slot][2]
  int *cell_start, *cell_cur;                     // [slot][2][gcap+1]
  float4* cell_pts;                               // [slot][2][map_cap_s] map points in cell order (w = index in the ds map)
  const unsigned* vox_bbox;                       // VoxCtx::bbox of round 1 of this stream group (the map VoxelGrid context: jobs (slot-vox_slot0)*2 + {0,1})
  int vox_slot0;                                  // first slot of the stream group
  int* knn;                                       // [slot][qcap][5] neighbour indices of every query (lm_knn -> lm_fit)
  // residual blocks
  double* blocks;                                 // [slot][qcap][8]: a/normal (3), b (3), d, type (0 = none)
  double* crows;                                  // [slot][qcap][10]: the accepted rows only, packed for lm_solve: the 8 doubles above + the query point (float4)
};

#endif
