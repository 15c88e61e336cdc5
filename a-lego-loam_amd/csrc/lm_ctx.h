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
  LI_NU_C, LI_NU_S,      // occupied voxels of the corner / surf local map (size of the sorted voxel-key lists U)
  LI_UVALID,       // U / Ucnt describe the window in rec_prev (cleared by everything that changes key frames behind their back)
  LI_PREV_CNT,     // entries of rec_prev
  LI_KF_PENDING,   // a key frame was written to kf_tmp_* and still has to be sorted into ring entry LI_KF_PEND_RING
  LI_KF_PEND_RING,
  LI_TMPN_C, LI_TMPN_S,  // points in kf_tmp_c / kf_tmp_s
  LI_REBUILD_FB,   // LI_REBUILD on the concat + radix VoxelGrid path (ALEGO_MAP_MERGE=0)
  LI_MAP_PASS,     // bit m: map m took PCL's "leaf size too small" path in map_update (output = input), map_accum skips it
  LI_SORT_N,       // (2 entries) point counts written by the key-frame sort jobs
  LI_SORT_N1,
  LI_COUNT = 48
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

// Packed voxel key of a point: PCL's VoxelGrid orders the output by idx = i + j dx + k dx dy with (i, j, k) the integer voxel
// coordinates relative to the cloud's bounding box, i.e. lexicographically by (floor(z inv), floor(y inv), floor(x inv)) — an order that
// does not depend on the bounding box.  21 bits per axis (|coordinate| < 2^20 voxels: 419 km at a 0.4 m leaf).
DEV_INLINE unsigned long long vkey_pack(int ix, int iy, int iz) {
  const int B = 1 << 20;
  const unsigned long long x = (unsigned long long)(unsigned)min(max(ix + B, 0), 2 * B - 1), y = (unsigned long long)(unsigned)min(max(iy + B, 0), 2 * B - 1),
                           z = (unsigned long long)(unsigned)min(max(iz + B, 0), 2 * B - 1);
  return (z << 42) | (y << 21) | x;
}
DEV_INLINE unsigned long long vkey_of(const float4& p, float inv) {   // floor(p * inverse_leaf_size) as pcl::VoxelGrid computes it (f32)
  return vkey_pack((int)floorf(p.x * inv), (int)floorf(p.y * inv), (int)floorf(p.z * inv));
}
DEV_INLINE unsigned vbox_enc(float f) {   // order-preserving u32 code of a float (the VoxelGrid kernels' bounding-box format)
  const unsigned b = (unsigned)__float_as_int(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// f32 4x4 of transformPointCloud (laserMapping.h:166-173): AngleAxisf(yaw,Z)*AngleAxisf(pitch,Y)*AngleAxisf(roll,X).
// sin/cos of the half angles are glibc's sinf / cosf (dev_common.h), as Eigen's Quaternionf(AngleAxisf) calls them.
DEV_INLINE void keypose_matrix(const float* kp, float m[3][4]) {
  const float hz = 0.5f * kp[5], hy = 0.5f * kp[4], hx = 0.5f * kp[3];
  const float qz[4] = {d_cosf(hz), 0.f, 0.f, d_sinf(hz)};
  const float qy[4] = {d_cosf(hy), 0.f, d_sinf(hy), 0.f};
  const float qx[4] = {d_cosf(hx), d_sinf(hx), 0.f, 0.f};
  float t[4], q[4];
  t[0] = qz[0] * qy[0] - qz[1] * qy[1] - qz[2] * qy[2] - qz[3] * qy[3];
  t[1] = qz[0] * qy[1] + qz[1] * qy[0] + qz[2] * qy[3] - qz[3] * qy[2];
  t[2] = qz[0] * qy[2] + qz[2] * qy[0] + qz[3] * qy[1] - qz[1] * qy[3];
  t[3] = qz[0] * qy[3] + qz[3] * qy[0] + qz[1] * qy[2] - qz[2] * qy[1];
  q[0] = t[0] * qx[0] - t[1] * qx[1] - t[2] * qx[2] - t[3] * qx[3];
  q[1] = t[0] * qx[1] + t[1] * qx[0] + t[2] * qx[3] - t[3] * qx[2];
  q[2] = t[0] * qx[2] + t[2] * qx[0] + t[3] * qx[1] - t[1] * qx[3];
  q[3] = t[0] * qx[3] + t[3] * qx[0] + t[1] * qx[2] - t[2] * qx[1];
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  const float tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  m[0][0] = 1 - (tyy + tzz); m[0][1] = txy - twz; m[0][2] = txz + twy; m[0][3] = kp[0];
  m[1][0] = txy + twz; m[1][1] = 1 - (txx + tzz); m[1][2] = tyz - twx; m[1][3] = kp[1];
  m[2][0] = txz - twy; m[2][1] = tyz + twx; m[2][2] = 1 - (txx + tyy); m[2][3] = kp[2];
}

DEV_INLINE float4 kf_transform(const float m[3][4], const float4& p) {   // laserMapping.h:175 (pcl::transformPointCloud, f32)
  float4 o;
  o.x = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3];
  o.y = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3];
  o.z = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3];
  o.w = p.w;
  return o;
}

struct LmCtx {
  int K;                         // recent_keyframe_num
  int KR;                        // ring entries per slot = K + 1: frame f lives in entry f % KR, so the frame a full window pops
                                 // (f - K) is still intact when the frame that pushes it out (f) has been stored
  int kf_cap_c, kf_cap_s, kf_cap_o;
  int in_cap_c, in_cap_s, in_cap_o;
  int map_cap_c, map_cap_s;      // K*kf_cap_c, K*(kf_cap_s+kf_cap_o)
  int total_cap;                 // kf_cap_s + kf_cap_o
  int gcap;                      // grid cells capacity per map
  int qcap;                      // residual rows capacity: kf_cap_c + total_cap
  int solve_row_bytes;           // LDS lm_solve keeps its accepted rows in (ALEGO_LM_ROW_LDS overrides; rows beyond it are read from crows)
  int* li;                       // [slot][LI_COUNT]
  double* ld;                    // [slot][LD_COUNT]
  // staged inputs
  float4 *in_corner, *in_surf, *in_outl;          // [slot][in_cap_*]
  double* stage_odom;                             // [slot][2][8] /odom/lidar of a scan (t 3, q 4, valid) handed over by lm_stage when LaserMapping runs on a
                                                  // HIP stream of its own behind the front end (double-buffered by scan parity)
  // key-frame ring (clouds already transformed into the map frame, laserMapping.cpp:216-218) and SORTED by voxel key of the
  // map's leaf size (stable: input order inside a voxel): corner, and surf followed by outlier (:240-242) as one run
  float4 *kfs_c, *kfs_s;                          // [slot][KR][kf_cap_c] / [slot][KR][total_cap]
  int* kfs_n;                                     // [slot][KR][2] points per run
  float* kfs_box;                                 // [slot][KR][2][8] min xyz (0..2) / max xyz (4..6) of the run
  float4 *kf_tmp_c, *kf_tmp_s;                    // [slot][kf_cap_c] / [slot][total_cap] transformed clouds of the key frame waiting to be sorted
  // sorted list of the occupied voxels of each local map (persistent, updated incrementally when the window changes)
  unsigned long long *U_c, *U_s;                  // [slot][map_cap_*]
  int *Ucnt_c, *Ucnt_s;                           // [slot][map_cap_*] points per voxel
  int* rec_prev;                                  // [slot][K] window the lists describe (ring frame ids, front first)
  float4* newkeys;                                // [slot][2][total_cap] scratch of map_update: (key lo, key hi, lower bound, count) of the voxels a run adds
  // ---- one registration sharded over the ranks of a communicator (alego_dist_*; kernels lm_shard_*) ----
  int shard_rank, shard_world;                    // world 0: not sharded
  double* shard_part;                             // [slot][32] this rank's partial sums of an evaluation: 21 J^T J, 6 J^T r, cost, corner / surf rows; all-reduced in place
  void* shard_state;                              // [slot] LmState of the solve in flight (dev_cost.h)
  int* shard_ctl;                                 // [slot][8]: 0 local rows, 1 action of the last step, 2 solve finished, 3 outer iteration, 4 guard failed
  unsigned* map_bbox;                             // [slot][2][8] bounding box of the window's points in the VoxelGrid kernels' encoding (merge path)
  // the same key frames as saveKeyFramesAndFactor stores them (sensor frame, :553-555): host read-back + pose correction
  float4 *kf_raw_c, *kf_raw_s, *kf_raw_o;         // [slot][KR][kf_cap_*]
  int* rec;                                       // [slot][K] frame ids held by recent_*_keyframes_, front first
  int* kf_cnt;                                    // [slot][KR][4]
  float* kf_pose;                                 // [slot][KR][8]  x y z roll pitch yaw (PointXYZIRPYT f32)
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
  double* crows;                                  // [slot][qcap][10]: accepted rows that do not fit lm_solve's LDS (all rows on the sharded path): the 8 doubles above + the query point (float4)
};

#endif
