// dev_common.h — device-side view of one handle's HBM-resident state.
//
// Every per-scan array is laid out [slot][capacity] (slot = independent stream),
// so one launch with blockIdx.y = slot advances all streams together.
#ifndef ALEGO_DEV_COMMON_H_
#define ALEGO_DEV_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/alego_params.h"

// per-slot integer scalars (DevCtx::scal, stride SC_COUNT)
enum {
  SC_FIRST = 0,   // smallest index of a valid input point (orientation, imageProjection.cpp:62)
  SC_LAST,        // largest index of a valid input point
  SC_PVALID,      // valid input points
  SC_M,           // segmented cloud size
  SC_NOUT,        // outlier cloud size
  SC_NFEAS,       // feasible segments (label_cnt_-1)
  SC_LO_INIT,     // system_initialized_ (laserOdometry.cpp:36)
  SC_LO_NSURF,    // surf correspondences of the last scan
  SC_LO_NCORNER,  // corner correspondences
  SC_LO_FLAGS,    // ALEGO_FLAG_* of the last LO step
  SC_LO_ITERS,    // packed solver summaries (surf: it | succ<<8 | term<<16 ; corner <<... in next)
  SC_LO_ITERS2,
  SC_CUR,         // feature double-buffer index holding the features of the last COMPLETED LO step;
                  // FE/LO of the scan in flight write/read buffer SC_CUR^1, lo_solve(phase 1) flips it
  SC_ODOM_VALID,
  SC_LM_FRAME,    // LaserMapping frame_cnt (laserMapping.cpp:111)
  SC_LM_FLAGS,
  SC_PVALID_OUT,  // valid input points of the last projected scan (SC_PVALID is an accumulator, cleared by ip_front)
  SC_FE_EPOCH,    // feature-extraction launches of this slot so far (fe_front increments it; tags the ring counts fe_ring_out's workgroups publish to each other)
  SC_FE_ERR,      // != 0: a workgroup of fe_ring_out gave up waiting for the counts of the rings below it; the slot's less_flat cloud is then treated as EMPTY by
                  // everything that reads it (lo_grid_build, lo_assoc, lm_stage) and the host gets ALEGO_ERR_HIP (fetch_pose).  Sticky until the host clears it
                  // with alego_debug_set_option("ALEGO_FE_ERR_CLEAR", slot) (-1: every slot) — a slot that gave up once keeps failing loudly, never silently
  SC_FE_TICKET,   // fe_ring_out: rings of this launch handed out so far (a workgroup's ring = its ticket, so the rings it waits for belong to workgroups that
                  // are already running whatever order the dispatchers place them in; fe_pickc resets it)
  SC_M_DSK,       // points lo_deskew wrote into seg_dsk (/undistorted): its own count, because ImageProjection of the NEXT scan may rewrite SC_M before a host fetches the cloud
  SC_COUNT = 32
};

#define IPB_NM 9
#define IP_OWNER_TAG 0x40000000   // any tagged entry beats every plain one (index or -1) in ip_project's atomicMax

// feature cloud kinds
enum { F_SHARP = 0, F_LSHARP = 1, F_FLAT = 2, F_LFLAT = 3 };

struct DevCtx {
  alego_params P;
  int n_slots, ring_len;
  int slot0, n_launch;  // slots [slot0, slot0+n_launch) are advanced by a launch (blockIdx.y + slot0)
  int N, H, NS;   // cells, columns, rings
  int Pcap;       // input points capacity per scan (= N)
  int cap_sharp, cap_lsharp, cap_flat;  // per-ring staging capacities: n_sharp*n_sectors, ...
  double sin_ax, cos_ax, sin_ay, cos_ay;  // sin/cos of seg_alpha_x / seg_alpha_y (host libm)
  double inv_res_x, inv_res_y;            // 1 / ang_res_x, 1 / ang_res_y (projection shortcut)
  // ip_project fast path: (cos, sin) of the cell-boundary angles, rows r = -3 .. NS+2 (index r + 3) and raw columns
  // c = ip_cmin .. ip_cmin + ip_ncb - 1; ip_fast bit 0 = rows usable (uniform laser), bit 1 = columns usable (H * ang_res_x == 360)
  const double2* ip_rowtab;
  const double2* ip_coltab;
  const double* ip_colfrac;               // [H] col / 10000.0 (host fp64 division): the fractional part of a segmented point's intensity (imageProjection.cpp:101)
  int ip_cmin, ip_ncb, ip_fast;
  unsigned h_magic;                       // floor(2^32 / H) + 1: cell / H == umulhi(cell, h_magic) for every cell < 2^32 / H
  double tan_g_lo, tan_g_hi;              // tan of (sensor_mount_ang -/+ ground_angle_thres) (ground test shortcut; NaN disables)
  double tan_theta;                       // tan(seg_theta) for the edge predicate shortcut; NaN disables the shortcut
  // ---- kernel-variant switches (read once from the environment by alego_create, alego_debug_set_option overrides) ----
  int opt_ip_fused;     // ALEGO_IP_FUSED   1: ImageProjection as one launch, one workgroup per stream (ip_fused; <= 16 rings, <= 32768 cells); 0: ip_project + ip_front + cc_*
  int opt_ip_half;      // ALEGO_IP_HALF    1: ip_fused_h (512 threads, 2 N + 8 H bytes of LDS: two workgroups per CU) where it applies; 0: ip_fused
  int opt_cc_fused;     // ALEGO_CC_FUSED   1: cc_lds16 also compacts; 0: ip_rowcount + ip_compact
  int opt_cc_tile;      // ALEGO_CC_TILE    1: images beyond the LDS paths are labelled band by band in LDS (cc_tile + cc_seam); 0: cc_runs + cc_link
  int opt_fe_pick1;     // ALEGO_FE_PICK1   1: one ring per wavefront (fe_pick) instead of fe_pick4
  int opt_fe_fused;     // ALEGO_FE_FUSED   1: feature extraction as fe_front + fe_ring_out (kernels_fe2.hip); 0: fe_curv + fe_pick* + fe_voxel + fe_collect
  int opt_fe_cand;      // ALEGO_FE_CAND    sharp candidates of a ring sector kept in LDS by fe_front (0: by geometry; the tests set 8 to drive the overflow path)
  int opt_lo_box_lds;   // ALEGO_LO_BOX_LDS boxes staged in LDS by lo_assoc (0: straight from HBM)
  int opt_map_merge;    // ALEGO_MAP_MERGE  1: local map from the pre-sorted key frames; 0: concat + radix VoxelGrid
  // ---- input ring ----
  float4* in_pts;  // [slot][ring][Pcap]
  int* in_n;       // [slot][ring]
  // ---- bag store (alego_replay_*): recorded streams shared by the slots; slot s replays bag bag_src[s].x from scan
  // bag_src[s].y on, cyclically.  A launch with replay_bag != 0 reads it; its `ring_pos` is then the step index.
  const float4* bag_pts;  // [bag][bag_len][Pcap]
  const int* bag_n;       // [bag][bag_len]
  const int2* bag_src;    // [slot]
  int n_bags, bag_len, replay_bag;
  // ---- look-ahead lanes (alego_stream_run): ImageProjection + feature extraction carry no state across scans, so a single stream
  // runs them for several scans ahead in one launch, each scan in a "lane" (a slot used only for its per-scan buffers).  LaserOdometry /
  // LaserMapping of the stream's own slot then take the features, outliers and odometry hand-over of the scan in flight from lane
  // fs_cur and the previous scan's features from lane fs_last (-1: the slot's own double buffers, the normal batch / single-scan mode).
  int fs_cur, fs_last;
  // ---- image projection ----
  int* owner;           // [slot][N] winning input index per cell (last writer = max index), -1 empty; ip_project writes
                        // IP_OWNER_TAG | index; ip_front (and, for its workgroups' first columns, the kernel after it) turns every
                        // cell back into the plain form (= the reset for the next scan)
  float* range_img;     // [slot][N] f32 range, -1 empty
  uint8_t* flag_img;    // [slot][N] bit0 ground, bit1 active (filled, non-ground), bit2 edge->right, bit3 edge->down
  int* parent;          // [slot][N] union-find parent (root = min linear index of the component)
  int* cc_size;         // [slot][N] per-root size, later per-root label
  unsigned long long* cc_rows;  // [slot][N] per-root row bitmask
  unsigned long long* ipb_col;   // [slot][IPB_NM][H] the banded path's 64-bit row masks of every column (kernels_ipb.hip): ground, active, down-edges, right-edges, band roots |
                                 // keep, outlier, feasible roots, final roots (null: not allocated — sensors of <= 16 rings)
  int* ipb_off;         // [slot][3][64][chunks of 64 columns] row-major exclusive offsets of the ordered compaction: kept cells, outliers, feasible roots
  int opt_ip_band;      // ALEGO_IP_BAND    1: sensors of 17 - 64 rings take the banded mask path (ip_project + ipb_band + ipb_merge + ipb_emit); 0: ip_front + cc_* + ip_rowcount + ip_compact
  unsigned* ipf_own;    // [slot][N / 2] ip_fused_h: the packed 16-bit owners of two adjacent columns between its phases B and D (null: not allocated)
  int* label_img;       // [slot][N] label_mat_
  int* cc_label;        // [slot][N] per-root label_cnt_ number (0 = infeasible)
  int* row_cnt;         // [slot][NS][4] per-row kept / outlier / feasible-root counts
  int* scal;            // [slot][SC_COUNT]
  float4* seg_pts;      // [slot][N]
  uint8_t* seg_ground;  // [slot][N]
  int* seg_col;         // [slot][N]
  float* seg_range;     // [slot][N]
  int* ring_start;      // [slot][NS]
  int* ring_end;        // [slot][NS]
  float* ori;           // [slot][4]
  float4* outlier;      // [slot][N]
  // ---- feature extraction ----
  float* cd;            // [slot][N] f32 11-tap sum (curvature = (double)cd^2)
  uint8_t* picked0;     // [slot][N] cloud_neighbor_picked_ after occlusion marking (single-scan entry points / tests only)
  uint8_t* fe_flag;     // [slot][N] per-point flag byte of the feature pick (fe_common.h)
  int* plabel;          // [slot][N] cloud_label_
  int* st_idx;          // [slot][NS][st_stride] per-ring staging: sharp | less_sharp | flat | less_flat_scan indices
  int* st_cnt;          // [slot][NS][8]  counts: sharp, less_sharp, flat, less_flat_scan, less_flat voxels
  float4* st_lfds;      // [slot][NS][H] per-ring voxel-filtered less_flat
  int st_stride;        // cap_sharp + cap_lsharp + cap_flat + H
  // ---- feature clouds, double-buffered by scan parity ----
  float4* feat[4];      // [slot][2][fcap[k]]
  int* feat_idx[3];     // [slot][2][fcap[k]] indices into the segmented cloud (sharp, less_sharp, flat)
  int fcap[4];
  int* feat_cnt;        // [slot][2][4]
  int* ring_off;        // [slot][2][2][NS+1] ring offsets of less_sharp ([..][0]) and less_flat ([..][1])
  int* ring_boff;       // [slot][2][2][NS+1] the same for their bounding boxes: ring r owns the boxes [ring_boff[r], ring_boff[r+1]) of lo_box
  unsigned* fe_sync;    // [slot][NS] fe_ring_out: (SC_FE_EPOCH << 16 | less_flat voxels of the ring), published for the rings above
  // ---- laser odometry ----
  int* lo_corr;         // [slot][qcap][4]  surf rows then corner rows: (query, closest, idx2, idx3) ; closest<0 = none
  int lo_qcap_surf, lo_qcap_corner;
  float4* lo_box;       // [slot][2 buffers][2 kinds][lo_box_cap][2]: min / max corner of up to LO_CH consecutive targets of one ring; .w of the
  // the previous scan's target clouds once more, sorted by the cells of a 2-D (x, y) grid of >= 1 m (lo_grid_build, kernels_lo.hip): the exact 1-NN of almost
  // every query is settled by the 3 x 3 cells around it; the boxes remain for the ring walks and for the queries whose neighbour is farther than a cell
  float4* lo_cpts[2];   // [slot][2 buffers][fcap]: kind 0 less_flat, 1 less_sharp; .w = the target's index (int bits)
  unsigned short* lo_cell;   // [slot][2 buffers][2 kinds][LO_GC + 2]: first sorted point of every cell (exclusive prefix, row-major in x)
  float* lo_geom;       // [slot][2 buffers][2 kinds][8]: origin x, y, 1 / cell size, settle threshold (0.999 cell^2), gx, gy (int bits; gx = 0: no grid), -, -
  int opt_lo_grid;      // ALEGO_LO_GRID    1: lo_assoc takes the 1-NN from the grid where that is exact; 0: boxes only
  int opt_fo_spin;      // ALEGO_FE_SPIN    development: polls a ring of fe_ring_out waits for a lower ring's count (0: FO_SPIN_LIMIT; the tests set 1 to drive the give-up path)
  int opt_fo_pad8;      // ALEGO_FE_PAD8    development: 1 pads fe_ring_out's grid to a multiple of 8 streams (a stream's rings on one XCD, round 4's work-around; 0 = default since the ring tickets)
  int lo_box_cap;       //   corners = first target / number of targets (int bits); kind 0: less_flat, 1: less_sharp; written by feature extraction
  double* lo_state;     // [slot][LO_STATE_N]
  // ---- motion de-skew (adjustDistortion, laserOdometry.cpp:557-726; alego_params.deskew_mode) ----
  double* imu_ring;     // [slot][ALEGO_IMU_Q][10]: time, roll, pitch, yaw, shift xyz, velo xyz (imu_time_ ... imu_velo_z_)
  int* imu_ptr;         // [slot][4]: imu_ptr_last_, imu_ptr_front_, imu_ptr_last_iter_
  double* scan_stamp;   // [slot] stamp of the segmented cloud being processed (t1, :111)
  float4* seg_dsk;      // [slot][N] the LaserOdometry's own copy of the segmented cloud after adjustDistortion (/undistorted)
  const float4* seg_lo; // what feature extraction reads: seg_dsk when deskew_mode != 0, else seg_pts
  // ---- outputs ----
  double* poses;        // [slot][16]: odom t(3) q(4), map t(3) q(4), pad
  double* traj;         // [slot][traj_cap][14] per-scan log of `poses` (alego_trajectory_enable), else null
  int* traj_n;          // [slot] scans logged so far (keeps counting past traj_cap; entries beyond it are dropped)
  int traj_cap;
};
#define ALEGO_IMU_Q 200   // imu_queue_length, utility.h:70

enum {
  LS_PARAMS = 0,        // params_[6]
  LS_TW = 6,            // t_w_cur_[3]
  LS_RW = 9,            // r_w_cur_[9] row-major
  LS_PARAMS_SURF = 18,  // params_ after the surf solve (debug)
  LS_COSTS = 24,        // initial/final cost of both solves
  LS_ROT = 32,          // rotation matrix of params_ (row-major), cached for transformToStart ...
  LS_ROT_P = 41,        // ... and the params_ it was computed from (recomputed by lo_assoc when they differ)
  LO_STATE_N = 48
};

#define DEV_INLINE __device__ __forceinline__

#ifndef LO_CH
#define LO_CH 32
#endif
#ifndef LO_GC
#define LO_GC 4096
#endif
// LO_GC: cells of the LaserOdometry target grid (64 x 64; the cell size starts at 1 m and doubles until the cloud's bounding box fits: 2 m for a 100 m scene.  16384 cells
// were measured: the 1 m cells halve lo_assoc's candidates, but lo_grid_build's scan and table are four times as long: 418 k against 425 k scans/s; 1024 cells: 420 k)
// LO_CH: targets per bounding box of the LaserOdometry 1-NN / ring-walk pruning

// the input scan of `slot` at ring position / replay step `pos`
DEV_INLINE size_t scan_slot(const DevCtx& d, int slot, int pos) {
  if (d.replay_bag) { const int2 s = d.bag_src[slot]; return (size_t)s.x * d.bag_len + (unsigned)(s.y + pos) % (unsigned)d.bag_len; }
  return (size_t)slot * d.ring_len + pos;
}
DEV_INLINE const float4* scan_pts(const DevCtx& d, int slot, int pos) { return (d.replay_bag ? d.bag_pts : d.in_pts) + scan_slot(d, slot, pos) * d.Pcap; }
DEV_INLINE int scan_count(const DevCtx& d, int slot, int pos) { return (d.replay_bag ? d.bag_n : d.in_n)[scan_slot(d, slot, pos)]; }

// row of a cell index without an integer division (exact for v < 2^32 / H, i.e. for every supported image)
DEV_INLINE int cell_row(const DevCtx& d, int v) { return (int)__umulhi((unsigned)v, d.h_magic); }

// buffer written by the scan in flight (valid from fe_collect until lo_solve phase 1 flips SC_CUR)
DEV_INLINE int cur_in_flight(const DevCtx& d, int slot) { return d.scal[slot * SC_COUNT + SC_CUR] ^ 1; }
// index into the [slot][2] feature arrays of the scan in flight / of the previous scan, as LaserOdometry sees them.  A lane never
// runs lo_solve, so its SC_CUR stays at its initial 1 and feature extraction always fills its buffer 0.
DEV_INLINE size_t fidx_cur(const DevCtx& d, int slot) { return d.fs_cur >= 0 ? (size_t)d.fs_cur * 2 : (size_t)slot * 2 + cur_in_flight(d, slot); }
DEV_INLINE size_t fidx_last(const DevCtx& d, int slot) { return d.fs_last >= 0 ? (size_t)d.fs_last * 2 : (size_t)slot * 2 + (cur_in_flight(d, slot) ^ 1); }
// slot whose per-scan outputs (poses, outlier cloud, outlier count) belong to the scan in flight
DEV_INLINE int scan_slot_of(const DevCtx& d, int slot) { return d.fs_cur >= 0 ? d.fs_cur : slot; }

DEV_INLINE int32_t d_f2i(float f) { return __float_as_int(f); }
DEV_INLINE float d_i2f(int32_t i) { return __int_as_float(i); }

// ---------------------------------------------------------------------------
// atan2f / atanf: the fdlibm algorithm glibc 2.35 uses for std::atan2(float,float)
// (what imageProjection.cpp:79,87 resolves to), written for the device with plain
// IEEE f32 + - * / only.  Compiled with -ffp-contract=off.  Algorithm and
// constants: Sun fdlibm e_atan2f.c / s_atanf.c, "Copyright (C) 1993 by Sun
// Microsystems, Inc. All rights reserved. ... Permission to use, copy, modify,
// and distribute this software is freely granted, provided that this notice is
// preserved."
// ---------------------------------------------------------------------------
DEV_INLINE float d_atanf(float x) {
  const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
  const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
  const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
              aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
              aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
  int32_t hx = d_f2i(x), ix = hx & 0x7fffffff;
  float ahi = 0.f, alo = 0.f;
  int id;
  if (ix >= 0x4c800000) {
    if (ix > 0x7f800000) return x + x;
    return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
  }
  if (ix < 0x3ee00000) {
    if (ix < 0x31000000) return x;
    id = -1;
  } else {
    x = fabsf(x);
    if (ix < 0x3f980000) {
      if (ix < 0x3f300000) { id = 0; ahi = hi0; alo = lo0; x = (2.0f * x - 1.0f) / (2.0f + x); }
      else { id = 1; ahi = hi1; alo = lo1; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000) { id = 2; ahi = hi2; alo = lo2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
      else { id = 3; ahi = hi3; alo = lo3; x = -1.0f / x; }
    }
  }
  float z = x * x, w = z * z;
  float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return x - x * (s1 + s2);
  z = ahi - ((x * (s1 + s2) - alo) - x);
  return hx < 0 ? -z : z;
}

DEV_INLINE float d_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
              pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  int32_t hx = d_f2i(x), ix = hx & 0x7fffffff, hy = d_f2i(y), iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return d_atanf(y);
  int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) { return m == 0 || m == 1 ? y : (m == 2 ? pi + tiny : -pi - tiny); }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
    return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  int32_t k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = d_atanf(fabsf(y / x));
  if (m == 0) return z;
  if (m == 1) return d_i2f(d_f2i(z) ^ (int32_t)0x80000000);
  if (m == 2) return pi - (z - pi_lo);
  return (z - pi_lo) - pi;
}

// glibc 2.35 hypotf: evaluated in double, rounded once
DEV_INLINE float d_hypotf(float x, float y) { return (float)sqrt((double)x * (double)x + (double)y * (double)y); }

// ---------------------------------------------------------------------------
// sinf / cosf of glibc >= 2.28 (what Eigen's Quaternionf(AngleAxisf) in transformPointCloud,
// laserMapping.h:166-173, calls on x86-64 Linux): fast quadrant reduction, fp64 minimax polynomial, one
// rounding.  Not the correctly rounded value (1.3 % of the inputs differ), so the algorithm itself is
// shared with the parity checker; plain fp64 + - * only, built with -ffp-contract=off.  |x| < 120 (a key
// pose's half angles lie in [-pi/2, pi/2]); beyond that the correctly rounded value is returned.
// ---------------------------------------------------------------------------
DEV_INLINE uint32_t d_abstop12(float x) { return ((uint32_t)d_f2i(x) >> 20) & 0x7ffu; }
DEV_INLINE float d_sincos_poly(double x, double x2, bool neg, int n) {
  if ((n & 1) == 0) {
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    const double x3 = x * x2, s1 = s2c + x2 * s3c, x7 = x3 * x2, s = x + x3 * s1c;
    return (float)(s + x7 * s1);
  }
  const double sg = neg ? -1.0 : 1.0;   // the second table of glibc holds the negated cosine coefficients (exact sign flips)
  const double c0 = sg * 0x1p0, c1c = sg * -0x1.ffffffd0c621cp-2, c2c = sg * 0x1.55553e1068f19p-5, c3c = sg * -0x1.6c087e89a359dp-10,
               c4c = sg * 0x1.99343027bf8c3p-16;
  const double x4 = x2 * x2, c2 = c3c + x2 * c4c, c1 = c0 + x2 * c1c, x6 = x4 * x2, c = c1 + x4 * c2c;
  return (float)(c + x6 * c2);
}
DEV_INLINE float d_sincosf(float y, int is_cos) {
  double x = (double)y;
  if (d_abstop12(y) < d_abstop12(0x1.921FB6p-1f)) {
    if (d_abstop12(y) < d_abstop12(0x1p-12f)) return is_cos ? 1.0f : y;
    return d_sincos_poly(x, x * x, false, is_cos);
  }
  if (d_abstop12(y) < d_abstop12(120.0f)) {
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * 0x1.921FB54442D18p0;
    const int q = n + is_cos;
    const double s = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0;
    return d_sincos_poly(x * s, x * x, (q & 2) != 0, n ^ is_cos);
  }
  return is_cos ? (float)cos((double)y) : (float)sin((double)y);
}
DEV_INLINE float d_sinf(float y) { return d_sincosf(y, 0); }
DEV_INLINE float d_cosf(float y) { return d_sincosf(y, 1); }

// ---------------------------------------------------------------------------
// wavefront (64 lanes) helpers
// ---------------------------------------------------------------------------
DEV_INLINE double wave_sum_f64(double v) {  // fixed butterfly order -> deterministic
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
DEV_INLINE int lane_id() { return threadIdx.x & 63; }

// Wave-wide max / min of a u32 returned as a uniform value: four DPP steps inside each 16-lane row
// (quad_perm xor1, xor2, row_half_mirror, row_mirror — valid because max/min are idempotent), then the four
// row results through v_readlane.  ~12 instructions instead of 6 dependent ds_bpermute round trips.
DEV_INLINE uint32_t wave_max_u32(uint32_t v) {
  int x = (int)v, t;
  t = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;   // quad_perm [1,0,3,2]
  t = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;   // quad_perm [2,3,0,1]
  t = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;  // row_half_mirror
  t = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;  // row_mirror
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane(x, 0), b = (uint32_t)__builtin_amdgcn_readlane(x, 16);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane(x, 32), e = (uint32_t)__builtin_amdgcn_readlane(x, 48);
  const uint32_t ab = a > b ? a : b, ce = c > e ? c : e;
  return ab > ce ? ab : ce;
}
DEV_INLINE uint32_t wave_min_u32(uint32_t v) { return ~wave_max_u32(~v); }
// u64 versions: high word first, then the low word among the lanes that hold the winning high word
DEV_INLINE unsigned long long wave_max_u64(unsigned long long v) {
  const uint32_t hi = wave_max_u32((uint32_t)(v >> 32));
  const uint32_t lo = wave_max_u32((uint32_t)(v >> 32) == hi ? (uint32_t)v : 0u);
  return ((unsigned long long)hi << 32) | lo;
}
DEV_INLINE unsigned long long wave_min_u64(unsigned long long v) {
  const uint32_t hi = wave_min_u32((uint32_t)(v >> 32));
  const uint32_t lo = wave_min_u32((uint32_t)(v >> 32) == hi ? (uint32_t)v : 0xFFFFFFFFu);
  return ((unsigned long long)hi << 32) | lo;
}

// 16-lane (one DPP row) versions: every lane of the row gets the result; no cross-row traffic at all
DEV_INLINE uint32_t row16_max_u32(uint32_t v) {
  int x = (int)v, t;
  t = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;
  t = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;
  t = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;
  t = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x;
  return (uint32_t)x;
}
DEV_INLINE uint32_t row16_min_u32(uint32_t v) { return ~row16_max_u32(~v); }
DEV_INLINE unsigned long long row16_min_u64(unsigned long long v) {
  const uint32_t hi = row16_min_u32((uint32_t)(v >> 32));
  const uint32_t lo = row16_min_u32((uint32_t)(v >> 32) == hi ? (uint32_t)v : 0xFFFFFFFFu);
  return ((unsigned long long)hi << 32) | lo;
}

#endif
