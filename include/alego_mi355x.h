/*
 * alego_mi355x.h — C ABI of the MI355X-native A-LeGO-LOAM hot path.
 *
 * The reference has no FFI: its boundary is three ROS nodelets exchanging
 * messages (nodelet_plugins.xml:1-10).  Each entry point below replaces the
 * numeric body of one nodelet callback and takes / returns exactly the payload
 * of the ROS messages that callback consumes / publishes, as plain pointers and
 * sizes.  A thin adapter keeps plugin names, topics and frames.
 *
 *   alego_ip_process   <- ImageProjection::pcCB            src/imageProjection.cpp:49-208
 *   alego_lo_process   <- LaserOdometry::mainLoop body     src/laserOdometry.cpp:111-553
 *   alego_lm_process   <- LaserMapping::mainLoop body +    src/laserMapping.cpp:112-123,
 *                         laserOdomHandler                 154-166
 *   alego_scan_process <- the three chained in one process (launch/test.launch: one
 *                         nodelet manager), intermediates stay in HBM
 *   alego_batch_*      <- bag replay at unbounded rate (README.md:33-37) over many
 *                         independent streams, inputs resident in HBM, no per-scan
 *                         host synchronisation
 *
 * All functions return 0 on success, >0 for the reference's "skip" conditions
 * (surfaced as flags, see ALEGO_FLAG_*), <0 for hard errors (ALEGO_ERR_*); nothing
 * throws across the boundary.  A handle is single-threaded (one caller at a time;
 * it owns its HIP streams); different handles are independent.  There is no CPU fallback:
 * alego_create fails with ALEGO_ERR_NO_DEVICE when no gfx950 device is visible.
 */
#ifndef ALEGO_MI355X_H_
#define ALEGO_MI355X_H_

#include <stdint.h>

#include "alego_params.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct alego_handle alego_handle;

enum {
  ALEGO_OK = 0,
  ALEGO_ERR_NO_DEVICE = -1, /* no HIP device / not gfx950 */
  ALEGO_ERR_HIP = -2,       /* a HIP call failed, see alego_last_error */
  ALEGO_ERR_CAPACITY = -3,  /* caller buffer too small */
  ALEGO_ERR_ARG = -4
};
/* bit flags returned by the *_process calls (reference guards, not errors) */
enum {
  ALEGO_FLAG_LO_INIT = 1,        /* first scan: features stored, no odometry   laserOdometry.cpp:316-324 */
  ALEGO_FLAG_FEW_SURF = 2,       /* < lo_min_corr surf correspondences         laserOdometry.cpp:422-425 */
  ALEGO_FLAG_FEW_CORNER = 4,     /* < lo_min_corr corner correspondences       laserOdometry.cpp:496-499 */
  ALEGO_FLAG_LM_SKIPPED = 8,     /* odd mapping frame                          laserMapping.cpp:112 */
  ALEGO_FLAG_LM_FEW_FEATURES = 16, /* registration guard                       laserMapping.cpp:350-354 */
  ALEGO_FLAG_LM_KEYFRAME = 32    /* a key frame was saved                      laserMapping.cpp:491-559 */
};

/* ---- messages ------------------------------------------------------------ */
/* /lslidar_point_cloud (sensor_msgs/PointCloud2 as PointXYZI) */
typedef struct alego_scan_in {
  const alego_point* pts;
  int32_t n;
  double stamp;
} alego_scan_in;

/* /segmented_cloud + /seg_info (msg/cloud_info.msg:1-12) + /outlier.
 * Caller owns every buffer; capacities are in elements; worst case is
 * n_scan*horizon_scan for seg/ground/col/range/outlier/label_image. */
typedef struct alego_seg_out {
  alego_point* seg;      int32_t seg_cap;  int32_t m;          /* segmented_cloud */
  uint8_t* ground;       /* segmentedCloudGroundFlag[m] */
  int32_t* col;          /* segmentedCloudColInd[m]     */
  float* range;          /* segmentedCloudRange[m]      */
  int32_t* ring_start;   /* startRingIndex[n_scan]      */
  int32_t* ring_end;     /* endRingIndex[n_scan]        */
  float orientation[3];  /* startOrientation, endOrientation, orientationDiff */
  alego_point* outlier;  int32_t outlier_cap;  int32_t n_outlier;
  int32_t* label_image;  /* optional (may be NULL): label_mat_ [n_scan][horizon_scan] */
  double stamp;          /* header stamp of /segmented_cloud + /seg_info = the input scan's (t1 of laserOdometry.cpp:111; read by the de-skew) */
} alego_seg_out;

/* /corner, /corner_less, /surf, /surf_less (laserOdometry.cpp:299-314); the
 * less_* clouds are also /corner_last and /surf_last (:531-546). */
typedef struct alego_feat_out {
  alego_point* sharp;       int32_t sharp_cap;       int32_t n_sharp;
  alego_point* less_sharp;  int32_t less_sharp_cap;  int32_t n_less_sharp;
  alego_point* flat;        int32_t flat_cap;        int32_t n_flat;
  alego_point* less_flat;   int32_t less_flat_cap;   int32_t n_less_flat;
  int32_t* point_label;     /* optional (may be NULL): cloud_label_[m] */
} alego_feat_out;

/* nav_msgs/Odometry pose (/odom/lidar, /odom_aft_mapped) + the 6-vector the solver works on */
typedef struct alego_pose {
  double t[3];
  double q[4];      /* w, x, y, z */
  double params[6]; /* LO: last relative transform; LM: absolute map pose (x,y,z,roll,pitch,yaw) */
  int32_t valid;
} alego_pose;

/* ---- lifecycle ------------------------------------------------------------ */
/* n_slots independent streams share one handle (batch path); the single-scan
 * entry points use slot 0.  ring_len = scans kept resident per slot for the
 * batch path (0 -> 1). */
int alego_create(const alego_params* params, int device, int n_slots, int ring_len, alego_handle** out);
void alego_destroy(alego_handle* h);
const char* alego_last_error(const alego_handle* h);
int alego_device_count(void);
/* size of alego_params this library was built with (binding sanity check) */
int alego_params_sizeof(void);
/* A handle is single-threaded: one caller at a time.  A host that drives ONE handle from several threads — the three nodelets of
 * launch/test.launch:6-10 share a process, the reference serialises them with per-node mutexes (laserOdometry.cpp:93,548;
 * laserMapping.cpp:114,729,769) — brackets every call (and every group of calls that must see a consistent handle) with this
 * recursive lock, which lives in the handle so that all users of the handle share it. */
int alego_handle_lock(alego_handle* h);
int alego_handle_unlock(alego_handle* h);

/* ---- nodelet-equivalent single-scan entry points (host buffers, slot 0) --- */
int alego_ip_process(alego_handle* h, const alego_scan_in* in, alego_seg_out* out);
/* `in` is the IP output as received on /segmented_cloud + /seg_info; `odom` is /odom/lidar */
int alego_lo_process(alego_handle* h, const alego_seg_out* in, alego_feat_out* feat, alego_pose* odom);
/* inputs: /corner_last, /surf_last, /outlier, /odom/lidar; output: /odom_aft_mapped and params_ */
int alego_lm_process(alego_handle* h, const alego_point* corner_last, int32_t n_corner,
                     const alego_point* surf_last, int32_t n_surf, const alego_point* outlier,
                     int32_t n_outlier, const alego_pose* odom, alego_pose* map_pose);
/* IP -> LO -> LM for one scan with intermediates kept on the device.  stages: bit0 IP,
 * bit1 LO, bit2 LM.  seg/feat may be NULL (then nothing but the poses is copied back). */
int alego_scan_process(alego_handle* h, int slot, const alego_scan_in* in, int stages,
                       alego_seg_out* seg, alego_feat_out* feat, alego_pose* odom, alego_pose* map_pose);

/* ---- device-resident batch path ------------------------------------------ */
/* copy one scan into ring position `ring_pos` of `slot` (host -> HBM, outside the timed region) */
int alego_batch_load(alego_handle* h, int slot, int ring_pos, const alego_point* pts, int32_t n);
/* advance every slot by n_scans scans (ring positions first_pos, first_pos+1, ... mod ring_len),
 * all kernels enqueued on the handle's streams; returns without synchronising when sync == 0.  With sync == 0 LaserMapping of the
 * last scans may still be in flight on the stream groups' second ("back") HIP streams when the call returns: every entry point that
 * touches LaserOdometry / LaserMapping state waits for it first (per-slot calls for their own group, alego_lo_process /
 * alego_lm_process / alego_stream_run / alego_dist_init / alego_dist_shutdown for all groups); alego_batch_load, which only writes the
 * input ring, does not.  A handle of G stream groups drives 2 G HIP streams: export GPU_MAX_HW_QUEUES=16 in the host process before
 * its first HIP call (the runtime's default of 4 hardware queues makes pairs of streams serialise); the library never sets it. */
int alego_batch_run(alego_handle* h, int first_pos, int n_scans, int stages, int sync);
int alego_synchronize(alego_handle* h);
/* OR into `stages` of alego_batch_run: replay the resident ring back and forth (0..R-1,R-2..0,1..)
 * instead of wrapping, so that consecutive scans of a stream are always trajectory neighbours */
#define ALEGO_REPLAY_PINGPONG 0x100
/* Bag store: n_bags recorded streams of bag_len scans each, resident in HBM once and SHARED by the slots (a 560-scan lap of a
 * 16x1800 sensor is 258 MB; a private copy per slot would not fit).  alego_replay_assign makes `slot` replay `bag` from
 * scan `start_scan` on, cyclically; alego_batch_run with ALEGO_REPLAY_BAG in `stages` then advances every slot by n_scans
 * scans of its bag: first_pos is the step index (slot s processes scan (start_scan + first_pos + i) mod bag_len at step i). */
#define ALEGO_REPLAY_BAG 0x200
int alego_replay_create(alego_handle* h, int n_bags, int bag_len);
int alego_replay_load(alego_handle* h, int bag, int scan, const alego_point* pts, int32_t n);
int alego_replay_assign(alego_handle* h, int slot, int bag, int start_scan);
/* ONE stream replayed from the bag store as fast as the device allows (BASELINE config 3 as written: a single bag).  Slot 0 carries
 * the stream's state; the other slots of the handle are look-ahead lanes in two sets of W = (n_slots - 1) / 2: ImageProjection and feature
 * extraction have no state across scans, so they run for W scans ahead in one launch per kernel (one scan per lane), while LaserOdometry
 * (sequential by nature: params_, the previous scan's features) and LaserMapping (the key-frame map) follow scan by scan on two
 * further HIP streams, each started by an event of its producer — the three-nodelet pipeline of launch/test.launch on the device.
 * Results are bit-identical to alego_batch_run on a one-slot handle.  The handle needs n_slots >= 3 (odd) and a bag store;
 * `first_step` / `n_scans` as in alego_batch_run with ALEGO_REPLAY_BAG (scan (start_scan + first_step + i) mod bag_len at step i). */
int alego_stream_setup(alego_handle* h, int bag, int start_scan);
int alego_stream_run(alego_handle* h, int first_step, int n_scans, int stages, int sync);
/* poses of the last processed scan of `slot` */
int alego_batch_get_pose(alego_handle* h, int slot, alego_pose* odom, alego_pose* map_pose);
/* Per-scan pose log: what a bag replay publishes on /odom/lidar (laserOdometry.cpp:513-529) and /odom_aft_mapped
 * (laserMapping.cpp:167-181) for every scan, kept on the device so that alego_batch_run needs no host synchronisation per scan
 * (the bench-path export with a poses_out argument that SURVEY.md 8b proposes).  After alego_trajectory_enable(h, capacity) every scan a slot processes through
 * alego_batch_run / alego_scan_process appends 14 doubles: odometry t(3) q(4: w x y z), map pose t(3) q(4).  alego_trajectory_get
 * copies entries [first, first + n) of `slot` and returns the number of scans logged so far (entries beyond the capacity are
 * dropped, the count keeps running). */
int alego_trajectory_enable(alego_handle* h, int32_t capacity_scans);
int alego_trajectory_get(alego_handle* h, int slot, int32_t first, int32_t n, double* out14);
/* per-scan device counters of the last processed scan of `slot`:
 * out[0..] = P (valid input points), M, n_outlier, n_sharp, n_less_sharp, n_flat, n_less_flat,
 *            n_surf_corr, n_corner_corr, lm: Kraw_corner, Kraw_surf, Kds_corner, Kds_surf, Lc, Ls,
 *            map rebuilds so far */
int alego_batch_get_counts(alego_handle* h, int slot, int32_t* out, int cap);
/* the HIP stream (hipStream_t) the handle enqueues slot 0 on, for event timing by the caller */
void* alego_stream(alego_handle* h);
/* The slots of a handle are split into contiguous groups, each enqueued on its own HIP stream so that the kernels of
 * different groups overlap (slots never interact).  Returns the number of groups; *slots_per_group (may be NULL) =
 * slots one kernel launch covers.  Default: one group per 64 slots, at most 4; ALEGO_STREAM_GROUPS=<n> overrides. */
int alego_stream_groups(const alego_handle* h, int* slots_per_group);

/* ---- per-kernel timing (bench.py's roofline leg) --------------------------- */
/* When enabled every kernel launch of this handle is bracketed by hipEventRecord on the handle's
 * stream.  Off by default; the benchmark's timed region runs with it off. */
int alego_profile_enable(alego_handle* h, int on);
/* names: ';'-separated kernel names; total_ms / launches per kernel.  Returns the kernel count. */
int alego_profile_report(alego_handle* h, char* names, int names_cap, double* total_ms, int* launches, int cap);

/* ---- motion de-skew (LaserOdometry::adjustDistortion, laserOdometry.cpp:557-726; alego_params.deskew_mode = 1) --------------
 * sensor_msgs/Imu as LaserOdometry::imuHandler (:761-802) receives it.  The handler's ring of 200 dead-reckoned samples lives on
 * the device; with deskew_mode = 1 every scan's segmented cloud is de-skewed against it before feature extraction (the call the
 * reference has commented out at :115), by alego_lo_process / alego_scan_process (the stamps come from alego_seg_out.stamp /
 * alego_scan_in.stamp).  Stamps must not decrease. */
typedef struct alego_imu {
  double stamp;
  double orientation[4];          /* w, x, y, z */
  double linear_acceleration[3];
  double angular_velocity[3];     /* carried by the message, unused by the reference (:787-789) */
} alego_imu;
int alego_lo_push_imu(alego_handle* h, int slot, const alego_imu* samples, int32_t n);
/* /undistorted (src/laserOdometry.cpp:56,718-725; src/LO.cpp:127,797-804): the de-skewed segmented cloud of `slot`'s last scan, as adjustDistortion
 * publishes it when somebody subscribes.  Returns the number of points, ALEGO_ERR_ARG when deskew_mode = 0 (the reference then never publishes on the
 * topic either: the call is commented out at laserOdometry.cpp:115), ALEGO_ERR_CAPACITY when `cap` is too small. */
int alego_lo_get_undistorted(alego_handle* h, int slot, alego_point* out, int32_t cap);
/* The STANDALONE LaserOdometry node's frame convention (src/LO.cpp:588-608): /odom/lidar is published as /odom -> /base_link with
 * tf_o2b = tf_o2l * tf_b2l^-1 (tf_b2l: base_link -> laser mount, row-major 4 x 4; identity at LO.cpp:121), its quaternion taken from the rotation
 * block; the nodelet (src/laserOdometry.cpp:513-529) publishes /odom -> /laser unchanged.  Host arithmetic: `o2l` is what alego_lo_process returned,
 * `o2b` gets t and q (params / valid copied).  ALEGO_ERR_ARG for a singular tf_b2l. */
int alego_pose_o2b(const alego_pose* o2l, const double* tf_b2l, alego_pose* o2b);

/* ---- state access for parity tests (teacher forcing) ---------------------- */
int alego_set_lo_params(alego_handle* h, int slot, const double* p6);
int alego_set_lm_params(alego_handle* h, int slot, const double* p6);
/* copy a named device intermediate of `slot` to host; *count = number of scalars written.
 * dtype: 0 f32, 1 f64, 2 i32, 3 u8.  Names mirror oracle_get(). */
int alego_debug_get(alego_handle* h, int slot, const char* name, void* out, int cap_bytes,
                    int* count, int* dtype);
/* the device pcl::VoxelGrid replacement on a host cloud (the LDS-resident radix sort for <= 8192 points, the HBM-scratch
 * radix sort above), for direct parity tests against the oracle; returns the output count */
int alego_debug_voxel(alego_handle* h, const alego_point* pts, int n, float leaf, alego_point* out, int cap);
/* device atan2f / hypotf used by the projection kernel, for the libm-equivalence test */
int alego_debug_atan2f(alego_handle* h, const float* y, const float* x, float* out, int n);
/* libstdc++ std::sort(idx, idx + n, [](a, b) { return keys[a] < keys[b]; }) on idx = 0..n-1 (n <= 4096) as the device reproduces
 * it for alego_params.sort_mode = 2 (laserOdometry.cpp:185): order[k] = the element at sorted position k.  depth_limit < 0 = std::sort's
 * own 2 floor(log2 n); >= 0 overrides __introsort_loop's depth limit (0 = heap sort at once) so that tests reach that branch */
int alego_debug_std_sort(alego_handle* h, const uint32_t* keys, int n, int depth_limit, int32_t* order);
/* the device's shared single-precision functions on arrays: mode 0 atan2f(a, b), 1 hypotf(a, b), 2 sinf(a), 3 cosf(a) */
int alego_debug_math(alego_handle* h, int mode, const float* a, const float* b, float* out, int n);
/* The four cost functors of include/alego/utility.h:122-349 evaluated on the device exactly as the solvers evaluate them
 * (csrc/dev_cost.h): type 0 SurfCostFunction, 1 CornerCostFunction, 2 LidarEdgeCostFunction, 3 LidarPlaneCostFunction;
 * geom13[i] = cp(3), lpj | normal(3), lpl(3), lpm(3), negative_OA_dot_norm; out: res[i], jac6[i][6] */
int alego_debug_eval_blocks(alego_handle* h, int type, int n, const double* geom13, const double* params6, double* res, double* jac6);
/* transformToStart (laserOdometry.cpp:728-740) of n points with LaserOdometry params_ = params6, as lo_assoc applies it */
int alego_debug_transform_to_start(alego_handle* h, const double* params6, const alego_point* pts, int n, alego_point* out);
/* development aid: with ALEGO_DEBUG_CANARY=1 in the environment every device allocation of the library is framed by 4 KB guard
 * pages of a known pattern; returns how many allocations have a damaged guard (0 = none, -1 = guards not enabled) and describes
 * them in `report` */
int alego_debug_check_guards(char* report, int cap);
/* Run-time switches of kernel variants (parity tests run both variants of a kernel inside one process).  Read once from
 * the environment at alego_create (ALEGO_CC_FUSED, ALEGO_FE_PICK1, ALEGO_LO_BOX_LDS, ALEGO_MAP_MERGE, ALEGO_IP_FAST); this
 * call overrides one of them by its environment name.  Not a hot-path call. */
int alego_debug_set_option(alego_handle* h, const char* name, int value);

/* ---- key-frame pass-through for a host-side pose graph (laserMapping.cpp:491-596) -------------------------------
 * LaserMapping keeps the recent_keyframe_num newest key frames of every slot on the device: the down-sampled clouds as
 * saveKeyFramesAndFactor stores them (corner_frames_ / surf_frames_ / outlier_frames_, sensor frame, :553-555) and the
 * f32 key pose (cloud_keyposes_6d_, :531-537).  A host pose graph (GTSAM in the reference) reads every new key frame
 * when ALEGO_FLAG_LM_KEYFRAME is returned, and writes corrected poses back after a loop closure. */
typedef struct alego_keyframe {
  int32_t id;            /* index in cloud_keyposes_3d_ (0-based); the reference's intensity field is id + 0.1 (:529) */
  float pose[6];         /* x y z roll pitch yaw (PointXYZIRPYT, utility.h:83-92) */
  alego_point* corner;   int32_t corner_cap;   int32_t n_corner;   /* laser_corner_ds_  of the frame (may be NULL) */
  alego_point* surf;     int32_t surf_cap;     int32_t n_surf;     /* laser_surf_ds_ */
  alego_point* outlier;  int32_t outlier_cap;  int32_t n_outlier;  /* laser_outlier_ds_ */
} alego_keyframe;
/* number of key frames saved so far by `slot` (cloud_keyposes_3d_->size()) */
int alego_lm_keyframe_count(alego_handle* h, int slot);
/* copy key frame kf_id (-1 = the newest) to the host; only the recent_keyframe_num newest ones are resident
 * (older ids return ALEGO_ERR_ARG: the host keeps its own copy, as the reference does).  publish() (:586-596) is the
 * `pose` of every frame. */
int alego_lm_get_keyframe(alego_handle* h, int slot, int kf_id, alego_keyframe* out);
/* correctPoses (:569-578): overwrite the key pose of a resident frame; the device re-transforms its clouds.  The local
 * map is rebuilt at the next mapping frame once alego_lm_reset_window has been called (the reference clears the
 * recent_* deques at :563-565 and refills them from the newest frames at :208-223). */
int alego_lm_set_keypose(alego_handle* h, int slot, int kf_id, const float pose6[6]);
int alego_lm_reset_window(alego_handle* h, int slot);
/* correctPoses (:579-580): q_map2odom <- R q_map2odom, t_map2odom <- R t_map2odom + c, with the row-major 3x4 [R | c] */
int alego_lm_apply_correction(alego_handle* h, int slot, const double rc[12]);
/* push_back a key frame from host data (clouds in the sensor frame + pose): restores a saved session or lets the host
 * pose graph insert a frame; equivalent to :531-555 with the given pose and clouds.  The device keeps the recent_keyframe_num + 1
 * newest frames.  A FULL window only advances by one frame per mapping frame (:224-237) and so falls behind the newest frames by
 * one for every extra frame inserted; it may lag by one.  An insertion that would make it lag further returns ALEGO_ERR_CAPACITY —
 * call alego_lm_reset_window first when inserting in bulk (the window is then rebuilt from the newest frames, :208-223). */
int alego_lm_add_keyframe(alego_handle* h, int slot, const float pose6[6], const alego_point* corner, int32_t n_corner,
                          const alego_point* surf, int32_t n_surf, const alego_point* outlier, int32_t n_outlier);

/* ---- loop closure (laserMapping.cpp:633-824): the host keeps the pose graph (GTSAM in the reference) and every key frame; the
 * library does the per-point work of one closure attempt.
 *   alego_loop_detect        detectLoopClosure's choice (:771-790), plain host code: the key pose nearest to `cur_xyz` within
 *                            lc_search_radius whose stamp is more than lc_min_time_gap older than the newest key frame's; -1 = none
 *   alego_loop_closure_icp   sub-map assembly (:794-812: the newest key frame as ICP source; the history frames
 *                            closest - lc_search_num .. closest + lc_search_num, transformed by their key poses, concatenated and
 *                            VoxelGrid(lc_leaf)-filtered as target) and pcl::IterativeClosestPoint as configured at :670-692.
 *                            `correction` = getFinalTransformation() (row-major 4x4, f32), `fitness` = getFitnessScore().  The caller
 *                            applies :697 (converged && fitness <= lc_fitness_max), adds the Between factor (:716-733) and, after the
 *                            graph update, writes the corrected poses back (alego_lm_set_keypose / reset_window / apply_correction).
 *                            target_out (may be NULL): near_history_keyframes_ for /history_keyframes. */
typedef struct alego_kf_in {
  float pose[6];                                   /* x y z roll pitch yaw of the key frame */
  const alego_point* corner;   int32_t n_corner;   /* corner_frames_[id], surf_frames_[id], outlier_frames_[id] (sensor frame) */
  const alego_point* surf;     int32_t n_surf;
  const alego_point* outlier;  int32_t n_outlier;
} alego_kf_in;
typedef struct alego_icp_result {
  int32_t converged, iterations, n_source, n_target;
  double fitness;
  float correction[16];
} alego_icp_result;
int alego_loop_detect(const alego_params* params, const float* keyposes6, const double* stamps, int32_t n, const double cur_xyz[3]);
int alego_loop_closure_icp(alego_handle* h, const alego_kf_in* latest, const alego_kf_in* history, int32_t n_history,
                           alego_icp_result* out, alego_point* target_out, int32_t target_cap);

/* ---- one scan-to-map registration sharded over the GPUs of a node (BASELINE.json config 5) ----------------
 * One process per GPU; every rank feeds its handle the SAME scans and so keeps a bit-identical replica of the stream's state
 * (ImageProjection, feature extraction, LaserOdometry and the local map are cheap and are computed redundantly).  What is split
 * is scan2MapOptimization (laserMapping.cpp:360-478): rank r owns the queries [r T / G, (r + 1) T / G) of laser_corner_ds_ ++
 * laser_surf_total_ds_ (5-NN, line / plane fit, residual + Jacobian rows); for every solver evaluation the 28 normal-equation
 * scalars J^T J (21), J^T r (6), cost (1) (+ 2 correspondence counts) are summed with ncclAllReduce(f64) — RCCL over xGMI — on
 * the handle's stream, and every rank takes the identical trust-region step.  The handle must have a single stream group
 * (fewer than 128 slots).  world = 1 is allowed: same kernel sequence, the collective is a copy (hardware tests on one GPU). */
#define ALEGO_DIST_ID_BYTES 128
int alego_dist_unique_id(char id[ALEGO_DIST_ID_BYTES]);   /* rank 0: ncclGetUniqueId; distribute the bytes to every rank */
int alego_dist_init(alego_handle* h, int rank, int world, const char id[ALEGO_DIST_ID_BYTES]);
int alego_dist_shutdown(alego_handle* h);
/* the collective of ONE solver evaluation measured on its own (a mapping frame has ~42 of them): `iters` in-place all-reduces of 32 doubles
 * back to back on the registration's stream; microseconds each.  A collective: every rank of the communicator calls it. */
int alego_dist_allreduce_probe(alego_handle* h, int iters, double* usec_per_allreduce);

/* ---- sensor_msgs/PointCloud2 to alego_point: pcl::fromROSMsg<PointXYZI>, imageProjection.cpp:54-55, IP.cpp:109-110 ----
 * ROS-free mirror of sensor_msgs/PointField + the PointCloud2 layout fields.  Fields are matched by name ("x", "y", "z",
 * "intensity") and must be FLOAT32 (datatype 7) with count 1 (or 0: unset), as PCL's field mapper requires; a missing intensity gives 0.
 * Returns the number of points written (width * height), or ALEGO_ERR_ARG / ALEGO_ERR_CAPACITY. */
typedef struct alego_pc2_field { const char* name; uint32_t offset; uint8_t datatype; uint32_t count; } alego_pc2_field;
int alego_pc2_to_points(const uint8_t* data, uint64_t data_len, uint32_t width, uint32_t height, uint32_t point_step,
                        uint32_t row_step, int is_bigendian, const alego_pc2_field* fields, int n_fields,
                        alego_point* out, int32_t cap);

/* ---- rosbag format 2.0 reader (host side, no ROS / libbz2 / liblz4 needed; csrc/rosbag.cpp) -------------------------------
 * Replaces `rosbag play <file>.bag` + the `/lslidar_point_cloud` subscription + pcl::fromROSMsg at the head of ImageProjection
 * (README.md:33-37, launch/test2.launch:6-14, src/IP.cpp:106-133, src/imageProjection.cpp:45,49-55): open the bag, hand out the
 * messages of a topic in time order, deserialise sensor_msgs/PointCloud2 and run it through alego_pc2_to_points.  Chunks may be
 * uncompressed, bz2 or lz4; a bag without index records (never closed) is scanned.  Errors: ALEGO_ERR_ARG (+ alego_bag_last_error). */
typedef struct alego_bag alego_bag;
int alego_bag_open(const char* path, alego_bag** out);
void alego_bag_close(alego_bag* b);
const char* alego_bag_last_error(const alego_bag* b);
int alego_bag_topic_count(const alego_bag* b);
int alego_bag_topic_info(const alego_bag* b, int i, const char** topic, const char** datatype, int64_t* n_messages);
int64_t alego_bag_message_count(const alego_bag* b, const char* topic);
/* serialized message `index` of `topic` (pointer valid until the next read / close); bag_time = the record's receive time */
int alego_bag_read_raw(alego_bag* b, const char* topic, int64_t index, const uint8_t** data, uint64_t* len, double* bag_time);
/* PointCloud2 message `index` of `topic` -> points (returns their number); header_stamp = msg.header.stamp, is_dense = msg.is_dense */
int alego_bag_read_pc2(alego_bag* b, const char* topic, int64_t index, alego_point* out, int32_t cap, double* header_stamp, int32_t* is_dense);

#ifdef __cplusplus
}
#endif
#endif /* ALEGO_MI355X_H_ */
