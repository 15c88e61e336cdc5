// alego_api.hip — host side of the C ABI (include/alego_mi355x.h): HBM allocation, the
// per-handle HIP stream, kernel sequencing and the nodelet-shaped entry points.
// No numerics live here and there is no CPU fallback.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/alego_mi355x.h"
#include "dev_common.h"
#include "guard_alloc.h"
#include "lm_host.h"
#include "prof.h"
#include "voxel.h"

thread_local Profiler* g_prof = nullptr;

// A handle of several stream groups drives two HIP streams per group (front end + LaserMapping); the runtime's default of 4 hardware
// queues makes pairs of them share a queue and serialise.  The host process exports GPU_MAX_HW_QUEUES=16 before its first HIP call
// (bench.py, tests/conftest.py, binding.py do; include/alego_mi355x.h documents it) — the library does not touch the environment.

void launch_ip(const DevCtx& d, int ring_pos, bool want_labels, hipStream_t st);
void launch_fe(const DevCtx& d, hipStream_t st);
void launch_lo(const DevCtx& d, hipStream_t st);
void launch_lo_grid(const DevCtx& d, hipStream_t st);
void launch_atan2f_probe(const float* y, const float* x, float* out, int n, int mode, hipStream_t st);
int launch_stdsort_probe(const uint32_t* keys, int n, int depth_limit, int* pos_out, hipStream_t st);
void launch_lo_imu_push(const DevCtx& d, int slot, const double* smp_dev, int n, hipStream_t st);
void launch_lo_deskew(const DevCtx& d, hipStream_t st);
void launch_traj_log(const DevCtx& d, hipStream_t st, const double* staged_odom = nullptr, int par = 0);
const double* lm_host_stage_odom(LmHost* lm);
void launch_dbg_eval_blocks(int type, int n, const double* geom13, const double* params6, double* res, double* jac6, hipStream_t st);
int icp_run(const alego_params& P, const alego_kf_in* latest, const alego_kf_in* history, int n_history, alego_icp_result* out,
            alego_point* target_out, int target_cap, hipStream_t st, std::string* err);
void launch_dbg_transform_to_start(const double* params6, const float4* pts, int n, float4* out, hipStream_t st);
int ip_configure(const DevCtx& d);
int lo_configure();
int lm_configure();

struct alego_handle {
  std::recursive_mutex host_lock;   // alego_handle_lock / alego_handle_unlock: for hosts that drive ONE handle from several threads (the three nodelets)
  alego_params P;
  int device = 0;
  hipStream_t stream = nullptr;       // = streams[0]
  // Slots are split into contiguous groups of `gsize`; each group has its own HIP stream (and its own VoxelGrid scratch in
  // LmHost), so the latency-bound kernels of one group overlap with the kernels of the others.  Slots never interact.
  std::vector<hipStream_t> streams;
  int gsize = 1;
  // alego_batch_run: LaserMapping of scan k runs on a second HIP stream of the group (back[g]) while the group's front end (ImageProjection,
  // feature extraction, LaserOdometry — none of which reads anything LaserMapping writes) already works on scans k + 1, k + 2.  lm_stage
  // hands a scan over (odometry + the three clouds, double-buffered by scan parity); ev_back[g][p] = LaserMapping of the last scan of parity p done.
  bool lm_async = false;
  std::vector<hipStream_t> back;
  std::vector<hipEvent_t> ev_stage, ev_back;   // [group][2]
  std::vector<long> grp_scans;                  // scans handed over per group
  DevCtx d;
  std::vector<void*> allocs;
  std::string err;
  std::vector<long> lo_scans;  // LO steps enqueued per slot (the first one only initialises, laserOdometry.cpp:316-324)
  LmHost* lm = nullptr;
  Profiler prof;
  int ip_fast_capable = 0;  // which of ip_project's table fast paths this geometry supports
  bool replay_assigned = false;
  // alego_stream_run: look-ahead lanes
  bool stream_mode = false;
  int lanes = 0;               // W: lanes per set (slots 1 .. W and W + 1 .. 2 W)
  int pose_slot = 0;           // slot holding the poses of the last processed scan of slot 0 (its lane)
  int last_lane = -1;          // lane of the previous scan (LaserOdometry's surf_last_ / corner_last_)
  std::vector<double> imu_last_stamp;   // per slot: stamp of the newest IMU sample (alego_lo_push_imu wants them non-decreasing)
  int next_set = 0;            // lane set the next group of scans uses (the other one still holds the previous scan's features)
  hipStream_t s_lo = nullptr, s_lm = nullptr;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_next = 0;
};

namespace {

#define HIP_TRY(h, call)                                                                       \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
      return ALEGO_ERR_HIP;                                                                    \
    }                                                                                          \
  } while (0)

template <class T>
int dalloc(alego_handle* h, T** p, size_t count, bool zero = true) {
  void* q = nullptr;
  size_t bytes = count * sizeof(T);
  if (bytes == 0) bytes = sizeof(T);
  hipError_t e = guard_malloc(&q, bytes);
  if (e != hipSuccess) { h->err = std::string("hipMalloc: ") + hipGetErrorString(e); return ALEGO_ERR_HIP; }
  h->allocs.push_back(q);
  if (zero) { e = hipMemset(q, 0, bytes); if (e != hipSuccess) { h->err = "hipMemset failed"; return ALEGO_ERR_HIP; } }
  *p = (T*)q;
  return 0;
}

DevCtx view(const alego_handle* h, int slot0, int n) {
  DevCtx d = h->d;
  d.slot0 = slot0;
  d.n_launch = n;
  return d;
}

hipStream_t stream_of(const alego_handle* h, int slot) { return h->streams[slot / h->gsize]; }
hipError_t sync_all(const alego_handle* h) {
  hipError_t r = hipSuccess;
  for (hipStream_t s : h->streams) { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) r = e; }
  for (hipStream_t s : h->back) { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) r = e; }
  for (hipStream_t s : {h->s_lo, h->s_lm}) if (s) { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) r = e; }
  return r;
}

// frees temporary device buffers on every exit path of the debug entries
struct DevTemps {
  std::vector<void*> p;
  template <class T> hipError_t get(T** q, size_t bytes) { void* v = nullptr; hipError_t e = hipMalloc(&v, bytes ? bytes : 16); if (e == hipSuccess) p.push_back(v); *q = (T*)v; return e; }
  ~DevTemps() { for (void* v : p) (void)hipFree(v); }
};

int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

// LaserMapping work still in flight on the groups' back streams (alego_batch_run without sync): every entry point that reads or writes
// LaserOdometry / LaserMapping state waits for it first — per-slot calls for their own group's back stream (check_slot), whole-handle
// calls (alego_lo_process, alego_lm_process, alego_stream_run, alego_dist_*) for all of them.  alego_batch_load only writes the
// front end's input ring and does not wait (a host-fed batch_load + batch_run(sync = 0) loop keeps its overlap).
void drain_back(alego_handle* h, int slot = -1) {
  if (slot >= 0 && !h->back.empty()) { (void)hipStreamSynchronize(h->back[std::min<size_t>((size_t)(slot / h->gsize), h->back.size() - 1)]); return; }
  for (hipStream_t s : h->back) (void)hipStreamSynchronize(s);
}
int check_slot(alego_handle* h, int slot, bool drain = true) {
  if (!h) return ALEGO_ERR_ARG;
  if (slot < 0 || slot >= h->d.n_slots) { h->err = "slot out of range"; return ALEGO_ERR_ARG; }
  if (drain) drain_back(h, slot);
  return 0;
}

}  // namespace

extern "C" {

int alego_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int alego_params_sizeof(void) { return (int)sizeof(alego_params); }
int alego_handle_lock(alego_handle* h) { if (!h) return ALEGO_ERR_ARG; h->host_lock.lock(); return 0; }
int alego_handle_unlock(alego_handle* h) { if (!h) return ALEGO_ERR_ARG; h->host_lock.unlock(); return 0; }
const char* alego_last_error(const alego_handle* h) { return h ? h->err.c_str() : "null handle"; }
void* alego_stream(alego_handle* h) { return h ? (void*)h->stream : nullptr; }
int alego_stream_groups(const alego_handle* h, int* slots_per_group) {
  if (!h) return ALEGO_ERR_ARG;
  if (slots_per_group) *slots_per_group = h->gsize;
  return (int)h->streams.size();
}

int alego_create(const alego_params* params, int device, int n_slots, int ring_len, alego_handle** out) {
  if (!params || !out) return ALEGO_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return ALEGO_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ALEGO_ERR_NO_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    std::fprintf(stderr, "alego_create: device %d is %s, this library is built for gfx950 only\n", device, prop.gcnArchName);
    return ALEGO_ERR_NO_DEVICE;
  }
  if (n_slots <= 0) n_slots = 1;
  if (ring_len <= 0) ring_len = 1;
  if (params->n_scan < 1 || params->n_scan > 64 || params->horizon_scan < 64 || params->horizon_scan > 4096) return ALEGO_ERR_ARG;
  if (params->sort_mode != 0 && params->sort_mode != 2) { std::fprintf(stderr, "alego_create: sort_mode %d is an oracle-only setting (0 = (curvature, index) order, 2 = libstdc++ std::sort tie order in the sector sort)\n", params->sort_mode); return ALEGO_ERR_ARG; }
  if (params->deskew_mode != 0 && params->deskew_mode != 1) { std::fprintf(stderr, "alego_create: deskew_mode must be 0 or 1\n"); return ALEGO_ERR_ARG; }
  if (params->deskew_mode && !(params->scan_period > 0)) { std::fprintf(stderr, "alego_create: scan_period must be positive\n"); return ALEGO_ERR_ARG; }
  if (params->kf_cap_surf < 0 || params->kf_cap_outlier < 0) { std::fprintf(stderr, "alego_create: negative key-frame capacity\n"); return ALEGO_ERR_ARG; }
  if (params->recent_keyframe_num > 512) { std::fprintf(stderr, "alego_create: recent_keyframe_num > 512 is not supported\n"); return ALEGO_ERR_ARG; }
  // The feature pick marks up to suppress_radius neighbours on either side of a picked point; the segmented cloud only
  // guarantees the reference's 5-point margin at both ends of a ring (laserOdometry.cpp:124,211-234 index i +- 5 unchecked).
  if (params->suppress_radius < 0 || params->suppress_radius > 5 || params->n_sectors < 1 || params->n_sharp < 0 ||
      params->n_less_sharp < params->n_sharp || params->n_flat < 1) {
    std::fprintf(stderr, "alego_create: feature-pick parameters out of range (0 <= suppress_radius <= 5, n_sectors >= 1, 0 <= n_sharp <= n_less_sharp, n_flat >= 1)\n");
    return ALEGO_ERR_ARG;
  }
  {
    // the feature pick keeps a sector's candidates in registers: 16 lanes x 43 (fe_pick4) or 64 lanes x 12 (fe_pick) elements
    const int sector_max = (params->horizon_scan + params->n_sectors - 1) / params->n_sectors + 2;
    if (sector_max > 64 * 12) {
      std::fprintf(stderr, "alego_create: ceil(horizon_scan / n_sectors) + 2 = %d exceeds the 768 sector elements the feature pick holds\n", sector_max);
      return ALEGO_ERR_ARG;
    }
  }
  alego_handle* h = new alego_handle();
  h->P = *params;
  h->device = device;
  h->lo_scans.assign(n_slots, 0);
  if (hipSetDevice(device) != hipSuccess) { delete h; return ALEGO_ERR_HIP; }
  {
    // stream groups: ALEGO_STREAM_GROUPS overrides; default one group per 64 slots, at most 4 (the HIP runtime's hardware queues)
    int G = n_slots / 64;
    if (const char* e = getenv("ALEGO_STREAM_GROUPS")) G = atoi(e);
    if (G > 8) G = 8;
    if (G > n_slots) G = n_slots;
    if (G < 1) G = 1;
    if (!getenv("ALEGO_STREAM_GROUPS") && G > 4) G = 4;
    h->gsize = (n_slots + G - 1) / G;
    G = (n_slots + h->gsize - 1) / h->gsize;
    for (int g = 0; g < G; ++g) {
      hipStream_t s = nullptr;
      if (hipStreamCreate(&s) != hipSuccess) { for (hipStream_t t : h->streams) hipStreamDestroy(t); delete h; return ALEGO_ERR_HIP; }
      h->streams.push_back(s);
    }
    h->stream = h->streams[0];
    h->lm_async = env_int("ALEGO_LM_ASYNC", 1) != 0;
    if (h->lm_async) {
      for (int g = 0; g < G; ++g) {
        hipStream_t s = nullptr;
        hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
        bool ok = hipStreamCreate(&s) == hipSuccess;
        for (int k = 0; k < 4 && ok; ++k) ok = hipEventCreateWithFlags(&e[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) { h->lm_async = false; break; }
        h->back.push_back(s);
        h->ev_stage.push_back(e[0]); h->ev_stage.push_back(e[1]); h->ev_back.push_back(e[2]); h->ev_back.push_back(e[3]);
      }
      h->grp_scans.assign(G, 0);
    }
  }
  DevCtx& d = h->d;
  std::memset(&d, 0, sizeof(d));
  d.P = *params;
  d.n_slots = n_slots; d.ring_len = ring_len; d.slot0 = 0; d.n_launch = n_slots;
  d.fs_cur = d.fs_last = -1;
  d.NS = params->n_scan; d.H = params->horizon_scan; d.N = d.NS * d.H; d.Pcap = d.N;
  d.cap_sharp = params->n_sharp * params->n_sectors;
  d.cap_lsharp = params->n_less_sharp * params->n_sectors;
  d.cap_flat = params->n_flat * params->n_sectors;
  d.st_stride = d.cap_sharp + d.cap_lsharp + d.cap_flat + d.H;
  d.sin_ax = std::sin(params->seg_alpha_x); d.cos_ax = std::cos(params->seg_alpha_x);  // imageProjection.cpp:269
  d.sin_ay = std::sin(params->seg_alpha_y); d.cos_ay = std::cos(params->seg_alpha_y);
  {
    const double lo = params->sensor_mount_ang - params->ground_angle_thres, hi = params->sensor_mount_ang + params->ground_angle_thres;
    const bool ok = lo > -89.0 && hi < 89.0 && lo < hi;
    d.tan_g_lo = ok ? std::tan(lo * M_PI / 180.0) : std::nan("");
    d.tan_g_hi = ok ? std::tan(hi * M_PI / 180.0) : std::nan("");
  }
  d.h_magic = (unsigned)((1ull << 32) / (unsigned long long)d.H) + 1u;
  d.inv_res_x = 1.0 / params->ang_res_x; d.inv_res_y = 1.0 / params->ang_res_y;
  d.tan_theta = (params->seg_theta > 0.0 && params->seg_theta < 1.5) ? std::tan(params->seg_theta) : std::nan("");
  d.opt_ip_fused = env_int("ALEGO_IP_FUSED", 1) != 0;
  d.opt_ip_half = env_int("ALEGO_IP_HALF", 1) != 0;
  d.opt_cc_fused = env_int("ALEGO_CC_FUSED", 1) != 0;
  d.opt_cc_tile = env_int("ALEGO_CC_TILE", 1) != 0;
  d.opt_ip_band = env_int("ALEGO_IP_BAND", 1) != 0;
  d.opt_fe_pick1 = env_int("ALEGO_FE_PICK1", 0) != 0;
  d.opt_fe_fused = env_int("ALEGO_FE_FUSED", 1) != 0;
  d.opt_fe_cand = env_int("ALEGO_FE_CAND", 0);
  d.opt_lo_box_lds = env_int("ALEGO_LO_BOX_LDS", 1 << 20);
  d.opt_lo_grid = env_int("ALEGO_LO_GRID", 1) != 0;
  d.opt_fo_spin = env_int("ALEGO_FE_SPIN", 0);
  d.opt_fo_pad8 = env_int("ALEGO_FE_PAD8", 0) != 0;
  d.opt_map_merge = env_int("ALEGO_MAP_MERGE", 1) != 0;
  const size_t B = n_slots, N = d.N, NS = d.NS;
  int rc = 0;
  rc |= dalloc(h, &d.in_pts, B * ring_len * d.Pcap, false);
  rc |= dalloc(h, &d.in_n, B * ring_len);
  rc |= dalloc(h, &d.owner, B * N); rc |= dalloc(h, &d.range_img, B * N); rc |= dalloc(h, &d.flag_img, B * N);
  rc |= dalloc(h, &d.parent, B * N); rc |= dalloc(h, &d.cc_size, B * N); rc |= dalloc(h, &d.cc_rows, B * N);
  rc |= dalloc(h, &d.label_img, B * N); rc |= dalloc(h, &d.cc_label, B * N); rc |= dalloc(h, &d.row_cnt, B * NS * 4);
  rc |= dalloc(h, &d.scal, B * SC_COUNT);
  d.ipb_col = nullptr; d.ipb_off = nullptr;
  if (d.NS > 16 && d.NS <= 64) { rc |= dalloc(h, &d.ipb_col, B * IPB_NM * d.H); rc |= dalloc(h, &d.ipb_off, B * 3 * 64 * ((d.H + 63) / 64)); }   // the banded mask path (kernels_ipb.hip)
  d.ipf_own = nullptr;
  if (d.NS <= 16 && d.N <= 65535 && (d.H & 1) == 0) rc |= dalloc(h, &d.ipf_own, B * (N / 2), false);
  rc |= dalloc(h, &d.seg_pts, B * N); rc |= dalloc(h, &d.seg_ground, B * N); rc |= dalloc(h, &d.seg_col, B * N);
  rc |= dalloc(h, &d.seg_range, B * N); rc |= dalloc(h, &d.ring_start, B * NS); rc |= dalloc(h, &d.ring_end, B * NS);
  rc |= dalloc(h, &d.ori, B * 4); rc |= dalloc(h, &d.outlier, B * N);
  rc |= dalloc(h, &d.cd, B * N); rc |= dalloc(h, &d.picked0, B * N); rc |= dalloc(h, &d.fe_flag, B * N); rc |= dalloc(h, &d.plabel, B * N);
  rc |= dalloc(h, &d.st_idx, B * NS * d.st_stride); rc |= dalloc(h, &d.st_cnt, B * NS * 8);
  rc |= dalloc(h, &d.st_lfds, B * NS * d.H);
  d.fcap[F_SHARP] = d.cap_sharp * d.NS; d.fcap[F_LSHARP] = d.cap_lsharp * d.NS; d.fcap[F_FLAT] = d.cap_flat * d.NS; d.fcap[F_LFLAT] = d.N;
  for (int k = 0; k < 4; ++k) rc |= dalloc(h, &d.feat[k], B * 2 * d.fcap[k]);
  for (int k = 0; k < 3; ++k) rc |= dalloc(h, &d.feat_idx[k], B * 2 * d.fcap[k]);
  rc |= dalloc(h, &d.feat_cnt, B * 2 * 4); rc |= dalloc(h, &d.ring_off, B * 2 * 2 * (NS + 1)); rc |= dalloc(h, &d.ring_boff, B * 2 * 2 * (NS + 1));
  rc |= dalloc(h, &d.fe_sync, B * NS);
  d.lo_qcap_surf = d.fcap[F_FLAT]; d.lo_qcap_corner = d.fcap[F_SHARP];
  rc |= dalloc(h, &d.lo_corr, B * (d.lo_qcap_surf + d.lo_qcap_corner) * 4);
  d.lo_box_cap = (d.N + LO_CH - 1) / LO_CH + d.NS;   // boxes never straddle rings: up to one partly filled box per ring
  rc |= dalloc(h, &d.lo_box, B * 2 * 2 * d.lo_box_cap * 2);
  rc |= dalloc(h, &d.lo_cpts[0], B * 2 * d.fcap[F_LFLAT], false); rc |= dalloc(h, &d.lo_cpts[1], B * 2 * d.fcap[F_LSHARP], false);
  rc |= dalloc(h, &d.lo_cell, B * 2 * 2 * (LO_GC + 2)); rc |= dalloc(h, &d.lo_geom, B * 2 * 2 * 8);
  rc |= dalloc(h, &d.lo_state, B * LO_STATE_N);
  rc |= dalloc(h, &d.poses, B * 16);
  rc |= dalloc(h, &d.imu_ring, B * ALEGO_IMU_Q * 10); rc |= dalloc(h, &d.imu_ptr, B * 4); rc |= dalloc(h, &d.scan_stamp, B);
  rc |= dalloc(h, &d.seg_dsk, h->P.deskew_mode ? B * N : 1);
  {  // boundary tables of ip_project's fast path (kernels_ip.hip)
    const double rx = params->ang_res_x, ry = params->ang_res_y;
    std::vector<double> rt(2 * (NS + 6)), ct;
    for (int k = 0; k < (int)NS + 6; ++k) {   // r = k - 3: the vertical angle at which (va + ang_bottom) / ang_res_y + 0.5 == r
      const double phi = (((double)(k - 3) - 0.5) * ry - params->ang_bottom) * M_PI / 180.0;
      rt[2 * k] = std::cos(phi); rt[2 * k + 1] = std::sin(phi);
    }
    d.ip_cmin = (int)std::floor(180.0 / rx) - 2;
    d.ip_ncb = (int)std::floor(540.0 / rx) + 3 - d.ip_cmin;
    ct.resize(2 * (size_t)d.ip_ncb);
    for (int k = 0; k < d.ip_ncb; ++k) {      // raw column c: the horizontal angle at which ha / ang_res_x == c
      const double A = (double)(d.ip_cmin + k) * rx * M_PI / 180.0;
      ct[2 * k] = std::cos(A); ct[2 * k + 1] = std::sin(A);
    }
    double2 *rtd = nullptr, *ctd = nullptr;
    rc |= dalloc(h, &rtd, NS + 6); rc |= dalloc(h, &ctd, (size_t)d.ip_ncb);
    if (!rc && (hipMemcpy(rtd, rt.data(), rt.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(ctd, ct.data(), ct.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)) { h->err = "upload of the projection tables failed"; rc = ALEGO_ERR_HIP; }
    d.ip_rowtab = rtd; d.ip_coltab = ctd;
    {  // col / 10000.0 of every column (the fraction of the intensity row + col / 10000.0, imageProjection.cpp:101): one IEEE fp64 division each, done here once
       // instead of ~35 instructions per emitted cell on the device; the sum with the row and the rounding to f32 stay where they were
      std::vector<double> cf(d.H);
      for (int c = 0; c < d.H; ++c) cf[c] = c / 10000.0;
      double* cfd = nullptr;
      rc |= dalloc(h, &cfd, (size_t)d.H);
      if (!rc && hipMemcpy(cfd, cf.data(), cf.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { h->err = "upload of the column table failed"; rc = ALEGO_ERR_HIP; }
      d.ip_colfrac = cfd;
    }
    d.ip_fast = 0;
    if (params->laser_type == ALEGO_LASER_UNIFORM && ry > 1e-3 && std::fabs(((double)NS + 2.5) * ry - params->ang_bottom) < 80.0 && std::fabs(-3.5 * ry - params->ang_bottom) < 80.0) d.ip_fast |= 1;
    if (rx > 1e-3 && std::fabs((double)d.H * rx - 360.0) < 1e-9 && d.ip_ncb < (1 << 20)) d.ip_fast |= 2;
    h->ip_fast_capable = d.ip_fast;
    d.ip_fast &= env_int("ALEGO_IP_FAST", 3);   // tests: 0 forces the reference expressions for every point
  }
  if (rc) { *out = h; int e = ALEGO_ERR_HIP; std::fprintf(stderr, "alego_create: %s\n", h->err.c_str()); alego_destroy(h); *out = nullptr; return e; }
  // r_w_cur_ = identity, pose quaternions = identity (laserOdometry.cpp:46-47)
  std::vector<double> st(B * LO_STATE_N, 0.0), po(B * 16, 0.0);
  for (size_t b = 0; b < B; ++b) {
    st[b * LO_STATE_N + LS_RW + 0] = st[b * LO_STATE_N + LS_RW + 4] = st[b * LO_STATE_N + LS_RW + 8] = 1.0;
    for (int k = 0; k < 6; ++k) st[b * LO_STATE_N + LS_ROT_P + k] = std::nan("");  // no cached rotation yet
    po[b * 16 + 3] = 1.0; po[b * 16 + 10] = 1.0;
  }
  std::vector<int> sc0(B * SC_COUNT, 0);
  for (size_t b = 0; b < B; ++b) {
    sc0[b * SC_COUNT + SC_CUR] = 1;  // the first scan writes feature buffer 0
    sc0[b * SC_COUNT + SC_FIRST] = 0x7fffffff; sc0[b * SC_COUNT + SC_LAST] = -1;   // accumulators of ip_project, re-armed by ip_front
  }
  d.seg_lo = h->P.deskew_mode ? d.seg_dsk : d.seg_pts;
  std::vector<int> ip0(B * 4, 0);
  for (size_t b = 0; b < B; ++b) ip0[b * 4] = -1;   // imu_ptr_last_ = -1, imu_ptr_front_ = imu_ptr_last_iter_ = 0 (laserOdometry.cpp:17-19)
  if (hipMemcpy(d.imu_ptr, ip0.data(), ip0.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { std::fprintf(stderr, "alego_create: initial state upload failed\n"); alego_destroy(h); return ALEGO_ERR_HIP; }
  h->imu_last_stamp.assign(B, -1e300);
  if (hipMemcpy(d.scal, sc0.data(), sc0.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d.lo_state, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d.poses, po.data(), po.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    std::fprintf(stderr, "alego_create: initial state upload failed\n"); alego_destroy(h); return ALEGO_ERR_HIP;
  }
  if (ip_configure(d) != 0 || lo_configure() != 0 || lm_configure() != 0) { h->err = "hipFuncSetAttribute failed"; std::fprintf(stderr, "alego_create: %s\n", h->err.c_str()); alego_destroy(h); return ALEGO_ERR_HIP; }
  h->lm = lm_host_create(h->P, d, n_slots, h->gsize, h->streams, &h->err);
  if (!h->lm) { std::fprintf(stderr, "alego_create: %s\n", h->err.c_str()); alego_destroy(h); return ALEGO_ERR_HIP; }
  *out = h;
  return ALEGO_OK;
}

void alego_destroy(alego_handle* h) {
  if (!h) return;
  hipSetDevice(h->device);
  (void)sync_all(h);
  if (g_prof == &h->prof) g_prof = nullptr;
  if (h->lm) lm_host_destroy(h->lm);
  for (void* p : h->allocs) (void)guard_free(p);
  for (hipStream_t s : h->streams) hipStreamDestroy(s);
  for (hipStream_t s : h->back) hipStreamDestroy(s);
  for (hipEvent_t e : h->ev_stage) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev_back) (void)hipEventDestroy(e);
  if (h->s_lo) hipStreamDestroy(h->s_lo);
  if (h->s_lm) hipStreamDestroy(h->s_lm);
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  delete h;
}

int alego_synchronize(alego_handle* h) {
  if (!h) return ALEGO_ERR_ARG;
  HIP_TRY(h, sync_all(h));
  return 0;
}

int alego_batch_load(alego_handle* h, int slot, int ring_pos, const alego_point* pts, int32_t n) {
  if (int r = check_slot(h, slot, false)) return r;
  if (ring_pos < 0 || ring_pos >= h->d.ring_len || n < 0 || (n > 0 && !pts)) { h->err = "ring_pos / n out of range or null points"; return ALEGO_ERR_ARG; }
  if (n > h->d.Pcap) { h->err = "scan larger than n_scan*horizon_scan"; return ALEGO_ERR_CAPACITY; }
  hipSetDevice(h->device);
  float4* dst = h->d.in_pts + ((size_t)slot * h->d.ring_len + ring_pos) * h->d.Pcap;
  if (n > 0) HIP_TRY(h, hipMemcpyAsync(dst, pts, (size_t)n * sizeof(alego_point), hipMemcpyHostToDevice, stream_of(h, slot)));
  HIP_TRY(h, hipMemcpyAsync(h->d.in_n + slot * h->d.ring_len + ring_pos, &n, sizeof(int), hipMemcpyHostToDevice, stream_of(h, slot)));
  HIP_TRY(h, hipStreamSynchronize(stream_of(h, slot)));
  return 0;
}

int alego_replay_create(alego_handle* h, int n_bags, int bag_len) {
  if (!h || n_bags < 1 || bag_len < 1) return ALEGO_ERR_ARG;
  if (h->d.bag_pts) { h->err = "alego_replay_create: the bag store exists already"; return ALEGO_ERR_ARG; }
  hipSetDevice(h->device);
  float4* pts = nullptr; int* cnt = nullptr; int2* src = nullptr;
  if (dalloc(h, &pts, (size_t)n_bags * bag_len * h->d.Pcap, false) || dalloc(h, &cnt, (size_t)n_bags * bag_len) || dalloc(h, &src, (size_t)h->d.n_slots)) return ALEGO_ERR_HIP;
  h->d.bag_pts = pts; h->d.bag_n = cnt; h->d.bag_src = src; h->d.n_bags = n_bags; h->d.bag_len = bag_len;
  return 0;
}
int alego_replay_load(alego_handle* h, int bag, int scan, const alego_point* pts, int32_t n) {
  if (!h || !h->d.bag_pts || bag < 0 || bag >= h->d.n_bags || scan < 0 || scan >= h->d.bag_len || n < 0 || (n > 0 && !pts)) return ALEGO_ERR_ARG;
  if (n > h->d.Pcap) { h->err = "scan larger than n_scan*horizon_scan"; return ALEGO_ERR_CAPACITY; }
  hipSetDevice(h->device);
  const size_t idx = (size_t)bag * h->d.bag_len + scan;
  HIP_TRY(h, hipMemcpy(const_cast<float4*>(h->d.bag_pts) + idx * h->d.Pcap, pts, (size_t)n * sizeof(alego_point), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(const_cast<int*>(h->d.bag_n) + idx, &n, sizeof(int), hipMemcpyHostToDevice));
  return 0;
}
int alego_replay_assign(alego_handle* h, int slot, int bag, int start_scan) {
  if (int r = check_slot(h, slot)) return r;
  if (!h->d.bag_pts || bag < 0 || bag >= h->d.n_bags || start_scan < 0) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  const int2 v = make_int2(bag, start_scan % h->d.bag_len);
  HIP_TRY(h, hipMemcpy(const_cast<int2*>(h->d.bag_src) + slot, &v, sizeof(v), hipMemcpyHostToDevice));
  h->replay_assigned = true;
  return 0;
}

int alego_stream_setup(alego_handle* h, int bag, int start_scan) {
  if (!h) return ALEGO_ERR_ARG;
  if (!h->d.bag_pts || bag < 0 || bag >= h->d.n_bags || start_scan < 0) { h->err = "alego_stream_setup: needs alego_replay_create and a valid bag"; return ALEGO_ERR_ARG; }
  if (h->d.n_slots < 3 || h->streams.size() != 1) { h->err = "alego_stream_setup: the handle needs n_slots = 1 + 2 W >= 3 (one stream group)"; return ALEGO_ERR_ARG; }
  if (h->d.traj) { h->err = "alego_stream_setup: the per-scan pose log belongs to the batch path"; return ALEGO_ERR_ARG; }
  if (h->P.deskew_mode) { h->err = "alego_stream_setup: bags carry no stamps / IMU data; the motion de-skew runs through alego_scan_process / alego_lo_process"; return ALEGO_ERR_ARG; }
  hipSetDevice(h->device);
  const int W = (h->d.n_slots - 1) / 2;
  if (int r = alego_replay_assign(h, 0, bag, start_scan)) return r;
  for (int set = 0; set < 2; ++set)
    for (int j = 0; j < W; ++j)   // lane j of either set processes scan (start + group base + j)
      if (int r = alego_replay_assign(h, 1 + set * W + j, bag, start_scan + j)) return r;
  if (!h->s_lo && (hipStreamCreate(&h->s_lo) != hipSuccess || hipStreamCreate(&h->s_lm) != hipSuccess)) { h->err = "alego_stream_setup: hipStreamCreate failed"; return ALEGO_ERR_HIP; }
  h->stream_mode = true; h->lanes = W; h->pose_slot = 0; h->last_lane = -1; h->next_set = 0;
  return 0;
}

int alego_stream_run(alego_handle* h, int first_step, int n_scans, int stages, int sync) {
  if (!h) return ALEGO_ERR_ARG;
  if (!h->stream_mode) { h->err = "alego_stream_run: call alego_stream_setup first"; return ALEGO_ERR_ARG; }
  hipSetDevice(h->device);
  drain_back(h);
  g_prof = &h->prof;
  const int W = h->lanes;
  hipStream_t sA = h->streams[0], sB = h->s_lo, sC = h->s_lm;
  auto ev = [&]() -> hipEvent_t {
    if (h->ev_next == h->ev_pool.size()) { hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); h->ev_pool.push_back(e); }
    return h->ev_pool[h->ev_next++];
  };
  h->ev_next = 0;
  // the three streams start behind whatever any of them did before (earlier runs still in flight, host entry points)
  {
    hipEvent_t eb = ev(), ec = ev(), ea = ev();
    HIP_TRY(h, hipEventRecord(eb, sB)); HIP_TRY(h, hipEventRecord(ec, sC));
    HIP_TRY(h, hipStreamWaitEvent(sA, eb, 0)); HIP_TRY(h, hipStreamWaitEvent(sA, ec, 0));
    HIP_TRY(h, hipEventRecord(ea, sA)); HIP_TRY(h, hipStreamWaitEvent(sB, ea, 0)); HIP_TRY(h, hipStreamWaitEvent(sC, ea, 0));
  }
  hipEvent_t lm_done[2] = {nullptr, nullptr};    // LaserMapping has consumed every lane of the set's previous use
  hipEvent_t lo_first_done = nullptr;            // LaserOdometry of the first scan of the previous group (it still reads the last lane of the group before)
  const bool do_lo = (stages & 2) != 0, do_lm = do_lo && (stages & 4);
  int g = 0;
  for (int base = 0; base < n_scans; base += W, ++g) {
    const int w = std::min(W, n_scans - base), set = h->next_set, lane0 = 1 + set * W;
    h->next_set ^= 1;
    // ---- ImageProjection + feature extraction of w scans, one per lane (stream A)
    if (lm_done[set]) HIP_TRY(h, hipStreamWaitEvent(sA, lm_done[set], 0));
    if (lo_first_done) HIP_TRY(h, hipStreamWaitEvent(sA, lo_first_done, 0));
    DevCtx dl = view(h, lane0, w);
    dl.replay_bag = 1;
    const int pos = (int)(((long long)first_step + base) % h->d.bag_len);
    if (stages & 1) launch_ip(dl, pos, false, sA);
    if (do_lo) launch_fe(dl, sA);
    hipEvent_t fe_done = ev();
    HIP_TRY(h, hipEventRecord(fe_done, sA));
    if (!do_lo) { h->pose_slot = lane0 + w - 1; continue; }
    HIP_TRY(h, hipStreamWaitEvent(sB, fe_done, 0));
    for (int j = 0; j < w; ++j) {
      // ---- LaserOdometry of the stream (slot 0) on the features of lane (stream B)
      DevCtx ds = view(h, 0, 1);
      ds.fs_cur = lane0 + j; ds.fs_last = h->last_lane >= 0 ? h->last_lane : lane0 + j;   // (first scan ever: LaserOdometry only initialises)
      launch_lo(ds, sB);
      hipEvent_t lo_done = ev();
      HIP_TRY(h, hipEventRecord(lo_done, sB));
      if (j == 0) lo_first_done = lo_done;
      const bool odom_valid = h->lo_scans[0]++ > 0;
      if (do_lm) {
        // ---- LaserMapping of that scan (stream C)
        HIP_TRY(h, hipStreamWaitEvent(sC, lo_done, 0));
        if (int r = lm_host_enqueue(h->lm, ds, std::vector<char>(1, odom_valid ? 1 : 0), &h->err, sC)) return r;
      }
      h->last_lane = lane0 + j;
      h->pose_slot = lane0 + j;
    }
    hipEvent_t done = ev();
    HIP_TRY(h, hipEventRecord(done, do_lm ? sC : sB));
    lm_done[set] = done;
  }
  HIP_TRY(h, hipGetLastError());
  if (sync) HIP_TRY(h, sync_all(h));
  return 0;
}

// enqueue IP -> FE -> LO -> LM for slots [slot0, slot0+n) on ring position `pos`
static int enqueue_scan(alego_handle* h, int slot0, int n, int pos, int stages, bool want_labels, bool async_lm = false) {
  DevCtx d = view(h, slot0, n);
  d.replay_bag = (stages & ALEGO_REPLAY_BAG) ? 1 : 0;
  hipStream_t S = stream_of(h, slot0);  // [slot0, slot0+n) lies inside one stream group
  g_prof = &h->prof;
  static const bool dbg = getenv("ALEGO_DEBUG_SYNC") != nullptr;
  auto chk = [&](const char* what) { if (dbg) { hipError_t e = hipStreamSynchronize(S); fprintf(stderr, "[alego dbg] %s: %s\n", what, hipGetErrorString(e)); } };
  if (stages & 1) { launch_ip(d, pos, want_labels, S); chk("ip"); }
  if (stages & 2) {
    if (d.P.deskew_mode) { launch_lo_deskew(d, S); chk("deskew"); }   // adjustDistortion(segmented_cloud, t1), laserOdometry.cpp:115
    launch_fe(d, S); chk("fe");
    launch_lo(d, S); chk("lo");
    std::vector<char> odom_valid(n);
    for (int i = 0; i < n; ++i) odom_valid[i] = h->lo_scans[slot0 + i]++ > 0;
    if ((stages & 4) && async_lm && h->lm_async) {
      // LaserMapping of this scan on the group's back stream, behind the hand-over kernel on S (lm_host.hip)
      const int g = slot0 / h->gsize;
      hipStream_t B = h->back[g];
      const long k = h->grp_scans[g]++;
      const int par = (int)(k & 1);
      if (int r = lm_host_enqueue_async(h->lm, d, odom_valid, &h->err, S, B, h->ev_stage[g * 2 + par], h->ev_back[g * 2 + par], h->ev_back[g * 2 + (par ^ 1)], k)) return r;
      if (d.traj) launch_traj_log(d, B, lm_host_stage_odom(h->lm), par);
      HIP_TRY(h, hipEventRecord(h->ev_back[g * 2 + par], B));
    } else {
      if (stages & 4) { if (int r = lm_host_enqueue(h->lm, d, odom_valid, &h->err)) return r; }
      if (d.traj) launch_traj_log(d, S);
    }
  }
  HIP_TRY(h, hipGetLastError());
  return 0;
}

int alego_batch_run(alego_handle* h, int first_pos, int n_scans, int stages, int sync) {
  if (!h || n_scans < 0) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  const int R = h->d.ring_len;
  if (h->P.deskew_mode && (stages & 2)) { h->err = "alego_batch_run: the motion de-skew needs every scan's stamp (alego_scan_process / alego_lo_process)"; return ALEGO_ERR_ARG; }
  if ((stages & ALEGO_REPLAY_BAG) && (!h->d.bag_pts || !h->replay_assigned)) { h->err = "alego_batch_run: ALEGO_REPLAY_BAG without alego_replay_create / alego_replay_assign"; return ALEGO_ERR_ARG; }
  for (int s = 0; s < n_scans; ++s) {
    int pos;
    if (stages & ALEGO_REPLAY_BAG) {
      pos = (int)(((long long)first_pos + s) % h->d.bag_len);   // every slot adds its own start scan on the device
    } else if ((stages & ALEGO_REPLAY_PINGPONG) && R > 1) {  // 0,1,..,R-1,R-2,..,1,0,1,.. : consecutive scans stay neighbours
      const int period = 2 * (R - 1);
      const int t = ((first_pos + s) % period + period) % period;
      pos = t < R ? t : period - t;
    } else {
      pos = ((first_pos + s) % R + R) % R;
    }
    for (int s0 = 0; s0 < h->d.n_slots; s0 += h->gsize)
      if (int r = enqueue_scan(h, s0, std::min(h->gsize, h->d.n_slots - s0), pos, stages & (7 | ALEGO_REPLAY_BAG), false, /*async_lm=*/true)) return r;
  }
  if (sync) HIP_TRY(h, sync_all(h));
  return 0;
}

static int fetch_pose(alego_handle* h, int slot, alego_pose* odom, alego_pose* map_pose) {
  double po[16], st[LO_STATE_N];
  int sc[SC_COUNT];
  const int pslot = (h->stream_mode && slot == 0) ? h->pose_slot : slot;   // alego_stream_run: the last scan's poses live in its lane
  if (h->stream_mode) HIP_TRY(h, sync_all(h));
  HIP_TRY(h, hipMemcpyAsync(po, h->d.poses + (size_t)pslot * 16, sizeof(po), hipMemcpyDeviceToHost, stream_of(h, slot)));
  HIP_TRY(h, hipMemcpyAsync(st, h->d.lo_state + (size_t)slot * LO_STATE_N, sizeof(st), hipMemcpyDeviceToHost, stream_of(h, slot)));
  HIP_TRY(h, hipMemcpyAsync(sc, h->d.scal + (size_t)slot * SC_COUNT, sizeof(sc), hipMemcpyDeviceToHost, stream_of(h, slot)));
  HIP_TRY(h, hipStreamSynchronize(stream_of(h, slot)));
  if (odom) {
    for (int i = 0; i < 3; ++i) odom->t[i] = po[i];
    for (int i = 0; i < 4; ++i) odom->q[i] = po[3 + i];
    for (int i = 0; i < 6; ++i) odom->params[i] = st[LS_PARAMS + i];
    odom->valid = sc[SC_ODOM_VALID];
  }
  if (map_pose) {
    for (int i = 0; i < 3; ++i) map_pose->t[i] = po[7 + i];
    for (int i = 0; i < 4; ++i) map_pose->q[i] = po[10 + i];
    lm_host_get_params(h->lm, slot, map_pose->params);
    map_pose->valid = sc[SC_ODOM_VALID];
  }
  if (sc[SC_FE_ERR]) { h->err = "feature extraction: a ring's workgroup gave up waiting for the voxel counts of the rings below it (fe_ring_out); the slot's feature clouds are incomplete"; return ALEGO_ERR_HIP; }
  const int lmf = lm_host_get_flags(h->lm, slot);
  if (lmf < 0) { h->err = "LaserMapping device capacity exceeded / launch logic out of sync"; return lmf; }
  return sc[SC_LO_FLAGS] | lmf;
}

int alego_batch_get_pose(alego_handle* h, int slot, alego_pose* odom, alego_pose* map_pose) {
  if (int r = check_slot(h, slot)) return r;
  hipSetDevice(h->device);
  return fetch_pose(h, slot, odom, map_pose);
}

int alego_trajectory_enable(alego_handle* h, int32_t capacity_scans) {
  if (!h || capacity_scans <= 0) return ALEGO_ERR_ARG;
  if (h->stream_mode) { h->err = "alego_trajectory_enable: not available with alego_stream_setup (the poses of a look-ahead stream live in its lanes)"; return ALEGO_ERR_ARG; }
  if (h->d.traj) { h->err = "alego_trajectory_enable: already enabled"; return ALEGO_ERR_ARG; }
  hipSetDevice(h->device);
  HIP_TRY(h, sync_all(h));
  double* t = nullptr;
  int* n = nullptr;
  if (dalloc(h, &t, (size_t)h->d.n_slots * capacity_scans * 14) || dalloc(h, &n, (size_t)h->d.n_slots)) return ALEGO_ERR_HIP;
  h->d.traj = t; h->d.traj_n = n; h->d.traj_cap = capacity_scans;
  return 0;
}
int alego_trajectory_get(alego_handle* h, int slot, int32_t first, int32_t n, double* out14) {
  if (int r = check_slot(h, slot)) return r;
  if (!h->d.traj || first < 0 || n < 0 || (n > 0 && !out14)) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  hipStream_t S = stream_of(h, slot);
  int logged = 0;
  HIP_TRY(h, hipMemcpyAsync(&logged, h->d.traj_n + slot, sizeof(int), hipMemcpyDeviceToHost, S));
  HIP_TRY(h, hipStreamSynchronize(S));
  const int have = std::min(logged, h->d.traj_cap);
  if (first > have || n > have - first) { h->err = "alego_trajectory_get: range beyond the scans logged"; return ALEGO_ERR_ARG; }   // (not first + n: it overflows for large arguments)
  if (n > 0) {
    HIP_TRY(h, hipMemcpyAsync(out14, h->d.traj + ((size_t)slot * h->d.traj_cap + first) * 14, (size_t)n * 14 * sizeof(double), hipMemcpyDeviceToHost, S));
    HIP_TRY(h, hipStreamSynchronize(S));
  }
  return logged;
}

int alego_batch_get_counts(alego_handle* h, int slot, int32_t* out, int cap) {
  if (int r = check_slot(h, slot)) return r;
  hipSetDevice(h->device);
  int sc[SC_COUNT], fc[8], sl[SC_COUNT];
  const bool lane = h->stream_mode && slot == 0;   // alego_stream_run: the per-scan counters of the last scan live in its lane (buffer 0)
  const int cslot = lane ? h->pose_slot : slot;
  if (lane) HIP_TRY(h, sync_all(h));
  HIP_TRY(h, hipMemcpyAsync(sc, h->d.scal + (size_t)cslot * SC_COUNT, sizeof(sc), hipMemcpyDeviceToHost, stream_of(h, slot)));
  HIP_TRY(h, hipMemcpyAsync(sl, h->d.scal + (size_t)slot * SC_COUNT, sizeof(sl), hipMemcpyDeviceToHost, stream_of(h, slot)));
  HIP_TRY(h, hipMemcpyAsync(fc, h->d.feat_cnt + (size_t)cslot * 8, sizeof(fc), hipMemcpyDeviceToHost, stream_of(h, slot)));
  HIP_TRY(h, hipStreamSynchronize(stream_of(h, slot)));
  sc[SC_LO_NSURF] = sl[SC_LO_NSURF]; sc[SC_LO_NCORNER] = sl[SC_LO_NCORNER];   // LaserOdometry's counters belong to the stream's own slot
  const int cur = lane ? 0 : sc[SC_CUR];  // buffer written by the last processed scan
  int v[16] = {sc[SC_PVALID_OUT], sc[SC_M], sc[SC_NOUT], fc[cur * 4 + 0], fc[cur * 4 + 1], fc[cur * 4 + 2], fc[cur * 4 + 3],
               sc[SC_LO_NSURF], sc[SC_LO_NCORNER], 0, 0, 0, 0, 0, 0, 0};  // v[15] = map rebuilds so far
  lm_host_get_counts(h->lm, slot, v + 9);
  for (int i = 0; i < cap && i < 16; ++i) out[i] = v[i];
  return 0;
}

static int download_seg(alego_handle* h, int slot, alego_seg_out* out) {
  const DevCtx& d = h->d;
  int sc[SC_COUNT];
  HIP_TRY(h, hipMemcpyAsync(sc, d.scal + (size_t)slot * SC_COUNT, sizeof(sc), hipMemcpyDeviceToHost, stream_of(h, slot)));
  HIP_TRY(h, hipStreamSynchronize(stream_of(h, slot)));
  const int M = sc[SC_M], NO = sc[SC_NOUT];
  out->m = M; out->n_outlier = NO;
  if (M > out->seg_cap || NO > out->outlier_cap) { h->err = "seg/outlier capacity too small"; return ALEGO_ERR_CAPACITY; }
  const size_t base = (size_t)slot * d.N;
  if (out->seg) HIP_TRY(h, hipMemcpyAsync(out->seg, d.seg_pts + base, (size_t)M * 16, hipMemcpyDeviceToHost, stream_of(h, slot)));
  if (out->ground) HIP_TRY(h, hipMemcpyAsync(out->ground, d.seg_ground + base, (size_t)M, hipMemcpyDeviceToHost, stream_of(h, slot)));
  if (out->col) HIP_TRY(h, hipMemcpyAsync(out->col, d.seg_col + base, (size_t)M * 4, hipMemcpyDeviceToHost, stream_of(h, slot)));
  if (out->range) HIP_TRY(h, hipMemcpyAsync(out->range, d.seg_range + base, (size_t)M * 4, hipMemcpyDeviceToHost, stream_of(h, slot)));
  if (out->ring_start) HIP_TRY(h, hipMemcpyAsync(out->ring_start, d.ring_start + (size_t)slot * d.NS, d.NS * 4, hipMemcpyDeviceToHost, stream_of(h, slot)));
  if (out->ring_end) HIP_TRY(h, hipMemcpyAsync(out->ring_end, d.ring_end + (size_t)slot * d.NS, d.NS * 4, hipMemcpyDeviceToHost, stream_of(h, slot)));
  HIP_TRY(h, hipMemcpyAsync(out->orientation, d.ori + (size_t)slot * 4, 12, hipMemcpyDeviceToHost, stream_of(h, slot)));
  if (out->outlier) HIP_TRY(h, hipMemcpyAsync(out->outlier, d.outlier + base, (size_t)NO * 16, hipMemcpyDeviceToHost, stream_of(h, slot)));
  if (out->label_image) HIP_TRY(h, hipMemcpyAsync(out->label_image, d.label_img + base, (size_t)d.N * 4, hipMemcpyDeviceToHost, stream_of(h, slot)));
  HIP_TRY(h, hipStreamSynchronize(stream_of(h, slot)));
  return 0;
}

static int download_feat(alego_handle* h, int slot, alego_feat_out* f) {
  const DevCtx& d = h->d;
  int fc[4], M, cur;
  HIP_TRY(h, hipMemcpy(&cur, d.scal + (size_t)slot * SC_COUNT + SC_CUR, 4, hipMemcpyDeviceToHost));
  HIP_TRY(h, hipMemcpyAsync(fc, d.feat_cnt + ((size_t)slot * 2 + cur) * 4, sizeof(fc), hipMemcpyDeviceToHost, stream_of(h, slot)));
  HIP_TRY(h, hipMemcpyAsync(&M, d.scal + (size_t)slot * SC_COUNT + SC_M, 4, hipMemcpyDeviceToHost, stream_of(h, slot)));
  HIP_TRY(h, hipStreamSynchronize(stream_of(h, slot)));
  f->n_sharp = fc[0]; f->n_less_sharp = fc[1]; f->n_flat = fc[2]; f->n_less_flat = fc[3];
  if (fc[0] > f->sharp_cap || fc[1] > f->less_sharp_cap || fc[2] > f->flat_cap || fc[3] > f->less_flat_cap) { h->err = "feature capacity too small"; return ALEGO_ERR_CAPACITY; }
  alego_point* dst[4] = {f->sharp, f->less_sharp, f->flat, f->less_flat};
  for (int k = 0; k < 4; ++k)
    if (dst[k]) HIP_TRY(h, hipMemcpyAsync(dst[k], d.feat[k] + ((size_t)slot * 2 + cur) * d.fcap[k], (size_t)fc[k] * 16, hipMemcpyDeviceToHost, stream_of(h, slot)));
  if (f->point_label) HIP_TRY(h, hipMemcpyAsync(f->point_label, d.plabel + (size_t)slot * d.N, (size_t)M * 4, hipMemcpyDeviceToHost, stream_of(h, slot)));
  HIP_TRY(h, hipStreamSynchronize(stream_of(h, slot)));
  return 0;
}

int alego_ip_process(alego_handle* h, const alego_scan_in* in, alego_seg_out* out) {
  if (!h || !in || !out) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  if (int r = alego_batch_load(h, 0, 0, in->pts, in->n)) return r;
  const DevCtx d = view(h, 0, 1);
  g_prof = &h->prof;
  launch_ip(d, 0, out->label_image != nullptr, h->stream);
  HIP_TRY(h, hipGetLastError());
  out->stamp = in->stamp;   // /segmented_cloud and /seg_info carry the input's stamp (imageProjection.cpp:318-336)
  return download_seg(h, 0, out);
}

int alego_lo_process(alego_handle* h, const alego_seg_out* in, alego_feat_out* feat, alego_pose* odom) {
  if (!h || !in) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  drain_back(h);
  const DevCtx& d = h->d;
  if (in->m < 0 || in->m > d.N) { h->err = "segmented cloud larger than n_scan*horizon_scan"; return ALEGO_ERR_CAPACITY; }
  if (!in->ring_start || !in->ring_end || (in->m > 0 && (!in->seg || !in->ground || !in->col || !in->range))) { h->err = "alego_lo_process: null input array"; return ALEGO_ERR_ARG; }
  const size_t M = in->m;
  HIP_TRY(h, hipMemcpyAsync(d.seg_pts, in->seg, M * 16, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d.seg_ground, in->ground, M, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d.seg_col, in->col, M * 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d.seg_range, in->range, M * 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d.ring_start, in->ring_start, d.NS * 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d.ring_end, in->ring_end, d.NS * 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d.scal + SC_M, &in->m, 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d.ori, in->orientation, 12, hipMemcpyHostToDevice, h->stream));       // seg_info's start / end orientation: adjustDistortion reads them
  HIP_TRY(h, hipMemcpyAsync(d.scan_stamp, &in->stamp, 8, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (int r = enqueue_scan(h, 0, 1, 0, 2, false)) return r;
  if (feat) { if (int r = download_feat(h, 0, feat)) return r; }
  const int r = fetch_pose(h, 0, odom, nullptr);
  return r < 0 ? r : (r & 7);
}

int alego_lm_process(alego_handle* h, const alego_point* corner_last, int32_t n_corner, const alego_point* surf_last,
                     int32_t n_surf, const alego_point* outlier, int32_t n_outlier, const alego_pose* odom, alego_pose* map_pose) {
  if (!h || !odom) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  drain_back(h);
  g_prof = &h->prof;
  return lm_host_process_host(h->lm, h->d, corner_last, n_corner, surf_last, n_surf, outlier, n_outlier, odom, map_pose, &h->err);
}

int alego_scan_process(alego_handle* h, int slot, const alego_scan_in* in, int stages, alego_seg_out* seg,
                       alego_feat_out* feat, alego_pose* odom, alego_pose* map_pose) {
  if (int r = check_slot(h, slot)) return r;
  if (!in) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  if (int r = alego_batch_load(h, slot, 0, in->pts, in->n)) return r;
  HIP_TRY(h, hipMemcpyAsync(h->d.scan_stamp + slot, &in->stamp, 8, hipMemcpyHostToDevice, stream_of(h, slot)));
  HIP_TRY(h, hipStreamSynchronize(stream_of(h, slot)));   // (`in` is the caller's)
  if (int r = enqueue_scan(h, slot, 1, 0, stages, seg && seg->label_image)) return r;
  if (seg) { if (int r = download_seg(h, slot, seg)) return r; seg->stamp = in->stamp; }
  if (feat && (stages & 2)) { if (int r = download_feat(h, slot, feat)) return r; }
  return fetch_pose(h, slot, odom, map_pose);
}

int alego_set_lo_params(alego_handle* h, int slot, const double* p6) {
  if (int r = check_slot(h, slot)) return r;
  hipSetDevice(h->device);
  HIP_TRY(h, hipMemcpyAsync(h->d.lo_state + (size_t)slot * LO_STATE_N + LS_PARAMS, p6, 48, hipMemcpyHostToDevice, stream_of(h, slot)));
  HIP_TRY(h, hipStreamSynchronize(stream_of(h, slot)));
  return 0;
}
int alego_set_lm_params(alego_handle* h, int slot, const double* p6) {
  if (int r = check_slot(h, slot)) return r;
  hipSetDevice(h->device);
  return lm_host_set_params(h->lm, slot, p6, &h->err);
}

// /undistorted (laserOdometry.cpp:56,718-725): the de-skewed segmented cloud of the slot's last scan
int alego_lo_get_undistorted(alego_handle* h, int slot, alego_point* out, int32_t cap) {
  if (int r = check_slot(h, slot)) return r;
  if (!h->P.deskew_mode) { h->err = "alego_lo_get_undistorted: deskew_mode is 0 (adjustDistortion is not run: laserOdometry.cpp:115)"; return ALEGO_ERR_ARG; }
  if (cap < 0 || (cap > 0 && !out)) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  int M = 0;
  HIP_TRY(h, hipMemcpyAsync(&M, h->d.scal + (size_t)slot * SC_COUNT + SC_M_DSK, 4, hipMemcpyDeviceToHost, stream_of(h, slot)));   // lo_deskew's own count, not SC_M
  HIP_TRY(h, hipStreamSynchronize(stream_of(h, slot)));
  if (M > cap) return ALEGO_ERR_CAPACITY;
  if (M > 0) HIP_TRY(h, hipMemcpy(out, h->d.seg_dsk + (size_t)slot * h->d.N, (size_t)M * sizeof(alego_point), hipMemcpyDeviceToHost));
  return M;
}

// The standalone LaserOdometry node publishes /odom/lidar as /odom -> /base_link: tf_o2b = tf_o2l * tf_b2l^-1, quaternion from its rotation block
// (LO.cpp:588-608; the nodelet publishes /odom -> /laser as it is, laserOdometry.cpp:513-529).  Host arithmetic only (a publishing convention, not part
// of the hot path): general 4 x 4 inverse by Gauss-Jordan with partial pivoting, Eigen's Quaterniond(Matrix3d) (largest-diagonal branch selection).
int alego_pose_o2b(const alego_pose* o2l, const double* tf_b2l, alego_pose* o2b) {
  if (!o2l || !tf_b2l || !o2b) return ALEGO_ERR_ARG;
  double A[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { A[i][j] = tf_b2l[4 * i + j]; A[i][4 + j] = i == j ? 1.0 : 0.0; }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    for (int r = c + 1; r < 4; ++r) if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
    if (!(std::fabs(A[piv][c]) > 0.0)) return ALEGO_ERR_ARG;   // singular (or NaN) mount transform
    if (piv != c) for (int j = 0; j < 8; ++j) std::swap(A[piv][j], A[c][j]);
    const double inv = 1.0 / A[c][c];
    for (int j = 0; j < 8; ++j) A[c][j] *= inv;
    for (int r = 0; r < 4; ++r) if (r != c) { const double f = A[r][c]; if (f != 0.0) for (int j = 0; j < 8; ++j) A[r][j] -= f * A[c][j]; }
  }
  const double w = o2l->q[0], x = o2l->q[1], y = o2l->q[2], z = o2l->q[3];
  const double T[4][4] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), o2l->t[0]},
                          {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x), o2l->t[1]},
                          {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y), o2l->t[2]},
                          {0, 0, 0, 1}};
  double M[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double acc = 0; for (int k = 0; k < 4; ++k) acc += T[i][k] * A[k][4 + j]; M[i][j] = acc; }
  *o2b = *o2l;
  for (int i = 0; i < 3; ++i) o2b->t[i] = M[i][3];
  double q[4];   // w, x, y, z
  double t = M[0][0] + M[1][1] + M[2][2];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0); q[0] = 0.5 * t; t = 0.5 / t;
    q[1] = (M[2][1] - M[1][2]) * t; q[2] = (M[0][2] - M[2][0]) * t; q[3] = (M[1][0] - M[0][1]) * t;
  } else {
    int i = 0;
    if (M[1][1] > M[0][0]) i = 1;
    if (M[2][2] > M[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(M[i][i] - M[j][j] - M[k][k] + 1.0);
    q[1 + i] = 0.5 * t; t = 0.5 / t;
    q[0] = (M[k][j] - M[j][k]) * t; q[1 + j] = (M[j][i] + M[i][j]) * t; q[1 + k] = (M[k][i] + M[i][k]) * t;
  }
  for (int i = 0; i < 4; ++i) o2b->q[i] = q[i];
  return ALEGO_OK;
}

// imuHandler (laserOdometry.cpp:761-802): the per-sample trigonometry here, ring bookkeeping + dead reckoning on the device
int alego_lo_push_imu(alego_handle* h, int slot, const alego_imu* smp, int32_t n) {
  if (int r = check_slot(h, slot)) return r;
  if (n < 0 || (n > 0 && !smp)) return ALEGO_ERR_ARG;
  if (n == 0) return 0;
  hipSetDevice(h->device);
  std::vector<double> pre((size_t)n * 7);
  double prev = h->imu_last_stamp[slot];
  for (int i = 0; i < n; ++i) {
    const alego_imu& m = smp[i];
    if (!(m.stamp >= prev)) { h->err = "alego_lo_push_imu: IMU stamps must not decrease"; return ALEGO_ERR_ARG; }
    prev = m.stamp;
    const double qw = m.orientation[0], qx = m.orientation[1], qy = m.orientation[2], qz = m.orientation[3];
    // tf::Matrix3x3(ori).getRPY(roll, pitch, yaw) (:765-767)  [upstream tf: setRotation + getEulerYPR, solution 1]
    double roll, pitch, yaw;
    {
      const double dd = qx * qx + qy * qy + qz * qz + qw * qw, sc = 2.0 / dd;
      const double xs = qx * sc, ys = qy * sc, zs = qz * sc;
      const double wx = qw * xs, wy = qw * ys, wz = qw * zs, xx = qx * xs, xy = qx * ys, xz = qx * zs, yy = qy * ys, yz = qy * zs, zz = qz * zs;
      const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
      if (std::fabs(m20) >= 1) {
        yaw = 0;
        const double delta = std::atan2(m21, m22);
        if (m20 < 0) { pitch = M_PI / 2.0; roll = delta; } else { pitch = -M_PI / 2.0; roll = delta; }
      } else {
        pitch = -std::asin(m20);
        roll = std::atan2(m21 / std::cos(pitch), m22 / std::cos(pitch));
        yaw = std::atan2(m10 / std::cos(pitch), m00 / std::cos(pitch));
      }
    }
    const double acc_x = m.linear_acceleration[0] + 9.81 * std::sin(pitch);                      // :768-770
    const double acc_y = m.linear_acceleration[1] - 9.81 * std::cos(pitch) * std::sin(roll);
    const double acc_z = m.linear_acceleration[2] - 9.81 * std::cos(pitch) * std::cos(roll);
    // Eigen::Quaternionf(w, x, y, z).toRotationMatrix() * Vector3f(acc) in f32 (:785-786)
    const float w = (float)qw, x = (float)qx, y = (float)qy, z = (float)qz;
    const float tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    const float R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
    const float a[3] = {(float)acc_x, (float)acc_y, (float)acc_z};
    double* o = &pre[(size_t)i * 7];
    o[0] = m.stamp; o[1] = roll; o[2] = pitch; o[3] = yaw;
    for (int k = 0; k < 3; ++k) o[4 + k] = (double)(R[k][0] * a[0] + (R[k][1] * a[1] + R[k][2] * a[2]));
  }
  DevTemps T;
  double* dp;
  HIP_TRY(h, T.get(&dp, pre.size() * 8));
  hipStream_t S = stream_of(h, slot);
  HIP_TRY(h, hipMemcpyAsync(dp, pre.data(), pre.size() * 8, hipMemcpyHostToDevice, S));
  launch_lo_imu_push(h->d, slot, dp, n, S);
  HIP_TRY(h, hipStreamSynchronize(S));
  h->imu_last_stamp[slot] = prev;
  return 0;
}

int alego_profile_enable(alego_handle* h, int on) {
  if (!h) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  HIP_TRY(h, sync_all(h));
  h->prof.reset();
  h->prof.on = on != 0;
  return 0;
}

int alego_profile_report(alego_handle* h, char* names, int names_cap, double* total_ms, int* launches, int cap) {
  if (!h) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  HIP_TRY(h, sync_all(h));
  Profiler& P = h->prof;
  const int nk = (int)P.names.size();
  std::vector<double> tot(nk, 0.0);
  std::vector<int> cnt(nk, 0);
  for (auto& r : P.recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { tot[r.id] += ms; cnt[r.id]++; }
  }
  std::string joined;
  for (int i = 0; i < nk; ++i) { if (i) joined += ";"; joined += P.names[i]; }
  if ((int)joined.size() + 1 > names_cap || nk > cap) { h->err = "profile_report: buffers too small"; return ALEGO_ERR_CAPACITY; }
  std::memcpy(names, joined.c_str(), joined.size() + 1);
  for (int i = 0; i < nk; ++i) { total_ms[i] = tot[i]; launches[i] = cnt[i]; }
  return nk;
}

int alego_debug_voxel(alego_handle* h, const alego_point* pts, int n, float leaf, alego_point* out, int cap) {
  if (!h || n < 0 || (n > 0 && !pts)) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  g_prof = &h->prof;
  DevTemps T;
  float4 *din = nullptr, *dout = nullptr;
  int* cnt = nullptr;
  const int c = n > 0 ? n : 1;
  HIP_TRY(h, T.get(&din, (size_t)c * 16)); HIP_TRY(h, T.get(&dout, (size_t)c * 16)); HIP_TRY(h, T.get(&cnt, 8));
  const int hc[2] = {n, 0};
  HIP_TRY(h, hipMemcpy(cnt, hc, 8, hipMemcpyHostToDevice));
  if (n) HIP_TRY(h, hipMemcpy(din, pts, (size_t)n * 16, hipMemcpyHostToDevice));
  VoxJob job{din, cnt, dout, cnt + 1, nullptr, leaf, c, c, nullptr, 0};
  VoxCtx V;
  if (vox_create(&V, &job, 1, &h->err)) return ALEGO_ERR_HIP;
  int rc = vox_run(V, h->stream, &h->err);
  hipError_t e = hipStreamSynchronize(h->stream);
  int nout = 0;
  if (e == hipSuccess) e = hipMemcpy(&nout, cnt + 1, 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == 0 && nout > cap) { h->err = "debug_voxel: output capacity"; rc = ALEGO_ERR_CAPACITY; }
  if (e == hipSuccess && rc == 0 && nout) e = hipMemcpy(out, dout, (size_t)nout * 16, hipMemcpyDeviceToHost);
  vox_destroy(&V);
  if (e != hipSuccess) { h->err = std::string("debug_voxel: ") + hipGetErrorString(e); return ALEGO_ERR_HIP; }
  return rc ? rc : nout;
}

int alego_debug_math(alego_handle* h, int mode, const float* a, const float* b, float* out, int n) {
  if (!h || n < 0 || mode < 0 || mode > 3 || (n > 0 && (!a || !out || (mode < 2 && !b)))) return ALEGO_ERR_ARG;
  if (n == 0) return 0;
  hipSetDevice(h->device);
  DevTemps T;
  float *da, *db, *dout;
  HIP_TRY(h, T.get(&da, (size_t)n * 4)); HIP_TRY(h, T.get(&db, (size_t)n * 4)); HIP_TRY(h, T.get(&dout, (size_t)n * 4));
  HIP_TRY(h, hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice));
  if (b) HIP_TRY(h, hipMemcpy(db, b, (size_t)n * 4, hipMemcpyHostToDevice));
  // the probe kernel takes (y, x): atan2f(y, x), hypotf(x, y), sinf(y), cosf(y)
  launch_atan2f_probe(da, db, dout, n, mode, h->stream);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost));
  return 0;
}
int alego_debug_atan2f(alego_handle* h, const float* y, const float* x, float* out, int n) { return alego_debug_math(h, 0, y, x, out, n); }

int alego_debug_std_sort(alego_handle* h, const uint32_t* keys, int n, int depth_limit, int32_t* order) {
  if (!h || n < 0 || n > 4096 || (n > 0 && (!keys || !order))) return ALEGO_ERR_ARG;
  if (n == 0) return 0;
  hipSetDevice(h->device);
  DevTemps T;
  uint32_t* dk;
  int* dp;
  HIP_TRY(h, T.get(&dk, (size_t)n * 4)); HIP_TRY(h, T.get(&dp, (size_t)n * 4));
  HIP_TRY(h, hipMemcpy(dk, keys, (size_t)n * 4, hipMemcpyHostToDevice));
  if (launch_stdsort_probe(dk, n, depth_limit, dp, h->stream)) return ALEGO_ERR_ARG;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::vector<int> pos(n), arr(n);
  HIP_TRY(h, hipMemcpy(pos.data(), dp, (size_t)n * 4, hipMemcpyDeviceToHost));
  // the device returns where the partition phase left every element; __final_insertion_sort is a stable sort of that arrangement
  for (int i = 0; i < n; ++i) { if (pos[i] < 0 || pos[i] >= n) { h->err = "alego_debug_std_sort: arrangement out of range"; return ALEGO_ERR_HIP; } arr[pos[i]] = i; }
  std::stable_sort(arr.begin(), arr.end(), [&](int a, int b) { return keys[a] < keys[b]; });
  for (int i = 0; i < n; ++i) order[i] = arr[i];
  return 0;
}

int alego_debug_eval_blocks(alego_handle* h, int type, int n, const double* geom13, const double* params6, double* res, double* jac6) {
  if (!h || n < 0 || type < 0 || type > 3 || !params6 || (n > 0 && (!geom13 || !res || !jac6))) return ALEGO_ERR_ARG;
  if (n == 0) return 0;
  hipSetDevice(h->device);
  DevTemps T;
  double *dg, *dp, *dr, *dj;
  HIP_TRY(h, T.get(&dg, (size_t)n * 13 * 8)); HIP_TRY(h, T.get(&dp, 48)); HIP_TRY(h, T.get(&dr, (size_t)n * 8)); HIP_TRY(h, T.get(&dj, (size_t)n * 48));
  HIP_TRY(h, hipMemcpy(dg, geom13, (size_t)n * 13 * 8, hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(dp, params6, 48, hipMemcpyHostToDevice));
  launch_dbg_eval_blocks(type, n, dg, dp, dr, dj, h->stream);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(res, dr, (size_t)n * 8, hipMemcpyDeviceToHost));
  HIP_TRY(h, hipMemcpy(jac6, dj, (size_t)n * 48, hipMemcpyDeviceToHost));
  return 0;
}

int alego_debug_transform_to_start(alego_handle* h, const double* params6, const alego_point* pts, int n, alego_point* out) {
  if (!h || n < 0 || !params6 || (n > 0 && (!pts || !out))) return ALEGO_ERR_ARG;
  if (n == 0) return 0;
  hipSetDevice(h->device);
  DevTemps T;
  double* dp;
  float4 *di, *dout;
  HIP_TRY(h, T.get(&dp, 48)); HIP_TRY(h, T.get(&di, (size_t)n * 16)); HIP_TRY(h, T.get(&dout, (size_t)n * 16));
  HIP_TRY(h, hipMemcpy(dp, params6, 48, hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(di, pts, (size_t)n * 16, hipMemcpyHostToDevice));
  launch_dbg_transform_to_start(dp, di, n, dout, h->stream);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(out, dout, (size_t)n * 16, hipMemcpyDeviceToHost));
  return 0;
}

int alego_debug_check_guards(char* report, int cap) {
  std::string r;
  const int bad = guard_check(&r);
  if (report && cap > 0) { std::strncpy(report, r.c_str(), (size_t)cap - 1); report[cap - 1] = 0; }
  return bad;
}

int alego_debug_set_option(alego_handle* h, const char* name, int value) {
  if (!h || !name) return ALEGO_ERR_ARG;
  const std::string s(name);
  DevCtx& d = h->d;
  if (s == "ALEGO_CC_FUSED") d.opt_cc_fused = value != 0;
  else if (s == "ALEGO_IP_FUSED") d.opt_ip_fused = value != 0;
  else if (s == "ALEGO_IP_HALF") d.opt_ip_half = value != 0;
  else if (s == "ALEGO_CC_TILE") d.opt_cc_tile = value != 0;
  else if (s == "ALEGO_IP_BAND") {   // the banded path expects the component statistics at zero between scans (it re-zeroes what it touched); the cc_stats path leaves them set
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemset(d.cc_size, 0, (size_t)d.n_slots * d.N * sizeof(int)));
    HIP_TRY(h, hipMemset(d.cc_rows, 0, (size_t)d.n_slots * d.N * sizeof(unsigned long long)));
    d.opt_ip_band = value != 0;
  }
  else if (s == "ALEGO_FE_PICK1") d.opt_fe_pick1 = value != 0;
  else if (s == "ALEGO_FE_FUSED") d.opt_fe_fused = value != 0;
  else if (s == "ALEGO_FE_CAND") d.opt_fe_cand = value;
  else if (s == "ALEGO_LO_BOX_LDS") d.opt_lo_box_lds = value;
  else if (s == "ALEGO_LO_GRID") d.opt_lo_grid = value != 0;
  else if (s == "ALEGO_FE_SPIN") d.opt_fo_spin = value;
  else if (s == "ALEGO_FE_PAD8") d.opt_fo_pad8 = value != 0;
  else if (s == "ALEGO_FE_ERR_CLEAR") {   // the per-slot reset of the sticky SC_FE_ERR (dev_common.h): value = slot, -1 = every slot; the caller has drained the handle's streams
    if (value < -1 || value >= d.n_slots) return ALEGO_ERR_ARG;
    HIP_TRY(h, hipDeviceSynchronize());
    if (value >= 0) HIP_TRY(h, hipMemset(d.scal + (size_t)value * SC_COUNT + SC_FE_ERR, 0, 4));
    else HIP_TRY(h, hipMemset2D(d.scal + SC_FE_ERR, SC_COUNT * sizeof(int), 0, 4, d.n_slots));
  }
  else if (s == "ALEGO_MAP_MERGE") { if (int r = lm_host_set_map_merge(h->lm, value != 0, &h->err)) return r; d.opt_map_merge = value != 0; }
  else if (s == "ALEGO_IP_FAST") d.ip_fast = h->ip_fast_capable & value;
  else if (s == "ALEGO_POKE_GUARD") { HIP_TRY(h, hipMemset(d.scal + (size_t)d.n_slots * SC_COUNT + value, 0xFF, 4)); }   // tests of the guard pages: a write `value` ints past the end of an array
  else if (s == "ALEGO_SHARD_SLICE") return lm_host_debug_slice(h->lm, value & 0xff, value >> 8, &h->err);   // tests: rank | world << 8 without a communicator
  else { h->err = "unknown option " + s; return ALEGO_ERR_ARG; }
  return 0;
}

// ---- loop closure ---------------------------------------------------------------------------------------------------
int alego_loop_detect(const alego_params* P, const float* keyposes6, const double* stamps, int32_t n, const double cur_xyz[3]) {
  // detectLoopClosure :771-790 (pcl::KdTreeFLANN::radiusSearch: f32 squared distances, ascending; ties by index)
  if (!P || n <= 0 || !keyposes6 || !stamps || !cur_xyz) return -1;
  const float cx = (float)cur_xyz[0], cy = (float)cur_xyz[1], cz = (float)cur_xyz[2];
  const float r2 = (float)(P->lc_search_radius * P->lc_search_radius);
  std::vector<std::pair<float, int>> cand;
  for (int i = 0; i < n; ++i) {
    float r = 0.f, df;
    df = keyposes6[i * 6 + 0] - cx; r += df * df; df = keyposes6[i * 6 + 1] - cy; r += df * df; df = keyposes6[i * 6 + 2] - cz; r += df * df;
    if (r < r2) cand.emplace_back(r, i);
  }
  std::sort(cand.begin(), cand.end());
  for (const auto& c : cand)
    if (stamps[n - 1] - stamps[c.second] > P->lc_min_time_gap) return c.second;
  return -1;
}
int alego_loop_closure_icp(alego_handle* h, const alego_kf_in* latest, const alego_kf_in* history, int32_t n_history, alego_icp_result* out,
                           alego_point* target_out, int32_t target_cap) {
  if (!h || !latest || !out || n_history < 0 || (n_history > 0 && !history)) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  g_prof = &h->prof;
  return icp_run(h->P, latest, history, n_history, out, target_out, target_out ? target_cap : 0, h->stream, &h->err);
}

// ---- one registration sharded over the GPUs of a node ---------------------------------------------------------------
int alego_dist_unique_id(char id[ALEGO_DIST_ID_BYTES]) { return id ? lm_host_dist_unique_id(id) : ALEGO_ERR_ARG; }
int alego_dist_init(alego_handle* h, int rank, int world, const char id[ALEGO_DIST_ID_BYTES]) {
  if (!h || !id) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  drain_back(h);
  return lm_host_dist_init(h->lm, rank, world, id, &h->err);
}
int alego_dist_allreduce_probe(alego_handle* h, int iters, double* usec_per_allreduce) {
  if (!h) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  drain_back(h);
  return lm_host_dist_probe(h->lm, iters, usec_per_allreduce, &h->err);
}
int alego_dist_shutdown(alego_handle* h) {
  if (!h) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  drain_back(h);
  return lm_host_dist_shutdown(h->lm);
}

// ---- key-frame pass-through ---------------------------------------------------------------------------------------
int alego_lm_keyframe_count(alego_handle* h, int slot) {
  if (int r = check_slot(h, slot)) return r;
  hipSetDevice(h->device);
  return lm_host_keyframe_count(h->lm, slot);
}
int alego_lm_get_keyframe(alego_handle* h, int slot, int kf_id, alego_keyframe* out) {
  if (int r = check_slot(h, slot)) return r;
  if (!out) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  return lm_host_get_keyframe(h->lm, slot, kf_id, out, &h->err);
}
int alego_lm_set_keypose(alego_handle* h, int slot, int kf_id, const float pose6[6]) {
  if (int r = check_slot(h, slot)) return r;
  if (!pose6) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  g_prof = &h->prof;
  return lm_host_set_keypose(h->lm, h->d, slot, kf_id, pose6, &h->err);
}
int alego_lm_reset_window(alego_handle* h, int slot) {
  if (int r = check_slot(h, slot)) return r;
  hipSetDevice(h->device);
  return lm_host_reset_window(h->lm, slot, &h->err);
}
int alego_lm_apply_correction(alego_handle* h, int slot, const double rc[12]) {
  if (int r = check_slot(h, slot)) return r;
  if (!rc) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  g_prof = &h->prof;
  return lm_host_apply_correction(h->lm, h->d, slot, rc, &h->err);
}
int alego_lm_add_keyframe(alego_handle* h, int slot, const float pose6[6], const alego_point* corner, int32_t n_corner,
                          const alego_point* surf, int32_t n_surf, const alego_point* outlier, int32_t n_outlier) {
  if (int r = check_slot(h, slot)) return r;
  if (!pose6) return ALEGO_ERR_ARG;
  hipSetDevice(h->device);
  g_prof = &h->prof;
  return lm_host_add_keyframe(h->lm, h->d, slot, pose6, corner, n_corner, surf, n_surf, outlier, n_outlier, &h->err);
}

int alego_debug_get(alego_handle* h, int slot, const char* name, void* out, int cap_bytes, int* count, int* dtype) {
  if (int r = check_slot(h, slot)) return r;
  hipSetDevice(h->device);
  const DevCtx& d = h->d;
  int sc[SC_COUNT];
  HIP_TRY(h, hipMemcpyAsync(sc, d.scal + (size_t)slot * SC_COUNT, sizeof(sc), hipMemcpyDeviceToHost, stream_of(h, slot)));
  HIP_TRY(h, hipStreamSynchronize(stream_of(h, slot)));
  const std::string s(name);
  const size_t base = (size_t)slot * d.N;
  const int M = sc[SC_M];
  const int cur = sc[SC_CUR];  // buffer written by the last processed scan
  int fc[8];
  HIP_TRY(h, hipMemcpy(fc, d.feat_cnt + (size_t)slot * 8, sizeof(fc), hipMemcpyDeviceToHost));
  const void* src = nullptr;
  size_t n = 0;
  int dt = 0, esz = 4;
  auto set = [&](const void* p, size_t cnt, int t) { src = p; n = cnt; dt = t; esz = t == 1 ? 8 : t == 3 ? 1 : 4; };
  if (s == "range_img") set(d.range_img + base, d.N, 0);
  else if (s == "label_img") set(d.label_img + base, d.N, 2);
  else if (s == "flag_img") set(d.flag_img + base, d.N, 3);
  else if (s == "owner") set(d.owner + base, d.N, 2);
  else if (s == "parent") set(d.parent + base, d.N, 2);
  else if (s == "seg_cloud") set(d.seg_pts + base, (size_t)M * 4, 0);
  else if (s == "outlier") set(d.outlier + base, (size_t)sc[SC_NOUT] * 4, 0);
  else if (s == "undistorted" && d.P.deskew_mode) set(d.seg_dsk + base, (size_t)sc[SC_M_DSK] * 4, 0);
  else if (s == "imu_ptr") set(d.imu_ptr + (size_t)slot * 4, 3, 2);
  else if (s == "imu_ring") set(d.imu_ring + (size_t)slot * ALEGO_IMU_Q * 10, ALEGO_IMU_Q * 10, 1);
  else if (s == "seg_ground") set(d.seg_ground + base, M, 3);
  else if (s == "seg_col") set(d.seg_col + base, M, 2);
  else if (s == "seg_range") set(d.seg_range + base, M, 0);
  else if (s == "ring_start") set(d.ring_start + (size_t)slot * d.NS, d.NS, 2);
  else if (s == "ring_end") set(d.ring_end + (size_t)slot * d.NS, d.NS, 2);
  else if (s == "orientation") set(d.ori + (size_t)slot * 4, 3, 0);
  else if (s == "scal") set(d.scal + (size_t)slot * SC_COUNT, SC_COUNT, 2);
  else if (s == "curv_d") set(d.cd + base, M, 0);
  else if (s == "picked_occl") set(d.picked0 + base, M, 3);
  else if (s == "point_label") set(d.plabel + base, M, 2);
  else if (s == "sharp") set(d.feat[0] + ((size_t)slot * 2 + cur) * d.fcap[0], (size_t)fc[cur * 4 + 0] * 4, 0);
  else if (s == "less_sharp") set(d.feat[1] + ((size_t)slot * 2 + cur) * d.fcap[1], (size_t)fc[cur * 4 + 1] * 4, 0);
  else if (s == "flat") set(d.feat[2] + ((size_t)slot * 2 + cur) * d.fcap[2], (size_t)fc[cur * 4 + 2] * 4, 0);
  else if (s == "less_flat") set(d.feat[3] + ((size_t)slot * 2 + cur) * d.fcap[3], (size_t)fc[cur * 4 + 3] * 4, 0);
  else if (s == "sharp_idx") set(d.feat_idx[0] + ((size_t)slot * 2 + cur) * d.fcap[0], fc[cur * 4 + 0], 2);
  else if (s == "less_sharp_idx") set(d.feat_idx[1] + ((size_t)slot * 2 + cur) * d.fcap[1], fc[cur * 4 + 1], 2);
  else if (s == "flat_idx") set(d.feat_idx[2] + ((size_t)slot * 2 + cur) * d.fcap[2], fc[cur * 4 + 2], 2);
  else if (s == "lo_surf_corr") set(d.lo_corr + (size_t)slot * (d.lo_qcap_surf + d.lo_qcap_corner) * 4, (size_t)fc[cur * 4 + 2] * 4, 2);
  else if (s == "lo_corner_corr") set(d.lo_corr + ((size_t)slot * (d.lo_qcap_surf + d.lo_qcap_corner) + d.lo_qcap_surf) * 4, (size_t)fc[cur * 4 + 0] * 4, 2);
  else if (s == "lo_state") set(d.lo_state + (size_t)slot * LO_STATE_N, LO_STATE_N, 1);
  else if (s == "poses") set(d.poses + (size_t)slot * 16, 16, 1);
  else return lm_host_debug_get(h->lm, slot, name, out, cap_bytes, count, dtype, &h->err);
  if ((size_t)cap_bytes < n * esz) { h->err = "debug_get: buffer too small"; return ALEGO_ERR_CAPACITY; }
  if (n) HIP_TRY(h, hipMemcpy(out, src, n * esz, hipMemcpyDeviceToHost));
  *count = (int)n; *dtype = dt;
  return 0;
}

}  // extern "C"
