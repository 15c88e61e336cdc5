// lm_host.hip — LaserMapping: HBM allocation and kernel sequencing (no numerics here).
#include "lm_host.h"
#include "guard_alloc.h"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>

#include <rccl/rccl.h>

#include "lm_ctx.h"
#include "voxel.h"

void launch_lm_prepare(const DevCtx& d, const LmCtx& L, int stage, int run_hint, int par, hipStream_t st);
void launch_lm_stage(const DevCtx& d, const LmCtx& L, int run_hint, int par, hipStream_t st);
void launch_lm_concat(const DevCtx& d, const LmCtx& L, hipStream_t st);
void launch_lm_total(const DevCtx& d, const LmCtx& L, hipStream_t st);
void launch_lm_grid(const DevCtx& d, const LmCtx& L, hipStream_t st);
int launch_lm_register(const DevCtx& d, const LmCtx& L, hipStream_t st, int (*allreduce)(void*, double*, size_t, hipStream_t), void* ar_ctx);
size_t lm_solve_row_bytes_max();
size_t lm_solve_row_bytes_default();
void launch_lm_retransform(const DevCtx& d, const LmCtx& L, int ring, hipStream_t st);
struct MapWork { int* items; int* count; int cap; };   // kernels_map.hip
void launch_map_update(const DevCtx& d, const LmCtx& L, const MapWork& W, hipStream_t st);
void launch_map_accum(const DevCtx& d, const LmCtx& L, const MapWork& W, hipStream_t st);
void launch_lm_apply_correction(const DevCtx& d, const LmCtx& L, int slot, const double* rc_dev, hipStream_t st);

struct LmHost {
  alego_params P;
  int n_slots;
  int gsize;                   // slots per stream group
  std::vector<hipStream_t> st; // one HIP stream per group
  LmCtx L;
  // per group — vm: map corner, map surf of every slot (jobs 2 b, 2 b + 1), then scan corner, scan surf, scan outlier of every
  // slot: the filters of the current scan do not depend on the map, so they share the map round's three launches;
  // v2: scan surf_total (needs the first round's outputs)
  std::vector<VoxCtx> vm, v2;
  std::vector<VoxCtx> vk;      // per group: the two key-frame sort jobs of every slot alone (set_keypose / add_keyframe, outside the regular sequence)
  std::vector<MapWork> work;   // per group: work list of map_accum
  bool fallback_ok = true;     // buffers of the concat + radix VoxelGrid path (ALEGO_MAP_MERGE=0) are allocated
  ncclComm_t comm = nullptr;   // alego_dist_init: the registration of every slot is sharded over the ranks of this communicator
  std::string dist_err;
  std::vector<void*> allocs;
  std::vector<long> frames;  // host mirror of frame_cnt per slot: only used to skip launches
};

namespace {
template <class T>
bool A(LmHost* lm, T** p, size_t count, std::string* err) {
  void* q = nullptr;
  size_t bytes = count * sizeof(T);
  if (bytes == 0) bytes = 16;
  hipError_t e = guard_malloc(&q, bytes);
  if (e == hipSuccess) e = hipMemset(q, 0, bytes);
  if (e != hipSuccess) { *err = std::string("lm hipMalloc: ") + hipGetErrorString(e); return false; }
  lm->allocs.push_back(q);
  *p = (T*)q;
  return true;
}
}  // namespace

LmHost* lm_host_create(const alego_params& P, const DevCtx& d, int n_slots, int gsize, const std::vector<hipStream_t>& st, std::string* err) {
  LmHost* lm = new LmHost();
  lm->P = P; lm->n_slots = n_slots; lm->gsize = gsize; lm->st = st; lm->frames.assign(n_slots, 0);
  VoxCtx vz; std::memset(&vz, 0, sizeof(VoxCtx));
  lm->vm.assign(st.size(), vz); lm->v2.assign(st.size(), vz); lm->vk.assign(st.size(), vz);
  lm->work.assign(st.size(), MapWork{nullptr, nullptr, 0});
  LmCtx& L = lm->L;
  std::memset(&L, 0, sizeof(L));
  L.K = P.recent_keyframe_num > 0 ? P.recent_keyframe_num : 1;
  L.KR = L.K + 1;
  // the concat + radix-sort path needs the raw maps and 20 B of sort scratch per map point (~42 MB per stream at 16x1800 / K = 50): large
  // batches only carry it when they are configured to use it
  lm->fallback_ok = n_slots <= 64 || !d.opt_map_merge;
  L.in_cap_c = d.fcap[F_LSHARP]; L.in_cap_s = d.N; L.in_cap_o = d.N;
  L.kf_cap_c = d.fcap[F_LSHARP];
  L.kf_cap_s = P.kf_cap_surf > 0 ? std::min(P.kf_cap_surf, d.N) : d.N / 2;
  L.kf_cap_o = P.kf_cap_outlier > 0 ? std::min(P.kf_cap_outlier, d.N) : d.N / 4;
  L.kf_cap_s = std::max(L.kf_cap_s, L.kf_cap_c - L.kf_cap_o);   // (scratch shared by both maps is sized by surf + outlier)
  L.total_cap = L.kf_cap_s + L.kf_cap_o;
  L.map_cap_c = L.K * L.kf_cap_c; L.map_cap_s = L.K * L.total_cap;
  L.gcap = n_slots <= 64 ? (1 << 20) : (1 << 18);
  L.qcap = L.kf_cap_c + L.total_cap;
  {
    const size_t mx = lm_solve_row_bytes_max();
    const char* e = getenv("ALEGO_LM_ROW_LDS");   // development: a smaller budget sends rows through crows (tests: 0 = all of them)
    // more streams per launch than CUs: 72 KB, so that two lm_solve workgroups (254 VGPRs x 4 wavefronts each) share a CU and the few rows beyond the
    // budget come from the L2 (measured at 512 per launch: 446 k -> 452 k scans/s for anything between 56 and 76 KB; at 192 / 128 per launch 96 KB is 1 % better)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const size_t dflt = gsize > cus ? (size_t)72 * 1024 : lm_solve_row_bytes_default();
    L.solve_row_bytes = (int)(e ? std::min(mx, (size_t)std::max(0, atoi(e))) : std::min(mx, dflt));
  }
  const size_t B = n_slots;
  bool ok = true;
  ok = ok && A(lm, &L.li, B * LI_COUNT, err) && A(lm, &L.ld, B * LD_COUNT, err);
  ok = ok && A(lm, &L.stage_odom, B * 2 * 8, err);
  ok = ok && A(lm, &L.in_corner, B * L.in_cap_c, err) && A(lm, &L.in_surf, B * L.in_cap_s, err) && A(lm, &L.in_outl, B * L.in_cap_o, err);
  ok = ok && A(lm, &L.kfs_c, B * L.KR * L.kf_cap_c, err) && A(lm, &L.kfs_s, B * L.KR * L.total_cap, err);
  ok = ok && A(lm, &L.kfs_n, B * 2 * L.KR, err) && A(lm, &L.kfs_box, B * 2 * L.KR * 8, err);
  ok = ok && A(lm, &L.kf_tmp_c, B * L.kf_cap_c, err) && A(lm, &L.kf_tmp_s, B * L.total_cap, err);
  ok = ok && A(lm, &L.kf_raw_c, B * L.KR * L.kf_cap_c, err) && A(lm, &L.kf_raw_s, B * L.KR * L.kf_cap_s, err) && A(lm, &L.kf_raw_o, B * L.KR * L.kf_cap_o, err);
  ok = ok && A(lm, &L.rec, B * L.K, err) && A(lm, &L.rec_prev, B * L.K, err);
  ok = ok && A(lm, &L.kf_cnt, B * L.KR * 4, err) && A(lm, &L.kf_pose, B * L.KR * 8, err);
  ok = ok && A(lm, &L.U_c, B * L.map_cap_c, err) && A(lm, &L.U_s, B * L.map_cap_s, err) && A(lm, &L.Ucnt_c, B * L.map_cap_c, err) && A(lm, &L.Ucnt_s, B * L.map_cap_s, err);
  ok = ok && A(lm, &L.newkeys, B * 2 * L.total_cap, err) && A(lm, &L.map_bbox, B * 2 * 8, err);
  {
    double* part = nullptr; int* ctl = nullptr; double* state = nullptr;
    ok = ok && A(lm, &part, B * 32, err) && A(lm, &ctl, B * 8, err) && A(lm, &state, B * 64, err);   // LmState is < 64 doubles (checked in kernels_lm.hip)
    L.shard_part = part; L.shard_ctl = ctl; L.shard_state = state;
  }
  ok = ok && A(lm, &L.map_corner_raw, lm->fallback_ok ? B * L.map_cap_c : 1, err) && A(lm, &L.map_surf_raw, lm->fallback_ok ? B * L.map_cap_s : 1, err);
  ok = ok && A(lm, &L.map_corner_ds, B * L.map_cap_c, err) && A(lm, &L.map_surf_ds, B * L.map_cap_s, err);
  ok = ok && A(lm, &L.cur_corner_ds, B * L.kf_cap_c, err) && A(lm, &L.cur_surf_ds, B * L.kf_cap_s, err) && A(lm, &L.cur_outl_ds, B * L.kf_cap_o, err);
  ok = ok && A(lm, &L.cur_total, B * L.total_cap, err) && A(lm, &L.cur_total_ds, B * L.total_cap, err);
  ok = ok && A(lm, &L.grid, B * 2, err) && A(lm, &L.cell_start, B * 2 * ((size_t)L.gcap + 1), err) && A(lm, &L.cell_cur, B * 2 * ((size_t)L.gcap + 1), err);
  ok = ok && A(lm, &L.cell_pts, B * 2 * L.map_cap_s, err);
  ok = ok && A(lm, &L.blocks, B * L.qcap * 8, err) && A(lm, &L.crows, B * L.qcap * 10, err) && A(lm, &L.knn, B * L.qcap * 5, err);
  if (!ok) { lm_host_destroy(lm); return nullptr; }
  // identity quaternions (laserMapping.cpp:56-61)
  std::vector<double> ld(B * LD_COUNT, 0.0);
  for (size_t b = 0; b < B; ++b) { ld[b * LD_COUNT + LD_Q_M2O] = 1.0; ld[b * LD_COUNT + LD_Q_O2L] = 1.0; ld[b * LD_COUNT + LD_Q_M2L] = 1.0; }
  if (hipMemcpy(L.ld, ld.data(), ld.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { *err = "lm_host_create: upload failed"; lm_host_destroy(lm); return nullptr; }
  std::vector<int> li0(B * LI_COUNT, 0);
  for (size_t b = 0; b < B; ++b) li0[b * LI_COUNT + LI_LATEST] = -1;   // laserMapping.cpp:50
  if (hipMemcpy(L.li, li0.data(), li0.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { *err = "lm_host_create: upload failed"; lm_host_destroy(lm); return nullptr; }
  // VoxelGrid job tables (laserMapping.cpp:37-39,316-319,329-342)
  for (size_t g = 0; g < st.size(); ++g) {
  std::vector<VoxJob> jm, j1, j2, jk;
  for (size_t b = g * gsize; b < B && b < (g + 1) * (size_t)gsize; ++b) {
    int* li = L.li + b * LI_COUNT;
    if (lm->fallback_ok) {   // the maps by concat + radix sort (ALEGO_MAP_MERGE=0)
      jm.push_back(VoxJob{L.map_corner_raw + b * L.map_cap_c, li + LI_KRAW_C, L.map_corner_ds + b * L.map_cap_c, li + LI_KDS_C, li + LI_REBUILD_FB, P.lm_leaf_corner, L.map_cap_c, L.map_cap_c, li + LI_OVERFLOW, 0});
      jm.push_back(VoxJob{L.map_surf_raw + b * L.map_cap_s, li + LI_KRAW_S, L.map_surf_ds + b * L.map_cap_s, li + LI_KDS_S, li + LI_REBUILD_FB, P.lm_leaf_surf, L.map_cap_s, L.map_cap_s, li + LI_OVERFLOW, 0});
    }
    j1.push_back(VoxJob{L.in_corner + b * L.in_cap_c, li + LI_NIN_C, L.cur_corner_ds + b * L.kf_cap_c, li + LI_NCUR_C, li + LI_RUN, P.lm_leaf_corner, L.in_cap_c, L.kf_cap_c, li + LI_OVERFLOW, 0});
    j1.push_back(VoxJob{L.in_surf + b * L.in_cap_s, li + LI_NIN_S, L.cur_surf_ds + b * L.kf_cap_s, li + LI_NCUR_S, li + LI_RUN, P.lm_leaf_surf, L.in_cap_s, L.kf_cap_s, li + LI_OVERFLOW, 0});
    j1.push_back(VoxJob{L.in_outl + b * L.in_cap_o, li + LI_NIN_O, L.cur_outl_ds + b * L.kf_cap_o, li + LI_NCUR_O, li + LI_RUN, P.lm_leaf_outlier, L.in_cap_o, L.kf_cap_o, li + LI_OVERFLOW, 0});
    j2.push_back(VoxJob{L.cur_total + b * L.total_cap, li + LI_NTOTAL, L.cur_total_ds + b * L.total_cap, li + LI_NTOTAL_DS, li + LI_RUN, P.lm_leaf_surf, L.total_cap, L.total_cap, li + LI_OVERFLOW, 0});
    // the key frame waiting in kf_tmp_*: sorted by voxel key of the map's leaf into its ring entry (mode 1)
    VoxJob kc{L.kf_tmp_c + b * L.kf_cap_c, li + LI_TMPN_C, L.kfs_c + b * L.KR * L.kf_cap_c, li + LI_SORT_N, li + LI_KF_PENDING, P.lm_leaf_corner, L.kf_cap_c, L.kf_cap_c, li + LI_OVERFLOW, 0};
    kc.mode = 1; kc.out_sel = li + LI_KF_PEND_RING; kc.out_stride = L.kf_cap_c;
    kc.box_out = L.kfs_box + (b * 2 + 0) * L.KR * 8; kc.n_sel_out = L.kfs_n + (b * 2 + 0) * L.KR; kc.n_sel_stride = 1;
    VoxJob ks{L.kf_tmp_s + b * L.total_cap, li + LI_TMPN_S, L.kfs_s + b * L.KR * L.total_cap, li + LI_SORT_N + 1, li + LI_KF_PENDING, P.lm_leaf_surf, L.total_cap, L.total_cap, li + LI_OVERFLOW, 0};
    ks.mode = 1; ks.out_sel = li + LI_KF_PEND_RING; ks.out_stride = L.total_cap;
    ks.box_out = L.kfs_box + (b * 2 + 1) * L.KR * 8; ks.n_sel_out = L.kfs_n + (b * 2 + 1) * L.KR; ks.n_sel_stride = 1;
    jk.push_back(kc); jk.push_back(ks);
  }
  const int ns = (int)j1.size() / 3;
  jm.insert(jm.end(), j1.begin(), j1.end());
  jm.insert(jm.end(), jk.begin(), jk.end());
  if (vox_create(&lm->vm[g], jm.data(), (int)jm.size(), err) || vox_create(&lm->v2[g], j2.data(), (int)j2.size(), err) ||
      vox_create(&lm->vk[g], jk.data(), (int)jk.size(), err)) { lm_host_destroy(lm); return nullptr; }
  // Expected work per context: the current-scan clouds and key frames practically never exceed 8192 points (vox_small); a key frame
  // is sorted for ~1 stream in 5 per mapping frame.  Fewer persistent workgroups where little is expected (they loop).
  // (vox_big only has work on the radix path of the maps or for a scan cloud of more than 8192 points)
  // (with 64 rings the scan's own surf clouds exceed 8192 points on every mapping frame: as many workgroups for the large jobs as
  //  mapping frames are expected per round, instead of a handful for the rare one — 64x2048: 1.85 -> see profiles/r02_geo_64x2048*)
  const bool big_scans = d.N > 8 * 8192;
  lm->vm[g].grid_small = 3 * ns + std::max(2, ns / 2);
  if (const char* e = getenv("ALEGO_VOX_GRID_DIV")) { const int dv = std::max(1, atoi(e)); lm->vm[g].grid_small = std::max(2, lm->vm[g].grid_small / dv); lm->v2[g].grid_small = std::max(1, lm->v2[g].grid_small / dv); }
  lm->vm[g].grid_big = (!d.opt_map_merge || big_scans) ? std::max(2, ns / 2) : std::min(8, std::max(2, ns / 2));
  lm->v2[g].grid_big = big_scans ? std::max(2, ns / 2) : std::max(1, ns / 16);
  lm->vk[g].grid_small = 2; lm->vk[g].grid_big = 2;
  MapWork& W = lm->work[g];
  W.cap = std::max(4096, 64 * ns);
  if (!A(lm, &W.items, (size_t)W.cap, err) || !A(lm, &W.count, 2, err)) { lm_host_destroy(lm); return nullptr; }
  }
  return lm;
}

void lm_host_destroy(LmHost* lm) {
  if (!lm) return;
  if (lm->comm) { (void)ncclCommDestroy(lm->comm); lm->comm = nullptr; }
  for (auto& v : lm->vm) vox_destroy(&v);
  for (auto& v : lm->v2) vox_destroy(&v);
  for (auto& v : lm->vk) vox_destroy(&v);
  for (void* p : lm->allocs) (void)guard_free(p);
  delete lm;
}

static bool dbg_sync(hipStream_t st, const char* what, std::string* err) {
  static const bool on = getenv("ALEGO_DEBUG_SYNC") != nullptr;
  if (!on) return true;
  hipError_t e = hipStreamSynchronize(st);
  if (e == hipSuccess) e = hipGetLastError();
  fprintf(stderr, "[alego dbg] %s: %s\n", what, hipGetErrorString(e));
  if (e != hipSuccess) { *err = std::string(what) + ": " + hipGetErrorString(e); return false; }
  return true;
}

// A key frame's clouds were (re)written outside the regular sequence (set_keypose / add_keyframe): sort it into the ring now and
// make the next map update rebuild its voxel lists from scratch.
static void lm_kf_changed(LmHost* lm, const DevCtx& d, int /*ring*/, hipStream_t st) {
  const int g = d.slot0 / lm->gsize;
  std::string e;
  (void)vox_run(lm->vk[g], st, &e);
  const int zero = 0;
  int* li = lm->L.li + (size_t)d.slot0 * LI_COUNT;
  (void)hipMemcpyAsync(li + LI_KF_PENDING, &zero, sizeof(int), hipMemcpyHostToDevice, st);
  (void)hipMemcpyAsync(li + LI_UVALID, &zero, sizeof(int), hipMemcpyHostToDevice, st);
}

// the local maps of the slots whose window changed (LI_REBUILD, set by lm_prepare) + the VoxelGrid filters of the scan's three
// clouds + the sort of a pending key frame, then the k-NN grids
static int map_sequence(LmHost* lm, const DevCtx& d, const LmCtx& L, int g, hipStream_t st, std::string* err) {
  if (!d.opt_map_merge) {
    if (!lm->fallback_ok) { *err = "ALEGO_MAP_MERGE=0 needs the handle to be created with it (more than 64 slots)"; return ALEGO_ERR_ARG; }
    launch_lm_concat(d, L, st);
    if (!dbg_sync(st, "lm_concat", err)) return ALEGO_ERR_HIP;
  }
  if (int r = vox_run(lm->vm[g], st, err)) return r;
  if (!dbg_sync(st, "vox round 1", err)) return ALEGO_ERR_HIP;
  launch_map_update(d, L, lm->work[g], st);   // (also closes the key-frame sort's bookkeeping when the maps come from the radix path)
  if (!dbg_sync(st, "map_update", err)) return ALEGO_ERR_HIP;
  if (d.opt_map_merge) {
    launch_map_accum(d, L, lm->work[g], st);
    if (!dbg_sync(st, "map_accum", err)) return ALEGO_ERR_HIP;
  }
  launch_lm_grid(d, L, st);
  if (!dbg_sync(st, "lm_grid", err)) return ALEGO_ERR_HIP;
  return 0;
}

// sum of the partial normal equations over the ranks, in place, on the registration's stream (RCCL over xGMI)
static int lm_allreduce(void* ctx, double* buf, size_t count, hipStream_t st) {
  LmHost* lm = static_cast<LmHost*>(ctx);
  const ncclResult_t r = ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, lm->comm, st);
  if (r != ncclSuccess) { lm->dist_err = std::string("ncclAllReduce: ") + ncclGetErrorString(r); return ALEGO_ERR_HIP; }
  return 0;
}

// odom_valid[s - slot0]: whether slot s has an /odom/lidar message for this scan (false on its first scan)
// A (optional): LaserMapping of this scan runs on A->back, behind the hand-over kernel lm_stage on the front end's stream.
struct LmAsync { hipStream_t front, back; hipEvent_t staged, back_same, back_other; long k; };   // k: scans handed over before this one (its parity picks the staging buffer)
static int lm_sequence(LmHost* lm, const DevCtx& d, int stage, const std::vector<char>& odom_valid, std::string* err, hipStream_t st_override = nullptr, const LmAsync* A = nullptr) {
  // the slots of one launch view always belong to one stream group
  const int g = d.slot0 / lm->gsize;
  hipStream_t st = A ? A->back : (st_override ? st_override : lm->st[g]);
  LmCtx L = lm->L;
  L.vox_bbox = lm->vm[g].bbox; L.vox_slot0 = g * lm->gsize;
  int n_run = 0, n_norun = 0;
  for (int i = 0; i < d.n_launch; ++i) {
    long& f = lm->frames[d.slot0 + i];
    const bool run = odom_valid[i] && (f % lm->P.lm_every) == 0;
    if (odom_valid[i]) ++f;
    run ? ++n_run : ++n_norun;
  }
  const int hint = n_run == 0 ? 0 : (n_norun == 0 ? 1 : -1);  // -1: slots out of phase, no launch skipping
  int par = 0;
  if (A) {
    par = (int)(A->k & 1);
    // stage_odom[par] was last read by the LaserMapping of scan k - 2; the cloud inputs by the last mapping frame (k - 1 at the latest)
    if (A->k >= 2 && hipStreamWaitEvent(A->front, A->back_same, 0) != hipSuccess) { *err = "hipStreamWaitEvent failed"; return ALEGO_ERR_HIP; }
    if (hint != 0 && A->k >= 1 && hipStreamWaitEvent(A->front, A->back_other, 0) != hipSuccess) { *err = "hipStreamWaitEvent failed"; return ALEGO_ERR_HIP; }
    launch_lm_stage(d, L, hint, par, A->front);
    if (hipEventRecord(A->staged, A->front) != hipSuccess || hipStreamWaitEvent(st, A->staged, 0) != hipSuccess) { *err = "event hand-over to the LaserMapping stream failed"; return ALEGO_ERR_HIP; }
    stage = 2;
  }
  launch_lm_prepare(d, L, stage, hint, par, st);
  if (!dbg_sync(st, "lm_prepare", err)) return ALEGO_ERR_HIP;
  if (n_run == 0) return 0;
  if (int r = map_sequence(lm, d, L, g, st, err)) return r;   // (includes the VoxelGrid filters of the scan's three clouds)
  launch_lm_total(d, L, st);
  if (int r = vox_run(lm->v2[g], st, err)) return r;
  if (!dbg_sync(st, "vox total", err)) return ALEGO_ERR_HIP;
  if (int r = launch_lm_register(d, L, st, lm->comm ? &lm_allreduce : nullptr, lm)) { *err = "sharded registration: " + lm->dist_err; return r; }
  if (!dbg_sync(st, "lm_register", err)) return ALEGO_ERR_HIP;
  return 0;
}

// NOTE: the VoxelGrid rounds always cover every slot of the stream group; slots outside the launch view
// have LI_RUN == 0 from their own last prepare only if they were prepared in this call, so the
// single-slot entry points clear the run flags of the other slots of the group first.
static void clear_run_flags_outside(LmHost* lm, const DevCtx& d) {
  const int g = d.slot0 / lm->gsize;
  const int lo = g * lm->gsize, hi = std::min(lm->n_slots, lo + lm->gsize);
  if (d.slot0 == lo && d.slot0 + d.n_launch == hi) return;
  for (int s = lo; s < hi; ++s) {
    if (s >= d.slot0 && s < d.slot0 + d.n_launch) continue;
    (void)hipMemsetAsync(lm->L.li + (size_t)s * LI_COUNT + LI_RUN, 0, sizeof(int), lm->st[g]);  // LI_REBUILD is always 0 between map sequences
  }
}

int lm_host_enqueue(LmHost* lm, const DevCtx& d, const std::vector<char>& odom_valid, std::string* err, hipStream_t st_override) {
  if (!st_override) clear_run_flags_outside(lm, d);   // (alego_stream_run: the other slots of the group are look-ahead lanes, LaserMapping never ran on them)
  return lm_sequence(lm, d, 1, odom_valid, err, st_override);
}
// The batch path: the view covers a whole stream group; LaserMapping of the scan goes to `back`, lm_stage to `front`.
int lm_host_enqueue_async(LmHost* lm, const DevCtx& d, const std::vector<char>& odom_valid, std::string* err, hipStream_t front, hipStream_t back,
                          hipEvent_t staged, hipEvent_t back_same, hipEvent_t back_other, long k) {
  const LmAsync A{front, back, staged, back_same, back_other, k};
  return lm_sequence(lm, d, 1, odom_valid, err, nullptr, &A);
}

int lm_host_process_host(LmHost* lm, const DevCtx& dfull, const alego_point* corner_last, int n_corner, const alego_point* surf_last,
                         int n_surf, const alego_point* outlier, int n_outlier, const alego_pose* odom, alego_pose* map_pose,
                         std::string* err) {
  const LmCtx& L = lm->L;
  if (n_corner > L.in_cap_c || n_surf > L.in_cap_s || n_outlier > L.in_cap_o) { *err = "alego_lm_process: input cloud exceeds capacity"; return ALEGO_ERR_CAPACITY; }
  DevCtx d = dfull;
  d.slot0 = 0; d.n_launch = 1;
  hipStream_t st = lm->st[0];
  if ((n_corner > 0 && !corner_last) || (n_surf > 0 && !surf_last) || (n_outlier > 0 && !outlier) || n_corner < 0 || n_surf < 0 || n_outlier < 0) { *err = "alego_lm_process: null cloud / negative count"; return ALEGO_ERR_ARG; }
  hipError_t e = hipSuccess;
  auto up = [&](void* dst, const void* src, size_t bytes) { if (e == hipSuccess && bytes) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st); };
  up(L.in_corner, corner_last, (size_t)n_corner * 16);
  up(L.in_surf, surf_last, (size_t)n_surf * 16);
  up(L.in_outl, outlier, (size_t)n_outlier * 16);
  const int nin[3] = {n_corner, n_surf, n_outlier};
  up(L.li + LI_NIN_C, nin, sizeof(nin));
  double po[7] = {odom->t[0], odom->t[1], odom->t[2], odom->q[0], odom->q[1], odom->q[2], odom->q[3]};
  up(d.poses, po, sizeof(po));
  const int one = 1;
  up(d.scal + SC_ODOM_VALID, &one, sizeof(int));
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { *err = std::string("alego_lm_process: upload failed: ") + hipGetErrorString(e); return ALEGO_ERR_HIP; }
  clear_run_flags_outside(lm, d);
  if (int r = lm_sequence(lm, d, 0, std::vector<char>(1, 1), err)) return r;
  double out[16], ld[LD_COUNT];
  int li[LI_COUNT];
  e = hipMemcpyAsync(out, d.poses, sizeof(out), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(ld, L.ld, sizeof(ld), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(li, L.li, sizeof(li), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { *err = std::string("alego_lm_process: kernels failed: ") + hipGetErrorString(e); return ALEGO_ERR_HIP; }
  if (li[LI_OVERFLOW]) {   // reported by the call that caused it, then cleared
    const int zero = 0;
    (void)hipMemcpy(L.li + LI_OVERFLOW, &zero, sizeof(int), hipMemcpyHostToDevice);
    *err = li[LI_OVERFLOW] == 2 ? "alego_lm_process: launch logic out of sync" : "alego_lm_process: device capacity exceeded (cloud truncated)";
    return ALEGO_ERR_CAPACITY;
  }
  if (map_pose) {
    for (int i = 0; i < 3; ++i) map_pose->t[i] = out[7 + i];
    for (int i = 0; i < 4; ++i) map_pose->q[i] = out[10 + i];
    for (int i = 0; i < 6; ++i) map_pose->params[i] = ld[LD_PARAMS + i];
    map_pose->valid = 1;
  }
  return li[LI_FLAGS];
}

const double* lm_host_stage_odom(LmHost* lm) { return lm->L.stage_odom; }
void lm_host_get_params(LmHost* lm, int slot, double* p6) {
  (void)hipMemcpy(p6, lm->L.ld + (size_t)slot * LD_COUNT + LD_PARAMS, 48, hipMemcpyDeviceToHost);
}
int lm_host_set_params(LmHost* lm, int slot, const double* p6, std::string* err) {
  if (hipMemcpy(lm->L.ld + (size_t)slot * LD_COUNT + LD_PARAMS, p6, 48, hipMemcpyHostToDevice) != hipSuccess) { *err = "set_lm_params failed"; return ALEGO_ERR_HIP; }
  return 0;
}
// LI_OVERFLOW stays set until it has been reported to the host once (the batch path only looks at the end of a run)
int lm_host_get_flags(LmHost* lm, int slot) {
  int li[LI_COUNT];
  if (hipMemcpy(li, lm->L.li + (size_t)slot * LI_COUNT, sizeof(li), hipMemcpyDeviceToHost) != hipSuccess) return ALEGO_ERR_HIP;
  if (li[LI_OVERFLOW]) {
    const int zero = 0;
    (void)hipMemcpy(lm->L.li + (size_t)slot * LI_COUNT + LI_OVERFLOW, &zero, sizeof(int), hipMemcpyHostToDevice);
    return ALEGO_ERR_CAPACITY;
  }
  return li[LI_FLAGS];
}
void lm_host_get_counts(LmHost* lm, int slot, int* o) {  // o[7]
  int li[LI_COUNT];
  (void)hipMemcpy(li, lm->L.li + (size_t)slot * LI_COUNT, sizeof(li), hipMemcpyDeviceToHost);
  o[0] = li[LI_KRAW_C]; o[1] = li[LI_KRAW_S]; o[2] = li[LI_KDS_C]; o[3] = li[LI_KDS_S]; o[4] = li[LI_NCUR_C]; o[5] = li[LI_NTOTAL_DS]; o[6] = li[LI_NREBUILD];
}

// ---- key-frame pass-through (alego_lm_get_keyframe / set_keypose / reset_window / apply_correction / add_keyframe) ----
static hipStream_t stream_of_slot(LmHost* lm, int slot) { return lm->st[slot / lm->gsize]; }
int lm_host_keyframe_count(LmHost* lm, int slot) {
  int n = 0;
  if (hipStreamSynchronize(stream_of_slot(lm, slot)) != hipSuccess) return ALEGO_ERR_HIP;
  if (hipMemcpy(&n, lm->L.li + (size_t)slot * LI_COUNT + LI_NKF, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return ALEGO_ERR_HIP;
  return n;
}
int lm_host_get_keyframe(LmHost* lm, int slot, int kf_id, alego_keyframe* out, std::string* err) {
  const LmCtx& L = lm->L;
  const int nkf = lm_host_keyframe_count(lm, slot);
  if (nkf < 0) { *err = "get_keyframe: device error"; return nkf; }
  if (kf_id < 0) kf_id = nkf - 1;
  if (kf_id < 0 || kf_id >= nkf || kf_id < nkf - L.K) { *err = "get_keyframe: key frame not resident (only the recent_keyframe_num newest are)"; return ALEGO_ERR_ARG; }
  const size_t rs = (size_t)slot * L.KR + kf_id % L.KR;
  int cnt[4];
  float kp[8];
  if (hipMemcpy(cnt, L.kf_cnt + rs * 4, sizeof(cnt), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(kp, L.kf_pose + rs * 8, sizeof(kp), hipMemcpyDeviceToHost) != hipSuccess) { *err = "get_keyframe: copy failed"; return ALEGO_ERR_HIP; }
  out->id = kf_id;
  for (int k = 0; k < 6; ++k) out->pose[k] = kp[k];
  out->n_corner = cnt[0]; out->n_surf = cnt[1]; out->n_outlier = cnt[2];
  if ((out->corner && cnt[0] > out->corner_cap) || (out->surf && cnt[1] > out->surf_cap) || (out->outlier && cnt[2] > out->outlier_cap)) { *err = "get_keyframe: buffer too small"; return ALEGO_ERR_CAPACITY; }
  hipError_t e = hipSuccess;
  if (out->corner && cnt[0]) e = hipMemcpy(out->corner, L.kf_raw_c + rs * L.kf_cap_c, (size_t)cnt[0] * 16, hipMemcpyDeviceToHost);
  if (e == hipSuccess && out->surf && cnt[1]) e = hipMemcpy(out->surf, L.kf_raw_s + rs * L.kf_cap_s, (size_t)cnt[1] * 16, hipMemcpyDeviceToHost);
  if (e == hipSuccess && out->outlier && cnt[2]) e = hipMemcpy(out->outlier, L.kf_raw_o + rs * L.kf_cap_o, (size_t)cnt[2] * 16, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { *err = std::string("get_keyframe: ") + hipGetErrorString(e); return ALEGO_ERR_HIP; }
  return 0;
}
static int retransform(LmHost* lm, const DevCtx& dfull, int slot, int ring, std::string* err) {
  DevCtx d = dfull;
  d.slot0 = slot; d.n_launch = 1;
  hipStream_t st = stream_of_slot(lm, slot);
  const int g = slot / lm->gsize;
  const int zero = 0;
  int* li = lm->L.li + (size_t)slot * LI_COUNT;
  // a key frame saved by the last mapping frame may still wait in kf_tmp_* for its sort: flush it before the buffer is reused
  if (int r = vox_run(lm->vk[g], st, err)) return r;
  (void)hipMemcpyAsync(li + LI_KF_PENDING, &zero, sizeof(int), hipMemcpyHostToDevice, st);
  launch_lm_retransform(d, lm->L, ring, st);
  lm_kf_changed(lm, d, ring, st);
  if (hipStreamSynchronize(st) != hipSuccess) { *err = "key-frame transform failed"; return ALEGO_ERR_HIP; }
  return 0;
}
int lm_host_set_keypose(LmHost* lm, const DevCtx& dfull, int slot, int kf_id, const float* pose6, std::string* err) {
  const LmCtx& L = lm->L;
  const int nkf = lm_host_keyframe_count(lm, slot);
  if (nkf < 0) return nkf;
  if (kf_id < 0 || kf_id >= nkf || kf_id < nkf - L.K) { *err = "set_keypose: key frame not resident"; return ALEGO_ERR_ARG; }
  const int ring = kf_id % L.KR;
  if (hipMemcpy(L.kf_pose + ((size_t)slot * L.KR + ring) * 8, pose6, 6 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { *err = "set_keypose: copy failed"; return ALEGO_ERR_HIP; }
  return retransform(lm, dfull, slot, ring, err);
}
int lm_host_reset_window(LmHost* lm, int slot, std::string* err) {
  // recent_*_keyframes_.clear() (:563-565): the next mapping frame refills the window from the newest key frames (:208-223)
  const int v[2] = {0, 1};
  int* li = lm->L.li + (size_t)slot * LI_COUNT;
  if (hipStreamSynchronize(stream_of_slot(lm, slot)) != hipSuccess || hipMemcpy(li + LI_REC_CNT, &v[0], sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(li + LI_DIRTY, &v[1], sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(li + LI_UVALID, &v[0], sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { *err = "reset_window failed"; return ALEGO_ERR_HIP; }
  return 0;
}
int lm_host_apply_correction(LmHost* lm, const DevCtx& dfull, int slot, const double* rc12, std::string* err) {
  double* dev = nullptr;
  hipStream_t st = stream_of_slot(lm, slot);
  if (hipMalloc((void**)&dev, 12 * sizeof(double)) != hipSuccess) { *err = "apply_correction: hipMalloc"; return ALEGO_ERR_HIP; }
  hipError_t e = hipMemcpy(dev, rc12, 12 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) { launch_lm_apply_correction(dfull, lm->L, slot, dev, st); e = hipStreamSynchronize(st); }
  (void)hipFree(dev);
  if (e != hipSuccess) { *err = std::string("apply_correction: ") + hipGetErrorString(e); return ALEGO_ERR_HIP; }
  return 0;
}
int lm_host_add_keyframe(LmHost* lm, const DevCtx& dfull, int slot, const float* pose6, const alego_point* corner, int nc, const alego_point* surf, int ns,
                         const alego_point* outlier, int no, std::string* err) {
  const LmCtx& L = lm->L;
  if (nc < 0 || ns < 0 || no < 0 || (nc && !corner) || (ns && !surf) || (no && !outlier)) { *err = "add_keyframe: null cloud / negative count"; return ALEGO_ERR_ARG; }
  if (nc > L.kf_cap_c || ns > L.kf_cap_s || no > L.kf_cap_o) { *err = "add_keyframe: cloud exceeds the key-frame capacity"; return ALEGO_ERR_CAPACITY; }
  const int nkf = lm_host_keyframe_count(lm, slot);
  if (nkf < 0) return nkf;
  const int ring = nkf % L.KR;
  const size_t rs = (size_t)slot * L.KR + ring;
  float kp[8] = {pose6[0], pose6[1], pose6[2], pose6[3], pose6[4], pose6[5], 0.f, 0.f};
  const int cnt[4] = {nc, ns, no, 0};
  int* li = L.li + (size_t)slot * LI_COUNT;
  {
    // A full window advances by ONE frame per mapping frame (pop the oldest, push the newest, laserMapping.cpp:224-237): it keeps its
    // K - 1 other frames however many frames were added in between, and falls behind the newest frames by one for every extra frame.
    // The device holds the K + 1 newest frames, so the window may lag by one: the frame inserted here (id nkf) is refused when the
    // oldest frame the next window still needs, rec[1], would no longer be among the K + 1 newest, i.e. rec[1] < nkf - K.
    int rec_cnt = 0;
    if (hipMemcpy(&rec_cnt, li + LI_REC_CNT, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { *err = "add_keyframe: device read failed"; return ALEGO_ERR_HIP; }
    if (rec_cnt >= L.K && L.K >= 2) {
      int rec1 = 0;
      if (hipMemcpy(&rec1, L.rec + (size_t)slot * L.K + 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { *err = "add_keyframe: device read failed"; return ALEGO_ERR_HIP; }
      if (rec1 < nkf - L.K) {
        *err = "add_keyframe: the full local-map window already lags one frame behind the newest key frames (it advances by one frame per mapping frame); "
               "insert at most one extra key frame per window length, or call alego_lm_reset_window first (the window is then rebuilt from the newest frames)";
        return ALEGO_ERR_CAPACITY;
      }
    }
  }
  const int nkf1 = nkf + 1, one = 1;
  hipError_t e = hipMemcpy(L.kf_pose + rs * 8, kp, sizeof(kp), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(L.kf_cnt + rs * 4, cnt, sizeof(cnt), hipMemcpyHostToDevice);
  if (e == hipSuccess && nc) e = hipMemcpy(L.kf_raw_c + rs * L.kf_cap_c, corner, (size_t)nc * 16, hipMemcpyHostToDevice);
  if (e == hipSuccess && ns) e = hipMemcpy(L.kf_raw_s + rs * L.kf_cap_s, surf, (size_t)ns * 16, hipMemcpyHostToDevice);
  if (e == hipSuccess && no) e = hipMemcpy(L.kf_raw_o + rs * L.kf_cap_o, outlier, (size_t)no * 16, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(li + LI_NKF, &nkf1, sizeof(int), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(li + LI_DIRTY, &one, sizeof(int), hipMemcpyHostToDevice);
  if (e != hipSuccess) { *err = std::string("add_keyframe: ") + hipGetErrorString(e); return ALEGO_ERR_HIP; }
  return retransform(lm, dfull, slot, ring, err);
}

// ---- one registration sharded over the ranks of a communicator (alego_dist_*) ----
int lm_host_dist_unique_id(char* id128) {
  static_assert(sizeof(ncclUniqueId) <= 128, "ALEGO_DIST_ID_BYTES");
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return ALEGO_ERR_HIP;
  std::memset(id128, 0, 128);
  std::memcpy(id128, &u, sizeof(u));
  return 0;
}
int lm_host_dist_init(LmHost* lm, int rank, int world, const char* id128, std::string* err) {
  if (lm->comm) { *err = "alego_dist_init: already initialised"; return ALEGO_ERR_ARG; }
  if (world < 1 || rank < 0 || rank >= world) { *err = "alego_dist_init: rank / world out of range"; return ALEGO_ERR_ARG; }
  if (lm->st.size() != 1) { *err = "alego_dist_init: a sharded registration needs a handle with one stream group (collectives of one communicator must not overlap)"; return ALEGO_ERR_ARG; }
  ncclUniqueId u;
  std::memcpy(&u, id128, sizeof(u));
  const ncclResult_t r = ncclCommInitRank(&lm->comm, world, u, rank);
  if (r != ncclSuccess) { lm->comm = nullptr; *err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); return ALEGO_ERR_HIP; }
  lm->L.shard_rank = rank; lm->L.shard_world = world;
  return 0;
}
// the collective of one solver evaluation on its own: `iters` in-place ncclAllReduce(sum, 32 doubles) back to back on the registration's
// stream between two events (every rank has to call it: it IS a collective); microseconds per all-reduce, enqueue + completion included
int lm_host_dist_probe(LmHost* lm, int iters, double* usec, std::string* err) {
  if (!lm->comm) { *err = "alego_dist_allreduce_probe: no communicator (alego_dist_init first)"; return ALEGO_ERR_ARG; }
  if (iters < 1 || !usec) { *err = "alego_dist_allreduce_probe: iters / output"; return ALEGO_ERR_ARG; }
  hipStream_t st = lm->st[0];
  double* buf = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipMalloc(&buf, 32 * sizeof(double)) != hipSuccess || hipMemsetAsync(buf, 0, 32 * sizeof(double), st) != hipSuccess ||
      hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { *err = "alego_dist_allreduce_probe: allocation failed"; if (buf) (void)hipFree(buf); return ALEGO_ERR_HIP; }
  int rc = 0;
  for (int i = 0; i < 8 && !rc; ++i) rc = lm_allreduce(lm, buf, 32, st);   // warm the channels
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters && !rc; ++i) rc = lm_allreduce(lm, buf, 32, st);
  (void)hipEventRecord(e1, st);
  const hipError_t se = hipStreamSynchronize(st);
  float ms = 0.f;
  if (!rc && se == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(buf);
  if (rc) { *err = lm->dist_err; return rc; }
  if (se != hipSuccess) { *err = std::string("alego_dist_allreduce_probe: ") + hipGetErrorString(se); return ALEGO_ERR_HIP; }
  *usec = 1e3 * (double)ms / iters;
  return 0;
}
int lm_host_dist_shutdown(LmHost* lm) {
  if (!lm->comm) return 0;
  for (hipStream_t s : lm->st) (void)hipStreamSynchronize(s);
  (void)ncclCommDestroy(lm->comm);
  lm->comm = nullptr; lm->L.shard_rank = 0; lm->L.shard_world = 0;
  return 0;
}

// tests: only the query slice of rank `rank` of `world` is associated (no communicator: the fused solver then works on that slice alone)
int lm_host_debug_slice(LmHost* lm, int rank, int world, std::string* err) {
  if (lm->comm) { *err = "ALEGO_SHARD_SLICE: a communicator is active"; return ALEGO_ERR_ARG; }
  if (world < 0 || (world > 0 && (rank < 0 || rank >= world))) { *err = "ALEGO_SHARD_SLICE: rank / world out of range"; return ALEGO_ERR_ARG; }
  lm->L.shard_rank = rank; lm->L.shard_world = world;
  return 0;
}

// ALEGO_MAP_MERGE switched at run time (tests): the voxel lists no longer describe what the other path did in between
int lm_host_set_map_merge(LmHost* lm, int on, std::string* err) {
  if (!on && !lm->fallback_ok) { *err = "ALEGO_MAP_MERGE=0 needs the handle to be created with it (more than 64 slots)"; return ALEGO_ERR_ARG; }
  for (hipStream_t s : lm->st) (void)hipStreamSynchronize(s);
  const int zero = 0;
  for (int b = 0; b < lm->n_slots; ++b)
    if (hipMemcpy(lm->L.li + (size_t)b * LI_COUNT + LI_UVALID, &zero, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { *err = "set_map_merge: copy failed"; return ALEGO_ERR_HIP; }
  return 0;
}

int lm_host_debug_get(LmHost* lm, int slot, const char* name, void* out, int cap_bytes, int* count, int* dtype, std::string* err) {
  const LmCtx& L = lm->L;
  int li[LI_COUNT];
  (void)hipMemcpy(li, L.li + (size_t)slot * LI_COUNT, sizeof(li), hipMemcpyDeviceToHost);
  const std::string s(name);
  const void* src = nullptr;
  size_t n = 0;
  int dt = 0, esz = 4;
  auto set = [&](const void* p, size_t cnt, int t) { src = p; n = cnt; dt = t; esz = t == 1 ? 8 : t == 3 ? 1 : 4; };
  const size_t b = slot;
  if (s == "lm_info") set(L.li + b * LI_COUNT, LI_COUNT, 2);
  else if (s == "lm_state") set(L.ld + b * LD_COUNT, LD_COUNT, 1);
  else if (s == "lm_corner_map") set(L.map_corner_raw + b * L.map_cap_c, (size_t)li[LI_KRAW_C] * 4, 0);
  else if (s == "lm_surf_map") set(L.map_surf_raw + b * L.map_cap_s, (size_t)li[LI_KRAW_S] * 4, 0);
  else if (s == "lm_corner_map_ds") set(L.map_corner_ds + b * L.map_cap_c, (size_t)li[LI_KDS_C] * 4, 0);
  else if (s == "lm_surf_map_ds") set(L.map_surf_ds + b * L.map_cap_s, (size_t)li[LI_KDS_S] * 4, 0);
  else if (s == "lm_corner_ds") set(L.cur_corner_ds + b * L.kf_cap_c, (size_t)li[LI_NCUR_C] * 4, 0);
  else if (s == "lm_surf_ds") set(L.cur_surf_ds + b * L.kf_cap_s, (size_t)li[LI_NCUR_S] * 4, 0);
  else if (s == "lm_outlier_ds") set(L.cur_outl_ds + b * L.kf_cap_o, (size_t)li[LI_NCUR_O] * 4, 0);
  else if (s == "lm_surf_total_ds") set(L.cur_total_ds + b * L.total_cap, (size_t)li[LI_NTOTAL_DS] * 4, 0);
  else if (s == "lm_blocks") set(L.blocks + b * L.qcap * 8, (size_t)L.qcap * 8, 1);
  else if (s == "lm_keyposes") set(L.kf_pose + b * L.KR * 8, (size_t)L.KR * 8, 0);
  else if (s == "lm_window") set(L.rec + b * L.K, (size_t)li[LI_REC_CNT], 2);   // frame ids of recent_*_keyframes_
  else if (s == "lm_kf_corner_map" || s == "lm_kf_surf_map") {   // newest key frame in the map frame, sorted by voxel key (surf = surf + outlier)
    const int nkf = li[LI_NKF];
    if (nkf <= 0) { *count = 0; *dtype = 0; return 0; }
    const int e = (nkf - 1) % L.KR, m = s == "lm_kf_corner_map" ? 0 : 1;
    int n = 0;
    (void)hipMemcpy(&n, L.kfs_n + (b * 2 + m) * L.KR + e, sizeof(int), hipMemcpyDeviceToHost);
    if (m == 0) set(L.kfs_c + (b * L.KR + e) * L.kf_cap_c, (size_t)n * 4, 0);
    else set(L.kfs_s + (b * L.KR + e) * L.total_cap, (size_t)n * 4, 0);
  }
  else if (s == "lm_voxel_keys_c" || s == "lm_voxel_keys_s") {   // the sorted voxel-key list of a map as pairs of i32 (lo, hi)
    const int m = s == "lm_voxel_keys_c" ? 0 : 1;
    set(m == 0 ? (const void*)(L.U_c + b * L.map_cap_c) : (const void*)(L.U_s + b * L.map_cap_s), (size_t)li[LI_NU_C + m] * 2, 2);
  }
  else { *err = std::string("debug_get: unknown name ") + name; return ALEGO_ERR_ARG; }
  if ((size_t)cap_bytes < n * esz) { *err = "debug_get: buffer too small"; return ALEGO_ERR_CAPACITY; }
  if (n && hipMemcpy(out, src, n * esz, hipMemcpyDeviceToHost) != hipSuccess) { *err = "debug_get: copy failed"; return ALEGO_ERR_HIP; }
  *count = (int)n; *dtype = dt;
  return 0;
}
