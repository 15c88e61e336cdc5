// kernels_lm.hip — LaserMapping scan-to-map registration on gfx950
// (replaces src/laserMapping.cpp:154-166,188-192,194-244,315-559 and laserMapping.h:164-194).
//
//   lm_prepare     laserOdomHandler + mainLoop gate + transformAssociateToMap; stages the inputs
//   lm_concat      a19: concatenation of the <= K most recent key frames (already in the map frame)
//   (voxel.h)      a19/a20: VoxelGrid of the map (only when the key-frame set changed) and of the scan
//   lm_total       laser_surf_total_ = laser_surf_ds_ + laser_outlier_ds_ (:337-340)
//   lm_grid_*      uniform grid (cell >= 1 m) over the down-sampled maps: exact 5-NN for every
//                  accepted query because acceptance needs d5^2 < 1.0 (:376,:426) — replaces KdTreeFLANN
//   lm_assoc       a21/a22: 5-NN, 3x3 scatter eigen-decomposition (line) / 5x3 Householder LS (plane)
//   lm_solve       a23/a24: both ceres::Solve calls of :360-478 in one workgroup per stream
//   lm_finish / lm_store_kf  saveKeyFramesAndFactor (no-loop-closure pass-through) + transformUpdate
#include "dev_cost.h"
#include "prof.h"
#include "lm_ctx.h"

#define LM_BLOCK 256

DEV_INLINE double* ldp(const LmCtx& L, int slot) { return L.ld + (size_t)slot * LD_COUNT; }
DEV_INLINE int* lip(const LmCtx& L, int slot) { return L.li + (size_t)slot * LI_COUNT; }
DEV_INLINE DQuat ldq(const double* p) { return DQuat{p[0], p[1], p[2], p[3]}; }
DEV_INLINE void stq(double* p, const DQuat& q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }

// Map sequence = lm_concat, VoxelGrid of the two maps, lm_grid_*, at the start of every mapping frame.  lm_map_update
// (called by lm_prepare's bookkeeping thread) replays the deque bookkeeping of extractSurroundingKeyFrames
// (laserMapping.cpp:206-238) on the list of frame ids and decides whether the local map changed (the reference
// re-assembles and re-filters the identical map on every mapping frame; here only when its content changes):
//   deque not full yet  -> the newest min(n, K) key frames; content changes when a key frame was saved
//   deque full          -> if latest_frame_id_ != n-1: pop front, push frame n-1.  latest_frame_id_ starts at -1, so the
//                          first mapping frame after the deque filled up pushes frame K-1 a second time (and drops
//                          frame 0) although no key frame was saved: the duplicate stays in the window for K-1 further
//                          key frames and doubles that frame's weight in the voxel centroids.  Reproduced as is.
// LI_REBUILD is cleared again by lm_finish at the end of the frame.
DEV_INLINE void lm_map_update(const LmCtx& L, int slot, int* li, int merge) {
  const int nkf = li[LI_NKF];
  if (nkf == 0) return;   // :196-199
  int* rec = L.rec + (size_t)slot * L.K;
  const int cnt = li[LI_REC_CNT];
  bool changed = false;
  if (cnt < L.K) {
    const int nk = min(nkf, L.K);
    changed = li[LI_DIRTY] != 0;
    for (int j = 0; j < nk; ++j) rec[j] = nkf - nk + j;
    li[LI_REC_CNT] = nk;
  } else if (li[LI_LATEST] != nkf - 1) {
    for (int j = 0; j + 1 < L.K; ++j) rec[j] = rec[j + 1];
    rec[L.K - 1] = nkf - 1;
    li[LI_LATEST] = nkf - 1;
    changed = true;
  }
  li[LI_DIRTY] = 0;
  if (changed) { li[LI_REBUILD] = 1; li[LI_NREBUILD] += 1; }
}


// grid (8, 3, slots).  stage: copy /corner_last, /surf_last, /outlier of this scan into the LM inputs.
// stage 0: inputs uploaded by the host (alego_lm_process); 1: copy this scan's clouds and read its odometry from the front end's
// buffers; 2: lm_stage (below) has copied both — LaserMapping runs on its own HIP stream while the front end works on later scans.
__global__ void __launch_bounds__(LM_BLOCK) lm_prepare(DevCtx d, LmCtx L, int stage, int run_hint, int par) {
  const int slot = blockIdx.z + d.slot0, kind = blockIdx.y;
  const int cur = d.scal[slot * SC_COUNT + SC_CUR];  // LO has completed: features of this scan
  const int sslot = scan_slot_of(d, slot);           // where this scan's features, outliers and odometry hand-over live (its lane, or the slot itself)
  int* li = lip(L, slot);
  if (stage == 1 && run_hint != 0) {   // run_hint == 0: the host knows that no slot of this launch maps this scan (odd frame)
    // (written with unconditional loads + selects: a 3-way if/else chain here was lowered by hipcc 7.2 into
    //  a scalar switch that left the count pointer of the last arm undefined)
    const size_t fb = d.fs_cur >= 0 ? (size_t)d.fs_cur * 2 : (size_t)slot * 2 + cur;
    const int n_c = d.feat_cnt[fb * 4 + F_LSHARP], n_s = d.scal[sslot * SC_COUNT + SC_FE_ERR] ? 0 : d.feat_cnt[fb * 4 + F_LFLAT], n_o = d.scal[sslot * SC_COUNT + SC_NOUT];   // (SC_FE_ERR: dev_common.h)
    const float4* src_c = d.feat[F_LSHARP] + fb * d.fcap[F_LSHARP];
    const float4* src_s = d.feat[F_LFLAT] + fb * d.fcap[F_LFLAT];
    const float4* src_o = d.outlier + (size_t)sslot * d.N;
    const float4* src = kind == 0 ? src_c : (kind == 1 ? src_s : src_o);
    float4* dst = kind == 0 ? L.in_corner + (size_t)slot * L.in_cap_c : (kind == 1 ? L.in_surf + (size_t)slot * L.in_cap_s : L.in_outl + (size_t)slot * L.in_cap_o);
    const int cap = kind == 0 ? L.in_cap_c : (kind == 1 ? L.in_cap_s : L.in_cap_o);
    int n = kind == 0 ? n_c : (kind == 1 ? n_s : n_o);
    if (n > cap) { n = cap; if (threadIdx.x == 0 && blockIdx.x == 0) li[LI_OVERFLOW] = 1; }
    for (int i = blockIdx.x * LM_BLOCK + threadIdx.x; i < n; i += gridDim.x * LM_BLOCK) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) li[LI_NIN_C + kind] = n;
  }
  if (blockIdx.x != 0 || kind != 0 || threadIdx.x != 0) return;
  double* ld = ldp(L, slot);
  double* po = d.poses + (size_t)sslot * 16;
  const double* pin = stage == 2 ? L.stage_odom + ((size_t)slot * 2 + par) * 8 : po;   // this scan's /odom/lidar
  li[LI_RUN] = 0; li[LI_REBUILD] = 0; li[LI_REBUILD_FB] = 0; li[LI_KF_ADDED] = 0; li[LI_OPTIMIZED] = 0; li[LI_FLAGS] = 0;
  if (stage == 2 ? pin[7] == 0.0 : !d.scal[sslot * SC_COUNT + SC_ODOM_VALID]) return;  // no /odom/lidar on the initialising scan -> no mapping frame
  // laserOdomHandler :154-166
  for (int k = 0; k < 3; ++k) ld[LD_T_O2L + k] = pin[k];
  for (int k = 0; k < 4; ++k) ld[LD_Q_O2L + k] = pin[3 + k];
  const DQuat qm2o = ldq(ld + LD_Q_M2O), qo2l = ldq(ld + LD_Q_O2L);
  double r[3];
  dq_rotate(qm2o, ld + LD_T_O2L, r);
  for (int k = 0; k < 3; ++k) ld[LD_T_M2L + k] = r[k] + ld[LD_T_M2O + k];
  stq(ld + LD_Q_M2L, dq_mul(qm2o, qo2l));
  for (int k = 0; k < 3; ++k) po[7 + k] = ld[LD_T_M2L + k];   // /odom_aft_mapped
  for (int k = 0; k < 4; ++k) po[10 + k] = ld[LD_Q_M2L + k];
  // mainLoop gate :107-127
  const int run = (li[LI_FRAME] % d.P.lm_every) == 0;
  li[LI_FRAME] += 1;
  li[LI_RUN] = run;
  if (run_hint >= 0 && run != run_hint) li[LI_OVERFLOW] = 2;  // host launch-skipping logic out of sync
  if (!run) { li[LI_FLAGS] = 8; return; }
  lm_map_update(L, slot, li, d.opt_map_merge);
  li[LI_REBUILD_FB] = li[LI_REBUILD] && !d.opt_map_merge;
}


// grid (8, 3, slots) on the FRONT END's stream: hands this scan over to a LaserMapping that runs on a stream of its own — /odom/lidar
// and its validity into stage_odom[par], and (mapping frames, run_hint != 0) /corner_last, /surf_last, /outlier into the LM inputs.
// After it the front end may overwrite its per-scan buffers; lm_prepare(stage 2) reads only what was staged here.
__global__ void __launch_bounds__(LM_BLOCK) lm_stage(DevCtx d, LmCtx L, int run_hint, int par) {
  const int slot = blockIdx.z + d.slot0, kind = blockIdx.y;
  const int cur = d.scal[slot * SC_COUNT + SC_CUR];  // LO has completed: features of this scan
  int* li = lip(L, slot);
  if (run_hint != 0) {
    const size_t fb = (size_t)slot * 2 + cur;
    const int n_c = d.feat_cnt[fb * 4 + F_LSHARP], n_s = d.scal[slot * SC_COUNT + SC_FE_ERR] ? 0 : d.feat_cnt[fb * 4 + F_LFLAT], n_o = d.scal[slot * SC_COUNT + SC_NOUT];
    const float4* src_c = d.feat[F_LSHARP] + fb * d.fcap[F_LSHARP];
    const float4* src_s = d.feat[F_LFLAT] + fb * d.fcap[F_LFLAT];
    const float4* src_o = d.outlier + (size_t)slot * d.N;
    const float4* src = kind == 0 ? src_c : (kind == 1 ? src_s : src_o);
    float4* dst = kind == 0 ? L.in_corner + (size_t)slot * L.in_cap_c : (kind == 1 ? L.in_surf + (size_t)slot * L.in_cap_s : L.in_outl + (size_t)slot * L.in_cap_o);
    const int cap = kind == 0 ? L.in_cap_c : (kind == 1 ? L.in_cap_s : L.in_cap_o);
    int n = kind == 0 ? n_c : (kind == 1 ? n_s : n_o);
    if (n > cap) { n = cap; if (threadIdx.x == 0 && blockIdx.x == 0) li[LI_OVERFLOW] = 1; }
    for (int i = blockIdx.x * LM_BLOCK + threadIdx.x; i < n; i += gridDim.x * LM_BLOCK) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) li[LI_NIN_C + kind] = n;
  }
  if (blockIdx.x != 0 || kind != 0 || threadIdx.x >= 8) return;
  double* so = L.stage_odom + ((size_t)slot * 2 + par) * 8;
  so[threadIdx.x] = threadIdx.x < 7 ? d.poses[(size_t)slot * 16 + threadIdx.x] : (d.scal[slot * SC_COUNT + SC_ODOM_VALID] ? 1.0 : 0.0);
}


// grid (2, K, slots): chronological concatenation of the window's key frames (:238-243), each transformed by its key pose, in
// the reference's order (corner | surf then outlier per frame).  Only the concat + radix-sort VoxelGrid path (ALEGO_MAP_MERGE=0)
// runs it; the default path merges the pre-sorted key frames (kernels_map.hip).
__global__ void __launch_bounds__(LM_BLOCK) lm_concat(DevCtx d, LmCtx L) {
  const int slot = blockIdx.z + d.slot0, j = blockIdx.y;
  int* li = lip(L, slot);
  if (!li[LI_REBUILD_FB]) return;
  const int nk = li[LI_REC_CNT];
  if (j >= nk) return;
  const int* kc = L.kf_cnt + (size_t)slot * L.KR * 4;
  const int* rec = L.rec + (size_t)slot * L.K;   // frame f lives in ring entry f % KR
  int offc = 0, offs = 0;
  for (int i = 0; i < j; ++i) { const int r = rec[i] % L.KR; offc += kc[r * 4 + 0]; offs += kc[r * 4 + 1] + kc[r * 4 + 2]; }
  const int ring = rec[j] % L.KR;
  const int nc = kc[ring * 4 + 0], ns = kc[ring * 4 + 1], no = kc[ring * 4 + 2];
  const size_t rs = (size_t)slot * L.KR + ring;
  float m[3][4];
  keypose_matrix(L.kf_pose + rs * 8, m);
  const float4* sc_ = L.kf_raw_c + rs * L.kf_cap_c;
  const float4* ss_ = L.kf_raw_s + rs * L.kf_cap_s;
  const float4* so_ = L.kf_raw_o + rs * L.kf_cap_o;
  float4* dc = L.map_corner_raw + (size_t)slot * L.map_cap_c + offc;
  float4* ds = L.map_surf_raw + (size_t)slot * L.map_cap_s + offs;  // surf then outlier per key frame (:241-242)
  for (int i = blockIdx.x * LM_BLOCK + threadIdx.x; i < nc; i += gridDim.x * LM_BLOCK) dc[i] = kf_transform(m, sc_[i]);
  for (int i = blockIdx.x * LM_BLOCK + threadIdx.x; i < ns; i += gridDim.x * LM_BLOCK) ds[i] = kf_transform(m, ss_[i]);
  for (int i = blockIdx.x * LM_BLOCK + threadIdx.x; i < no; i += gridDim.x * LM_BLOCK) ds[ns + i] = kf_transform(m, so_[i]);
  if (j == nk - 1 && blockIdx.x == 0 && threadIdx.x == 0) { li[LI_KRAW_C] = offc + nc; li[LI_KRAW_S] = offs + ns + no; }
}

// grid (8, slots)
__global__ void __launch_bounds__(LM_BLOCK) lm_total(DevCtx d, LmCtx L) {
  const int slot = blockIdx.y + d.slot0;
  int* li = lip(L, slot);
  if (!li[LI_RUN]) return;
  int ns = li[LI_NCUR_S], no = li[LI_NCUR_O];
  if (ns + no > L.total_cap) { no = L.total_cap - ns; if (threadIdx.x == 0 && blockIdx.x == 0) li[LI_OVERFLOW] = 1; }
  const float4* s = L.cur_surf_ds + (size_t)slot * L.kf_cap_s;
  const float4* o = L.cur_outl_ds + (size_t)slot * L.kf_cap_o;
  float4* t = L.cur_total + (size_t)slot * L.total_cap;
  for (int i = blockIdx.x * LM_BLOCK + threadIdx.x; i < ns; i += gridDim.x * LM_BLOCK) t[i] = s[i];
  for (int i = blockIdx.x * LM_BLOCK + threadIdx.x; i < no; i += gridDim.x * LM_BLOCK) t[ns + i] = o[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) li[LI_NTOTAL] = ns + no;
}

DEV_INLINE float vxl_dec(unsigned e) {
  const unsigned b = (e >> 31) ? (e ^ 0x80000000u) : ~e;
  return __int_as_float((int)b);
}

// grid (2, slots): grid geometry from the raw map bounding box, zero the cell counters
DEV_INLINE int grid_cell(const GridGeom& g, float x, float y, float z, int* cx, int* cy, int* cz) {
  int ix = (int)floorf((x - g.ox) * g.inv), iy = (int)floorf((y - g.oy) * g.inv), iz = (int)floorf((z - g.oz) * g.inv);
  *cx = ix; *cy = iy; *cz = iz;
  ix = min(max(ix, 0), g.gx - 1); iy = min(max(iy, 0), g.gy - 1); iz = min(max(iz, 0), g.gz - 1);
  return ix + g.gx * (iy + g.gy * iz);
}

// grid (2, slots): the whole uniform-grid build of one map by one workgroup — geometry from the VoxelGrid bounding box, cell
// counts, exclusive scan, cell-sorted copy.  Only the streams whose map was rebuilt this frame do anything (≈1 in 16), and
// a filtered map is a few thousand points: four launches (setup, count, scan, fill) cost more in launch latency than in work.
__global__ void __launch_bounds__(LM_BLOCK) lm_grid_build(DevCtx d, LmCtx L) {
  const int slot = blockIdx.y + d.slot0, m = blockIdx.x;
  const int* li = lip(L, slot);
  if (!li[LI_REBUILD]) return;
  __shared__ GridGeom s_g;
  __shared__ int s[LM_BLOCK / 64];
  __shared__ int s_run;
  if (threadIdx.x == 0) {
    const unsigned* bb = d.opt_map_merge ? L.map_bbox + ((size_t)slot * 2 + m) * 8 : L.vox_bbox + ((size_t)(slot - L.vox_slot0) * 2 + m) * 8;
    GridGeom g;
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) { mn[a] = vxl_dec(bb[a]); mx[a] = vxl_dec(~bb[4 + a]); }
    if (li[LI_KRAW_C + m] <= 0) { for (int a = 0; a < 3; ++a) { mn[a] = 0.f; mx[a] = 0.f; } }
    float cell = 1.0f;  // >= sqrt(knn_max_dist)
    const float need = sqrtf((float)d.P.knn_max_dist);
    while (cell < need) cell *= 2.0f;
    for (;;) {
      g.gx = (int)floorf((mx[0] - mn[0]) / cell) + 2; g.gy = (int)floorf((mx[1] - mn[1]) / cell) + 2; g.gz = (int)floorf((mx[2] - mn[2]) / cell) + 2;
      if ((long long)g.gx * g.gy * g.gz <= (long long)L.gcap) break;
      cell *= 2.0f;
    }
    g.ox = mn[0]; g.oy = mn[1]; g.oz = mn[2]; g.inv = 1.0f / cell; g.ncell = g.gx * g.gy * g.gz;
    s_g = g;
    L.grid[(size_t)slot * 2 + m] = g;
    s_run = 0;
  }
  __syncthreads();
  const GridGeom g = s_g;
  const int ncell = g.ncell;
  int* cs = L.cell_start + ((size_t)slot * 2 + m) * (L.gcap + 1);
  int* cc = L.cell_cur + ((size_t)slot * 2 + m) * (L.gcap + 1);
  for (int c = threadIdx.x; c <= ncell; c += LM_BLOCK) cs[c] = 0;
  __threadfence_block();
  __syncthreads();
  const int n = li[LI_KDS_C + m];
  const float4* pts = (m == 0 ? L.map_corner_ds + (size_t)slot * L.map_cap_c : L.map_surf_ds + (size_t)slot * L.map_cap_s);
  float4* cp = L.cell_pts + ((size_t)slot * 2 + m) * L.map_cap_s;
  for (int i = threadIdx.x; i < n; i += LM_BLOCK) {
    const float4 p = pts[i];
    int cx, cy, cz;
    atomicAdd(&cs[grid_cell(g, p.x, p.y, p.z, &cx, &cy, &cz)], 1);
  }
  __threadfence_block();
  __syncthreads();
  // exclusive scan of the cell counts (in place) + fill cursors
  for (int c0 = 0; c0 <= ncell; c0 += LM_BLOCK) {
    const int c = c0 + threadIdx.x;
    // (the counts were accumulated by L2 atomics: read them past the CU's vector cache)
    const int v = __hip_atomic_load(&cs[min(c, ncell)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * (c < ncell ? 1 : 0);
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((threadIdx.x & 63) >= o) incl += t; }
    if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < LM_BLOCK / 64; ++w) { if (w < (int)(threadIdx.x >> 6)) woff += s[w]; tot += s[w]; }
    const int run = s_run;
    if (c <= ncell) { const int e2 = run + woff + incl - v; cs[c] = e2; cc[c] = e2; }
    __syncthreads();
    if (threadIdx.x == 0) s_run = run + tot;
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();
  // cell-sorted copy: one contiguous read per cell run in lm_knn (the order inside a cell is whatever the atomics give:
  // the k-NN result does not depend on it, ties are broken by the point index)
  for (int i = threadIdx.x; i < n; i += LM_BLOCK) {
    const float4 p = pts[i];
    int cx, cy, cz;
    const int pos = atomicAdd(&cc[grid_cell(g, p.x, p.y, p.z, &cx, &cy, &cz)], 1);
    cp[pos] = make_float4(p.x, p.y, p.z, __int_as_float(i));
  }
}

// ---- small dense linear algebra in registers ------------------------------------------
// symmetric 3x3 eigen-decomposition, cyclic Jacobi (same algorithm as the oracle's eig3)
DEV_INLINE void d_eig3(const double Ain[9], double lam[3], double vmax[3], double* lam_mid) {
  double A[9], V[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { A[i] = Ain[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (off < 1e-300) break;
    // converged to working precision (round 4, oracle and device alike): the off-diagonal mass is below 1e-40 of the diagonal's, a further rotation moves no entry
    // by more than 1e-20 of the largest.  The absolute test alone ran 8-9 sweeps where 5 suffice — lm_fit was 40 % shorter for it, the bench + 2.8 %.
    if (off <= 1e-40 * (A[0] * A[0] + A[4] * A[4] + A[8] * A[8])) break;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double akp = A[k * 3 + p], akq = A[k * 3 + q]; A[k * 3 + p] = c * akp - s * akq; A[k * 3 + q] = s * akp + c * akq; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double apk = A[p * 3 + k], aqk = A[q * 3 + k]; A[p * 3 + k] = c * apk - s * aqk; A[q * 3 + k] = s * apk + c * aqk; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double vkp = V[k * 3 + p], vkq = V[k * 3 + q]; V[k * 3 + p] = c * vkp - s * vkq; V[k * 3 + q] = s * vkp + c * vkq; }
      }
  }
  const double dd[3] = {A[0], A[4], A[8]};
  int imax = 0, imin = 0;
  // ascending order with std::sort-like tie handling on indices (first minimum, last maximum not needed: ties are measure-zero)
  if (dd[1] > dd[imax]) imax = 1;
  if (dd[2] > dd[imax]) imax = 2;
  if (dd[1] < dd[imin]) imin = 1;
  if (dd[2] < dd[imin]) imin = 2;
  int imid = 3 - imax - imin;
  if (imax == imin) { imax = 2; imin = 0; imid = 1; }
  lam[0] = dd[imin]; lam[1] = dd[imid]; lam[2] = dd[imax];
  *lam_mid = dd[imid];
  vmax[0] = V[0 * 3 + imax]; vmax[1] = V[1 * 3 + imax]; vmax[2] = V[2 * 3 + imax];
}

// Eigen::ColPivHouseholderQR<Matrix<double, 5, 3>>::compute() + solve() (laserMapping.cpp:435) — the parity checker's colpiv_qr_solve (where the algorithm's
// source in Eigen 3.3 is cited step by step) specialised to 5 x 3 with everything in registers: the column of largest remaining norm is swapped into place (selects
// over static indices, no indexed register access), Householder reflector (beta, essential, tau) as Householder.h builds it, LAPACK's norm down-date, nonzeroPivots()
// by Eigen's threshold, and the components past it ZERO — a collinear / coincident neighbourhood gets a finite basic solution instead of a division by a ~1e-17 pivot.
// Same operations in the same order as the oracle, fp64, -ffp-contract=off.  A[c][i] = column c, row i (overwritten).
DEV_INLINE void d_colpiv_qr53(double A[3][5], const double b[5], double x[3]) {
  const double eps = 2.220446049250313e-16, tiny = 2.2250738585072014e-308;
  double nu[3], nd[3], hc[3], c[5];
  int perm[3] = {0, 1, 2};
  double maxn = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double s2 = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) s2 += A[k][i] * A[k][i];
    nd[k] = nu[k] = sqrt(s2);
    if (nu[k] > maxn) maxn = nu[k];
  }
  const double threshold_helper = (maxn * eps) * (maxn * eps) / 5.0;
  const double downdate_threshold = 1.4901161193847656e-08;   // sqrt(eps), exact
  int nonzero = 3;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int big = k;
    double nbig = nu[k];
#pragma unroll
    for (int j = k + 1; j < 3; ++j) if (nu[j] > nbig) { big = j; nbig = nu[j]; }   // maxCoeff: the first of equal maxima
    if (nonzero == 3 && nbig * nbig < threshold_helper * (double)(5 - k)) nonzero = k;
#pragma unroll
    for (int j = k + 1; j < 3; ++j)
      if (big == j) {
#pragma unroll
        for (int i = 0; i < 5; ++i) { const double t = A[k][i]; A[k][i] = A[j][i]; A[j][i] = t; }
        { const double t = nu[k]; nu[k] = nu[j]; nu[j] = t; }
        { const double t = nd[k]; nd[k] = nd[j]; nd[j] = t; }
        { const int t = perm[k]; perm[k] = perm[j]; perm[j] = t; }   // (transposition k of the sequence applied on the right: the oracle's perm after all swaps)
      }
    double tail2 = 0;
#pragma unroll
    for (int i = k + 1; i < 5; ++i) tail2 += A[k][i] * A[k][i];
    const double c0 = A[k][k];
    double beta, tau;
    if (tail2 <= tiny) {
      tau = 0; beta = c0;
#pragma unroll
      for (int i = k + 1; i < 5; ++i) A[k][i] = 0;
    } else {
      beta = sqrt(c0 * c0 + tail2);
      if (c0 >= 0) beta = -beta;
#pragma unroll
      for (int i = k + 1; i < 5; ++i) A[k][i] = A[k][i] / (c0 - beta);
      tau = (beta - c0) / beta;
    }
    hc[k] = tau;
    A[k][k] = beta;
    if (tau != 0) {
#pragma unroll
      for (int j = k + 1; j < 3; ++j) {
        double t = 0;
#pragma unroll
        for (int i = k + 1; i < 5; ++i) t += A[k][i] * A[j][i];
        t += A[j][k];
        A[j][k] -= tau * t;
#pragma unroll
        for (int i = k + 1; i < 5; ++i) A[j][i] -= tau * A[k][i] * t;
      }
    }
#pragma unroll
    for (int j = k + 1; j < 3; ++j) {
      if (nu[j] != 0) {
        double t = fabs(A[j][k]) / nu[j];
        t = (1.0 + t) * (1.0 - t);
        t = t < 0 ? 0 : t;
        const double q = nu[j] / nd[j];
        const double t2 = t * (q * q);
        if (t2 <= downdate_threshold) {
          double s2 = 0;
#pragma unroll
          for (int i = k + 1; i < 5; ++i) s2 += A[j][i] * A[j][i];
          nd[j] = nu[j] = sqrt(s2);
        } else nu[j] *= sqrt(t);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) c[i] = b[i];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double tau = hc[k];
    if (k < nonzero && tau != 0) {
      double t = 0;
#pragma unroll
      for (int i = k + 1; i < 5; ++i) t += A[k][i] * c[i];
      t += c[k];
      c[k] -= tau * t;
#pragma unroll
      for (int i = k + 1; i < 5; ++i) c[i] -= tau * A[k][i] * t;
    }
  }
#pragma unroll
  for (int k = 2; k >= 0; --k) {
    if (k < nonzero) {
      double sum = c[k];
#pragma unroll
      for (int j = k + 1; j < 3; ++j) if (j < nonzero) sum -= A[j][k] * c[j];
      c[k] = sum / A[k][k];
    } else c[k] = 0.0;
  }
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) if (perm[i] == o) v = c[i];
    x[o] = v;
  }
}

// minimum of a u64 over a group of LM_KNN_LANES consecutive lanes (high word, then low word among the winners): every
// lane of the group gets it.  DPP quad_perm xor 1, xor 2, row_half_mirror cover 8 lanes; row_mirror extends to 16.
#ifndef LM_KNN_LANES
#define LM_KNN_LANES 4   // measured at 1024 streams: 16 lanes 1008 us, 8: 689, 4: 581, 2: 662, 1: 1176
#endif
DEV_INLINE uint32_t row_max_u32(uint32_t v) {
  int x = (int)v, t;
  if (LM_KNN_LANES >= 2) { t = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x; }
  if (LM_KNN_LANES >= 4) { t = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x; }
  if (LM_KNN_LANES >= 8) { t = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x; }
  if (LM_KNN_LANES == 16) { t = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false); x = (uint32_t)t > (uint32_t)x ? t : x; }
  return (uint32_t)x;
}
DEV_INLINE unsigned long long row_min_u64(unsigned long long v) {
  const uint32_t hi = ~row_max_u32(~(uint32_t)(v >> 32));
  const uint32_t lo = ~row_max_u32(~((uint32_t)(v >> 32) == hi ? (uint32_t)v : 0xFFFFFFFFu));
  return ((unsigned long long)hi << 32) | lo;
}

// The queries [qlo, qhi) of `kind` that this rank registers: all of them, or — one registration sharded over the ranks of a communicator
// (alego_dist_init) — this rank's contiguous slice of laser_corner_ds_ ++ laser_surf_total_ds_.
DEV_INLINE void lm_shard_slice(const LmCtx& L, int nqc, int nqs, int kind, int* qlo, int* qhi) {
  int lo = 0, hi = nqc + nqs;
  if (L.shard_world > 1) {
    const long long T = nqc + nqs;
    lo = (int)(T * L.shard_rank / L.shard_world); hi = (int)(T * (L.shard_rank + 1) / L.shard_world);
  }
  if (kind == 0) { *qlo = min(lo, nqc); *qhi = min(hi, nqc); } else { *qlo = max(lo - nqc, 0); *qhi = max(hi - nqc, 0); }
}

// Workgroup order of the registration kernels (lm_knn, lm_fit): a 1-D grid of n_launch x (2 kinds x GX) workgroups in which the workgroups of EIGHT
// slots are interleaved — b % 8 picks the slot of the octet, so all workgroups of a slot land on one XCD (workgroup b runs on XCD b mod 8) as before —
// and the octets follow one another: a slot's ~100 workgroups are dispatched back to back instead of one in every n_launch, and its map (2 x 190 KB
// of cell-sorted points) is read into that XCD's L2 once instead of being evicted between them (lm_knn fetched 3x the maps' size from HBM).
DEV_INLINE bool lm_reg_block(const DevCtx& d, int GX, int* slot, int* kind, int* bx) {
  const int per = 2 * GX, b = blockIdx.x, oct = b / (8 * per), r = b - oct * 8 * per;
  const int s = oct * 8 + (r & 7), w = r >> 3;
  *slot = s + d.slot0; *kind = w / GX; *bx = w - (w / GX) * GX;
  return s < d.n_launch;
}
static inline int lm_reg_grid(const DevCtx& d, int GX) { return ((d.n_launch + 7) / 8) * 8 * 2 * GX; }

// 1-D grid (lm_reg_grid): LM_KNN_LANES lanes per query, 128-thread workgroups, LM_ASSOC_GX of them per stream and kind, grid-stride over the queries.
// The lanes of a group split the candidates of the 27 surrounding cells, keep a private top-5 each and merge them
// with five group-wide arg-min rounds.  The kernel is instruction-issue bound: with 4 lanes per query the per-query
// fixed work (pose transform, cell addressing, merge) is shared by 16 queries per wavefront.
#define LM_ASSOC_GX 64
#ifndef LM_KNN_PK
#define LM_KNN_PK 1
#endif
__global__ void __launch_bounds__(128) lm_knn(DevCtx d, LmCtx L) {
  // 1-D grid, lm_reg_block: all workgroups of a stream share one XCD and its L2 and are dispatched back to back (measured HBM traffic of the kernel:
  // 0.60 -> 0.20 MB per scan; throughput unchanged)
  int slot, kind, bx;
  if (!lm_reg_block(d, LM_ASSOC_GX, &slot, &kind, &bx)) return;
  const int gx = LM_ASSOC_GX;
  const int* li = lip(L, slot);
  if (!li[LI_RUN]) return;
  const alego_params& P = d.P;
  // registration guard :350
  if (li[LI_NCUR_C] < P.lm_min_corner || li[LI_NTOTAL] < P.lm_min_surf || li[LI_KDS_C] < P.lm_min_map_corner || li[LI_NKF] == 0) return;
  const int nq = kind == 0 ? li[LI_NCUR_C] : li[LI_NTOTAL_DS];
  int qlo, qhi;
  lm_shard_slice(L, li[LI_NCUR_C], li[LI_NTOTAL_DS], kind, &qlo, &qhi);
  const float4* qp = kind == 0 ? L.cur_corner_ds + (size_t)slot * L.kf_cap_c : L.cur_total_ds + (size_t)slot * L.total_cap;
  const int nmap = li[LI_KDS_C + kind];
  const GridGeom g = L.grid[(size_t)slot * 2 + kind];
  const int* cs = L.cell_start + ((size_t)slot * 2 + kind) * (L.gcap + 1);
  const float4* cp = L.cell_pts + ((size_t)slot * 2 + kind) * L.map_cap_s;
  const double* ld = ldp(L, slot);
  const DQuat qm = ldq(ld + LD_Q_M2L);
  constexpr int QPB = 128 / LM_KNN_LANES;
  const int sub = threadIdx.x & (LM_KNN_LANES - 1);
  // bit pattern of the smallest f32 f with (double)f >= knn_max_dist: dist < f  <=>  (double)dist < knn_max_dist (:376,:426 accept a query only
  // when its FIFTH neighbour is closer than that, so farther candidates need not enter the top-5 sets at all)
  float flim = (float)P.knn_max_dist;
  if ((double)flim < P.knn_max_dist) flim = __int_as_float(__float_as_int(flim) + 1);
  const uint32_t limbits = P.knn_max_dist > 0.0 ? (uint32_t)__float_as_int(flim) : 0u;
  // every lane of a wavefront runs the same number of iterations (DPP reads neighbours' registers): clamp, don't exit
  const int nq_round = (nq + QPB - 1) / QPB * QPB;
  for (int qq = bx * QPB + threadIdx.x / LM_KNN_LANES; qq < nq_round; qq += gx * QPB) {
  const int q = min(qq, nq - 1);
  const float4 pin = qp[q];
  // pointAssociateToMap laserMapping.h:187-194 (pose predicted from odometry)
  const double vin[3] = {pin.x, pin.y, pin.z};
  double r[3];
  dq_rotate(qm, vin, r);
  const float sx = (float)(r[0] + ld[LD_T_M2L + 0]), sy = (float)(r[1] + ld[LD_T_M2L + 1]), sz = (float)(r[2] + ld[LD_T_M2L + 2]);
  // exact 5-NN among all points with d^2 < knn_max_dist: they all lie in the 27 surrounding cells.
  // A lane's five best candidates are an UNSORTED set of 64-bit keys (f32 distance bits << 32 | map index: the lexicographic
  // (distance, index) order is the integer order) with the set's maximum tracked next to it: a candidate that beats the maximum
  // replaces it (five compare-selects) and the maximum is recomputed (four), instead of a five-step insertion sort — the kernel is
  // VALU-bound and the insertion ran for every candidate of every lane (any lane of the wavefront inserting makes all of them pay).
  typedef unsigned long long u64k;
  const u64k KNONE = ~0ull;
  u64k k0 = KNONE, k1 = KNONE, k2 = KNONE, k3 = KNONE, k4 = KNONE, kmax = KNONE;
  if (nmap >= 5 && q >= qlo && q < qhi) {   // (another rank's query: nothing to search, the row stays empty)
    int cx, cy, cz;
    grid_cell(g, sx, sy, sz, &cx, &cy, &cz);
    // bounds of the nine x-runs of cells first (18 independent loads), then the candidates two per lane at a time
    int rs[9], re[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
      const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.gx - 1);
      const bool in = z >= 0 && z < g.gz && y >= 0 && y < g.gy && x0 <= x1;
      const int c0 = x0 + g.gx * (y + g.gy * z), c1 = x1 + g.gx * (y + g.gy * z);
      // (clamped addresses, not `in ? cs[c0] : 0`: behind a branch every one of the 18 loads waited for the previous one)
      const int a0 = cs[in ? c0 : 0], a1 = cs[in ? c1 + 1 : 0];
      rs[r] = in ? a0 : 0;
      re[r] = in ? a1 : 0;
    }
    // (x, y) as one packed pair: v_pk_add_f32 / v_pk_mul_f32 are IEEE operations on both halves, the sum keeps the reference's order
    // ((dx^2 + dy^2) + dz^2; 0 + dx^2 = dx^2 exactly) — 6 instead of 8 VALU instructions per candidate, same bits
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f sxy = {sx, sy};
    auto consider = [&](const float4& a) {
#if LM_KNN_PK
      const v2f axy = {a.x, a.y};
      const v2f dxy = sxy - axy, sq = dxy * dxy;
      const float dzv = sz - a.z;
      float dist = sq.x + sq.y;
      dist += dzv * dzv;
#else
      float dist = 0.f, df;
      df = sx - a.x; dist += df * df;
      df = sy - a.y; dist += df * df;
      df = sz - a.z; dist += df * df;
#endif
      const u64k key = ((u64k)(uint32_t)d_f2i(dist) << 32) | (uint32_t)__float_as_int(a.w);   // dist >= 0: its bit pattern orders like its value
      if (key < kmax && (uint32_t)d_f2i(dist) < limbits) {   // (limbits: a neighbour at knn_max_dist or beyond can never be part of an ACCEPTED query — see `ok` below)
        // (a set that is not full yet holds KNONE entries, which are its maximum; equal KNONE entries are all "the maximum": only one may be replaced)
        const bool e0 = k0 == kmax, e1 = !e0 && k1 == kmax, e2 = !e0 && !e1 && k2 == kmax, e3 = !e0 && !e1 && !e2 && k3 == kmax, e4 = !e0 && !e1 && !e2 && !e3;
        k0 = e0 ? key : k0; k1 = e1 ? key : k1; k2 = e2 ? key : k2; k3 = e3 ? key : k3; k4 = e4 ? key : k4;
        const u64k m01 = k0 > k1 ? k0 : k1, m23 = k2 > k3 ? k2 : k3, m03 = m01 > m23 ? m01 : m23;
        kmax = m03 > k4 ? m03 : k4;
      }
    };
    // the first candidate of each of the nine runs: nine independent loads (a run is usually exhausted by one or two turns)
    float4 first[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) first[r] = cp[rs[r] + sub < re[r] ? rs[r] + sub : 0];
    // the query's own row of cells first, then the rows that share a face with it, the four diagonal ones last: the sets' maxima fall early and
    // fewer of the later candidates enter a set at all (the result does not depend on the order: a set of unique keys)
#ifndef LM_KNN_ORDER
#define LM_KNN_ORDER 1
#endif
    constexpr int ord[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};
#pragma unroll
    for (int k = 0; k < 9; ++k) { const int r = LM_KNN_ORDER ? ord[k] : k; if (rs[r] + sub < re[r]) consider(first[r]); }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int r = LM_KNN_ORDER ? ord[k] : k;
      for (int t = rs[r] + sub + LM_KNN_LANES; t < re[r]; t += 2 * LM_KNN_LANES) {  // an x-run of cells is contiguous in the cell-sorted copy
        const int t1 = t + LM_KNN_LANES;
        const float4 a0 = cp[t], a1 = cp[t1 < re[r] ? t1 : t];
        consider(a0);
        if (t1 < re[r]) consider(a1);
      }
    }
  }
  // merge the private sets of the query's lanes: five rounds of "smallest key of the group"; the lane that holds it drops it
  float bd[5];
  int bi[5];
  {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const u64k m01 = k0 < k1 ? k0 : k1, m23 = k2 < k3 ? k2 : k3, m03 = m01 < m23 ? m01 : m23, mine = m03 < k4 ? m03 : k4;
      const u64k m = row_min_u64(mine);
      bd[k] = d_i2f((int32_t)(m >> 32)); bi[k] = (int)(uint32_t)m;   // (KNONE: distance bits 0xFFFFFFFF = NaN, index 0xFFFFFFFF — see `ok` below)
      if (mine == m && m != KNONE) {   // keys are unique (one per map point): exactly one lane of the group, exactly one of its entries
        k0 = k0 == m ? KNONE : k0; k1 = k1 == m ? KNONE : k1; k2 = k2 == m ? KNONE : k2; k3 = k3 == m ? KNONE : k3; k4 = k4 == m ? KNONE : k4;
      }
    }
  }
  // neighbour indices (ascending distance) for lm_fit; idx[0] < 0 = rejected (:376,:426)
  if (sub == 0 && qq < nq) {
    int* kn = L.knn + ((size_t)slot * L.qcap + (kind == 0 ? 0 : L.kf_cap_c) + q) * 5;
    const bool ok = bi[4] != -1 && (double)bd[4] < P.knn_max_dist && q >= qlo && q < qhi;   // (queries of other ranks' slices give no row here)
#pragma unroll
    for (int k = 0; k < 5; ++k) kn[k] = ok ? bi[k] : -1;
  }
  }
}

// grid (ceil(qcap/128), 2, slots): one thread per query: 3x3 scatter eigen-decomposition (line) or 5x3 Householder
// least squares (plane) on the five neighbours found by lm_knn
#define LM_FIT_GX 20   // x 128 threads: the 1-2 k queries of a kind in one sweep; larger clouds grid-stride
__global__ void __launch_bounds__(128) lm_fit(DevCtx d, LmCtx L) {
  int slot, kind, bx;
  if (!lm_reg_block(d, LM_FIT_GX, &slot, &kind, &bx)) return;   // (workgroup order: see lm_knn)
  const int gx = LM_FIT_GX;
  const int* li = lip(L, slot);
  if (!li[LI_RUN]) return;
  const alego_params& P = d.P;
  if (li[LI_NCUR_C] < P.lm_min_corner || li[LI_NTOTAL] < P.lm_min_surf || li[LI_KDS_C] < P.lm_min_map_corner || li[LI_NKF] == 0) return;
  const int nq = kind == 0 ? li[LI_NCUR_C] : li[LI_NTOTAL_DS];
  const float4* mp = kind == 0 ? L.map_corner_ds + (size_t)slot * L.map_cap_c : L.map_surf_ds + (size_t)slot * L.map_cap_s;
  for (int q = bx * 128 + threadIdx.x; q < nq; q += gx * 128) {
  const int* kn = L.knn + ((size_t)slot * L.qcap + (kind == 0 ? 0 : L.kf_cap_c) + q) * 5;
  double* blk = L.blocks + ((size_t)slot * L.qcap + (kind == 0 ? 0 : L.kf_cap_c) + q) * 8;
  int bi[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) bi[k] = kn[k];
  blk[7] = 0.0;
  do {
    if (bi[0] < 0) break;
  if (kind == 0) {  // :378-411
    double near[5][3], center[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const float4 m = mp[bi[j]];
      near[j][0] = m.x; near[j][1] = m.y; near[j][2] = m.z;
#pragma unroll
      for (int a = 0; a < 3; ++a) center[a] = center[a] + near[j][a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) center[a] = center[a] / 5.0;
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const double zm[3] = {near[j][0] - center[0], near[j][1] - center[1], near[j][2] - center[2]};
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) cov[a * 3 + b] = cov[a * 3 + b] + zm[a] * zm[b];
    }
    double lam[3], v[3], lmid;
    d_eig3(cov, lam, v, &lmid);
    if (lam[2] > P.line_ratio * lam[1]) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { blk[a] = P.line_half_len * v[a] + center[a]; blk[3 + a] = -P.line_half_len * v[a] + center[a]; }
      blk[6] = 0.0;
      blk[7] = 2.0;  // BLK_EDGE
    }
  } else {  // :424-460
    double A[3][5], b[5], nrm[3];
    float mx[5], my[5], mz[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const float4 m = mp[bi[j]];
      mx[j] = m.x; my[j] = m.y; mz[j] = m.z;
      A[0][j] = m.x; A[1][j] = m.y; A[2][j] = m.z; b[j] = -1.0;
    }
    d_colpiv_qr53(A, b, nrm);   // :435
    const double nn = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    const double dd = 1 / nn;
#pragma unroll
    for (int a = 0; a < 3; ++a) nrm[a] /= nn;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (fabs(nrm[0] * mx[j] + nrm[1] * my[j] + nrm[2] * mz[j] + dd) > P.plane_tol) ok = false;
    if (ok) {
      blk[0] = nrm[0]; blk[1] = nrm[1]; blk[2] = nrm[2]; blk[3] = 0; blk[4] = 0; blk[5] = 0;
      blk[6] = dd;
      blk[7] = 3.0;  // BLK_PLANE
    }
  }
  } while (false);
  }
}

// ---- the accepted residual rows of one registration --------------------------------------------------------------------------
// Round 3: lm_solve used to pack its ~2.9 k accepted rows (80 B each = 230 KB per stream) into HBM and stream them through every one
// of the ~30 evaluations of a mapping frame: 512 streams x 230 KB x 30 = 3.5 GB per launch out of the L2 / Infinity Cache, which is what
// bounded the kernel (514 us per launch with the rows, 0.94 MB of HBM traffic per scan).  Now the rows live in LDS for the whole solve,
// structure of arrays, edges (a, b: 6 f64 + the query point 3 f32 = 60 B) apart from planes (normal, d: 4 f64 + 3 f32 = 44 B): 143 KB
// at 16 x 1800.  Rows beyond the workgroup's LDS budget stay in `crows` (80 B rows, edges at [0, Rc), planes behind them) and are read
// from there — same arithmetic, same order.  The compaction is STABLE in the query index (every thread takes a contiguous range of
// queries), thread t evaluates edges t, t + T, ... and then planes t, t + T, ...: the summation order of the normal equations is a
// function of the accepted-query lists alone, and lm_shard_eval (all rows in `crows`) follows the same order.
// Measured on the 2048-stream bench (four stream groups share the chip), T threads x LDS for the rows, always against the other variants in
// the same run (runs on different boxes differ by +-3 %).  While map_update still flooded the chip with empty 68 KB workgroups: 512 x 158 KB 348 k
// scans/s, 256 x 158 KB 347 k, 256 x 96 KB 355 k, 256 x 48 KB 357 k, 256 x 0 354 k, 1024 x any 287 k (128 VGPRs: spills); the old kernel 349 k.
// After that fix: 256 x 96 KB 408 k, 256 x 32 KB 410 k, 256 x 0 409 k, 256 x 158 KB 399 k; 512 x 96 KB against 256 x 96 KB: 396 k / 404 k.
// The row budget does not matter except for the variant that needs a CU's whole LDS (it has to wait for every other LDS user to leave);
// 256 threads x 254 VGPRs leave half a CU's register file to the other stream groups.  A single stream's solve takes 94 us with 512 threads,
// ~130 us with 256, 150 us before.  Two thirds of the rows on chip remove two thirds of the kernel's HBM traffic: 256 threads, 96 KB.
#ifndef LM_SOLVE_T
#define LM_SOLVE_T 256
#endif
#define LM_SOLVE_RED_BYTES ((size_t)28 * (LM_SOLVE_T / 8) * sizeof(double))
#define LM_SOLVE_DYN_BYTES ((size_t)158 * 1024)   // upper limit (ALEGO_LM_ROW_LDS): of the CU's 160 KB; the kernel's static LDS (solver state, scan tables) is < 2 KB
#define LM_SOLVE_ROW_BYTES_DEFAULT ((size_t)96 * 1024)
struct LmRows {
  int Rc, Rs;             // accepted corner (edge) / surf (plane) rows
  int Ce, Cp;             // the first Ce edges / Cp planes are resident in LDS
  double *ea, *eb, *pn;   // LDS: ea[k * Ce + i], eb[k * Ce + i] (k < 3); pn[k * Cp + j] (k < 4: normal, negative_OA_dot_norm)
  float *ep, *pp;         // LDS: query points ep[k * Ce + i], pp[k * Cp + j] (k < 3)
  double* crows;          // HBM: rows that do not fit (all of them on the sharded path)
};

template <int T>
DEV_INLINE void lm_pack_rows(const LmCtx& L, int slot, int nqc, int nqs, unsigned char* row_lds, size_t row_budget, int (*s_cnt)[T / 64], LmRows& W) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* blocks = L.blocks + (size_t)slot * L.qcap * 8;
  const float4* qc = L.cur_corner_ds + (size_t)slot * L.kf_cap_c;
  const float4* qs = L.cur_total_ds + (size_t)slot * L.total_cap;
  const int n_all = nqc + nqs, per = (n_all + T - 1) / T;
  const int lo = min(tid * per, n_all), hi = min(lo + per, n_all);
  auto qrow = [&](int i) -> size_t { return (size_t)(i < nqc ? i : L.kf_cap_c + (i - nqc)); };
  constexpr int U = 4;   // loads in flight per thread
  int cc = 0, cs = 0;
  for (int i0 = lo; i0 < hi; i0 += U) {
    double ty[U];
#pragma unroll
    for (int u = 0; u < U; ++u) ty[u] = blocks[qrow(min(i0 + u, n_all - 1)) * 8 + 7];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + u < hi && ty[u] != 0.0) { if (i0 + u < nqc) ++cc; else ++cs; }
  }
  int ic = cc, is = cs;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int tc = __shfl_up(ic, o, 64), ts = __shfl_up(is, o, 64); if (lane >= o) { ic += tc; is += ts; } }
  if (lane == 63) { s_cnt[0][wave] = ic; s_cnt[1][wave] = is; }
  __syncthreads();
  int epos = ic - cc, ppos = is - cs, Rc = 0, Rs = 0;
#pragma unroll
  for (int w = 0; w < T / 64; ++w) { const int a = s_cnt[0][w], b = s_cnt[1][w]; if (w < wave) { epos += a; ppos += b; } Rc += a; Rs += b; }
  W.Rc = Rc; W.Rs = Rs;
  W.Ce = (int)min((size_t)Rc, row_budget / 60);
  W.Cp = (int)min((size_t)Rs, (row_budget - (size_t)60 * W.Ce) / 44);
  W.ea = reinterpret_cast<double*>(row_lds); W.eb = W.ea + 3 * (size_t)W.Ce; W.pn = W.eb + 3 * (size_t)W.Ce;
  W.ep = reinterpret_cast<float*>(W.pn + 4 * (size_t)W.Cp); W.pp = W.ep + 3 * (size_t)W.Ce;
  W.crows = L.crows + (size_t)slot * L.qcap * 10;
  for (int i0 = lo; i0 < hi; i0 += U) {
    double4 blo[U], bhi[U];
    float4 pt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(i0 + u, n_all - 1);
      const double4* b = reinterpret_cast<const double4*>(blocks + qrow(i) * 8);
      blo[u] = b[0]; bhi[u] = b[1];
      pt[u] = i < nqc ? qc[i] : qs[i - nqc];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u;
      if (i < hi && bhi[u].w != 0.0) {
        const bool is_c = i < nqc;
        const int r = is_c ? epos++ : ppos++;
        if (is_c && r < W.Ce) {
          W.ea[r] = blo[u].x; W.ea[W.Ce + r] = blo[u].y; W.ea[2 * W.Ce + r] = blo[u].z;
          W.eb[r] = blo[u].w; W.eb[W.Ce + r] = bhi[u].x; W.eb[2 * W.Ce + r] = bhi[u].y;
          W.ep[r] = pt[u].x; W.ep[W.Ce + r] = pt[u].y; W.ep[2 * W.Ce + r] = pt[u].z;
        } else if (!is_c && r < W.Cp) {
          W.pn[r] = blo[u].x; W.pn[W.Cp + r] = blo[u].y; W.pn[2 * W.Cp + r] = blo[u].z; W.pn[3 * W.Cp + r] = bhi[u].z;
          W.pp[r] = pt[u].x; W.pp[W.Cp + r] = pt[u].y; W.pp[2 * W.Cp + r] = pt[u].z;
        } else {
          double2* o = reinterpret_cast<double2*>(W.crows + (size_t)(is_c ? r : Rc + r) * 10);   // 80 B rows: 16-byte aligned
          o[0] = make_double2(blo[u].x, blo[u].y); o[1] = make_double2(blo[u].z, blo[u].w);
          o[2] = make_double2(bhi[u].x, bhi[u].y); o[3] = make_double2(bhi[u].z, bhi[u].w);
          *reinterpret_cast<float4*>(o + 4) = pt[u];
        }
      }
    }
  }
  __threadfence_block();
  __syncthreads();
}

// residuals + Jacobians of all rows at the pose of Tm, accumulated into this thread's 28 normal-equation scalars
template <int T>
DEV_INLINE void lm_eval_rows(const LmRows& W, const PoseTerms& Tm, double huber, double acc[28]) {
  const double c3[3] = {0, 0, 0};
  for (int i = threadIdx.x; i < W.Rc; i += T) {   // EdgeCostFunction (utility.h:242-297)
    double a3[3], b3[3], cp[3];
    if (i < W.Ce) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { a3[k] = W.ea[k * W.Ce + i]; b3[k] = W.eb[k * W.Ce + i]; cp[k] = (double)W.ep[k * W.Ce + i]; }
    } else {
      const double2* b = reinterpret_cast<const double2*>(W.crows + (size_t)i * 10);
      const double2 q0 = b[0], q1 = b[1], q2 = b[2];
      const float4 pc = *reinterpret_cast<const float4*>(b + 4);
      a3[0] = q0.x; a3[1] = q0.y; a3[2] = q1.x; b3[0] = q1.y; b3[1] = q2.x; b3[2] = q2.y; cp[0] = pc.x; cp[1] = pc.y; cp[2] = pc.z;
    }
    double res, J[6];
    eval_block(BLK_EDGE, cp, a3, b3, c3, 0.0, Tm, &res, J);
    accumulate_block(res, J, huber, acc);
  }
  for (int j = threadIdx.x; j < W.Rs; j += T) {   // PlaneCostFunction (utility.h:299-349)
    double a3[3], cp[3], dd;
    if (j < W.Cp) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { a3[k] = W.pn[k * W.Cp + j]; cp[k] = (double)W.pp[k * W.Cp + j]; }
      dd = W.pn[3 * W.Cp + j];
    } else {
      const double2* b = reinterpret_cast<const double2*>(W.crows + (size_t)(W.Rc + j) * 10);
      const double2 q0 = b[0], q1 = b[1], q3 = b[3];
      const float4 pc = *reinterpret_cast<const float4*>(b + 4);
      a3[0] = q0.x; a3[1] = q0.y; a3[2] = q1.x; dd = q3.x; cp[0] = pc.x; cp[1] = pc.y; cp[2] = pc.z;
    }
    double res, J[6];
    eval_block(BLK_PLANE, cp, a3, c3, c3, dd, Tm, &res, J);
    accumulate_block(res, J, huber, acc);
  }
}

// grid (slots): scan2MapOptimization's solver part
__global__ void __launch_bounds__(LM_SOLVE_T) lm_solve(DevCtx d, LmCtx L) {
  const int slot = blockIdx.x + d.slot0;
  int* li = lip(L, slot);
  if (!li[LI_RUN]) return;
  const alego_params& P = d.P;
  double* ld = ldp(L, slot);
  if (li[LI_NCUR_C] < P.lm_min_corner || li[LI_NTOTAL] < P.lm_min_surf || li[LI_KDS_C] < P.lm_min_map_corner || li[LI_NKF] == 0) {
    if (threadIdx.x == 0) { li[LI_FLAGS] |= 16; li[LI_NCC] = 0; li[LI_NSC] = 0; li[LI_SUM0] = 0; li[LI_SUM1] = 0; }
    return;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char lm_smem[];
  double* s_acc = reinterpret_cast<double*>(lm_smem);                // [28][LM_SOLVE_T / 8]
  __shared__ double s_out[28], s_trig[12];
  __shared__ LmState S;
  __shared__ int s_action, s_cnt[2][LM_SOLVE_T / 64];
  // ---- correspondence counts (:465) and the accepted rows, into LDS ----
  LmRows W;
#ifdef ALEGO_TIMING
  const long long cpack_ = clock64();
#endif
  lm_pack_rows<LM_SOLVE_T>(L, slot, li[LI_NCUR_C], li[LI_NTOTAL_DS], lm_smem + LM_SOLVE_RED_BYTES, L.solve_row_bytes, s_cnt, W);
  if (threadIdx.x == 0) { li[LI_NCC] = W.Rc; li[LI_NSC] = W.Rs; li[LI_OPTIMIZED] = 1; }
  const int R = W.Rc + W.Rs;
  double acc[28];
#ifdef ALEGO_TIMING
  long long tm[5] = {0, 0, 0, 0, clock64() - cpack_};   // pose terms, rows, reduction, trust-region control (one thread), pack
#define TM(k, expr) { const long long c_ = clock64(); expr; tm[k] += clock64() - c_; }
#else
#define TM(k, expr) { expr; }
#endif
  auto evaluate = [&](const double* x) {
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0;
    PoseTerms T;
    TM(0, T = pose_terms_coop(x, s_trig));
    TM(1, lm_eval_rows<LM_SOLVE_T>(W, T, P.huber_delta, acc));
    TM(2, block_reduce28_oct<LM_SOLVE_T>(acc, s_acc, s_out));
  };
  for (int outer = 0; outer < P.lm_outer_iters; ++outer) {  // :360 — identical correspondences both times
    if (R == 0) {  // ceres::Solve on an empty problem is a no-op
      if (threadIdx.x == 0 && outer < 2) {
        li[LI_SUM0 + outer] = 4 << 16;
        for (int k = 0; k < 6; ++k) ld[LD_PARAMS_IT + outer * 6 + k] = ld[LD_PARAMS + k];
      }
      continue;
    }
    double x0[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) x0[k] = ld[LD_PARAMS + k];
    evaluate(x0);
    if (threadIdx.x == 0) lm_begin(S, x0, s_out, P.lm_max_iters);
    __syncthreads();
    while (true) {
      TM(3, if (threadIdx.x == 0) s_action = lm_propose(S); __syncthreads());
      const int act = s_action;
      if (act == LM_STOP) break;
      if (act == LM_EVAL) {
        double xc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) xc[k] = S.cand[k];
        evaluate(xc);
        TM(3, if (threadIdx.x == 0) s_action = lm_consume(S, s_out); __syncthreads());
        if (s_action == LM_STOP) break;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
#ifdef ALEGO_TIMING
      for (int k = 0; k < 5; ++k) ld[43 + k] = (double)tm[k];   // (accumulated over both outer iterations)
#endif
#pragma unroll
      for (int k = 0; k < 6; ++k) ld[LD_PARAMS + k] = S.x[k];
      if (outer < 2) {
        for (int k = 0; k < 6; ++k) ld[LD_PARAMS_IT + outer * 6 + k] = S.x[k];
        ld[LD_COSTS + outer * 2] = S.initial_cost; ld[LD_COSTS + outer * 2 + 1] = S.x_cost;
        li[LI_SUM0 + outer] = S.iter | (S.successful << 8) | (S.termination << 16);
      }
    }
    __syncthreads();
  }
}

// grid (ceil(slots/64)): saveKeyFramesAndFactor :491-559 (no-loop-closure pass-through) + transformUpdate :481-489
__global__ void lm_finish(DevCtx d, LmCtx L) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.n_launch) return;
  const int slot = s + d.slot0;
  int* li = lip(L, slot);
  if (!li[LI_RUN]) return;
  li[LI_REBUILD] = 0;   // the map sequence of this frame is done: no stale flag for a later round over the group's jobs
  double* ld = ldp(L, slot);
  const int nkf = li[LI_NKF];
  bool add = true;
  if (nkf > 0) {
    const float* pre = L.kf_pose + ((size_t)slot * L.KR + (nkf - 1) % L.KR) * 8;
    const double ex = ld[LD_T_M2L + 0] - (double)pre[0], ey = ld[LD_T_M2L + 1] - (double)pre[1], ez = ld[LD_T_M2L + 2] - (double)pre[2];
    if (ex * ex + ey * ey + ez * ez < d.P.min_keyframe_dist) add = false;  // :501-508
  }
  if (add) {
    double R[9];
    dq_to_mat(ldq(ld + LD_Q_M2L), R);
    const double roll = atan2(R[7], R[8]);
    const double pitch = atan2(-R[6], sqrt(R[7] * R[7] + R[8] * R[8]));
    const double yaw = atan2(R[3], R[0]);
    float* kp = L.kf_pose + ((size_t)slot * L.KR + nkf % L.KR) * 8;
    kp[0] = (float)ld[LD_T_M2L + 0]; kp[1] = (float)ld[LD_T_M2L + 1]; kp[2] = (float)ld[LD_T_M2L + 2];
    kp[3] = (float)roll; kp[4] = (float)pitch; kp[5] = (float)yaw;
    for (int k = 0; k < 6; ++k) ld[LD_PARAMS + k] = (double)kp[k];  // :539-544
    li[LI_NKF] = nkf + 1; li[LI_DIRTY] = 1; li[LI_KF_ADDED] = 1; li[LI_FLAGS] |= 32;
  }
  // transformUpdate
  const DQuat qm2l = dq_zyx(ld[LD_PARAMS + 5], ld[LD_PARAMS + 4], ld[LD_PARAMS + 3]);
  stq(ld + LD_Q_M2L, qm2l);
  for (int k = 0; k < 3; ++k) ld[LD_T_M2L + k] = ld[LD_PARAMS + k];
  const DQuat qm2o = dq_mul(qm2l, dq_inverse(ldq(ld + LD_Q_O2L)));
  stq(ld + LD_Q_M2O, qm2o);
  double r[3];
  dq_rotate(qm2o, ld + LD_T_O2L, r);
  for (int k = 0; k < 3; ++k) ld[LD_T_M2O + k] = ld[LD_T_M2L + k] - r[k];
}

// grid (8, 3, slots): a key frame enters the ring (saveKeyFramesAndFactor :546-555 keeps the down-sampled clouds of the
// frame in the sensor frame; extractSurroundingKeyFrames :216-218,:240-242 transforms them by the f32 key pose,
// laserMapping.h:164-177).  The raw clouds go to ring entry f % KR (what the host pose graph reads back and what a corrected
// key pose is applied to); the transformed clouds go to kf_tmp_* (corner | surf followed by outlier) and are sorted by voxel key
// into the ring by the next VoxelGrid round (LI_KF_PENDING).
//   only_ring < 0: the slots whose LI_KF_ADDED is set store their current scan
//   only_ring >= 0: re-transform ring entry `only_ring` of every slot of the launch from its raw clouds (set_keypose / add_keyframe)
__global__ void __launch_bounds__(LM_BLOCK) lm_store_kf(DevCtx d, LmCtx L, int only_ring) {
  const int slot = blockIdx.z + d.slot0, kind = blockIdx.y;
  int* li = lip(L, slot);
  if (only_ring < 0 && !li[LI_KF_ADDED]) return;
  const int ring = only_ring < 0 ? (li[LI_NKF] - 1) % L.KR : only_ring;
  const size_t rs = (size_t)slot * L.KR + ring;
  float m[3][4];
  keypose_matrix(L.kf_pose + rs * 8, m);
  float4* raw = kind == 0 ? L.kf_raw_c + rs * L.kf_cap_c : (kind == 1 ? L.kf_raw_s + rs * L.kf_cap_s : L.kf_raw_o + rs * L.kf_cap_o);
  const float4* cur = kind == 0 ? L.cur_corner_ds + (size_t)slot * L.kf_cap_c : (kind == 1 ? L.cur_surf_ds + (size_t)slot * L.kf_cap_s : L.cur_outl_ds + (size_t)slot * L.kf_cap_o);
  const int cap = kind == 0 ? L.kf_cap_c : (kind == 1 ? L.kf_cap_s : L.kf_cap_o);
  const int* kc = L.kf_cnt + rs * 4;
  // counts of the three clouds of this key frame (new frame: the current scan's; re-transform: the stored ones)
  const int n_c = only_ring < 0 ? min(li[LI_NCUR_C], L.kf_cap_c) : kc[0];
  const int n_s = only_ring < 0 ? min(li[LI_NCUR_S], L.kf_cap_s) : kc[1];
  const int n_o = only_ring < 0 ? min(li[LI_NCUR_O], L.kf_cap_o) : kc[2];
  const int n = min(kind == 0 ? n_c : (kind == 1 ? n_s : n_o), cap);
  float4* dst = kind == 0 ? L.kf_tmp_c + (size_t)slot * L.kf_cap_c : L.kf_tmp_s + (size_t)slot * L.total_cap + (kind == 1 ? 0 : n_s);
  for (int i = blockIdx.x * LM_BLOCK + threadIdx.x; i < n; i += gridDim.x * LM_BLOCK) {
    float4 p;
    if (only_ring < 0) { p = cur[i]; raw[i] = p; } else { p = raw[i]; }
    dst[i] = kf_transform(m, p);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (only_ring < 0) L.kf_cnt[rs * 4 + kind] = n;
    if (kind == 0) { li[LI_TMPN_C] = n_c; li[LI_TMPN_S] = n_s + n_o; li[LI_KF_PEND_RING] = ring; li[LI_KF_PENDING] = 1; }
  }
}

// one thread: correctPoses :579-580 on map -> odom with the 3x4 [R | c] of the loop-closure correction
__global__ void lm_apply_correction(DevCtx d, LmCtx L, int slot, const double* rc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double* ld = ldp(L, slot);
  double R[9], M[9], t[3];
  dq_to_mat(ldq(ld + LD_Q_M2O), R);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i * 3 + j] = rc[i * 4 + 0] * R[0 * 3 + j] + rc[i * 4 + 1] * R[1 * 3 + j] + rc[i * 4 + 2] * R[2 * 3 + j];
  for (int i = 0; i < 3; ++i) t[i] = rc[i * 4 + 0] * ld[LD_T_M2O + 0] + rc[i * 4 + 1] * ld[LD_T_M2O + 1] + rc[i * 4 + 2] * ld[LD_T_M2O + 2] + rc[i * 4 + 3];
  stq(ld + LD_Q_M2O, dq_from_mat(M));
  for (int i = 0; i < 3; ++i) ld[LD_T_M2O + i] = t[i];
}


// ---------------------------------------------------------------------------------------------------------------------
// One registration sharded over the ranks of a communicator (BASELINE config 5): lm_knn / lm_fit above only fill the
// rows of this rank's query slice; the solver of lm_solve is cut at its evaluations so that the 28 normal-equation scalars (and,
// with the first evaluation, the two correspondence counts) can be summed over the ranks between the kernels:
//   lm_shard_pack            registration guard, packed rows of this rank, row counts
//   lm_shard_eval(which)     residual + Jacobian rows of this rank at params_ (which = 0) or at the candidate of the step in flight
//                            (which = 1) -> shard_part[slot][0..27] (+ counts in [28], [29])
//   ncclAllReduce(f64, sum)  over shard_part of all slots of the launch, on the same stream (host, lm_host.hip)
//   lm_shard_step(first)     the trust-region control of lm_solve on the summed scalars: identical on every rank, so all ranks
//                            take the same step; state (LmState) lives in HBM between the kernels
// The host enqueues the worst-case sequence (lm_outer_iters x (1 + lm_max_iters) evaluations); kernels of a finished solve return at once.
// Same evaluation code, same row order and same reduction as lm_solve: with one rank the result is bit-identical to it.
static_assert(sizeof(LmState) <= 64 * sizeof(double), "LmHost allocates 64 doubles per slot for the sharded solve's state");
DEV_INLINE void lm_shard_pack_dev(const DevCtx& d, const LmCtx& L, int slot) {
  int* li = lip(L, slot);
  int* ctl = L.shard_ctl + (size_t)slot * 8;
  double* part = L.shard_part + (size_t)slot * 32;
  if (threadIdx.x < 32) part[threadIdx.x] = 0.0;
  if (threadIdx.x == 0) { ctl[0] = 0; ctl[1] = LM_STOP; ctl[2] = 1; ctl[3] = 0; ctl[4] = 1; ctl[5] = 0; }
  if (!li[LI_RUN]) return;
  const alego_params& P = d.P;
  if (li[LI_NCUR_C] < P.lm_min_corner || li[LI_NTOTAL] < P.lm_min_surf || li[LI_KDS_C] < P.lm_min_map_corner || li[LI_NKF] == 0) {
    if (threadIdx.x == 0) { li[LI_FLAGS] |= 16; li[LI_NCC] = 0; li[LI_NSC] = 0; li[LI_SUM0] = 0; li[LI_SUM1] = 0; }
    return;
  }
  __shared__ int s_cnt[2][LM_SOLVE_T / 64];
  LmRows W;
  lm_pack_rows<LM_SOLVE_T>(L, slot, li[LI_NCUR_C], li[LI_NTOTAL_DS], nullptr, 0, s_cnt, W);   // every row to HBM, in lm_solve's order
  if (threadIdx.x == 0) {
    ctl[0] = W.Rc + W.Rs; ctl[5] = W.Rc; ctl[1] = LM_EVAL; ctl[2] = 0; ctl[4] = 0;
    part[28] = (double)W.Rc; part[29] = (double)W.Rs;   // summed over the ranks with the first evaluation
    li[LI_OPTIMIZED] = 1;
  }
}
__global__ void __launch_bounds__(LM_SOLVE_T) lm_shard_pack(DevCtx d, LmCtx L) { lm_shard_pack_dev(d, L, blockIdx.x + d.slot0); }

DEV_INLINE void lm_shard_step_dev(const DevCtx& d, const LmCtx& L, int slot, int first);
DEV_INLINE void lm_shard_next_outer_dev(const LmCtx& L, int slot);
// pre (round 6: what used to be separate one-thread-per-slot launches between two evaluations runs at the head of the next evaluation's own workgroup, so an
// evaluation is ONE launch around its all-reduce instead of two or three):  LM_PRE_PACK = lm_shard_pack (the first evaluation of a frame), LM_PRE_STEP_FIRST /
// LM_PRE_STEP = lm_shard_step on the sums the all-reduce has just delivered, LM_PRE_STEP_NEXT = that step followed by lm_shard_next_outer (the first evaluation of
// a further outer iteration).  A workgroup is one slot: its thread 0 does exactly what the slot's thread of the separate launch did.
enum { LM_PRE_NONE = 0, LM_PRE_PACK, LM_PRE_STEP_FIRST, LM_PRE_STEP, LM_PRE_STEP_NEXT };
__global__ void __launch_bounds__(LM_SOLVE_T) lm_shard_eval(DevCtx d, LmCtx L, int which, int pre) {
  const int slot = blockIdx.x + d.slot0;
  if (pre == LM_PRE_PACK) lm_shard_pack_dev(d, L, slot);
  else if (pre != LM_PRE_NONE && threadIdx.x == 0) {
    lm_shard_step_dev(d, L, slot, pre == LM_PRE_STEP_FIRST ? 1 : 0);
    if (pre == LM_PRE_STEP_NEXT) lm_shard_next_outer_dev(L, slot);
  }
  if (pre != LM_PRE_NONE) { __threadfence_block(); __syncthreads(); }   // (the control words, the candidate and the packed rows are read back by the whole workgroup)
  const int* ctl = L.shard_ctl + (size_t)slot * 8;
  if (ctl[2] || ctl[4] || ctl[1] != LM_EVAL) return;   // solve finished / guard failed: the all-reduce still runs, on stale partials nobody reads
  extern __shared__ __attribute__((aligned(16))) unsigned char lm_smem[];
  double* s_acc = reinterpret_cast<double*>(lm_smem);
  __shared__ double s_out[28], s_trig[12];
  const LmState* S = reinterpret_cast<const LmState*>(L.shard_state) + slot;
  const double* ld = ldp(L, slot);
  double x[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) x[k] = which == 0 ? ld[LD_PARAMS + k] : S->cand[k];
  LmRows W{};
  W.Rc = ctl[5]; W.Rs = ctl[0] - ctl[5];
  W.crows = L.crows + (size_t)slot * L.qcap * 10;
  double acc[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = 0;
  const PoseTerms T = pose_terms_coop(x, s_trig);
  lm_eval_rows<LM_SOLVE_T>(W, T, d.P.huber_delta, acc);
  block_reduce28_oct<LM_SOLVE_T>(acc, s_acc, s_out);
  double* part = L.shard_part + (size_t)slot * 32;
  if (threadIdx.x < 28) part[threadIdx.x] = s_out[threadIdx.x];
  if (threadIdx.x == 0 && which == 1) { part[28] = 0.0; part[29] = 0.0; }
}

// one thread per slot.  first: the evaluation just summed was the one at params_ (start of an outer iteration)
DEV_INLINE void lm_shard_step_dev(const DevCtx& d, const LmCtx& L, int slot, int first) {
  int* ctl = L.shard_ctl + (size_t)slot * 8;
  if (ctl[4]) return;
  int* li = lip(L, slot);
  double* ld = ldp(L, slot);
  LmState& S = reinterpret_cast<LmState*>(L.shard_state)[slot];
  const double* red = L.shard_part + (size_t)slot * 32;   // summed over the ranks
  const int outer = ctl[3];
  if (first) {
    if (outer == 0) { li[LI_NCC] = (int)(red[28] + 0.5); li[LI_NSC] = (int)(red[29] + 0.5); }
    if (li[LI_NCC] + li[LI_NSC] == 0) {   // ceres::Solve on an empty problem is a no-op
      if (outer < 2) { li[LI_SUM0 + outer] = 4 << 16; for (int k = 0; k < 6; ++k) ld[LD_PARAMS_IT + outer * 6 + k] = ld[LD_PARAMS + k]; }
      ctl[2] = 1; ctl[1] = LM_STOP;
      return;
    }
    double x0[6];
    for (int k = 0; k < 6; ++k) x0[k] = ld[LD_PARAMS + k];
    lm_begin(S, x0, red, d.P.lm_max_iters);
    ctl[2] = 0;
  } else {
    if (ctl[2] || ctl[1] != LM_EVAL) return;
    if (lm_consume(S, red) == LM_STOP) ctl[2] = 1;
  }
  int act = LM_STOP;
  if (!ctl[2]) { do { act = lm_propose(S); } while (act == LM_AGAIN); }
  ctl[1] = act;
  if (act == LM_STOP) {   // this outer iteration's ceres::Solve has returned
    ctl[2] = 1;
    for (int k = 0; k < 6; ++k) ld[LD_PARAMS + k] = S.x[k];
    if (outer < 2) {
      for (int k = 0; k < 6; ++k) ld[LD_PARAMS_IT + outer * 6 + k] = S.x[k];
      ld[LD_COSTS + outer * 2] = S.initial_cost; ld[LD_COSTS + outer * 2 + 1] = S.x_cost;
      li[LI_SUM0 + outer] = S.iter | (S.successful << 8) | (S.termination << 16);
    }
  }
}

__global__ void lm_shard_step(DevCtx d, LmCtx L, int first) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < d.n_launch) lm_shard_step_dev(d, L, s + d.slot0, first);
}
// between two outer iterations: the next ceres::Solve starts from the params_ the last one left (:360)
DEV_INLINE void lm_shard_next_outer_dev(const LmCtx& L, int slot) {
  int* ctl = L.shard_ctl + (size_t)slot * 8;
  if (ctl[4]) return;
  ctl[3] += 1; ctl[2] = 0; ctl[1] = LM_EVAL;
}

size_t lm_solve_row_bytes_max() { return LM_SOLVE_DYN_BYTES - LM_SOLVE_RED_BYTES; }
size_t lm_solve_row_bytes_default() { return LM_SOLVE_ROW_BYTES_DEFAULT; }
int lm_configure() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LM_SOLVE_DYN_BYTES) == hipSuccess ? 0 : -1;
}

// ---- launchers ---------------------------------------------------------------------
void launch_lm_prepare(const DevCtx& d, const LmCtx& L, int stage, int run_hint, int par, hipStream_t st) {
  // a frame that is not mapped only needs the odometry hand-over (one thread per slot); so does every frame whose clouds lm_stage copied
  const bool small = run_hint == 0 || stage == 2;
  const dim3 grid = small ? dim3(1, 1, d.n_launch) : dim3(8, 3, d.n_launch);
  ALEGO_LAUNCH(lm_prepare, grid, dim3(small ? 64 : LM_BLOCK), 0, st, d, L, stage, run_hint, par);
}
void launch_lm_stage(const DevCtx& d, const LmCtx& L, int run_hint, int par, hipStream_t st) {
  const dim3 grid = run_hint == 0 ? dim3(1, 1, d.n_launch) : dim3(8, 3, d.n_launch);
  ALEGO_LAUNCH(lm_stage, grid, dim3(run_hint == 0 ? 64 : LM_BLOCK), 0, st, d, L, run_hint, par);
}
void launch_lm_concat(const DevCtx& d, const LmCtx& L, hipStream_t st) {
  ALEGO_LAUNCH(lm_concat, dim3(2, L.K, d.n_launch), dim3(LM_BLOCK), 0, st, d, L);
}
void launch_lm_total(const DevCtx& d, const LmCtx& L, hipStream_t st) {
  ALEGO_LAUNCH(lm_total, dim3(8, d.n_launch), dim3(LM_BLOCK), 0, st, d, L);
}
void launch_lm_grid(const DevCtx& d, const LmCtx& L, hipStream_t st) {
  ALEGO_LAUNCH(lm_grid_build, dim3(2, d.n_launch), dim3(LM_BLOCK), 0, st, d, L);
}
// allreduce(buffer, count of doubles, stream): sums shard_part over the ranks in place (RCCL, lm_host.hip); nullptr = not sharded
// Returns the first non-zero allreduce code: a failed collective (communicator aborted, peer gone) would leave the ranks stepping on
// un-summed partials and hanging in the next one, so nothing further of this frame is enqueued and the caller reports the error.
int launch_lm_register(const DevCtx& d, const LmCtx& L, hipStream_t st, int (*allreduce)(void*, double*, size_t, hipStream_t), void* ar_ctx) {
  ALEGO_LAUNCH(lm_knn, dim3(lm_reg_grid(d, LM_ASSOC_GX)), dim3(128), 0, st, d, L);
  ALEGO_LAUNCH(lm_fit, dim3(lm_reg_grid(d, LM_FIT_GX)), dim3(128), 0, st, d, L);
  if (allreduce) {
    double* part = L.shard_part + (size_t)d.slot0 * 32;
    const size_t cnt = (size_t)d.n_launch * 32;
    const dim3 g1((d.n_launch + 63) / 64), b1(64);
    // per evaluation ONE launch + the all-reduce (round 6; before: pack | eval, all-reduce, step | next_outer as separate launches — 86 launches per mapping frame,
    // now 43): the step on the sums of evaluation k runs at the head of evaluation k + 1's workgroups, a last stand-alone step closes the frame
    for (int outer = 0; outer < d.P.lm_outer_iters; ++outer) {
      ALEGO_LAUNCH(lm_shard_eval, dim3(d.n_launch), dim3(LM_SOLVE_T), LM_SOLVE_RED_BYTES, st, d, L, 0, outer == 0 ? (int)LM_PRE_PACK : (int)LM_PRE_STEP_NEXT);
      if (int rc = allreduce(ar_ctx, part, cnt, st)) return rc;
      for (int it = 0; it < d.P.lm_max_iters; ++it) {
        ALEGO_LAUNCH(lm_shard_eval, dim3(d.n_launch), dim3(LM_SOLVE_T), LM_SOLVE_RED_BYTES, st, d, L, 1, it == 0 ? (int)LM_PRE_STEP_FIRST : (int)LM_PRE_STEP);
        if (int rc = allreduce(ar_ctx, part, cnt, st)) return rc;
      }
    }
    ALEGO_LAUNCH(lm_shard_step, g1, b1, 0, st, d, L, d.P.lm_max_iters == 0 ? 1 : 0);
  } else {
  ALEGO_LAUNCH(lm_solve, dim3(d.n_launch), dim3(LM_SOLVE_T), LM_SOLVE_RED_BYTES + L.solve_row_bytes, st, d, L);
  }
  ALEGO_LAUNCH(lm_finish, dim3((d.n_launch + 63) / 64), dim3(64), 0, st, d, L);
  ALEGO_LAUNCH(lm_store_kf, dim3(8, 3, d.n_launch), dim3(LM_BLOCK), 0, st, d, L, -1);
  return 0;
}
void launch_lm_retransform(const DevCtx& d, const LmCtx& L, int ring, hipStream_t st) {
  ALEGO_LAUNCH(lm_store_kf, dim3(8, 3, d.n_launch), dim3(LM_BLOCK), 0, st, d, L, ring);
}
void launch_lm_apply_correction(const DevCtx& d, const LmCtx& L, int slot, const double* rc_dev, hipStream_t st) {
  ALEGO_LAUNCH(lm_apply_correction, dim3(1), dim3(64), 0, st, d, L, slot, rc_dev);
}
