// kernels_lo.hip — scan-to-scan LaserOdometry on gfx950 (replaces src/laserOdometry.cpp:328-534).
//
//   lo_assoc   a11-a14: 16 lanes per query feature: transformToStart, exact f32 1-NN over the
//              previous scan's feature cloud (replaces pcl::KdTreeFLANN), then the +-2-ring walks
//              evaluated as a lexicographic arg-min over the ring interval; both pruned with the
//              bounding boxes of 32 consecutive targets
//   lo_solve   a15-a17: one workgroup per stream runs a whole ceres::Solve (trust-region LM, Huber
//              corrector, Jacobi scaling) on-chip: residual/Jacobian evaluation in fp64, wavefront
//              shuffle + LDS reduction of the 21+6+1 normal-equation scalars in a fixed order, 6x6
//              Cholesky by one lane, step acceptance, and (second call) the pose integration
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "dev_cost.h"
#include "lm_ctx.h"
#include "prof.h"

#define LO_BLOCK 256
#ifndef LO_SOLVE_BLOCK
#define LO_SOLVE_BLOCK 64    // threads of the per-stream solver workgroup (~300 rows): 64 measured 127 us per launch, 128: 133, 256: 199
#endif

// transformToStart (laserOdometry.cpp:728-740) = R(params_) * p + t.  R depends on the pose only, not on the point:
// lo_solve caches it next to the pose (three fp64 sin/cos pairs are a thousand instructions), lo_assoc applies it.
DEV_INLINE void pose_rotation(const double* p, double R[9]) {
  const DQuat q = dq_zyx(p[5], p[4], p[3]);
  dq_to_mat(q, R);
}
DEV_INLINE void store_pose_rotation(double* st) {
  double R[9];
  pose_rotation(st + LS_PARAMS, R);
#pragma unroll
  for (int k = 0; k < 9; ++k) st[LS_ROT + k] = R[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) st[LS_ROT_P + k] = st[LS_PARAMS + k];
}
DEV_INLINE void transform_to_start(const double* R, const double* p, const float4& pi, float out[3]) {
  const double x = pi.x, y = pi.y, z = pi.z;
  out[0] = (float)(R[0] * x + R[1] * y + R[2] * z + p[0]);
  out[1] = (float)(R[3] * x + R[4] * y + R[5] * z + p[1]);
  out[2] = (float)(R[6] * x + R[7] * y + R[8] * z + p[2]);
}

struct WalkBest { double dist; int rank; int idx; };
DEV_INLINE void walk_consider(WalkBest& b, double dist, int rank, int idx) {
  if (dist < b.dist || (dist == b.dist && rank < b.rank)) { b.dist = dist; b.rank = rank; b.idx = idx; }
}
// lexicographic (distance, visiting rank) arg-min over the wavefront: non-negative doubles order like their bit
// patterns, so this is a u64 min, then a u32 min of the rank among the lanes that hold the winning distance
DEV_INLINE WalkBest walk_reduce(WalkBest b) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(b.dist);
  const unsigned long long mn = wave_min_u64(bits);
  const uint32_t rk = wave_min_u32(bits == mn ? (uint32_t)b.rank : 0xFFFFFFFFu);
  const uint32_t ix = wave_min_u32((bits == mn && (uint32_t)b.rank == rk) ? (uint32_t)b.idx : 0xFFFFFFFFu);
  WalkBest r;
  r.dist = __longlong_as_double((long long)mn); r.rank = (int)rk; r.idx = (int)ix;  // idx -1 (0xFFFFFFFF) = none
  return r;
}

// kind 0: flat -> surf_last (less_flat of the previous scan); kind 1: sharp -> corner_last (less_sharp).
// One DPP row (16 lanes) per query, four queries per wavefront in lock-step, LO_QPB per workgroup: the kernel is
// instruction-issue bound, and every reduction below is four row-local DPP steps shared by the four queries.
//
// Both searches of a query are exact and pruned with bounding boxes of up to LO_CH consecutive targets OF ONE RING (written by feature
// extraction next to the clouds: box c covers the targets [start, start + len), start / len carried in the .w fields of its two corners;
// ring r owns the boxes [ring_boff[r], ring_boff[r + 1]) — a ring's feature-extraction workgroup writes its own boxes without waiting for
// any other ring, and a ring window is a contiguous range of boxes):
// a box is skipped when its lower bound exceeds the best distance found so far.  The lower bound is evaluated with
// the same operations, in the same order and precision as the distance itself (clamped per-axis difference, squares,
// left-to-right sum), and IEEE rounding is monotone, so bound <= distance of every point of the box: no candidate
// that could win or tie is ever skipped.
//   1-NN      (flann::L2_Simple<float>, ties -> lowest index): seed = the box with the smallest bound and its
//             neighbour, then every box whose bound <= seed distance
//   ring walk (:344-373,:433-475): the reference walks up and down from the closest point inside +-2.5 rings and keeps
//             the minimum of the double-precision distance per class, first visited on ties; here a lexicographic
//             (distance, visiting rank) arg-min over the surviving boxes of the ring interval
#define LO_QPB_OF(B) ((B) / 16)
#ifndef LO_NB
#define LO_NB 2       // surviving boxes evaluated per turn (2 x LO_NB loads in flight per lane); 3: same, 4: slower
#endif
#ifndef LO_BOX_LDS
#define LO_BOX_LDS_BIG 512   // sensors with more than 16 rings (64 x 2048: ~430 less_flat boxes)
#define LO_BOX_LDS 160   // boxes (of LO_CH targets) staged in LDS: 5 KB (16 x 1800 has ~135 less_flat boxes; larger feature sets read the boxes from the L2).
                         // 512 boxes = 16 KB made eight of these workgroups fill a CU's LDS: 351 k -> 354 k scans/s with 160
#endif
DEV_INLINE WalkBest walk_reduce_row(WalkBest b) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(b.dist);
  const unsigned long long mn = row16_min_u64(bits);
  const uint32_t rk = row16_min_u32(bits == mn ? (uint32_t)b.rank : 0xFFFFFFFFu);
  const uint32_t ix = row16_min_u32((bits == mn && (uint32_t)b.rank == rk) ? (uint32_t)b.idx : 0xFFFFFFFFu);
  WalkBest r;
  r.dist = __longlong_as_double((long long)mn); r.rank = (int)rk; r.idx = (int)ix;  // idx -1 (0xFFFFFFFF) = none
  return r;
}
// the 16 bits of a wave-wide ballot that belong to this lane's row
DEV_INLINE uint32_t row_bits(unsigned long long ballot, int lane) { return (uint32_t)(ballot >> (lane & 48)) & 0xFFFFu; }

#ifdef ALEGO_TIMING
__device__ long long la_times[12];
extern "C" void alego_la_times(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(la_times), sizeof(long long) * 12); }
__device__ unsigned long long la_cnt[8];   // [kind * 4 + ...]: wavefront sweeps, walk_eval turns of the pass, surviving boxes (summed over rows), window boxes (summed over rows)
extern "C" void alego_la_counts(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(la_cnt), sizeof(unsigned long long) * 8); }
#define LA_COUNT(i, v) do { atomicAdd(&la_cnt[kind * 4 + (i)], (unsigned long long)(v)); } while (0)
#define LA_TICK(k) do { if (threadIdx.x == 0 && slot == d.slot0 && qb0 == 0 && kind == 0) la_times[k] = wall_clock64(); } while (0)
#define LA_TICK0 LA_TICK(0)
#elif defined(LA_STOP_AFTER)
// development (instruction counts per phase, tools/la_phase_counts.sh): a sweep ends after phase LA_STOP_AFTER (its queries get no correspondence)
#define LA_TICK(k) if ((k) == LA_STOP_AFTER) continue
#define LA_TICK0
#define LA_COUNT(i, v)
#else
#define LA_TICK(k)
#define LA_TICK0
#define LA_COUNT(i, v)
#endif
// ---- the target grid (round 4) ----------------------------------------------------------------------------------------------------
// pcl::KdTreeFLANN::nearestKSearch(sel, 1, ...) (:341,:430) is an EXACT nearest neighbour; the boxes answer it with ~12 boxes of 32 targets per
// query whatever the bound (a box is a 43-degree arc of one ring: every ring has one that contains the query's (x, y)).  The neighbour a
// LiDAR feature had a tenth of a second ago is almost always within a metre: lo_grid_build sorts a scan's less_flat / less_sharp clouds by the
// cells of a 2-D grid over (x, y) — cell size 1 m, doubled until the cloud's bounding box fits LO_GC cells (2 m for a 100 m scene) — as soon as feature extraction has
// written them, and lo_assoc looks at the 3 x 3 cells around the query first: every target closer than a cell size lies there, so a best
// candidate closer than that IS the nearest neighbour (ties: lowest index, as before); only a query without one takes the box search.
// One workgroup per (stream, cloud): counts packed two cells per LDS word (a cloud has < 65536 points: otherwise no grid, gx = 0), exclusive
// scan, scatter; the order inside a cell is whatever the atomics give and does not matter (the key is (distance, index)).
#define LG_T 256
#ifndef LG_KEEP
#define LG_KEEP 20
#endif
// LG_KEEP:  // 5120 points: every cloud of a 16-ring sensor
__global__ void __launch_bounds__(LG_T) lo_grid_build(DevCtx d) {
  const int slot = blockIdx.x + d.slot0, kind = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t fb = (size_t)slot * 2 + cur_in_flight(d, slot);
  const int tk = kind == 0 ? F_LFLAT : F_LSHARP;
  const int n = (kind == 0 && d.scal[slot * SC_COUNT + SC_FE_ERR]) ? 0 : d.feat_cnt[fb * 4 + tk];   // (a slot whose fe_ring_out gave up has no less_flat cloud: dev_common.h)
  const float4* pts = d.feat[tk] + fb * d.fcap[tk];
  float4* cp = d.lo_cpts[kind] + fb * d.fcap[tk];
  unsigned short* cell = d.lo_cell + (fb * 2 + kind) * (LO_GC + 2);
  float* geom = d.lo_geom + (fb * 2 + kind) * 8;
  __shared__ unsigned s_cnt[LO_GC / 2];
  __shared__ float s_red[4][LG_T / 64];
  __shared__ int s_tot[LG_T / 64];
  if (n <= 0 || n > 65535) { if (tid == 0) geom[4] = __int_as_float(0); return; }
  // the cloud's (x, y) extent from its bounding boxes (feature extraction wrote them next to the cloud: ~150 boxes instead of every point once more)
  float mn[2] = {3.402823466e+38f, 3.402823466e+38f}, mx[2] = {-3.402823466e+38f, -3.402823466e+38f};
  {
    const float4* bx = d.lo_box + (fb * 2 + kind) * d.lo_box_cap * 2;
    const int nbox = d.ring_boff[(fb * 2 + (kind == 0 ? 1 : 0)) * (d.NS + 1) + d.NS];
    for (int b = tid; b < nbox; b += LG_T) { const float4 l = bx[2 * b], u = bx[2 * b + 1]; mn[0] = fminf(mn[0], l.x); mn[1] = fminf(mn[1], l.y); mx[0] = fmaxf(mx[0], u.x); mx[1] = fmaxf(mx[1], u.y); }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64)); }
    if (lane == 0) { s_red[a][wave] = mn[a]; s_red[2 + a][wave] = mx[a]; }
  }
  for (int w = tid; w < LO_GC / 2; w += LG_T) s_cnt[w] = 0u;
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    mn[a] = s_red[a][0]; mx[a] = s_red[2 + a][0];
#pragma unroll
    for (int w = 1; w < LG_T / 64; ++w) { mn[a] = fminf(mn[a], s_red[a][w]); mx[a] = fmaxf(mx[a], s_red[2 + a][w]); }
  }
  float csz = 1.0f;
  int gx = 0, gy = 0;
  bool ok = mx[0] - mn[0] < 1e6f && mx[1] - mn[1] < 1e6f;   // (also false for NaN / infinite extents)
  for (int it = 0; ok && it < 24; ++it) {
    gx = (int)floorf((mx[0] - mn[0]) / csz) + 2; gy = (int)floorf((mx[1] - mn[1]) / csz) + 2;
    if ((long long)gx * gy <= LO_GC) break;
    csz *= 2.0f;
  }
  ok = ok && (long long)gx * gy <= LO_GC;
  if (!ok) { if (tid == 0) geom[4] = __int_as_float(0); return; }
  const float inv = 1.0f / csz;   // (csz is a power of two: exact)
  const int ncell = gx * gy;
  auto cell_of = [&](const float4& p) -> int {
    const int ix = min(max((int)floorf((p.x - mn[0]) * inv), 0), gx - 1), iy = min(max((int)floorf((p.y - mn[1]) * inv), 0), gy - 1);
    return ix + gx * iy;
  };
  // the points stay in registers between the count and the scatter (LG_KEEP per thread; a larger cloud reads the rest again)
  float4 keep[LG_KEEP];
#pragma unroll
  for (int u = 0; u < LG_KEEP; ++u) {
    const int i = tid + u * LG_T;
    keep[u] = pts[min(i, n - 1)];
    if (i < n) { const int c = cell_of(keep[u]); atomicAdd(&s_cnt[c >> 1], 1u << ((c & 1) * 16)); }
  }
  for (int i = tid + LG_KEEP * LG_T; i < n; i += LG_T) { const int c = cell_of(pts[i]); atomicAdd(&s_cnt[c >> 1], 1u << ((c & 1) * 16)); }
  __syncthreads();
  // exclusive scan: thread t owns the cells [64 t, 64 t + 64) (32 words)
  constexpr int WPT = LO_GC / 2 / LG_T;
  int tsum = 0;
#pragma unroll 4
  for (int k = 0; k < WPT; ++k) { const unsigned w = s_cnt[tid * WPT + k]; tsum += (int)(w & 0xFFFFu) + (int)(w >> 16); }
  int incl = tsum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
  if (lane == 63) s_tot[wave] = incl;
  __syncthreads();
  int run = incl - tsum;
#pragma unroll
  for (int w = 0; w < LG_T / 64; ++w) if (w < wave) run += s_tot[w];
  for (int k = 0; k < WPT; ++k) {
    const int c0 = 2 * (tid * WPT + k);
    const unsigned w = s_cnt[tid * WPT + k];
    const int a = run, b = run + (int)(w & 0xFFFFu);
    run = b + (int)(w >> 16);
    s_cnt[tid * WPT + k] = (unsigned)a | ((unsigned)b << 16);   // the counts become the cells' cursors
    if (c0 <= ncell) cell[c0] = (unsigned short)a;
    if (c0 + 1 <= ncell) cell[c0 + 1] = (unsigned short)b;
  }
  if (tid == LG_T - 1 && ncell == LO_GC) cell[ncell] = (unsigned short)n;   // (a full grid: the end of the last cell is beyond the words the scan owns)
  __syncthreads();
  auto place = [&](int i, const float4& p) {
    const int c = cell_of(p);
    const unsigned old = atomicAdd(&s_cnt[c >> 1], 1u << ((c & 1) * 16));
    cp[(old >> ((c & 1) * 16)) & 0xFFFFu] = make_float4(p.x, p.y, p.z, __int_as_float(i));
  };
#pragma unroll
  for (int u = 0; u < LG_KEEP; ++u) { const int i = tid + u * LG_T; if (i < n) place(i, keep[u]); }
  for (int i = tid + LG_KEEP * LG_T; i < n; i += LG_T) place(i, pts[i]);
  if (tid == 0) {
    geom[0] = mn[0]; geom[1] = mn[1]; geom[2] = inv; geom[3] = 0.999f * csz * csz;
    geom[4] = __int_as_float(gx); geom[5] = __int_as_float(gy);
  }
}
// (ALEGO_LO_GRID toggled off: the buffer's clouds have just been rewritten, so the grid a later scan could find there — if the switch goes back on — is marked absent)
__global__ void lo_grid_off(DevCtx d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.n_launch * 2) return;
  const int slot = i / 2 + d.slot0, kind = i & 1;
  d.lo_geom[(((size_t)slot * 2 + cur_in_flight(d, slot)) * 2 + kind) * 8 + 4] = __int_as_float(0);
}
void launch_lo_grid(const DevCtx& d, hipStream_t st) {
  if (!d.opt_lo_grid) { ALEGO_LAUNCH(lo_grid_off, dim3((d.n_launch * 2 + 63) / 64), dim3(64), 0, st, d); return; }
  ALEGO_LAUNCH(lo_grid_build, dim3(d.n_launch, 2), dim3(LG_T), 0, st, d);
}

// kind (compile-time): the corner association has one class of second points, the surf one two.  BLKA threads per workgroup, BLKA / 16
// queries at a time; workgroup qb0 of qbn of the stream takes the query blocks qb0, qb0 + qbn, ...  (A device function: the batch path
// launches it as lo_assoc, one stream's chain kernel lo_chain calls it between its grid barriers.)
template <int kind, int BLKA, int BOXCAP = LO_BOX_LDS>
DEV_INLINE void lo_assoc_body(const DevCtx& d, int box_lds_max, int slot, int qb0, int qbn) {
  static_assert(LO_CH % 16 == 0, "a box is evaluated as LO_CH / 16 targets per lane of a 16-lane row");
  constexpr int TPL = LO_CH / 16;
  constexpr int LO_QPB = BLKA / 16;
  const size_t fc = fidx_cur(d, slot), fl = fidx_last(d, slot);
  const int* sc = d.scal + slot * SC_COUNT;
  if (!sc[SC_LO_INIT]) return;
  const int lane = lane_id(), l16 = lane & 15;
  const int qk = kind == 0 ? F_FLAT : F_SHARP, tk = kind == 0 ? F_LFLAT : F_LSHARP;
  const int nq = d.feat_cnt[fc * 4 + qk];
  if (qb0 * LO_QPB >= nq) return;
  const int nt = (kind == 0 && sc[SC_FE_ERR]) ? 0 : d.feat_cnt[fl * 4 + tk];   // (fe_ring_out gave up on this slot: no less_flat targets, dev_common.h)
  const float4* tg = d.feat[tk] + fl * d.fcap[tk];
  const float4* bx = d.lo_box + (fl * 2 + kind) * d.lo_box_cap * 2;
  const int* roff = d.ring_off + (fl * 2 + (kind == 0 ? 1 : 0)) * (d.NS + 1);
  const int* boff = d.ring_boff + (fl * 2 + (kind == 0 ? 1 : 0)) * (d.NS + 1);
  const int nch = nt > 0 ? boff[d.NS] : 0;
  const double* st = d.lo_state + (size_t)slot * LO_STATE_N;
  LA_TICK0;
  __shared__ float s_sel[LO_QPB][4];
  __shared__ double s_pose[12];
  __shared__ float4 s_box[2 * BOXCAP];   // the boxes are read by every query of the workgroup: LDS when they fit
  __shared__ int s_roff[65], s_boff[65];
  __shared__ float s_geom[8];
  const bool box_lds = nch <= box_lds_max;   // (<= LO_BOX_LDS; the parity tests also run with 0 = boxes straight from HBM)
  if (box_lds) for (int i = threadIdx.x; i < 2 * nch; i += BLKA) s_box[i] = bx[i];
  for (int i = threadIdx.x; i <= d.NS; i += BLKA) { s_roff[i] = roff[i]; s_boff[i] = boff[i]; }
  if (threadIdx.x < 8) s_geom[threadIdx.x] = d.opt_lo_grid ? d.lo_geom[(fl * 2 + kind) * 8 + threadIdx.x] : 0.f;
  const float4* gpts = d.lo_cpts[kind] + fl * d.fcap[tk];
  const unsigned short* gcell = d.lo_cell + (fl * 2 + kind) * (LO_GC + 2);
  if (threadIdx.x == 0) {
    bool same = true;   // NaN (nothing cached yet) or a pose written by alego_set_lo_params compares unequal
#pragma unroll
    for (int k = 0; k < 6; ++k) same = same && st[LS_ROT_P + k] == st[LS_PARAMS + k];
    double R[9];
    if (same) {
#pragma unroll
      for (int k = 0; k < 9; ++k) R[k] = st[LS_ROT + k];
    } else {
      pose_rotation(st + LS_PARAMS, R);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) s_pose[k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) s_pose[9 + k] = st[LS_PARAMS + k];
  }
  // the launch covers the typical query count in one sweep; larger feature sets take further sweeps
  for (int qb = qb0; qb * LO_QPB < nq; qb += qbn) {
  __syncthreads();
  if (threadIdx.x < LO_QPB) {  // transformToStart once per query, shared through LDS
    const int q = min((int)(qb * LO_QPB + threadIdx.x), nq - 1);
    float o[3];
    transform_to_start(s_pose, s_pose + 9, d.feat[qk][fc * d.fcap[qk] + q], o);
    s_sel[threadIdx.x][0] = o[0]; s_sel[threadIdx.x][1] = o[1]; s_sel[threadIdx.x][2] = o[2];
  }
  __syncthreads();
  LA_TICK(1);
  const double nfd = d.P.nearest_feature_dist;
  const float INF = __int_as_float(0x7f800000);
  // rows beyond the last query redo the last one (the rows of a wavefront run in lock-step) and do not store
  const int qi = threadIdx.x >> 4, q = qb * LO_QPB + qi;
  const bool store = q < nq;
  const float sx = s_sel[qi][0], sy = s_sel[qi][1], sz = s_sel[qi][2];
  int closest = -1, idx2 = -1, idx3 = -1;
  if (nt > 0) {
    // per-axis clamped differences to box c (0 inside the box)
    auto box_diff = [&](int c, float& dx, float& dy, float& dz) {
      float4 lo, hi;
      if (box_lds) { lo = s_box[2 * c]; hi = s_box[2 * c + 1]; } else { lo = bx[2 * c]; hi = bx[2 * c + 1]; }
      dx = fmaxf(fmaxf(lo.x - sx, sx - hi.x), 0.f);
      dy = fmaxf(fmaxf(lo.y - sy, sy - hi.y), 0.f);
      dz = fmaxf(fmaxf(lo.z - sz, sz - hi.z), 0.f);
    };
    // first target and number of targets of box c (< 0: none)
    auto box_span = [&](int c, int& start, int& len) {
      const int cc = max(c, 0);
      float ws, wl;
      if (box_lds) { ws = s_box[2 * cc].w; wl = s_box[2 * cc + 1].w; } else { ws = bx[2 * cc].w; wl = bx[2 * cc + 1].w; }
      start = __float_as_int(ws); len = c >= 0 ? __float_as_int(wl) : 0;
    };
    auto lb_f32 = [&](int c) -> float {
      if (c >= nch) return INF;
      float dx, dy, dz;
      box_diff(c, dx, dy, dz);
      float r = 0.f;
      r += dx * dx; r += dy * dy; r += dz * dz;
      return r;
    };
    // this lane's two targets of each of the boxes ca, cb (< 0: none) folded into its running (distance, index) minimum;
    // the four loads are issued together
    auto nn_eval = [&](const int (&cb)[LO_NB], unsigned long long best) -> unsigned long long {
      int t[TPL * LO_NB], bs[LO_NB], bl[LO_NB];
      bool v[TPL * LO_NB];
      float4 a[TPL * LO_NB];
#pragma unroll
      for (int u = 0; u < LO_NB; ++u) box_span(cb[u], bs[u], bl[u]);
#pragma unroll
      for (int u = 0; u < TPL * LO_NB; ++u) { t[u] = bs[u / TPL] + l16 + 16 * (u % TPL); v[u] = l16 + 16 * (u % TPL) < bl[u / TPL]; }
#pragma unroll
      for (int u = 0; u < TPL * LO_NB; ++u) a[u] = tg[v[u] ? t[u] : 0];
#pragma unroll
      for (int u = 0; u < TPL * LO_NB; ++u) {
        float r = 0.f, df;
        df = a[u].x - sx; r += df * df; df = a[u].y - sy; r += df * df; df = a[u].z - sz; r += df * df;
        const unsigned long long k = ((unsigned long long)(uint32_t)d_f2i(r) << 32) | (uint32_t)t[u];
        if (v[u]) best = k < best ? k : best;
      }
      return best;
    };
    // the next LO_NB set bits of a row's survivor mask as box indices (-1 when exhausted)
    auto pop_boxes = [&](uint32_t& surv, int c0, int (&cb)[LO_NB]) {
#pragma unroll
      for (int u = 0; u < LO_NB; ++u) { cb[u] = surv ? c0 + __ffs((int)surv) - 1 : -1; surv &= surv - 1; }   // 0 stays 0
    };
    // ---- the 3 x 3 grid cells around the query first: a candidate closer than (almost) a cell size is the nearest neighbour
    unsigned long long gbest = ~0ull;
    bool settled = false;
    {
      const int ggx = __float_as_int(s_geom[4]), ggy = __float_as_int(s_geom[5]);
      if (ggx > 0) {
        const int cx = (int)floorf((sx - s_geom[0]) * s_geom[2]), cy = (int)floorf((sy - s_geom[1]) * s_geom[2]);
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, ggx - 1);
        int a0[3], a1[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {   // unconditional loads from clamped addresses: all six in flight together
          const int y = cy + r - 1;
          const bool in = y >= 0 && y < ggy && x0 <= x1 && cx >= -1 && cx <= ggx;
          const int b0 = gcell[in ? y * ggx + x0 : 0], b1 = gcell[in ? y * ggx + x1 + 1 : 0];
          a0[r] = in ? b0 : 0; a1[r] = in ? b1 : 0;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          for (int t = a0[r] + l16; __ballot(t < a1[r]); t += 16) {
            const bool v = t < a1[r];
            const float4 a = gpts[v ? t : 0];
            float r2 = 0.f, df;
            df = a.x - sx; r2 += df * df; df = a.y - sy; r2 += df * df; df = a.z - sz; r2 += df * df;
            const unsigned long long k = ((unsigned long long)(uint32_t)d_f2i(r2) << 32) | (uint32_t)__float_as_int(a.w);
            if (v) gbest = k < gbest ? k : gbest;
          }
        }
        gbest = row16_min_u64(gbest);
        settled = gbest != ~0ull && d_i2f((int32_t)(gbest >> 32)) < s_geom[3];
      }
    }
    unsigned long long bj = gbest;
    if (__ballot(!settled)) {   // (rare: a query with no target within a cell size — its row, and with it the wavefront's other rows, search the boxes; both answers are exact)
    unsigned long long m1 = ~0ull;
    for (int c = l16; c < nch; c += 16) {
      const unsigned long long k = ((unsigned long long)(uint32_t)d_f2i(lb_f32(c)) << 32) | (uint32_t)c;
      m1 = k < m1 ? k : m1;
    }
    const int cs = (int)(uint32_t)row16_min_u64(m1);
    LA_TICK(2);   // box with the smallest bound: nt > 0, so it exists
    unsigned long long best;
    { int cb[LO_NB]; for (int u = 0; u < LO_NB; ++u) cb[u] = -1; cb[0] = cs; cb[1] = (cs ^ 1) < nch ? (cs ^ 1) : -1; best = nn_eval(cb, ~0ull); }
    const float bound = d_i2f((int32_t)(row16_min_u64(best) >> 32));
    LA_TICK(3);
    for (int c0 = 0; c0 < nch; c0 += 16) {
      const int c = c0 + l16;
      uint32_t surv = row_bits(__ballot(lb_f32(c) <= bound && (c >> 1) != (cs >> 1)), lane);
      while (__ballot(surv != 0)) {   // LO_NB surviving boxes per turn
        int cb[LO_NB];
        pop_boxes(surv, c0, cb);
        best = nn_eval(cb, best);
      }
    }
    bj = row16_min_u64(best);
    }
    LA_TICK(4);
    const bool found = (double)d_i2f((int32_t)(bj >> 32)) < nfd;
    if (found) closest = (int)(uint32_t)bj;
    // ---- ring walk; rows without a closest point walk an empty window
    const int cref = found ? closest : 0;
    const int cr = (int)tg[cref].w;  // int(intensity) = ring (:347,:436)
    const int W = d.P.ring_window;
    const int rlo = min(max(cr - W, 0), d.NS - 1), rhi = min(max(cr + W, 0), d.NS - 1), crc = min(max(cr, 0), d.NS - 1);
    const int lo = found ? s_roff[rlo] : 0, hi = found ? s_roff[rhi + 1] : 0;           // the walks stay inside [lo, hi)
    const int same_lo = s_roff[crc], same_hi = s_roff[crc + 1];
    WalkBest b2{nfd, 0x7fffffff, -1}, b3{nfd, 0x7fffffff, -1};
    auto walk_one = [&](int k, bool valid, const float4& a) {
      if (!valid || k < lo || k >= hi || k == closest) return;
      const double ex = (double)(a.x - sx), ey = (double)(a.y - sy), ez = (double)(a.z - sz);
      const double pd = ex * ex + ey * ey + ez * ez;  // pow(f32 diff, 2) summed in double (:354)
      if (!(pd < nfd)) return;
      // visiting order of the reference: closest+1, closest+2, ... then closest-1, closest-2, ...
      const int rank = k > closest ? k - closest - 1 : (hi - closest - 1) + (closest - 1 - k);
      const bool same = k >= same_lo && k < same_hi;
      // surf: same ring -> b2, other rings -> b3; corner: other rings -> b2 (strictly above going up / strictly below going
      // down, :446,:462).  Both updates are evaluated and committed by value: choosing the struct to update at run time
      // is a select of addresses, which put b2 / b3 into scratch memory.
      const bool to2 = kind == 0 ? same : !same, to3 = kind == 0 && !same;
      const bool w2 = to2 && (pd < b2.dist || (pd == b2.dist && rank < b2.rank));
      const bool w3 = to3 && (pd < b3.dist || (pd == b3.dist && rank < b3.rank));
      b2.dist = w2 ? pd : b2.dist; b2.rank = w2 ? rank : b2.rank; b2.idx = w2 ? k : b2.idx;
      b3.dist = w3 ? pd : b3.dist; b3.rank = w3 ? rank : b3.rank; b3.idx = w3 ? k : b3.idx;
    };
    auto walk_eval = [&](const int (&cb)[LO_NB]) {   // boxes cb[] (< 0: none): all their loads in flight together
      int k[TPL * LO_NB], bs[LO_NB], bl[LO_NB];
      bool v[TPL * LO_NB];
      float4 a[TPL * LO_NB];
#pragma unroll
      for (int u = 0; u < LO_NB; ++u) box_span(cb[u], bs[u], bl[u]);
#pragma unroll
      for (int u = 0; u < TPL * LO_NB; ++u) { k[u] = bs[u / TPL] + l16 + 16 * (u % TPL); v[u] = l16 + 16 * (u % TPL) < bl[u / TPL]; }
#pragma unroll
      for (int u = 0; u < TPL * LO_NB; ++u) a[u] = tg[v[u] ? k[u] : 0];
#pragma unroll
      for (int u = 0; u < TPL * LO_NB; ++u) walk_one(k[u], v[u], a[u]);
    };
    // class S: same ring (surf only); class O: the other rings of the window.  A box belongs to one ring.
    const int cw0 = found ? s_boff[rlo] : 0, cw1 = found ? s_boff[rhi + 1] - 1 : -1;   // empty window: no box
    const int same_b0 = s_boff[crc], same_b1 = s_boff[crc + 1];
    auto box_class = [&](int c, bool& inS, bool& inO) {
      const bool own = c >= same_b0 && c < same_b1;
      inS = kind == 0 && own;
      inO = !own;
    };
    auto lb_f64 = [&](int c) -> double {
      float dx, dy, dz;
      box_diff(c, dx, dy, dz);
      const double ex = (double)dx, ey = (double)dy, ez = (double)dz;
      return ex * ex + ey * ey + ez * ez;
    };
    // seeds: the box of the closest point (its ring neighbours) and the other-ring box with the smallest bound
    unsigned long long mo = ~0ull;
    for (int it = 0; __ballot(cw0 + it * 16 <= cw1); ++it) {
      const int c = cw0 + it * 16 + l16;
      if (c <= cw1) {
        bool inS, inO;
        box_class(c, inS, inO);
        if (inO) {
          const unsigned long long k = ((unsigned long long)__double_as_longlong(lb_f64(c)) & 0xFFFFFFFF00000000ull) | (uint32_t)c;
          mo = k < mo ? k : mo;   // ordered by the high word of the bound: any box is a valid seed
        }
      }
    }
    mo = row16_min_u64(mo);
    LA_TICK(5);
    const int cseedS = cw1 >= cw0 ? min(max(same_b0 + (closest - same_lo) / LO_CH, cw0), cw1) : -1, cseedO = mo == ~0ull ? -1 : (int)(uint32_t)mo;
    { int cb[LO_NB]; for (int u = 0; u < LO_NB; ++u) cb[u] = -1; cb[0] = cseedS; cb[1] = cseedO != cseedS ? cseedO : -1; walk_eval(cb); }
    // class bounds after the seeds (an upper bound of the final minimum; nfd when nothing was found)
    // (values, not a conditional on the two structs: `c ? b3.dist : b2.dist` is a select of addresses and sent both structs
    //  to scratch memory, in the middle of the hot loop)
    const double seenS = b2.dist, seenO3 = b3.dist;
    double inS = nfd, inO = seenS;
    if (kind == 0) { inS = seenS; inO = seenO3; }
    const double boundS = __longlong_as_double((long long)row16_min_u64((unsigned long long)__double_as_longlong(inS)));
    LA_TICK(6);
    const double boundO = __longlong_as_double((long long)row16_min_u64((unsigned long long)__double_as_longlong(inO)));
    // (round 6) the survivors of up to FOUR groups of 16 boxes are collected in one 64-bit mask per row before any of them is evaluated: the few boxes that survive
    // (one or two per ring of the window) lie in different groups, and evaluating group by group spent a half-empty walk_eval turn on most of them
    if (lane == 0) LA_COUNT(0, 1);
    if (l16 == 0) LA_COUNT(3, max(cw1 - cw0 + 1, 0));
    for (int it0 = 0; __ballot(cw0 + it0 * 16 <= cw1); it0 += 4) {
      const int cbase = cw0 + it0 * 16;
      unsigned long long surv = 0ull;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = cbase + g * 16 + l16;
        if (!__ballot(c <= cw1)) break;   // (uniform)
        bool take = false;
        if (c <= cw1 && c != cseedS && c != cseedO) {
          bool inS, inO;
          box_class(c, inS, inO);
          const double lb = lb_f64(c);
          take = (inS && lb <= boundS) || (inO && lb <= boundO);
        }
        surv |= (unsigned long long)row_bits(__ballot(take), lane) << (16 * g);
      }
      if (l16 == 0) LA_COUNT(2, __popcll(surv));
      while (__ballot(surv != 0ull)) {
        if (lane == 0) LA_COUNT(1, 1);
        int cb[LO_NB];
#pragma unroll
        for (int u = 0; u < LO_NB; ++u) { cb[u] = surv ? cbase + __ffsll((long long)surv) - 1 : -1; surv &= surv - 1ull; }   // 0 stays 0
        walk_eval(cb);
      }
    }
    LA_TICK(7);
    b2 = walk_reduce_row(b2);
    idx2 = b2.idx;
    if (kind == 0) { b3 = walk_reduce_row(b3); idx3 = b3.idx; }
  }
  LA_TICK(8);
  if (l16 == 0 && store) {
    int* row = d.lo_corr + ((size_t)slot * (d.lo_qcap_surf + d.lo_qcap_corner) + (kind == 0 ? 0 : d.lo_qcap_surf) + q) * 4;
    const bool ok = kind == 0 ? (idx2 >= 0 && idx3 >= 0) : (idx2 >= 0);
    row[0] = q; row[1] = ok ? closest : -1; row[2] = idx2; row[3] = idx3;
  }
  }
}

template <int kind, int BOXCAP = LO_BOX_LDS>
__global__ void __launch_bounds__(LO_BLOCK) lo_assoc(DevCtx d, int box_lds_max) {
  lo_assoc_body<kind, LO_BLOCK, BOXCAP>(d, box_lds_max, blockIdx.x + d.slot0, blockIdx.y, gridDim.y);   // slot fastest: a stream's workgroups share an XCD / L2 (see lm_knn)
}

// evaluate every valid correspondence row of [row0, row0+n) at pose p
template <int BLK>
DEV_INLINE void lo_eval_rows(const DevCtx& d, int slot, int kind, int n, const PoseTerms& T, double acc[28]) {
  const int qk = kind == 0 ? F_FLAT : F_SHARP, tk = kind == 0 ? F_LFLAT : F_LSHARP;
  const float4* qpts = d.feat[qk] + fidx_cur(d, slot) * d.fcap[qk];
  const float4* tg = d.feat[tk] + fidx_last(d, slot) * d.fcap[tk];
  const int* rows = d.lo_corr + ((size_t)slot * (d.lo_qcap_surf + d.lo_qcap_corner) + (kind == 0 ? 0 : d.lo_qcap_surf)) * 4;
  for (int i = threadIdx.x; i < n; i += BLK) {
    const int4 r = *reinterpret_cast<const int4*>(rows + (size_t)i * 4);
    if (r.y < 0) continue;
    const float4 pc = qpts[r.x], pa = tg[r.y], pb = tg[r.z];
    const double cp[3] = {pc.x, pc.y, pc.z}, a[3] = {pa.x, pa.y, pa.z}, b[3] = {pb.x, pb.y, pb.z};
    double c[3] = {0, 0, 0};
    if (kind == 0) { const float4 pm = tg[r.w]; c[0] = pm.x; c[1] = pm.y; c[2] = pm.z; }
    double res, J[6];
    eval_block(kind == 0 ? BLK_SURF : BLK_CORNER, cp, a, b, c, 0.0, T, &res, J);
    accumulate_block(res, J, d.P.huber_delta, acc);
  }
}

// phase 0: ceres::Solve #1 on the surf blocks (:410-421); phase 1: Solve #2 on surf + corner
// blocks (:484-495) followed by the pose integration (:504-508).
#ifdef ALEGO_TIMING
__device__ long long lo_times[8];
extern "C" void alego_lo_times(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lo_times), sizeof(long long) * 8); }
#define LO_T0 const long long t0_ = wall_clock64()
#define LO_ACC(k) do { if (threadIdx.x == 0 && slot == d.slot0) lo_times[k] += wall_clock64() - t0_; } while (0)
#else
#define LO_T0
#define LO_ACC(k)
#endif
// The correspondence rows of a solve do not change between its evaluations (six per ceres::Solve): the points of the rows are
// gathered ONCE into LDS (structure of arrays: lane-consecutive rows read consecutive words) instead of from HBM / L2 on every
// evaluation (four dependent loads per row and evaluation: 20 of lo_solve's 53 us for one stream).  Rows keep their order, so
// every thread sums the same rows in the same order as lo_eval_rows does.  Problems with more than LO_LDS_ROWS rows stream.
#define LO_LDS_ROWS 640   // >= n_flat * n_sectors * 16 rings + n_sharp * n_sectors * 16 rings at the reference's parameters (576)
struct LoRowsLds { float v[12][LO_LDS_ROWS]; };   // c xyz, a xyz, b xyz, m xyz
template <int BLK>
DEV_INLINE void lo_stage_rows(const DevCtx& d, int slot, int kind, int n, int off, LoRowsLds& R, unsigned& okm) {
  const int qk = kind == 0 ? F_FLAT : F_SHARP, tk = kind == 0 ? F_LFLAT : F_LSHARP;
  const float4* qpts = d.feat[qk] + fidx_cur(d, slot) * d.fcap[qk];
  const float4* tg = d.feat[tk] + fidx_last(d, slot) * d.fcap[tk];
  const int* rows = d.lo_corr + ((size_t)slot * (d.lo_qcap_surf + d.lo_qcap_corner) + (kind == 0 ? 0 : d.lo_qcap_surf)) * 4;
  okm = 0;
  int k = 0;
  for (int i = threadIdx.x; i < n; i += BLK, ++k) {
    const int4 r = *reinterpret_cast<const int4*>(rows + (size_t)i * 4);
    const bool ok = r.y >= 0;
    const float4 pc = qpts[ok ? r.x : 0], pa = tg[ok ? r.y : 0], pb = tg[ok ? r.z : 0], pm = tg[(ok && kind == 0) ? r.w : 0];
    const int j = off + i;
    R.v[0][j] = pc.x; R.v[1][j] = pc.y; R.v[2][j] = pc.z; R.v[3][j] = pa.x; R.v[4][j] = pa.y; R.v[5][j] = pa.z;
    R.v[6][j] = pb.x; R.v[7][j] = pb.y; R.v[8][j] = pb.z; R.v[9][j] = pm.x; R.v[10][j] = pm.y; R.v[11][j] = pm.z;
    if (ok) okm |= 1u << k;
  }
}
template <int BLK, int kind>
DEV_INLINE void lo_eval_staged(int n, int off, const LoRowsLds& R, unsigned okm, const PoseTerms& T, double huber, double acc[28]) {
  int k = 0;
  for (int i = threadIdx.x; i < n; i += BLK, ++k) {
    if (!((okm >> k) & 1u)) continue;
    const int j = off + i;
    const double cp[3] = {R.v[0][j], R.v[1][j], R.v[2][j]}, a[3] = {R.v[3][j], R.v[4][j], R.v[5][j]}, b[3] = {R.v[6][j], R.v[7][j], R.v[8][j]};
    const double c[3] = {kind == 0 ? (double)R.v[9][j] : 0.0, kind == 0 ? (double)R.v[10][j] : 0.0, kind == 0 ? (double)R.v[11][j] : 0.0};
    double res, J[6];
    eval_block(kind == 0 ? BLK_SURF : BLK_CORNER, cp, a, b, c, 0.0, T, &res, J);
    accumulate_block(res, J, huber, acc);
  }
}

template <int BLK, bool STAGED>   // STAGED: the handle's row capacities fit LO_LDS_ROWS (decided at launch: fixed by the geometry / parameters)
DEV_INLINE void lo_solve_body(const DevCtx& d, int phase, int slot) {
  const int cur = cur_in_flight(d, slot);
  int* sc = d.scal + slot * SC_COUNT;
  double* st = d.lo_state + (size_t)slot * LO_STATE_N;
  extern __shared__ __attribute__((aligned(16))) unsigned char lo_smem[];
  double* s_acc = reinterpret_cast<double*>(lo_smem);                // [28][BLK]
  double* s_seg = s_acc + 28 * (BLK / 4);                        // [28][BLK/128]
  __shared__ double s_out[28], s_trig[12];
  __shared__ LmState S;
  __shared__ int s_action, s_cnt[BLK / 64];
  __shared__ typename std::conditional<STAGED, LoRowsLds, char>::type s_rows;
  if (!sc[SC_LO_INIT]) {  // :316-324
    if (phase == 1 && threadIdx.x == 0) {
      sc[SC_LO_INIT] = 1; sc[SC_LO_FLAGS] = 1; sc[SC_LO_NSURF] = 0; sc[SC_LO_NCORNER] = 0; sc[SC_CUR] = cur;
      sc[SC_ODOM_VALID] = 0; d.scal[scan_slot_of(d, slot) * SC_COUNT + SC_ODOM_VALID] = 0;   // (the scan's own entry is what LaserMapping reads)
    }
    return;
  }
#ifdef ALEGO_TIMING
  const long long tk0_ = wall_clock64();
  if (threadIdx.x == 0 && slot == d.slot0) lo_times[7] += 1;
#endif
  const int nq_s = d.feat_cnt[fidx_cur(d, slot) * 4 + F_FLAT];
  const int nq_c = d.feat_cnt[fidx_cur(d, slot) * 4 + F_SHARP];
  // count the correspondences of the kind associated just before this call
  {
    const int n = phase == 0 ? nq_s : nq_c;
    const int* rows = d.lo_corr + ((size_t)slot * (d.lo_qcap_surf + d.lo_qcap_corner) + (phase == 0 ? 0 : d.lo_qcap_surf)) * 4;
    int c = 0;
    for (int i = threadIdx.x; i < n; i += BLK) c += rows[(size_t)i * 4 + 1] >= 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < BLK / 64; ++w) t += s_cnt[w];
      s_cnt[0] = t;
      sc[phase == 0 ? SC_LO_NSURF : SC_LO_NCORNER] = t;
      if (phase == 0) sc[SC_LO_FLAGS] = 0;
      if (t < d.P.lo_min_corr) sc[SC_LO_FLAGS] |= (phase == 0 ? 2 : 4);
    }
    __syncthreads();
  }
  const bool do_solve = s_cnt[0] >= d.P.lo_min_corr;
  if (do_solve) {
    double acc[28];
    unsigned oks = 0, okc = 0;
    if constexpr (STAGED) {
      LO_T0;
      lo_stage_rows<BLK>(d, slot, 0, nq_s, 0, s_rows, oks);
      if (phase == 1) lo_stage_rows<BLK>(d, slot, 1, nq_c, nq_s, s_rows, okc);
      LO_ACC(1);
      // (every thread reads back only what it wrote itself: no barrier needed)
    }
    auto evaluate = [&](const double* x) {
#pragma unroll
      for (int k = 0; k < 28; ++k) acc[k] = 0;
      PoseTerms T;
      { LO_T0; T = pose_terms_coop(x, s_trig); LO_ACC(0); }
      { LO_T0;
        if constexpr (STAGED) {
          lo_eval_staged<BLK, 0>(nq_s, 0, s_rows, oks, T, d.P.huber_delta, acc);
          if (phase == 1) lo_eval_staged<BLK, 1>(nq_c, nq_s, s_rows, okc, T, d.P.huber_delta, acc);
        } else {
          lo_eval_rows<BLK>(d, slot, 0, nq_s, T, acc);
          if (phase == 1) lo_eval_rows<BLK>(d, slot, 1, nq_c, T, acc);
        }
        LO_ACC(1); }
      { LO_T0; block_reduce28_lds<BLK>(acc, s_acc, s_seg, s_out); LO_ACC(2); }
    };
    double x0[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) x0[k] = st[LS_PARAMS + k];
    evaluate(x0);
    if (threadIdx.x == 0) lm_begin(S, x0, s_out, phase == 0 ? d.P.lo_iters_surf : d.P.lo_iters_corner);
    __syncthreads();
    while (true) {
      { LO_T0; if (threadIdx.x == 0) s_action = lm_propose(S);
      __syncthreads(); LO_ACC(3); }
      const int act = s_action;
      if (act == LM_STOP) break;
      if (act == LM_EVAL) {
        double xc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) xc[k] = S.cand[k];
        evaluate(xc);
        { LO_T0; if (threadIdx.x == 0) s_action = lm_consume(S, s_out);
        __syncthreads(); LO_ACC(4); }
        if (s_action == LM_STOP) break;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) st[LS_PARAMS + k] = S.x[k];
      { LO_T0; store_pose_rotation(st); LO_ACC(5); }   // for the next lo_assoc (corner association / next scan)
      st[LS_COSTS + phase * 2] = S.initial_cost; st[LS_COSTS + phase * 2 + 1] = S.x_cost;
      sc[phase == 0 ? SC_LO_ITERS : SC_LO_ITERS2] = S.iter | (S.successful << 8) | (S.termination << 16);
    }
  } else if (threadIdx.x == 0) {
    sc[phase == 0 ? SC_LO_ITERS : SC_LO_ITERS2] = 0;
  }
  if (threadIdx.x == 0) {
    if (phase == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) st[LS_PARAMS_SURF + k] = st[LS_PARAMS + k];
    } else {
      // pose integration :504-508 (roll/pitch ignored)
      const double p0 = st[LS_PARAMS + 0], p1 = st[LS_PARAMS + 1], p2 = st[LS_PARAMS + 2], yaw = st[LS_PARAMS + 5];
      const double c = cos(yaw), s = sin(yaw);
      const double rl[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
      double rw[9], nt[3], nr[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) rw[k] = st[LS_RW + k];
#pragma unroll
      for (int i = 0; i < 3; ++i) nt[i] = st[LS_TW + i] + (rw[i * 3 + 0] * p0 + rw[i * 3 + 1] * p1 + rw[i * 3 + 2] * p2);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) nr[i * 3 + j] = rw[i * 3 + 0] * rl[j] + rw[i * 3 + 1] * rl[3 + j] + rw[i * 3 + 2] * rl[6 + j];
#pragma unroll
      for (int i = 0; i < 3; ++i) st[LS_TW + i] = nt[i];
#pragma unroll
      for (int k = 0; k < 9; ++k) st[LS_RW + k] = nr[k];
      const DQuat q = dq_from_mat(nr);
      double* po = d.poses + (size_t)scan_slot_of(d, slot) * 16;   // /odom/lidar of this scan (its lane's entry when scans are processed ahead)
      po[0] = nt[0]; po[1] = nt[1]; po[2] = nt[2]; po[3] = q.w; po[4] = q.x; po[5] = q.y; po[6] = q.z;
      sc[SC_ODOM_VALID] = 1; d.scal[scan_slot_of(d, slot) * SC_COUNT + SC_ODOM_VALID] = 1;
      sc[SC_CUR] = cur;  // surf_last_ / corner_last_ <- this scan's features (:531-534)
    }
  }
#ifdef ALEGO_TIMING
  if (threadIdx.x == 0 && slot == d.slot0) lo_times[6] += wall_clock64() - tk0_;
#endif
}

template <int BLK, bool STAGED = false>
__global__ void __launch_bounds__(BLK) lo_solve_t(DevCtx d, int phase) { lo_solve_body<BLK, STAGED>(d, phase, blockIdx.x + d.slot0); }

// (One stream, alego_stream_run: the four dependent launches of a scan's LaserOdometry were also tried as ONE launch of 48 one-wavefront
//  workgroups separated by grid barriers on a counter in HBM — bit-identical, and no faster: 5.60 k against 5.66 k scans/s.  The scan's
//  critical path is shared between this chain and LaserMapping's ~14 dependent launches on its own stream; removing three dispatch
//  gaps here moves nothing while that chain is as long.  Removed again.)

// ---- debug entries (alego_debug_eval_blocks / alego_debug_transform_to_start): the device functions the solvers and the
// association kernel call, on caller-provided data, for direct parity tests against the oracle
__global__ void dbg_eval_blocks(int type, int n, const double* geom13, const double* params6, double* res, double* jac6) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const PoseTerms T = pose_terms(params6);
  const double* g = geom13 + (size_t)i * 13;
  const double cp[3] = {g[0], g[1], g[2]}, a[3] = {g[3], g[4], g[5]}, b[3] = {g[6], g[7], g[8]}, c[3] = {g[9], g[10], g[11]};
  double r, J[6];
  eval_block(type, cp, a, b, c, g[12], T, &r, J);
  res[i] = r;
#pragma unroll
  for (int k = 0; k < 6; ++k) jac6[(size_t)i * 6 + k] = J[k];
}
__global__ void dbg_transform_to_start(const double* params6, const float4* pts, int n, float4* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double R[9];
  pose_rotation(params6, R);
  float o[3];
  transform_to_start(R, params6, pts[i], o);
  out[i] = make_float4(o[0], o[1], o[2], pts[i].w);
}
void launch_dbg_eval_blocks(int type, int n, const double* geom13, const double* params6, double* res, double* jac6, hipStream_t st) {
  hipLaunchKernelGGL(dbg_eval_blocks, dim3((n + 63) / 64), dim3(64), 0, st, type, n, geom13, params6, res, jac6);
}
void launch_dbg_transform_to_start(const double* params6, const float4* pts, int n, float4* out, hipStream_t st) {
  hipLaunchKernelGGL(dbg_transform_to_start, dim3((n + 63) / 64), dim3(64), 0, st, params6, pts, n, out);
}

#define LO_SOLVE_LDS_OF(B) ((size_t)(28 * ((B) / 4) + 28 * ((B) / 128 + 1)) * sizeof(double))
#define LO_SOLVE_WIDE 256   // solver workgroup of sensors with more than LO_WIDE_ROWS correspondence rows (64 rings: ~1.5 k rows, 23 per lane of one wavefront)
#define LO_WIDE_ROWS 1024
int lo_configure() {
  return (hipFuncSetAttribute(reinterpret_cast<const void*>(lo_solve_t<LO_SOLVE_BLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LO_SOLVE_LDS_OF(LO_SOLVE_BLOCK)) == hipSuccess &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(lo_solve_t<LO_SOLVE_BLOCK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LO_SOLVE_LDS_OF(LO_SOLVE_BLOCK)) == hipSuccess &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(lo_solve_t<LO_SOLVE_WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LO_SOLVE_LDS_OF(LO_SOLVE_WIDE)) == hipSuccess) ? 0 : -1;
}

// ---- IMU ring + motion de-skew (laserOdometry.cpp:557-726,761-802; dead in the reference: the call at :115 is commented out) ----
// imuHandler's ring bookkeeping and dead-reckoning for n samples of one slot; the per-sample trigonometry (tf getRPY, gravity
// removal, the f32 rotation of the acceleration) is done by the host in alego_lo_push_imu: smp[i] = time, roll, pitch, yaw, acc xyz.
__global__ void lo_imu_push(DevCtx d, int slot, const double* smp, int n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int* ptr = d.imu_ptr + (size_t)slot * 4;
  double* ring = d.imu_ring + (size_t)slot * ALEGO_IMU_Q * 10;
  int last = ptr[0], front = ptr[1];
  for (int i = 0; i < n; ++i) {
    const double* s = smp + 7 * i;
    last = (last + 1) % ALEGO_IMU_Q;                                                  // :773-777
    if ((last + 1) % ALEGO_IMU_Q == front) front = (front + 1) % ALEGO_IMU_Q;
    double* r = ring + last * 10;
    r[0] = s[0]; r[1] = s[1]; r[2] = s[2]; r[3] = s[3];
    const double* b = ring + ((last - 1 + ALEGO_IMU_Q) % ALEGO_IMU_Q) * 10;
    const double td = r[0] - b[0];
    if (td < 1.) {                                                                    // :793-802
      for (int k = 0; k < 3; ++k) {
        r[4 + k] = b[4 + k] + b[7 + k] * td + s[4 + k] * td * td * 0.5;
        r[7 + k] = b[7 + k] + s[4 + k] * td;
      }
    }
  }
  ptr[0] = last; ptr[1] = front;
}

DEV_INLINE void dsk_mul3(const float m[3][3], const float v[3], float o[3]) {   // Eigen's unrolled 3-term reduction: a0 + (a1 + a2)
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = m[i][0] * v[0] + (m[i][1] * v[1] + m[i][2] * v[2]);
}

// adjustDistortion, IMU branch, one workgroup per slot.  The reference walks the points in order with a forward-only cursor into
// the IMU ring (imu_ptr_front_ starts at imu_ptr_last_iter_ and only advances while cur_time >= imu_time_[front]); with
// non-decreasing IMU stamps (alego_lo_push_imu enforces them) point i's cursor is the running maximum over points 0..i of each
// point's own first ring step with cur_time < imu_time_: a binary search per point + a max-scan in point order.  The first point
// whose cursor entry is more than scan_period away aborts the function (:604-608): it and everything after it stay as they are.
#define DSK_T 256
__global__ void __launch_bounds__(DSK_T) lo_deskew(DevCtx d) {
  const int slot = blockIdx.x + d.slot0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t base = (size_t)slot * d.N;
  const int M = d.scal[(size_t)slot * SC_COUNT + SC_M];
  const float4* in = d.seg_pts + base;
  float4* out = d.seg_dsk + base;
  int* ptr = d.imu_ptr + (size_t)slot * 4;
  const double* ring = d.imu_ring + (size_t)slot * ALEGO_IMU_Q * 10;
  const int last = ptr[0], it0 = ptr[2];
  if (tid == 0) d.scal[(size_t)slot * SC_COUNT + SC_M_DSK] = M;   // (what alego_lo_get_undistorted copies: SC_M belongs to ImageProjection, which may already be a scan ahead)
  if (!(last > 0)) {                                                                  // :593
    for (int i = tid; i < M; i += DSK_T) out[i] = in[i];
    return;
  }
  __shared__ double s_time[ALEGO_IMU_Q];
  __shared__ int s_wmax[DSK_T / 64], s_abort, s_front_a, s_iter_a;
  __shared__ float s_start[15];   // shift_start, velo_start, r_s_i
  for (int j = tid; j < ALEGO_IMU_Q; j += DSK_T) s_time[j] = ring[j * 10];
  if (tid == 0) { s_abort = 0x7fffffff; s_front_a = -1; s_iter_a = -1; }
  const int H = d.H;
  const float so = d.ori[(size_t)slot * 4], eo = d.ori[(size_t)slot * 4 + 1];
  int start_ori = (int)(((double)so + 2 * M_PI) / H), end_ori = (int)(((double)eo + 2 * M_PI) / H);   // :562-563 (sic)
  if (start_ori >= H) start_ori -= H;
  if (end_ori >= H) end_ori -= H;
  int ori_diff = end_ori - start_ori;
  if (ori_diff <= 0) ori_diff = H;
  const double scan_time = d.scan_stamp[slot], period = d.P.scan_period;
  const int span = (last - it0 + ALEGO_IMU_Q) % ALEGO_IMU_Q;
  __syncthreads();
  int carry = 0;   // cursor (ring steps from it0) after the points before this chunk
  for (int c0 = 0; c0 < M; c0 += DSK_T) {
    const int i = c0 + tid;
    const bool valid = i < M;
    const int col = valid ? d.seg_col[base + i] : 0;
    const double rel_time = (col - start_ori) * period / ori_diff;                    // :588
    const double cur_time = scan_time + rel_time;
    int k = 0;
    if (valid) {   // first step k in [0, span) with cur_time < imu_time_[(it0 + k) % Q], else span (the cursor stops at imu_ptr_last_)
      int lo = 0, hi = span;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (cur_time < s_time[(it0 + mid) % ALEGO_IMU_Q]) hi = mid; else lo = mid + 1; }
      k = lo;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(k, o, 64); if (lane >= o) k = max(k, t); }
    if (lane == 63) s_wmax[wave] = k;
    __syncthreads();
    int before = carry, total = carry;
#pragma unroll
    for (int w = 0; w < DSK_T / 64; ++w) { const int v = s_wmax[w]; if (w < wave) before = max(before, v); total = max(total, v); }
    k = max(k, before);
    const int front = (it0 + k) % ALEGO_IMU_Q;
    const bool viol = valid && fabs(cur_time - s_time[front]) > period;               // :604
    const unsigned long long vm = __ballot(viol);
    if (vm && lane == 0) atomicMin(&s_abort, c0 + wave * 64 + (__ffsll((long long)vm) - 1));
    __syncthreads();
    const int a = s_abort;
    const bool process = valid && i < a;
    float rpy[3] = {0, 0, 0}, sh[3] = {0, 0, 0}, ve[3] = {0, 0, 0}, rc[3][3];
    if (process) {
      const double* R = ring + front * 10;
      if (cur_time > s_time[front]) {                                                 // :610-621
        rpy[0] = (float)R[1]; rpy[1] = (float)R[2]; rpy[2] = (float)R[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { sh[q] = (float)R[4 + q]; ve[q] = (float)R[7 + q]; }
      } else {                                                                        // :622-637
        const double* B = ring + ((front - 1 + ALEGO_IMU_Q) % ALEGO_IMU_Q) * 10;
        const double rf = (cur_time - B[0]) / (R[0] - B[0]), rb = 1. - rf;
        rpy[0] = (float)(R[1] * rf + B[1] * rb); rpy[1] = (float)(R[2] * rf + B[2] * rb); rpy[2] = (float)(R[3] * rf + B[3] * rb);
#pragma unroll
        for (int q = 0; q < 3; ++q) { sh[q] = (float)(R[4 + q] * rf + B[4 + q] * rb); ve[q] = (float)(R[7 + q] * rf + B[7 + q] * rb); }
      }
      const float kp[6] = {0.f, 0.f, 0.f, rpy[0], rpy[1], rpy[2]};
      float m[3][4];
      keypose_matrix(kp, m);                                                          // :639: (AngleAxisf(yaw, Z) * AngleAxisf(pitch, Y) * AngleAxisf(roll, X)).toRotationMatrix()
#pragma unroll
      for (int r = 0; r < 3; ++r) { rc[r][0] = m[r][0]; rc[r][1] = m[r][1]; rc[r][2] = m[r][2]; }
      if (i == 0) {                                                                   // :641-647; r_c.inverse(): cofactors / determinant (Eigen Inverse.h, size 3)
        auto cof = [&](int r, int c) { return rc[(r + 1) % 3][(c + 1) % 3] * rc[(r + 2) % 3][(c + 2) % 3] - rc[(r + 1) % 3][(c + 2) % 3] * rc[(r + 2) % 3][(c + 1) % 3]; };
        const float c0v[3] = {cof(0, 0), cof(1, 0), cof(2, 0)};
        const float det = c0v[0] * rc[0][0] + (c0v[1] * rc[1][0] + c0v[2] * rc[2][0]);
        const float invdet = 1.0f / det;
        s_start[0] = sh[0]; s_start[1] = sh[1]; s_start[2] = sh[2]; s_start[3] = ve[0]; s_start[4] = ve[1]; s_start[5] = ve[2];
        s_start[6] = c0v[0] * invdet; s_start[7] = c0v[1] * invdet; s_start[8] = c0v[2] * invdet;
        s_start[9] = cof(0, 1) * invdet; s_start[10] = cof(1, 1) * invdet; s_start[11] = cof(2, 1) * invdet;
        s_start[12] = cof(0, 2) * invdet; s_start[13] = cof(1, 2) * invdet; s_start[14] = cof(2, 2) * invdet;
      }
    }
    if (valid && i == a) s_front_a = front;                   // the cursor had already moved for the point that aborts (:595-603)
    if (valid && i == a - 1) s_iter_a = front;                // imu_ptr_last_iter_ of the last point that completed (:653)
    __syncthreads();
    if (valid) {
      float4 p = in[i];
      if (process && i > 0) {                                                         // :648-654
        const float rt = (float)rel_time;
        float rsi[3][3], sfs[3], v[3], w[3], o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { rsi[r][0] = s_start[6 + 3 * r]; rsi[r][1] = s_start[7 + 3 * r]; rsi[r][2] = s_start[8 + 3 * r]; }
#pragma unroll
        for (int q = 0; q < 3; ++q) sfs[q] = (sh[q] - s_start[q]) - s_start[3 + q] * rt;
        const float pv[3] = {p.x, p.y, p.z};
        dsk_mul3(rc, pv, v);
#pragma unroll
        for (int q = 0; q < 3; ++q) w[q] = v[q] + sfs[q];
        dsk_mul3(rsi, w, o);
        p.x = o[0]; p.y = o[1]; p.z = o[2];
      }
      out[i] = p;
    }
    if (a != 0x7fffffff) {   // aborted inside this chunk: the rest of the cloud stays as it is
      for (int j = c0 + DSK_T + tid; j < M; j += DSK_T) out[j] = in[j];
      if (tid == 0) {
        ptr[1] = s_front_a;
        ptr[2] = s_iter_a >= 0 ? s_iter_a : (a == 0 ? it0 : (it0 + carry) % ALEGO_IMU_Q);
      }
      return;
    }
    carry = total;
    __syncthreads();
  }
  if (tid == 0 && M > 0) { ptr[1] = (it0 + carry) % ALEGO_IMU_Q; ptr[2] = (it0 + carry) % ALEGO_IMU_Q; }
}

// per-scan pose log (alego_trajectory_*): what a bag replay publishes on /odom/lidar and /odom_aft_mapped, kept on the device so that
// a batch replay needs no host synchronisation per scan
// staged_odom != null: LaserMapping runs behind the front end on its own stream (alego_batch_run); the scan's /odom/lidar is then taken
// from the hand-over buffer (LaserOdometry may already have written a later scan's pose into `poses`)
__global__ void traj_log(DevCtx d, const double* staged_odom, int par) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.n_launch) return;
  const int slot = i + d.slot0;
  const int k = d.traj_n[slot];
  d.traj_n[slot] = k + 1;
  if (k >= d.traj_cap) return;
  const double* po = d.poses + (size_t)slot * 16;
  const double* od = staged_odom ? staged_odom + ((size_t)slot * 2 + par) * 8 : po;
  double* o = d.traj + ((size_t)slot * d.traj_cap + k) * 14;
#pragma unroll
  for (int j = 0; j < 7; ++j) o[j] = od[j];
#pragma unroll
  for (int j = 7; j < 14; ++j) o[j] = po[j];
}
void launch_traj_log(const DevCtx& d, hipStream_t st, const double* staged_odom, int par) {
  hipLaunchKernelGGL(traj_log, dim3((d.n_launch + 63) / 64), dim3(64), 0, st, d, staged_odom, par);
}

void launch_lo_imu_push(const DevCtx& d, int slot, const double* smp_dev, int n, hipStream_t st) {
  hipLaunchKernelGGL(lo_imu_push, dim3(1), dim3(1), 0, st, d, slot, smp_dev, n);
}
void launch_lo_deskew(const DevCtx& d, hipStream_t st) {
  ALEGO_LAUNCH(lo_deskew, dim3(d.n_launch), dim3(DSK_T), 0, st, d);
}

void launch_lo(const DevCtx& d, hipStream_t st) {
  const bool big_boxes = d.NS > 16;   // (which instantiation holds the boxes does not change a result: the boxes only prune)
  const int box_lds_max = std::min(d.opt_lo_box_lds, big_boxes ? (int)LO_BOX_LDS_BIG : (int)LO_BOX_LDS);
  const bool wide = d.lo_qcap_surf + d.lo_qcap_corner > LO_WIDE_ROWS;   // (fixed by the geometry: every handle of a sensor sums its rows in the same order)
  // the rows of a solve staged in LDS: only when every possible row count fits (capacities, not counts: the variant is a property of the handle)
  const bool staged = !wide && d.lo_qcap_surf + d.lo_qcap_corner <= LO_LDS_ROWS && d.lo_qcap_surf <= 32 * LO_SOLVE_BLOCK && d.lo_qcap_corner <= 32 * LO_SOLVE_BLOCK;
  const dim3 g0(d.n_launch, std::min((d.lo_qcap_surf + LO_QPB_OF(LO_BLOCK) - 1) / LO_QPB_OF(LO_BLOCK), 8)), g1(d.n_launch, std::min((d.lo_qcap_corner + LO_QPB_OF(LO_BLOCK) - 1) / LO_QPB_OF(LO_BLOCK), 12));
  if (big_boxes) { ALEGO_LAUNCH((lo_assoc<0, LO_BOX_LDS_BIG>), g0, dim3(LO_BLOCK), 0, st, d, box_lds_max); }
  else { ALEGO_LAUNCH(lo_assoc<0>, g0, dim3(LO_BLOCK), 0, st, d, box_lds_max); }
  auto solve = [&](int phase) {
    if (wide) { ALEGO_LAUNCH(lo_solve_t<LO_SOLVE_WIDE>, dim3(d.n_launch), dim3(LO_SOLVE_WIDE), LO_SOLVE_LDS_OF(LO_SOLVE_WIDE), st, d, phase); }
    else if (staged) { ALEGO_LAUNCH((lo_solve_t<LO_SOLVE_BLOCK, true>), dim3(d.n_launch), dim3(LO_SOLVE_BLOCK), LO_SOLVE_LDS_OF(LO_SOLVE_BLOCK), st, d, phase); }
    else { ALEGO_LAUNCH(lo_solve_t<LO_SOLVE_BLOCK>, dim3(d.n_launch), dim3(LO_SOLVE_BLOCK), LO_SOLVE_LDS_OF(LO_SOLVE_BLOCK), st, d, phase); }
  };
  solve(0);
  if (big_boxes) { ALEGO_LAUNCH((lo_assoc<1, LO_BOX_LDS_BIG>), g1, dim3(LO_BLOCK), 0, st, d, box_lds_max); }
  else { ALEGO_LAUNCH(lo_assoc<1>, g1, dim3(LO_BLOCK), 0, st, d, box_lds_max); }
  solve(1);
}
