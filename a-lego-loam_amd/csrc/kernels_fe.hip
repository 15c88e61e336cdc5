// kernels_fe.hip — feature extraction on gfx950 (replaces src/laserOdometry.cpp:122-293).
//
//   fe_curv    a7,a8: 11-tap f32 curvature sum through an LDS window + occlusion / parallel-beam
//              marking written as a gather (no atomics) (:122-159), and the flag byte of every point for the pick
//   fe_pick4   a9: four rings per wavefront (one 16-lane DPP row each); the reference's "sort, then scan
//              descending/ascending" is evaluated as repeated row-wide arg-max / arg-min over the
//              not-yet-picked candidates of the sector (identical result for the total order
//              (curvature, index)); +-5 neighbour suppression by ballot         (:172-286)
//   fe_pick    the same with one ring per wavefront (suppress_radius > 7, very long sectors)
//   fe_voxel   a10: per-ring pcl::VoxelGrid(0.4): runs of consecutive equal voxel ids ordered through
//              monotone buckets in LDS, centroid in sorted (= original) order    (:288-293)
//   fe_collect ring-ascending concatenation into the four feature clouds (:199-205,:245,:293) + bounding boxes of 32 consecutive
//              less_flat / less_sharp points for the next scan's LaserOdometry
#include <cstdlib>
#include <type_traits>
#include "dev_common.h"
#include "fe_common.h"
#include "prof.h"
#include "stdsort_emu.h"

#define FE_BLOCK 256
#define FE_CW 1024   // points per fe_curv workgroup
#define FE_MAXH 4096  // largest horizon_scan supported by the per-ring LDS staging
// fe_pick<FE_T>: sector elements per lane kept in registers (sector length <= 64*FE_T): 6 covers 16x1800
// (<= 300 points per sector), 12 covers horizon_scan 4096 (683)

__global__ void __launch_bounds__(FE_BLOCK) fe_curv(DevCtx d) {
  const int slot = blockIdx.y + d.slot0;
  const int M = d.scal[slot * SC_COUNT + SC_M];
  const int t0 = blockIdx.x * FE_CW;   // FE_CW points per workgroup, FE_CW / FE_BLOCK per thread
  if (t0 >= M) return;
  __shared__ float s_r[FE_CW + 2 * FE_HALO];
  __shared__ int s_c[FE_CW + 2 * FE_HALO];
  __shared__ uint8_t s_f[FE_CW + 2 * FE_HALO];
  fe_curv_chunk<FE_BLOCK, FE_CW>(d, slot, M, t0, s_r, s_c, s_f);
}

// one wavefront per (ring, slot).  Dynamic LDS: 3 bytes per ring point (column u16, flags + label u8); the
// kernel's duration under load is set by how many rings fit a CU next to the other streams' workgroups.
// flag bits: 0 picked, 1 ground, 2 curvature > edge_thres, 3 curvature < surf_thres, 4-5 cloud_label_ + 1
// STDSORT (alego_params.sort_mode = 2): candidates with EQUAL curvature are taken in the order libstdc++'s std::sort leaves them in
// (laserOdometry.cpp:185 sorts with a comparator on the curvature alone) instead of by index.  The picks only ever ask for the
// best remaining candidate, so the sort itself is not needed — only, when several candidates tie for it, where std::sort's
// partition phase would have left each of them (stdsort_emu.h); that arrangement is computed once per sector, on the first tie.
template <int FE_T, bool STDSORT>
__global__ void __launch_bounds__(64) fe_pick(DevCtx d) {
  const int slot = blockIdx.y + d.slot0, ring = blockIdx.x, lane = threadIdx.x;
  __shared__ typename std::conditional<STDSORT, SortEmu<64 * FE_T>, char>::type s_emu;
  const size_t base = (size_t)slot * d.N;
  const alego_params& P = d.P;
  const int S = d.ring_start[slot * d.NS + ring], E = d.ring_end[slot * d.NS + ring];
  const int rf = S - 5, rl = E + 5;  // first / last point of this ring in the segmented cloud
  const int cnt = rl - rf + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char fe_smem[];
  uint16_t* s_col = reinterpret_cast<uint16_t*>(fe_smem);
  uint8_t* s_flag = fe_smem + 2 * (size_t)d.H;
  const float* cdv = d.cd + base + rf;  // |cd| bit pattern = sort key: curvature order == unsigned order
#pragma unroll 4
  for (int k = lane; k < cnt; k += 64) {
    const float a = fabsf(d.cd[base + rf + k]);
    const double ad = (double)a;
    const double curv = ad * ad;  // (double)diff_range * diff_range, exact (:125)
    s_col[k] = (uint16_t)d.seg_col[base + rf + k];
    s_flag[k] = (uint8_t)((d.fe_flag[base + rf + k] & 1) | (d.seg_ground[base + rf + k] ? 2 : 0) |
                          (curv > P.edge_thres ? 4 : 0) | (curv < P.surf_thres ? 8 : 0) | (1 << 4));  // bits 4-5: label + 1
  }
  __syncthreads();
  int* st = d.st_idx + ((size_t)slot * d.NS + ring) * d.st_stride;
  int* st_sharp = st, *st_lsharp = st + d.cap_sharp, *st_flat = st_lsharp + d.cap_lsharp, *st_lfs = st_flat + d.cap_flat;
  int n_sharp = 0, n_ls = 0, n_flat = 0, n_lfs = 0;
  const int NSEC = P.n_sectors, SR = P.suppress_radius;
  for (int j = 0; j < NSEC; ++j) {
    int sp, ep;
    if (P.sector_formula == 0) { sp = (S * (NSEC - j) + E * j) / NSEC; ep = (S * (NSEC - 1 - j) + E * (j + 1)) / NSEC - 1; }
    else { const int diff = E - S; sp = S + j * diff / NSEC; ep = S + (j + 1) * diff / NSEC - 1; }
    if (sp >= ep) continue;
    const int lsp = sp - rf, lep = ep - rf;
    // Lane l owns the sector elements lsp + l + 64 t; their keys and candidate bits live in registers for the
    // whole sector, so an iteration of the greedy pick touches LDS only for the +-SR column test.
    uint32_t key[FE_T];
    uint32_t sharp_m = 0, flat_m = 0;
#pragma unroll
    for (int t = 0; t < FE_T; ++t) {
      const int c = lsp + lane + 64 * t;
      const bool ok = c <= lep;   // unconditional loads (clamped address): all of them in flight together
      const uint8_t f = s_flag[ok ? c : lsp];
      const uint32_t kv = (uint32_t)d_f2i(fabsf(cdv[ok ? c : lsp]));  // straight from HBM/L2: the LDS footprint decides how many rings fit a CU
      key[t] = ok ? kv : 0u;
      if (ok && (f & 7) == 4) sharp_m |= 1u << t;    // not picked, not ground, curvature > edge_thres
      if (ok && (f & 11) == 10) flat_m |= 1u << t;   // not picked, ground, curvature < surf_thres
    }
    // marks local index c and its +-SR neighbours picked (:211-234: stop at the first column jump), in LDS for
    // the later sectors and in the owners' candidate masks for this one
    auto mark = [&](int c, bool spread) {
      int nf = 0, nbk = 0;
      if (spread) {
        bool bad = false;
        if (lane < SR) { const int cdf = (int)s_col[c + lane + 1] - (int)s_col[c + lane]; bad = (cdf < 0 ? -cdf : cdf) > P.suppress_col_diff; }
        else if (lane >= 32 && lane < 32 + SR) { const int l = lane - 32; const int cdf = (int)s_col[c - l - 1] - (int)s_col[c - l]; bad = (cdf < 0 ? -cdf : cdf) > P.suppress_col_diff; }
        const unsigned long long mb = __ballot(bad);
        const unsigned lo = (unsigned)(mb & 0xffffffffull), hi = (unsigned)(mb >> 32);
        nf = lo ? min(SR, __ffs((int)lo) - 1) : SR;
        nbk = hi ? min(SR, __ffs((int)hi) - 1) : SR;
      }
      const int first = c - nbk, last = c + nf;
      if (lane <= last - first) s_flag[first + lane] |= 1;
      // the (single) owned element inside [first, last]
      const int off = (lane - (first - lsp)) & 63;
      const int ct = first + off;
      if (ct <= last && ct >= lsp && ct <= lep) { const uint32_t bit = 1u << ((ct - lsp - lane) / 64); sharp_m &= ~bit; flat_m &= ~bit; }
    };
    bool have_arr = false;
    auto sector_arrangement = [&]() {
      if constexpr (STDSORT) {
        const int n = lep - lsp + 1;
        __syncthreads();
        for (int i = lane; i < n; i += 64) s_emu.ak[i] = (uint32_t)d_f2i(fabsf(cdv[lsp + i]));   // |cd| bits order like the f64 curvature, equal iff it is
        __syncthreads();
        stdsort_arrangement(s_emu, n);
      }
    };
    // ---- sharp / less-sharp: descending curvature, ties -> larger index (:189-236) ----
    int picked_num = 0;
    while (true) {
      uint32_t bk = 0;
      int bt = 0;
#pragma unroll
      for (int t = 0; t < FE_T; ++t) if (((sharp_m >> t) & 1) && key[t] >= bk) { bk = key[t]; bt = t; }
      const uint32_t kmax = wave_max_u32(bk);
      if (kmax == 0) break;
      // almost always exactly one lane holds the maximum: its index comes through v_readlane; equal keys in several
      // lanes (ties -> larger index) take the second reduction
      const unsigned long long tie = __ballot(sharp_m && bk == kmax);
      int c = 0;
      bool by_arrangement = false;
      if constexpr (STDSORT) {
        int cnt = 0;
#pragma unroll
        for (int t = 0; t < FE_T; ++t) cnt += (((sharp_m >> t) & 1) && key[t] == kmax) ? 1 : 0;
        const unsigned long long a1 = __ballot(cnt > 0), a2 = __ballot(cnt > 1);
        by_arrangement = a2 != 0 || (a1 & (a1 - 1)) != 0;
        if (by_arrangement) {   // the scan runs k = ep .. sp: of the tied candidates the one std::sort placed last comes first
          if (!have_arr) { sector_arrangement(); have_arr = true; }
          uint32_t best = 0;
#pragma unroll
          for (int t = 0; t < FE_T; ++t)
            if (((sharp_m >> t) & 1) && key[t] == kmax) best = max(best, (((uint32_t)s_emu.pos[lane + 64 * t] + 1u) << 16) | (uint32_t)(lane + 64 * t));
          c = lsp + (int)(wave_max_u32(best) & 0xFFFFu);
        }
      }
      if (by_arrangement) {
      } else if ((tie & (tie - 1)) == 0) {
        const int wl = __ffsll((long long)tie) - 1;
        c = lsp + wl + 64 * __builtin_amdgcn_readlane(bt, wl);
      } else {
        const uint32_t cand = (sharp_m && bk == kmax) ? (uint32_t)(lsp + lane + 64 * bt) : 0u;
        c = (int)wave_max_u32(cand);
      }
      ++picked_num;
      int lab = 0;
      if (picked_num <= P.n_sharp) lab = 2; else if (picked_num <= P.n_less_sharp) lab = 1;
      if (lane == 0) {
        if (lab) s_flag[c] = (uint8_t)((s_flag[c] & 0xCF) | ((lab + 1) << 4));
        if (lab == 2) st_sharp[n_sharp] = c + rf;
        if (lab) st_lsharp[n_ls] = c + rf;
      }
      if (lab == 2) ++n_sharp;
      if (lab) ++n_ls;
      mark(c, lab != 0);  // the 21st pick is marked but breaks before the suppression (:207-210)
      if (!lab) break;
    }
    // ---- flat: ascending curvature, ground only, ties -> smaller index (:238-277) ----
    picked_num = 0;
    while (true) {
      uint32_t bk = 0xFFFFFFFFu;
      int bt = 0;
#pragma unroll
      for (int t = 0; t < FE_T; ++t) if (((flat_m >> t) & 1) && key[t] < bk) { bk = key[t]; bt = t; }
      const uint32_t kmin = wave_min_u32(bk);
      if (kmin == 0xFFFFFFFFu) break;
      const unsigned long long tie = __ballot(flat_m && bk == kmin);
      int c = 0;
      bool by_arrangement = false;
      if constexpr (STDSORT) {
        int cnt = 0;
#pragma unroll
        for (int t = 0; t < FE_T; ++t) cnt += (((flat_m >> t) & 1) && key[t] == kmin) ? 1 : 0;
        const unsigned long long a1 = __ballot(cnt > 0), a2 = __ballot(cnt > 1);
        by_arrangement = a2 != 0 || (a1 & (a1 - 1)) != 0;
        if (by_arrangement) {   // k = sp .. ep: the tied candidate std::sort placed first
          if (!have_arr) { sector_arrangement(); have_arr = true; }
          uint32_t best = 0xFFFFFFFFu;
#pragma unroll
          for (int t = 0; t < FE_T; ++t)
            if (((flat_m >> t) & 1) && key[t] == kmin) best = min(best, ((uint32_t)s_emu.pos[lane + 64 * t] << 16) | (uint32_t)(lane + 64 * t));
          c = lsp + (int)(wave_min_u32(best) & 0xFFFFu);
        }
      }
      if (by_arrangement) {
      } else if ((tie & (tie - 1)) == 0) {
        const int wl = __ffsll((long long)tie) - 1;
        c = lsp + wl + 64 * __builtin_amdgcn_readlane(bt, wl);
      } else {
        const uint32_t cand = (flat_m && bk == kmin) ? (uint32_t)(lsp + lane + 64 * bt) : 0xFFFFFFFFu;
        c = (int)wave_min_u32(cand);
      }
      ++picked_num;
      if (lane == 0) { s_flag[c] = (uint8_t)(s_flag[c] & 0xCF); st_flat[n_flat] = c + rf; }   // label -1
      ++n_flat;
      const bool stop = picked_num >= P.n_flat;
      mark(c, !stop);  // the n_flat-th pick breaks before the suppression (:248-251)
      if (stop) break;
    }
    __syncthreads();
    // ---- less-flat candidates in position order (:279-285) ----
    for (int c0 = lsp; c0 <= lep; c0 += 64) {
      const int c = c0 + lane;
      const bool take = c <= lep && ((s_flag[c] >> 4) & 3) <= 1;   // label <= 0
      const unsigned long long m = __ballot(take);
      if (take) st_lfs[n_lfs + (int)__popcll(m & ((1ull << lane) - 1ull))] = c + rf;
      n_lfs += (int)__popcll(m);
    }
  }
  __syncthreads();
  for (int k = lane; k < cnt; k += 64) d.plabel[base + rf + k] = (int)((s_flag[k] >> 4) & 3) - 1;
  if (lane == 0) {
    int* c = d.st_cnt + ((size_t)slot * d.NS + ring) * 8;
    c[0] = n_sharp; c[1] = n_ls; c[2] = n_flat; c[3] = n_lfs;
  }
}

// fe_pick4<FE_T>: the same greedy pick with FOUR rings per wavefront, one DPP row (16 lanes) each.  The per-pick
// overhead of the one-ring kernel (wave-wide arg-max, ballot, suppression window, uniform control flow: ~70 of its
// ~100 instructions per pick) is issued once for four rings; the reductions stay inside a DPP row.  Lane gl of a row owns
// the sector elements lsp + gl + 16 t (sector length <= 16 * FE_T).  Needs suppress_radius <= 7: the forward checks sit in
// lanes 0-7 of the row, the backward checks in lanes 8-15, and the 2 * radius + 1 marked elements get one lane each.  Dynamic LDS: 1 B per ring point, 4 rings.
// STDSORT (sort_mode 2): as in fe_pick — when several candidates of a row tie for the best curvature, the whole wavefront computes
// where libstdc++'s std::sort would have left that ring's sector (stdsort_emu.h), once per ring and sector, and the row picks by it.
#define FP_G 4
template <int FE_T, bool STDSORT = false>
__global__ void __launch_bounds__(64) fe_pick4(DevCtx d) {
  using mask_t = typename std::conditional<(FE_T > 32), unsigned long long, uint32_t>::type;
  __shared__ typename std::conditional<STDSORT, SortEmu<16 * FE_T>, char>::type s_emu;
  __shared__ uint16_t s_pos[STDSORT ? FP_G : 1][STDSORT ? 16 * FE_T : 1];
  static_assert(FE_T <= 64, "one candidate bit per owned element");
  const int slot = blockIdx.y + d.slot0, lane = threadIdx.x, g = lane >> 4, gl = lane & 15;
  const int ring0 = blockIdx.x * FP_G, ring = ring0 + g;
  const bool rv = ring < d.NS;
  const size_t base = (size_t)slot * d.N;
  const alego_params& P = d.P;
  extern __shared__ __attribute__((aligned(16))) unsigned char fe_smem[];
  // 1 B per ring point: the kernel's LDS footprint (one wavefront, four rings) is what limits how many of these workgroups —
  // and how much of the other stream groups' work — fit a CU.  The suppression only ever asks whether |col[k+1] - col[k]|
  // exceeds suppress_col_diff: that is bit 6 of the point's flag byte.
  uint8_t* s_flag = fe_smem;                                  // [FP_G][H]
  for (int r = 0; r < FP_G && ring0 + r < d.NS; ++r) {       // whole wavefront stages one ring after the other
    const int Sr = d.ring_start[slot * d.NS + ring0 + r], Er = d.ring_end[slot * d.NS + ring0 + r];
    const int rfr = Sr - 5, cntr = Er - Sr + 11;
    uint8_t* sf = s_flag + (size_t)r * d.H;
    // the flag byte of every point comes ready from fe_curv_chunk (picked | ground << 1 | curvature > edge_thres << 2 | curvature <
    // surf_thres << 3 | label 0 << 4 | column jump to the next point << 6); the jump bit of the ring's last point compares with
    // the point itself in the reference's suppression loop (clamped index): cleared here
    const uint8_t* ff = d.fe_flag + base + rfr;
#pragma unroll 8
    for (int k = lane; k < cntr; k += 64) sf[k] = k == cntr - 1 ? (uint8_t)(ff[k] & ~64) : ff[k];
  }
  __syncthreads();
  const int S = rv ? d.ring_start[slot * d.NS + ring] : 0, E = rv ? d.ring_end[slot * d.NS + ring] : 0;
  const int rf = S - 5;
  uint8_t* sf = s_flag + (size_t)g * d.H;
  int* st = d.st_idx + ((size_t)slot * d.NS + (rv ? ring : 0)) * d.st_stride;
  int* st_sharp = st, *st_lsharp = st + d.cap_sharp, *st_flat = st_lsharp + d.cap_lsharp;
  int n_sharp = 0, n_ls = 0, n_flat = 0;
  const int NSEC = P.n_sectors, SR = P.suppress_radius;
  for (int j = 0; j < NSEC; ++j) {
    int sp, ep;
    if (P.sector_formula == 0) { sp = (S * (NSEC - j) + E * j) / NSEC; ep = (S * (NSEC - 1 - j) + E * (j + 1)) / NSEC - 1; }
    else { const int diff = E - S; sp = S + j * diff / NSEC; ep = S + (j + 1) * diff / NSEC - 1; }
    const bool act0 = rv && sp < ep;
    if (!__any(act0)) continue;
    const int lsp = sp - rf, lep = ep - rf;
    uint32_t key[FE_T];
    mask_t sharp_m = 0, flat_m = 0;
#pragma unroll
    for (int t = 0; t < FE_T; ++t) {
      // unconditional loads (clamped address): all FE_T of them in flight together.  Behind a branch the compiler waited
      // for each load before it issued the next one: 19 global round trips per sector.
      const int c = lsp + gl + 16 * t;
      const bool ok = act0 && c <= lep;
      const uint8_t f = sf[ok ? c : 0];
      const uint32_t kv = (uint32_t)d_f2i(fabsf(d.cd[base + (ok ? rf + c : 0)]));
      key[t] = ok ? kv : 0u;
      if (ok && (f & 7) == 4) sharp_m |= (mask_t)1 << t;
      if (ok && (f & 11) == 10) flat_m |= (mask_t)1 << t;
    }
    // One greedy pick is a few hundred DEPENDENT instructions of a single wavefront (126 of them per ring quad): the loop body
    // is written branch-free.  LDS flag bytes are changed with no-return 32-bit atomics on the containing word (no read /
    // wait / write round trip; an operand of 0 or a clamped address makes a lane's update a no-op), loads use clamped
    // addresses, and only the list stores sit behind a (single) branch.
    unsigned* fw = reinterpret_cast<unsigned*>(s_flag);
    const int fo = g * d.H;                               // this ring's first flag byte
    // marks c and its +-SR neighbours picked (:211-234) for the rows where `on`; `spread` rows look for column jumps first
    auto mark = [&](int c, bool on, bool spread) {
      const int role = gl & 7, rr = max(min(role, SR - 1), 0);   // lanes 0-7 look forward, 8-15 backward
      const int a = spread ? (gl < 8 ? c + rr : c - rr - 1) : 0;
      const bool bad = spread && role < SR && (sf[a] & 64);
      const unsigned long long mb = __ballot(bad);
      const unsigned gb = (unsigned)(mb >> (16 * g)) & 0xffffu;
      const unsigned lo = gb & 0xffu, hi = gb >> 8;
      const int nf = spread ? (lo ? min(SR, __ffs((int)lo) - 1) : SR) : 0, nbk = spread ? (hi ? min(SR, __ffs((int)hi) - 1) : SR) : 0;
      const int first = c - nbk, last = c + nf;
      const bool w = on && gl <= last - first;
      const int bo = fo + (w ? first + gl : 0);
      atomicOr(&fw[bo >> 2], (w ? 1u : 0u) << ((bo & 3) * 8));
      const int off = (gl - (first - lsp)) & 15;   // the (single) owned element inside [first, last]
      const int ct = first + off;
      const mask_t bit = (on && ct <= last && ct >= lsp && ct <= lep) ? (mask_t)1 << (((ct - lsp - gl) >> 4) & (8 * (int)sizeof(mask_t) - 1)) : (mask_t)0;
      sharp_m &= ~bit; flat_m &= ~bit;
    };
    // the label bits (4-5) of an unlabelled element hold 1 (label 0): 1 ^ 2 = 3 (sharp), 1 ^ 3 = 2 (less sharp), 1 ^ 1 = 0 (flat);
    // an element is labelled at most once (it is marked picked with it)
    auto relabel = [&](int c, unsigned x) {
      const int bo = fo + (x ? c : 0);
      atomicXor(&fw[bo >> 2], (x << 4) << ((bo & 3) * 8));
    };
    int* st_dummy = d.st_cnt + ((size_t)slot * d.NS + (rv ? ring : 0)) * 8 + 7;   // unused field
    // (STDSORT) the sector arrangement of every ring whose row reports a tie and has none yet, ring by ring with all 64 lanes
    unsigned have_arr = 0;   // wavefront-uniform: bit r = ring r's arrangement of this sector is in s_pos[r]
    auto ring_arrangements = [&](bool tied) {
      if constexpr (STDSORT) {
        for (int r = 0; r < FP_G; ++r) {
          const bool want = __shfl((int)tied, 16 * r, 64) != 0;
          if (!want || ((have_arr >> r) & 1u)) continue;
          const int lsp_r = __shfl(lsp, 16 * r, 64), lep_r = __shfl(lep, 16 * r, 64), rf_r = __shfl(rf, 16 * r, 64);
          const int n = lep_r - lsp_r + 1;
          __syncthreads();
          for (int i = lane; i < n; i += 64) s_emu.ak[i] = (uint32_t)d_f2i(fabsf(d.cd[base + rf_r + lsp_r + i]));
          __syncthreads();
          stdsort_arrangement(s_emu, n);
          for (int i = lane; i < n; i += 64) s_pos[r][i] = s_emu.pos[i];
          __syncthreads();
          have_arr |= 1u << r;
        }
      }
    };
    // ---- sharp / less-sharp: descending curvature, ties -> larger index (:189-236) ----
    int picked_num = 0;
    bool act = act0;
    while (true) {
      uint32_t bk = 0;
      int bt = 0;
#pragma unroll
      for (int t = 0; t < FE_T; ++t) if (((sharp_m >> t) & 1) && key[t] >= bk) { bk = key[t]; bt = t; }
      if (!act) bk = 0;
      const uint32_t kmax = row16_max_u32(bk);
      act = act && kmax != 0;
      if (!__any(act)) break;
      int c = (int)row16_max_u32(bk == kmax ? (uint32_t)(lsp + gl + 16 * bt) : 0u);   // ties -> larger index
      if constexpr (STDSORT) {
        int cnt = 0;
#pragma unroll
        for (int t = 0; t < FE_T; ++t) cnt += (((sharp_m >> t) & 1) && key[t] == kmax) ? 1 : 0;
        if (!act) cnt = 0;
        const bool tied = row16_max_u32((uint32_t)cnt) > 1u || __popc((unsigned)(__ballot(cnt > 0) >> (16 * g)) & 0xffffu) > 1;
        if (__any(tied)) {
          ring_arrangements(tied);
          uint32_t best = 0;   // k = ep .. sp: of the tied candidates the one std::sort placed last comes first
#pragma unroll
          for (int t = 0; t < FE_T; ++t)
            if (((sharp_m >> t) & 1) && key[t] == kmax) best = max(best, (((uint32_t)s_pos[g][gl + 16 * t] + 1u) << 16) | (uint32_t)(gl + 16 * t));
          const int ca = lsp + (int)(row16_max_u32(tied ? best : 0u) & 0xFFFFu);
          c = tied ? ca : c;
        }
      }
      picked_num += act ? 1 : 0;
      const int lab = picked_num <= P.n_sharp ? 2 : (picked_num <= P.n_less_sharp ? 1 : 0);
      const bool w1 = act && gl == 0 && lab != 0;
      relabel(c, w1 ? (lab == 2 ? 2u : 3u) : 0u);
      if (w1) {
        int* ps = lab == 2 ? st_sharp + n_sharp : st_dummy;
        *ps = c + rf;
        st_lsharp[n_ls] = c + rf;
      }
      n_sharp += (act && lab == 2) ? 1 : 0;
      n_ls += (act && lab) ? 1 : 0;
      mark(c, act, act && lab != 0);  // the 21st pick is marked but breaks before the suppression (:207-210)
      act = act && lab != 0;
    }
    // ---- flat: ascending curvature, ground only, ties -> smaller index (:238-277) ----
    picked_num = 0;
    act = act0;
    while (true) {
      uint32_t bk = 0xFFFFFFFFu;
      int bt = 0;
#pragma unroll
      for (int t = 0; t < FE_T; ++t) if (((flat_m >> t) & 1) && key[t] < bk) { bk = key[t]; bt = t; }
      if (!act) bk = 0xFFFFFFFFu;
      const uint32_t kmin = row16_min_u32(bk);
      act = act && kmin != 0xFFFFFFFFu;
      if (!__any(act)) break;
      int c = (int)row16_min_u32(bk == kmin ? (uint32_t)(lsp + gl + 16 * bt) : 0xFFFFFFFFu);   // ties -> smaller index
      if constexpr (STDSORT) {
        int cnt = 0;
#pragma unroll
        for (int t = 0; t < FE_T; ++t) cnt += (((flat_m >> t) & 1) && key[t] == kmin) ? 1 : 0;
        if (!act) cnt = 0;
        const bool tied = row16_max_u32((uint32_t)cnt) > 1u || __popc((unsigned)(__ballot(cnt > 0) >> (16 * g)) & 0xffffu) > 1;
        if (__any(tied)) {
          ring_arrangements(tied);
          uint32_t best = 0xFFFFFFFFu;   // k = sp .. ep: the tied candidate std::sort placed first
#pragma unroll
          for (int t = 0; t < FE_T; ++t)
            if (((flat_m >> t) & 1) && key[t] == kmin) best = min(best, ((uint32_t)s_pos[g][gl + 16 * t] << 16) | (uint32_t)(gl + 16 * t));
          const int ca = lsp + (int)(row16_min_u32(tied ? best : 0xFFFFFFFFu) & 0xFFFFu);
          c = tied ? ca : c;
        }
      }
      picked_num += act ? 1 : 0;
      const bool w1 = act && gl == 0;
      relabel(c, w1 ? 1u : 0u);   // label -1
      if (w1) st_flat[n_flat] = c + rf;
      n_flat += act ? 1 : 0;
      const bool stop = picked_num >= P.n_flat;
      mark(c, act, act && !stop);  // the n_flat-th pick breaks before the suppression (:248-251)
      act = act && !stop;
    }
  }
  __syncthreads();
  if (rv && gl == 0) {
    int* c = d.st_cnt + ((size_t)slot * d.NS + ring) * 8;
    c[0] = n_sharp; c[1] = n_ls; c[2] = n_flat;
  }
  // ---- less-flat candidates in position order (:279-285) and the labels, one ring after the other with all 64 lanes.
  // A sector's labels only change while that sector is picked, so reading them at the end is the same as reading them
  // after the sector.
  for (int r = 0; r < FP_G && ring0 + r < d.NS; ++r) {
    const int Sr = d.ring_start[slot * d.NS + ring0 + r], Er = d.ring_end[slot * d.NS + ring0 + r];
    const int rfr = Sr - 5, cntr = Er - Sr + 11;
    const uint8_t* sfr = s_flag + (size_t)r * d.H;
    int* st_lfs = d.st_idx + ((size_t)slot * d.NS + ring0 + r) * d.st_stride + d.cap_sharp + d.cap_lsharp + d.cap_flat;
    int n_lfs = 0;
    for (int j = 0; j < NSEC; ++j) {
      int sp, ep;
      if (P.sector_formula == 0) { sp = (Sr * (NSEC - j) + Er * j) / NSEC; ep = (Sr * (NSEC - 1 - j) + Er * (j + 1)) / NSEC - 1; }
      else { const int diff = Er - Sr; sp = Sr + j * diff / NSEC; ep = Sr + (j + 1) * diff / NSEC - 1; }
      if (sp >= ep) continue;
      const int lsp = sp - rfr, lep = ep - rfr;
      for (int c0 = lsp; c0 <= lep; c0 += 64) {
        const int c = c0 + lane;
        const bool take = c <= lep && ((sfr[c] >> 4) & 3) <= 1;   // label <= 0
        const unsigned long long m = __ballot(take);
        if (take) st_lfs[n_lfs + (int)__popcll(m & ((1ull << lane) - 1ull))] = c + rfr;
        n_lfs += (int)__popcll(m);
      }
    }
    for (int k = lane; k < cntr; k += 64) d.plabel[base + rfr + k] = (int)((sfr[k] >> 4) & 3) - 1;
    if (lane == 0) d.st_cnt[((size_t)slot * d.NS + ring0 + r) * 8 + 3] = n_lfs;
  }
}

// one block per (ring, slot): pcl::VoxelGrid on the ring's less_flat_scan.
// Consecutive points of a ring mostly fall into the same 0.4 m voxel, so the (voxel id, position) sort is done on
// RUNS of equal consecutive voxel ids (a few hundred per ring instead of ~1000 points): rank-by-counting of the
// runs in LDS, then the first run of every voxel accumulates all runs of that voxel in order — the same f32
// summation order as a stable sort of the points.  Dynamic LDS: 10 B per ring point (+ 4 KB of bucket counters): the
// number of these workgroups that fit a CU decides the kernel's duration.
#ifdef ALEGO_TIMING
__device__ long long fv_times[12];
extern "C" void alego_fv_times(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(fv_times), sizeof(long long) * 12); }
#define FV_TICK(k) do { if (threadIdx.x == 0 && blockIdx.x == 8 && blockIdx.y == 0) fv_times[k] = wall_clock64(); } while (0)
#else
#define FV_TICK(k)
#endif
#ifndef FV_NB
#define FV_NB 256
#endif
#ifndef FV_BLOCK
#define FV_BLOCK 256   // threads per ring
#endif
#ifndef FV_STAGE
#define FV_STAGE 1     // 1: the ring's candidate points are gathered once into LDS (26 B per column); 0: every pass gathers them from HBM / L2 (10 B per column)
#endif
// (ranking the ~300 runs of a ring by counting — every run sweeps all runs in LDS — was measured: 346 k -> 317 k scans/s.  The
//  pipeline as a whole is VALU-issue bound; 90 k compares per ring cost more than the bucket tables' latency.)
#define FV_LDS_PER_COL (FV_STAGE ? 26 : 10)
// Points staged in LDS per ring: a ring of H columns holds ~0.6 H less_flat_scan points (the rest are empty, ground-only or picked cells); the
// staging area is sized for 0.72 H of them and a fuller ring reads its tail from the L2 on every pass.  10 H + 16 x 0.72 H + 2.2 KB = 40.9 KB at
// H = 1800: FOUR rings per CU instead of three (26 H + 4.2 KB = 51 KB).  Same run, scans/s: 512 buckets + everything staged 395 k; 512 buckets +
// 0.64 H staged 406 k; 256 buckets + 0.72 H staged 407 k; nothing staged (three gathers through the L2 per point) 402 k.
#ifndef FV_CAP_PCT
#define FV_CAP_PCT 72
#endif
// Wide images (H = 4000: the key arrays alone are 40 KB) stage less or nothing, so that the workgroup stays near 40 KB: a ring of 16 x 4000 with
// everything staged took 88 KB — one workgroup per CU, fe_voxel a quarter of that geometry's device time.
__host__ __device__ inline int fv_stage_cap(int H) {
  if (!FV_STAGE) return 0;
  const int by_pct = (H * FV_CAP_PCT / 100 + 15) & ~15, by_lds = (40960 - 2200 - 10 * H) / 16;
  return by_lds <= 0 ? 0 : (by_pct < by_lds ? by_pct : (by_lds & ~15));
}
static size_t fv_lds_bytes(int H) { return (size_t)10 * H + (size_t)16 * fv_stage_cap(H); }
#define FV_U 4    // gathers kept in flight per thread
__global__ void __launch_bounds__(FV_BLOCK) fe_voxel(DevCtx d) {
  const int slot = blockIdx.y + d.slot0, ring = blockIdx.x, tid = threadIdx.x;
  const size_t base = (size_t)slot * d.N;
  int* cnts = d.st_cnt + ((size_t)slot * d.NS + ring) * 8;
  const int n = cnts[3];
  const int* lfs = d.st_idx + ((size_t)slot * d.NS + ring) * d.st_stride + d.cap_sharp + d.cap_lsharp + d.cap_flat;
  float4* out = d.st_lfds + ((size_t)slot * d.NS + ring) * d.H;
  const float4* seg = d.seg_lo + base;
  extern __shared__ __attribute__((aligned(16))) unsigned char fv_smem[];
  float4* s_pt = reinterpret_cast<float4*>(fv_smem);                            // the ring's less_flat_scan points, gathered ONCE [cap] (FV_STAGE)
  const int cap = fv_stage_cap(d.H);
  unsigned char* fv2 = fv_smem + (size_t)16 * cap;
  auto point = [&](int i) -> float4 { if (i < cap) return s_pt[i]; return seg[lfs[i]]; };
  uint32_t* s_key = reinterpret_cast<uint32_t*>(fv2);                           // voxel id per point      [H]
  uint32_t* s_rvid = s_key;                                                     // voxel id per run, compacted in place (run r <= its first point)
  uint16_t* s_rstart = reinterpret_cast<uint16_t*>(fv2 + 4 * (size_t)d.H);      // first point of the run  [H]
  uint16_t* s_order = reinterpret_cast<uint16_t*>(fv2 + 6 * (size_t)d.H);       // runs sorted by (voxel id, run) [H]
  uint16_t* s_tmp = reinterpret_cast<uint16_t*>(fv2 + 8 * (size_t)d.H);         // runs dealt into buckets [H]
  __shared__ float s_red[6][FV_BLOCK / 64];
  __shared__ int s_scan[FV_BLOCK / 64];
  __shared__ int s_boff[FV_NB + 1], s_bcur[FV_NB + 1];
  if (n == 0) { if (tid == 0) cnts[4] = 0; return; }
  const float inv = 1.0f / d.P.less_flat_leaf;
  FV_TICK(0);
  // getMinMax3D
  float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  // gathers through the index list: FV_U index loads, then FV_U point loads in flight together
  for (int i0 = tid; i0 < n; i0 += FV_BLOCK * FV_U) {
    int ix[FV_U];
    float4 pt[FV_U];
#pragma unroll
    for (int u = 0; u < FV_U; ++u) ix[u] = lfs[min(i0 + u * FV_BLOCK, n - 1)];
#pragma unroll
    for (int u = 0; u < FV_U; ++u) pt[u] = seg[ix[u]];
#pragma unroll
    for (int u = 0; u < FV_U; ++u) {   // (a clamped duplicate of the last point does not change min / max)
      if (i0 + u * FV_BLOCK < min(n, cap)) s_pt[i0 + u * FV_BLOCK] = pt[u];   // every later pass reads the point from LDS: one global round trip instead of three
      mn[0] = fminf(mn[0], pt[u].x); mn[1] = fminf(mn[1], pt[u].y); mn[2] = fminf(mn[2], pt[u].z);
      mx[0] = fmaxf(mx[0], pt[u].x); mx[1] = fmaxf(mx[1], pt[u].y); mx[2] = fmaxf(mx[2], pt[u].z);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64)); }
    if ((tid & 63) == 0) { s_red[a][tid >> 6] = mn[a]; s_red[3 + a][tid >> 6] = mx[a]; }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mn[a] = s_red[a][0]; mx[a] = s_red[3 + a][0];
#pragma unroll
    for (int w = 1; w < FV_BLOCK / 64; ++w) { mn[a] = fminf(mn[a], s_red[a][w]); mx[a] = fmaxf(mx[a], s_red[3 + a][w]); }
  }
  const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > 2147483647LL) {  // "leaf size too small": the input is returned unchanged
    for (int i = tid; i < n; i += FV_BLOCK) out[i] = point(i);
    if (tid == 0) cnts[4] = n;
    return;
  }
  FV_TICK(1);
  int minb[3], divb[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    minb[a] = (int)floorf(mn[a] * inv);
    divb[a] = (int)floorf(mx[a] * inv) - minb[a] + 1;
  }
  const int mul1 = divb[0], mul2 = divb[0] * divb[1];
#pragma unroll 4
  for (int i = tid; i < n; i += FV_BLOCK) {
    const float4 q = point(i);
    const int i0 = (int)(floorf(q.x * inv) - (float)minb[0]);
    const int i1 = (int)(floorf(q.y * inv) - (float)minb[1]);
    const int i2 = (int)(floorf(q.z * inv) - (float)minb[2]);
    s_key[i] = (uint32_t)(i0 + i1 * mul1 + i2 * mul2);
  }
  __syncthreads();
  FV_TICK(2);
  // runs of consecutive equal voxel ids
  int nruns = 0;
  for (int c0 = 0; c0 < n; c0 += FV_BLOCK) {
    const int i = c0 + tid;
    const uint32_t mykey = i < n ? s_key[i] : 0u;   // read before the barrier: the run ids are compacted into the same array
    const bool head = i < n && (i == 0 || mykey != s_key[i - 1]);
    const unsigned long long m = __ballot(head);
    if ((tid & 63) == 0) s_scan[tid >> 6] = (int)__popcll(m);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < FV_BLOCK / 64; ++w) { if (w < (tid >> 6)) woff += s_scan[w]; tot += s_scan[w]; }
    if (head) {
      const int r = nruns + woff + (int)__popcll(m & ((1ull << (tid & 63)) - 1ull));
      s_rvid[r] = mykey;
      s_rstart[r] = (uint16_t)i;
    }
    nruns += tot;
    __syncthreads();
  }
  FV_TICK(3);
  // Order the runs by (voxel id, run index).  Voxel ids are bounded by the grid size T, so the runs are first dealt
  // into <= FV_NB buckets that are monotone in the voxel id (LDS atomics; arbitrary order inside a bucket), then
  // every run ranks itself among the one or two runs of its bucket — instead of against all runs of the ring.
  {
    unsigned T = (unsigned)divb[0] * (unsigned)divb[1] * (unsigned)divb[2];
    if (T == 0) T = 1;
    int shift = 0;
    while (((T - 1) >> shift) >= (unsigned)FV_NB) ++shift;
    const int nb = (int)((T - 1) >> shift) + 1;
    for (int b = tid; b <= nb; b += FV_BLOCK) s_boff[b] = 0;
    __syncthreads();
    for (int r = tid; r < nruns; r += FV_BLOCK) atomicAdd(&s_boff[min((int)(s_rvid[r] >> shift), nb - 1) + 1], 1);
    __syncthreads();
    // inclusive scan of s_boff[1..nb] (FV_NB / FV_BLOCK consecutive entries per thread) -> bucket b = [s_boff[b], s_boff[b+1])
    {
      constexpr int PER = FV_NB / FV_BLOCK;
      int v[PER], sum = 0;
#pragma unroll
      for (int k = 0; k < PER; ++k) { const int b = tid * PER + k; v[k] = b < nb ? s_boff[b + 1] : 0; sum += v[k]; }
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += t; }
      if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
      __syncthreads();
      int run = incl - sum;
#pragma unroll
      for (int w = 0; w < FV_BLOCK / 64; ++w) if (w < (tid >> 6)) run += s_scan[w];
#pragma unroll
      for (int k = 0; k < PER; ++k) { const int b = tid * PER + k; run += v[k]; if (b < nb) { s_boff[b + 1] = run; s_bcur[b + 1] = run; } }
      if (tid == 0) s_bcur[0] = 0;
    }
    __syncthreads();
    for (int r = tid; r < nruns; r += FV_BLOCK) {
      const int b = min((int)(s_rvid[r] >> shift), nb - 1);
      s_tmp[atomicAdd(&s_bcur[b], 1)] = (uint16_t)r;  // s_bcur[b] starts at s_boff[b] (written one slot up, read one down)
    }
    __syncthreads();
    for (int t = tid; t < nruns; t += FV_BLOCK) {
      const int r = s_tmp[t];
      const uint32_t v = s_rvid[r];
      const int b = min((int)(v >> shift), nb - 1);
      const int bs = s_boff[b], be = s_boff[b + 1];
      int rank = bs;
      for (int q = bs; q < be; ++q) { const int o = s_tmp[q]; const uint32_t u = s_rvid[o]; rank += (u < v) || (u == v && o < r); }
      s_order[rank] = (uint16_t)r;
    }
  }
  __syncthreads();
  FV_TICK(4);
  // first run of every voxel -> output rank; it accumulates all runs of the voxel in order
  int nvox = 0;
  for (int c0 = 0; c0 < nruns; c0 += FV_BLOCK) {
    const int j = c0 + tid;
    const bool head = j < nruns && (j == 0 || s_rvid[s_order[j]] != s_rvid[s_order[j - 1]]);
    const unsigned long long m = __ballot(head);
    if ((tid & 63) == 0) s_scan[tid >> 6] = (int)__popcll(m);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < FV_BLOCK / 64; ++w) { if (w < (tid >> 6)) woff += s_scan[w]; tot += s_scan[w]; }
    if (head) {
      const int rank = nvox + woff + (int)__popcll(m & ((1ull << (tid & 63)) - 1ull));
      const uint32_t vid = s_rvid[s_order[j]];
      float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
      int c = 0;
      for (int jj = j; jj < nruns && s_rvid[s_order[jj]] == vid; ++jj) {
        const int r = s_order[jj], i0 = s_rstart[r], len = (r + 1 < nruns ? (int)s_rstart[r + 1] : n) - i0;
        for (int i = i0; i < i0 + len; ++i) { const float4 q = point(i); sx += q.x; sy += q.y; sz += q.z; si += q.w; ++c; }   // strictly in order
      }
      const float fn = (float)c;
      out[rank] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
    }
    nvox += tot;
    __syncthreads();
  }
  if (tid == 0) cnts[4] = nvox;
  FV_TICK(5);
}

// fe_collect = ring-ascending concatenation of the four feature clouds (:199-205,:245,:293) + the bounding boxes of every LO_CH
// consecutive less_flat / less_sharp points for the next scan's LaserOdometry (kernels_lo.hip), one workgroup per stream (was
// fe_gather + fe_boxes: 16 + 48 workgroups per stream in two launches, the second one re-reading what the first had written).
// Output point i of a cloud belongs to the ring r with off[r] <= i < off[r + 1] (binary search in the per-ring prefix, LDS);
// 32 consecutive threads hold the 32 points of a box.
#define FC_T 512
__global__ void __launch_bounds__(FC_T) fe_collect(DevCtx d) {
  const int slot = blockIdx.x + d.slot0, tid = threadIdx.x, lane = tid & 63;
  const int cur = cur_in_flight(d, slot);
  const size_t base = (size_t)slot * d.N;
  const int NS = d.NS;
  __shared__ int s_off[4][65], s_boff[2][65];   // s_boff: first box of every ring (less_sharp, less_flat): boxes never straddle rings
  const int* allc = d.st_cnt + (size_t)slot * NS * 8;
  if (tid < 64) {
    const int r = tid;
    int c[4];
    c[0] = r < NS ? allc[r * 8 + 0] : 0; c[1] = r < NS ? allc[r * 8 + 1] : 0;
    c[2] = r < NS ? allc[r * 8 + 2] : 0; c[3] = r < NS ? allc[r * 8 + 4] : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int incl = c[k];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      s_off[k][r] = incl - c[k];
      if (r == 63) s_off[k][64] = incl;
      if (k == 1 || k == 3) {
        const int nb = (c[k] + LO_CH - 1) / LO_CH;
        int bi = nb;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(bi, o, 64); if (lane >= o) bi += t; }
        s_boff[k == 1 ? 0 : 1][r] = bi - nb;
        if (r == 63) s_boff[k == 1 ? 0 : 1][64] = bi;
      }
    }
  }
  __syncthreads();
  const int tot[4] = {s_off[0][64], s_off[1][64], s_off[2][64], s_off[3][64]};
  if (tid <= NS) {
    int* ro = d.ring_off + (((size_t)slot * 2 + cur) * 2) * (NS + 1);
    ro[tid] = tid < NS ? s_off[1][tid] : tot[1];
    ro[(NS + 1) + tid] = tid < NS ? s_off[3][tid] : tot[3];
    int* rb = d.ring_boff + (((size_t)slot * 2 + cur) * 2) * (NS + 1);
    rb[tid] = tid < NS ? s_boff[0][tid] : s_boff[0][64];
    rb[(NS + 1) + tid] = tid < NS ? s_boff[1][tid] : s_boff[1][64];
  }
  if (tid == 0) {
    int* fc = d.feat_cnt + ((size_t)slot * 2 + cur) * 4;
    fc[0] = tot[0]; fc[1] = tot[1]; fc[2] = tot[2]; fc[3] = tot[3];
  }
  const float4* seg = d.seg_lo + base;
  auto ring_of = [&](int k, int i) -> int {   // largest r with s_off[k][r] <= i (rings beyond NS hold the total: never chosen for i < total)
    int lo = 0, hi = NS - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_off[k][mid] <= i) lo = mid; else hi = mid - 1; }
    return lo;
  };
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float4* dst = d.feat[k] + ((size_t)slot * 2 + cur) * d.fcap[k];
    int* dsti = k < 3 ? d.feat_idx[k] + ((size_t)slot * 2 + cur) * d.fcap[k] : nullptr;
    const bool boxes = k == F_LSHARP || k == F_LFLAT;
    float4* bx = boxes ? d.lo_box + (((size_t)slot * 2 + cur) * 2 + (k == F_LFLAT ? 0 : 1)) * d.lo_box_cap * 2 : nullptr;
    const int stoff = k == 0 ? 0 : (k == 1 ? d.cap_sharp : d.cap_sharp + d.cap_lsharp);
    // clouds with boxes are walked box by box (LO_CH consecutive threads = the up to LO_CH points of one ring's box), the others point by point
    const int kb = k == F_LSHARP ? 0 : 1;
    const int nwork = boxes ? s_boff[kb][64] * LO_CH : tot[k];
    for (int i0 = 0; i0 < nwork; i0 += FC_T) {
      int i = i0 + tid, r = 0, j = 0, bxi = 0;
      bool v = i < nwork;
      if (boxes) {
        bxi = min(i / LO_CH, s_boff[kb][64] - 1);
        int lo = 0, hi = NS - 1;   // largest r with s_boff[r] <= bxi
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_boff[kb][mid] <= bxi) lo = mid; else hi = mid - 1; }
        r = lo; j = (bxi - s_boff[kb][r]) * LO_CH + (tid % LO_CH);
        v = v && j < s_off[k][r + 1] - s_off[k][r];
        i = s_off[k][r] + j;
      }
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v) {
        if (!boxes) { r = ring_of(k, i); j = i - s_off[k][r]; }
        if (k < 3) {
          const int idx = d.st_idx[((size_t)slot * NS + r) * d.st_stride + stoff + j];
          p = seg[idx];
          dsti[i] = idx;
        } else {
          p = d.st_lfds[((size_t)slot * NS + r) * d.H + j];
        }
        dst[i] = p;
      }
      if (boxes) {
        float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
        if (v) { mn[0] = mx[0] = p.x; mn[1] = mx[1] = p.y; mn[2] = mx[2] = p.z; }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int o = LO_CH / 2; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64)); }
        if ((tid % LO_CH) == 0 && v) {   // (thread 0 of a box holds its first point: every box has one)
          const int len = min(LO_CH, s_off[k][r + 1] - i);
          bx[2 * bxi] = make_float4(mn[0], mn[1], mn[2], __int_as_float(i));
          bx[2 * bxi + 1] = make_float4(mx[0], mx[1], mx[2], __int_as_float(len));
        }
      }
    }
  }
}

// alego_debug_std_sort: the emulation on its own (one wavefront, up to 4096 keys)
#define STDSORT_PROBE_MAX 4096
__global__ void __launch_bounds__(64) stdsort_probe(const uint32_t* keys, int n, int depth_limit, int* pos_out) {
  __shared__ SortEmu<STDSORT_PROBE_MAX> E;
  for (int i = threadIdx.x; i < n; i += 64) E.ak[i] = keys[i];
  __syncthreads();
  stdsort_arrangement(E, n, depth_limit);
  for (int i = threadIdx.x; i < n; i += 64) pos_out[i] = E.pos[i];
}
int launch_stdsort_probe(const uint32_t* keys, int n, int depth_limit, int* pos_out, hipStream_t st) {
  if (n > STDSORT_PROBE_MAX) return -1;
  hipLaunchKernelGGL(stdsort_probe, dim3(1), dim3(64), 0, st, keys, n, depth_limit, pos_out);
  return 0;
}

void launch_lo_grid(const DevCtx& d, hipStream_t st);        // kernels_lo.hip: the target grid of the clouds just written, for the next scan's LaserOdometry
bool fe_fused_eligible(const DevCtx& d);                    // kernels_fe2.hip
void launch_fe_fused(const DevCtx& d, hipStream_t st);
void launch_fe_curv_debug(const DevCtx& d, hipStream_t st) { ALEGO_LAUNCH(fe_curv, dim3((d.N + FE_CW - 1) / FE_CW, d.n_launch), dim3(FE_BLOCK), 0, st, d); }

void launch_fe(const DevCtx& d, hipStream_t st) {
  if (fe_fused_eligible(d)) { launch_fe_fused(d, st); launch_lo_grid(d, st); return; }   // fe_cand + fe_pickc + fe_ring_out; below: the four-kernel path (ALEGO_FE_FUSED=0, sort_mode 2)
  // dynamic LDS above 64 KB has to be requested explicitly (fe_voxel: 26 B per column, horizon_scan <= 4096)
  static const bool cfg = hipFuncSetAttribute(reinterpret_cast<const void*>(fe_voxel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fv_lds_bytes(FE_MAXH)) == hipSuccess;
  (void)cfg;
  ALEGO_LAUNCH(fe_curv, dim3((d.N + FE_CW - 1) / FE_CW, d.n_launch), dim3(FE_BLOCK), 0, st, d);
  // the longest sector holds at most ceil(H / n_sectors) + 1 points
  const int sector_max = (d.H + d.P.n_sectors - 1) / (d.P.n_sectors > 0 ? d.P.n_sectors : 1) + 2;
  // (padding this allocation by 16 KB cost 7 % of the whole pipeline: the LDS footprint decides how many rings share a CU)
  const int extra = 0;
  const bool one_ring = d.opt_fe_pick1 || d.P.suppress_radius > 7 || sector_max > 16 * 43;   // (2 * radius + 1 marked elements <= 16 lanes)
  const dim3 g4((d.NS + FP_G - 1) / FP_G, d.n_launch);
  const size_t lds4 = (size_t)FP_G * d.H;
  const bool ss = d.P.sort_mode == 2;
  if (!one_ring && ss && sector_max <= 16 * 19) { ALEGO_LAUNCH((fe_pick4<19, true>), g4, dim3(64), lds4, st, d); }
  else if (!one_ring && ss && sector_max <= 16 * 24) { ALEGO_LAUNCH((fe_pick4<24, true>), g4, dim3(64), lds4, st, d); }
  else if (!one_ring && ss) { ALEGO_LAUNCH((fe_pick4<43, true>), g4, dim3(64), lds4, st, d); }
  else if (!one_ring && sector_max <= 16 * 19) { ALEGO_LAUNCH(fe_pick4<19>, g4, dim3(64), lds4, st, d); }
  else if (!one_ring && sector_max <= 16 * 24) { ALEGO_LAUNCH(fe_pick4<24>, g4, dim3(64), lds4, st, d); }
  else if (!one_ring) { ALEGO_LAUNCH(fe_pick4<43>, g4, dim3(64), lds4, st, d); }
  else if (d.P.sort_mode == 2 && sector_max <= 64 * 6) { ALEGO_LAUNCH((fe_pick<6, true>), dim3(d.NS, d.n_launch), dim3(64), (size_t)3 * d.H, st, d); }
  else if (d.P.sort_mode == 2) { ALEGO_LAUNCH((fe_pick<12, true>), dim3(d.NS, d.n_launch), dim3(64), (size_t)3 * d.H, st, d); }
  else if (sector_max <= 64 * 6) { ALEGO_LAUNCH((fe_pick<6, false>), dim3(d.NS, d.n_launch), dim3(64), (size_t)3 * d.H + extra, st, d); }
  else { ALEGO_LAUNCH((fe_pick<12, false>), dim3(d.NS, d.n_launch), dim3(64), (size_t)3 * d.H, st, d); }
  ALEGO_LAUNCH(fe_voxel, dim3(d.NS, d.n_launch), dim3(FV_BLOCK), fv_lds_bytes(d.H), st, d);
  ALEGO_LAUNCH(fe_collect, dim3(d.n_launch), dim3(FC_T), 0, st, d);
  launch_lo_grid(d, st);
}
