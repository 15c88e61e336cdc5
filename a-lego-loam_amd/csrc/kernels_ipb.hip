// kernels_ipb.hip — ImageProjection for sensors with more than 16 rings (64 x 2048 = BASELINE config 5's geometry, 32 x 2048, 40 x 1800): the
// column-banded, row-mask path (round 6).  Replaces ip_front + cc_tile + cc_seam + cc_stats + ip_rowcount + ip_compact (kernels_ip.hip), which
// walk every cell of the image five times through per-cell images in HBM / L2 (flag 1 B, parent 4 B, size 4 B, row mask 8 B, label 4 B per cell):
// 9.6 wavefront instructions per cell where the fused 16-ring kernel (kernels_ipf.hip) spends 5.  Here, as there, a column's state is a handful of
// ROW MASKS (64-bit: one bit per ring) and everything after the range / ground / edge pass works on masks and on vertical RUNS, not on cells:
//
//   ip_project   (kernels_ip.hip, unchanged)  points -> owner image, last writer wins (imageProjection.cpp:76-104)
//   ipb_band     one workgroup per band of 256 columns, one thread per column: owner -> point gather, ranges (in registers), ground test bottom-up,
//                right / down edge predicates (the right neighbour's range by a DPP lane shift; LDS only at wavefront boundaries and for the halo
//                column), vertical runs by bit operations, lock-free 16-bit union-find in LDS over the right-edges that join different run pairs,
//                then per RUN: the band root as a global linear index (min index of the piece = BFS discovery order restricted to the band, :147-156),
//                size and row mask added to the root's entries.  Written: five masks per column (ground, active, down-edges, right-edges, band roots)
//                and one parent entry per run head — nothing per cell.                                                     (:107-143, :210-281)
//   ipb_merge    one workgroup per stream: the right-edges that cross a band boundary (incl. the wrap-around column, :241-248) linked with the global
//                union, statistics of linked band roots added up, feasibility (:282-301), per column the KEEP / OUTLIER masks (:164-188), per (row,
//                64-column chunk) counts by ballots and the row-major exclusive offsets of the ordered compaction, startRingIndex / endRingIndex.
//   ipb_emit     one wavefront per 64-column chunk: second (and last) owner -> point gather of the kept cells, cloud_info written once (:158-191),
//                label_cnt_ numbers of the feasible roots when a label image is wanted (:303-314).
//
// Results are bit-identical to the seven-kernel path (ALEGO_IP_BAND=0 keeps it): same arithmetic per cell (ip_common.h), same roots (minimum linear index),
// same ordered compaction.  tests/test_gpu_parity.py::test_ip_bit_exact[(64,2048) / (32,2048) / (40,1800) x band / tile].
#include <cstdlib>
#include "dev_common.h"
#include "ip_common.h"
#include "prof.h"

#define IPB_TW 256            // columns per band = threads of ipb_band; a band has <= 256 x 64 = 16384 cells: 16-bit parents
#define IPB_NWV (IPB_TW / 64)
#ifndef IPB_MT
#define IPB_MT 512            // threads of ipb_merge (256 / 512 / 1024 measured at 64x2048: 122.3 / 123.1 / 121.7 k scans/s)
#endif
#define IPB_LINK_CAP 1024      // seam edges of an image: bands x rings <= (4096 / 256) x 64
#define IPB_ET 256            // threads of ipb_emit (4 chunks of 64 columns)
#ifndef IPB_EP
#define IPB_EP 4               // row parts of ipb_emit
#endif

typedef unsigned long long u64;
#ifdef ALEGO_TIMING   // development (tools/ipb_timing.py): wall-clock ticks (100 MHz) of workgroup 0 / slot 0 at the phase boundaries of the three kernels
__device__ long long ipb_times[48];
extern "C" void alego_ipb_times(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ipb_times), sizeof(long long) * 48); }
#define IPB_TICK(k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) ipb_times[k] = wall_clock64(); } while (0)
#else
#define IPB_TICK(k)
#endif
DEV_INLINE u64 ipb_low(int r) { return (2ull << r) - 1ull; }                 // bits 0..r (r = 63: all)
DEV_INLINE int ipb_head(u64 rs, int r) { return 63 - __clzll((long long)(rs & ipb_low(r))); }   // the run start at or below row r
DEV_INLINE u64* ipb_mask(const DevCtx& d, int slot, int k) { return (u64*)d.ipb_col + ((size_t)slot * IPB_NM + k) * d.H; }
DEV_INLINE int ipb_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV_INLINE u64 ipb_ld64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV_INLINE void ipb_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ECL-CC find / union on the global parent array (entries exist at run heads only; parents only ever decrease): as cc_find / cc_union of kernels_ip.hip
DEV_INLINE int ipb_find(int* parent, int v) {
  int curr = ipb_ld(parent + v);
  if (curr != v) {
    int prev = v, next;
    while (curr > (next = ipb_ld(parent + curr))) { ipb_st(parent + prev, next); prev = curr; curr = next; }
  }
  return curr;
}
DEV_INLINE int ipb_find_ro(const int* parent, int v) {
  int curr = ipb_ld(parent + v), next;
  while (curr > (next = ipb_ld(parent + curr))) curr = next;
  return curr;
}
// returns the root this call put under another one (every root is linked at most once, by exactly one successful compare-and-swap), -1 if none
DEV_INLINE int ipb_union(int* parent, int a, int b) {
  int ra = ipb_find(parent, a), rb = ipb_find(parent, b);
  bool repeat;
  int linked = -1;
  do {
    repeat = false;
    if (ra != rb) {
      int ret;
      if (ra < rb) { if ((ret = atomicCAS(parent + rb, rb, ra)) != rb) { rb = ret; repeat = true; } else linked = rb; }
      else { if ((ret = atomicCAS(parent + ra, ra, rb)) != ra) { ra = ret; repeat = true; } else linked = ra; }
    }
  } while (repeat);
  return linked;
}
DEV_INLINE bool ipb_feasible(const alego_params& P, int sz, u64 rows) {   // imageProjection.cpp:282-301
  return sz >= P.seg_big_num || (sz >= P.seg_valid_point_num && (int)__popcll(rows) >= P.seg_valid_line_num);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ipb_band: grid (bands, slots).  keep bit 0: also write the range and flag images (single-scan entry points / tests read them back)
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(IPB_TW) ipb_band(DevCtx d, int ring_pos, int keep) {
  const int slot = blockIdx.y + d.slot0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = d.H, NS = d.NS, c0 = blockIdx.x * IPB_TW;
  const int TW = min(IPB_TW, H - c0);
  const bool have = tid < TW;
  const int col = have ? c0 + tid : c0;
  const size_t base = (size_t)slot * d.N;
  const alego_params& P = d.P;
  const float4* pts = scan_pts(d, slot, ring_pos);
  __shared__ uint16_t par[IPB_TW * 64];               // entries exist at run heads only: index = row * 256 + column in band
  __shared__ float s_first[IPB_NWV + 1][64];          // ranges of the first column of every wavefront; [IPB_NWV] = the halo column (right of the band, wrap-around)
  __shared__ u64 s_a[IPB_TW + 1], s_y[IPB_TW + 1], s_rs[IPB_TW + 1];
  if (blockIdx.x == 0 && tid == 0) {  // orientation, :62-72 (as ip_front)
    int* sc = d.scal + slot * SC_COUNT;
    float* ori = d.ori + slot * 4;
    const int first = sc[SC_FIRST], last = sc[SC_LAST];
    sc[SC_PVALID_OUT] = sc[SC_PVALID];
    sc[SC_FIRST] = 0x7fffffff; sc[SC_LAST] = -1; sc[SC_PVALID] = 0;   // re-armed for the next scan's ip_project
    if (last >= 0) {
      const float4 p0 = pts[first], p1 = pts[last];
      float so = -d_atan2f(p0.y, p0.x);
      float eo = (float)((double)(-d_atan2f(p1.y, p1.x)) + 2 * M_PI);
      if ((double)(eo - so) > 3 * M_PI) eo = (float)((double)eo - 2 * M_PI);
      else if ((double)(eo - so) < M_PI) eo = (float)((double)eo + 2 * M_PI);
      ori[0] = so; ori[1] = eo; ori[2] = eo - so;
    }
  }
  IPB_TICK(0);
  int* owner = d.owner + base;
  // ---- ranges + ground, bottom-up; rows in batches of 8: the owner indices of a batch, then its point gathers, are independent loads ----
  float rng[64];
  u64 filled = 0, ground = 0;
  {
    float lx = 0, ly = 0, lz = 0;
    bool lower_ok = false;
    int obn[8];   // the NEXT batch's owner words: in flight while this batch's points are gathered (and issued before this batch's write-backs, which the compiler must keep behind them)
#pragma unroll
    for (int u = 0; u < 8; ++u) obn[u] = (have && u < NS) ? owner[u * H + col] : -1;
#pragma unroll
    for (int row0 = 0; row0 < 64; row0 += 8) {
      if (row0 < NS) {   // (uniform)
        int ob[8];
        float4 pb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ob[u] = obn[u];
        if (row0 + 8 < 64 && row0 + 8 < NS) {
#pragma unroll
          for (int u = 0; u < 8; ++u) obn[u] = (have && row0 + 8 + u < NS) ? owner[(row0 + 8 + u) * H + col] : -1;
        }
        // entries of this scan carry the tag, everything else is stale.  The plain form is written back (= the reset for the next scan) except for the band's
        // FIRST column, which the band to the left reads as its halo, possibly much later: it stays tagged here and ipb_merge strips it (cf. ip_front)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          ob[u] = (ob[u] >= 0 && (ob[u] & IP_OWNER_TAG)) ? (ob[u] & ~IP_OWNER_TAG) : -1;
          if (have && tid != 0 && row0 + u < NS) owner[(row0 + u) * H + col] = ob[u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) pb[u] = pts[max(ob[u], 0)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int row = row0 + u;
          const bool ok = ob[u] >= 0;
          const float x = pb[u].x, y = pb[u].y, z = pb[u].z;
          const float r = ok ? sqrtf(x * x + y * y + z * z) : -1.0f;   // :99
          rng[row] = r;
          if (ok) filled |= 1ull << row;
          if (row >= 1 && row - 1 < P.ground_scan_id && ok && lower_ok) {   // :111-131, pair (row - 1, row)
            if (ip_is_ground(d, x - lx, y - ly, z - lz)) ground |= 3ull << (row - 1);
          }
          lx = x; ly = y; lz = z; lower_ok = ok;
          if ((keep & 1) && have && row < NS) d.range_img[base + row * H + col] = r;
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) rng[row0 + u] = -1.0f;
      }
    }
  }
  IPB_TICK(1);
  const u64 act = filled & ~ground;
  s_a[tid] = have ? act : 0ull;
  if (lane == 0) {
#pragma unroll
    for (int row = 0; row < 64; ++row) s_first[wave][row] = rng[row];
  }
  if (wave == 0) {   // the halo column, one row per lane: range, ground pairs through a lane shift, masks by ballot
    const int hc = (c0 + TW == H) ? 0 : c0 + TW;
    int o = lane < NS ? owner[lane * H + hc] : -1;
    o = (o >= 0 && (o & IP_OWNER_TAG)) ? (o & ~IP_OWNER_TAG) : -1;   // (the halo is the first column of a band: its entries of this scan are still tagged)
    const float4 p = pts[max(o, 0)];
    const bool ok = o >= 0;
    const float px = __shfl_up(p.x, 1, 64), py = __shfl_up(p.y, 1, 64), pz = __shfl_up(p.z, 1, 64);
    const int okl = __shfl_up(ok ? 1 : 0, 1, 64);
    bool gp = false;
    if (lane >= 1 && lane - 1 < P.ground_scan_id && ok && okl) gp = ip_is_ground(d, p.x - px, p.y - py, p.z - pz);
    const u64 gb = __ballot(gp), fb = __ballot(ok);
    s_first[IPB_NWV][lane] = ok ? sqrtf(p.x * p.x + p.y * p.y + p.z * p.z) : -1.0f;
    if (lane == 0) s_a[TW] = fb & ~(gb | (gb >> 1));
  }
  __syncthreads();
  IPB_TICK(2);
  // ---- edge predicates: right (seg_alpha_x, :258-261) and down (seg_alpha_y, :262-265) ----
  u64 ex = 0, ey = 0;
  {
    const u64 act_r = s_a[min(tid + 1, TW)];
    const bool edge_lane = lane == 63 || tid == TW - 1;
    const float* sf = s_first[(tid >= TW - 1) ? IPB_NWV : wave + 1];
#pragma unroll
    for (int row = 0; row < 64; ++row) {
      const float nbd = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(rng[row]), 0x130, 0xF, 0xF, false));   // wave_shl:1: lane i <- lane i + 1
      const float nbl = sf[row];
      const float nb = edge_lane ? nbl : nbd;
      if ((act >> row) & 1ull) {
        const double r0 = (double)rng[row];
        if ((act_r >> row) & 1ull) {
          const double r1 = (double)nb;
          const double d1 = fmax(r0, r1), d2 = fmin(r0, r1);
          if (edge_angle_gt(d2 * d.sin_ax, d1 - d2 * d.cos_ax, P.seg_theta, d.tan_theta)) ex |= 1ull << row;
        }
        if (row + 1 < 64 && ((act >> (row + 1)) & 1ull)) {
          const double r1 = (double)rng[row + 1 < 64 ? row + 1 : 63];
          const double d1 = fmax(r0, r1), d2 = fmin(r0, r1);
          if (edge_angle_gt(d2 * d.sin_ay, d1 - d2 * d.cos_ay, P.seg_theta, d.tan_theta)) ey |= 1ull << row;
        }
      }
    }
  }
  if (H <= 1) ex = 0;
  IPB_TICK(3);
  const u64 rs = act & ~(ey << 1);   // run starts: active cells the cell below has no down-edge to
  s_y[tid] = ey; s_rs[tid] = rs;
  for (u64 m = rs; m; m &= m - 1) { const int r = __ffsll((long long)m) - 1; par[r * IPB_TW + tid] = (uint16_t)(r * IPB_TW + tid); }
  __syncthreads();
  IPB_TICK(4);
  // ---- unions over the right-edges inside the band that join different run pairs (an edge is skipped when the cell below already links the same two runs) ----
  if (tid + 1 < TW) {
    const u64 rs_n = s_rs[tid + 1], ey_n = s_y[tid + 1];
    for (u64 m = ex & ~((ex << 1) & (ey << 1) & (ey_n << 1)); m; m &= m - 1) {
      const int r = __ffsll((long long)m) - 1;
      ccl16_union(par, ipb_head(rs, r) * IPB_TW + tid, ipb_head(rs_n, r) * IPB_TW + tid + 1);
    }
  }
  __syncthreads();
  IPB_TICK(5);
  // ---- per run: band root (global linear index), size and row mask into the root's entries ----
  u64 rootm = 0;
  if (have) {
    int* gp = d.parent + base;
    for (u64 m = rs; m; m &= m - 1) {
      const int r = __ffsll((long long)m) - 1, l = r * IPB_TW + tid;
      int rr = par[l], nx;
      while (rr > (nx = par[rr])) rr = nx;   // read-only find: nobody writes any more
      const int vroot = (rr >> 8) * H + c0 + (rr & (IPB_TW - 1));
      const int e = __ffsll((long long)(~ey & ~(ipb_low(r) >> 1))) - 1;   // the run's last row: first row >= r without a down-edge
      const u64 runm = ipb_low(e) & ~(ipb_low(r) >> 1);
      gp[r * H + col] = vroot;
      atomicAdd(&d.cc_size[base + vroot], e - r + 1);
      atomicOr(&d.cc_rows[base + vroot], runm);
      if (rr == l) rootm |= 1ull << r;
    }
    ipb_mask(d, slot, 0)[col] = ground; ipb_mask(d, slot, 1)[col] = act; ipb_mask(d, slot, 2)[col] = ey; ipb_mask(d, slot, 3)[col] = ex; ipb_mask(d, slot, 4)[col] = rootm;
    if (keep & 1) {
      uint8_t* fimg = d.flag_img + base;
      for (int row = 0; row < NS; ++row)
        fimg[row * H + col] = (uint8_t)(((ground >> row) & 1ull) | (((act >> row) & 1ull) << 1) | (((ex >> row) & 1ull) << 2) | (((ey >> row) & 1ull) << 3));
    }
  }
  IPB_TICK(6);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ipb_merge: grid (slots).  keep bit 0: also write the per-cell root image (ip_labels, alego_debug_get("parent"))
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(IPB_MT) ipb_merge(DevCtx d, int keep) {
  const int slot = blockIdx.x + d.slot0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = d.H, NS = d.NS, nb = (H + IPB_TW - 1) / IPB_TW, nch = (H + 63) / 64;
  const size_t base = (size_t)slot * d.N;
  const alego_params& P = d.P;
  int* parent = d.parent + base;
  int* owner = d.owner + base;
  int* csz = d.cc_size + base;
  u64* crw = d.cc_rows + base;
  const u64 *Mg = ipb_mask(d, slot, 0), *Ma = ipb_mask(d, slot, 1), *My = ipb_mask(d, slot, 2), *Mx = ipb_mask(d, slot, 3), *Mr = ipb_mask(d, slot, 4);
  u64 *MK = ipb_mask(d, slot, 5), *MO = ipb_mask(d, slot, 6), *MF = ipb_mask(d, slot, 7), *MRf = ipb_mask(d, slot, 8);
  extern __shared__ __attribute__((aligned(16))) unsigned char ipb_smem[];
  unsigned short* s_cnt = reinterpret_cast<unsigned short*>(ipb_smem);   // [3][64][nch] counts, then exclusive prefixes inside a row
  __shared__ int s_rowtot[3][64], s_rowbase[3][64], s_tot[3], s_nlink;
  __shared__ int s_link[IPB_LINK_CAP];
  IPB_TICK(10);
  // the owner entries ipb_band left tagged (the first column of every band) -> plain form / -1
  for (int j = tid; j < nb * NS; j += IPB_MT) {
    const int row = j / nb, col = (j - row * nb) * IPB_TW;
    const int v = owner[row * H + col];
    owner[row * H + col] = (v >= 0 && (v & IP_OWNER_TAG)) ? (v & ~IP_OWNER_TAG) : -1;
  }
  // seams: the right-edges of every band's last column (incl. the wrap-around column, :241-248).  A root that a seam puts under another one is noted (once: by the
  // thread whose compare-and-swap linked it)
  if (tid == 0) s_nlink = 0;
  __syncthreads();
  for (int e = tid; e < nb * NS; e += IPB_MT) {
    const int t = e / NS, row = e - t * NS;
    const int col = min((t + 1) * IPB_TW, H) - 1, cn = col + 1 == H ? 0 : col + 1;
    if ((Mx[col] >> row) & 1ull) {
      const u64 rl = Ma[col] & ~(My[col] << 1), rr = Ma[cn] & ~(My[cn] << 1);
      const int linked = ipb_union(parent, ipb_head(rl, row) * H + col, ipb_head(rr, row) * H + cn);
      if (linked >= 0) s_link[atomicAdd(&s_nlink, 1)] = linked;   // (at most one per seam edge: nb * NS <= IPB_LINK_CAP)
    }
  }
  __syncthreads();
  IPB_TICK(11);
  // statistics of the linked roots go to their final root, and the totals of a joined component back into every root of it: whichever band root a run's parent
  // entry names, its entries then hold its component's size and row mask (ipb_band left every other band root's own, final, figures there)
  const int nlink = s_nlink;
  for (int j = tid; j < nlink; j += IPB_MT) {
    const int v = s_link[j], R = ipb_find_ro(parent, v);
    atomicAdd(&csz[R], ipb_ld(csz + v)); atomicOr(&crw[R], ipb_ld64(crw + v));
  }
  __syncthreads();
  IPB_TICK(12);
  for (int j = tid; j < nlink; j += IPB_MT) {
    const int v = s_link[j], R = ipb_find_ro(parent, v);
    const int sz = ipb_ld(csz + R);
    const u64 rw = ipb_ld64(crw + R);
    ipb_st(csz + v, sz); __hip_atomic_store(crw + v, rw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  IPB_TICK(13);
  // per column: keep / outlier masks (:164-188), feasible roots; per (row, chunk) counts by ballots
  const u64 above = P.ground_scan_id >= 63 ? 0ull : ~ipb_low(P.ground_scan_id);   // rows > ground_scan_id
  for (int ch = wave; ch < nch; ch += IPB_MT / 64) {
    const int col = ch * 64 + lane;
    u64 K = 0, O = 0, F = 0, Rf = 0;
    if (col < H) {
      const u64 g = Mg[col], a = Ma[col], y = My[col];
      for (u64 m = a & ~(y << 1); m;) {   // four runs per iteration: their loads are independent (a run = two dependent L2 round trips)
        int r[4], v[4], b[4], z[4];
        u64 w[4];
        bool on[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { on[q] = m != 0; r[q] = on[q] ? __ffsll((long long)m) - 1 : r[0]; m &= m - 1; v[q] = r[q] * H + col; }
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = ipb_ld(parent + v[q]);   // a band root (ipb_band wrote it; path halving only ever replaces it by a root further up the chain)
#pragma unroll
        for (int q = 0; q < 4; ++q) { z[q] = ipb_ld(csz + b[q]); w[q] = ipb_ld64(crw + b[q]); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (!on[q]) continue;
          const bool f = ipb_feasible(P, z[q], w[q]);
          const int e = __ffsll((long long)(~y & ~(ipb_low(r[q]) >> 1))) - 1;
          if (f) K |= ipb_low(e) & ~(ipb_low(r[q]) >> 1);
          if (b[q] == v[q]) { Rf |= 1ull << r[q]; if (f) F |= 1ull << r[q]; }
          if (keep & 1) {   // the per-cell root image (ip_labels, alego_debug_get("parent"))
            const int R = ipb_find_ro(parent, v[q]);
            for (int k = r[q]; k <= e; ++k) ipb_st(parent + k * H + col, R);
          }
        }
      }
      if (col % 5 == 0) O = a & ~K & above;                              // :165-171
      if (col % 5 == 0 || col <= 4 || col >= H - 5) K |= g;               // :173-176
      MK[col] = K; MO[col] = O; MF[col] = F; MRf[col] = Rf;
    }
    int ck = 0, co = 0, cf = 0;
#pragma unroll
    for (int row = 0; row < 64; ++row) {
      const int a0 = (int)__popcll(__ballot((K >> row) & 1ull)), a1 = (int)__popcll(__ballot((O >> row) & 1ull)), a2 = (int)__popcll(__ballot((F >> row) & 1ull));
      if (lane == row) { ck = a0; co = a1; cf = a2; }
    }
    s_cnt[(0 * 64 + lane) * nch + ch] = (unsigned short)ck; s_cnt[(1 * 64 + lane) * nch + ch] = (unsigned short)co; s_cnt[(2 * 64 + lane) * nch + ch] = (unsigned short)cf;
  }
  __syncthreads();
  IPB_TICK(14);
  if (tid < 192) {   // exclusive prefix inside every row (chunks ascending = columns ascending)
    const int ty = tid >> 6, row = tid & 63;
    int run = 0;
    for (int ch = 0; ch < nch; ++ch) { const int c = s_cnt[(ty * 64 + row) * nch + ch]; s_cnt[(ty * 64 + row) * nch + ch] = (unsigned short)run; run += c; }
    s_rowtot[ty][row] = row < NS ? run : 0;
  }
  __syncthreads();
  if (tid < 3) {
    int run = 0;
    for (int row = 0; row < 64; ++row) { s_rowbase[tid][row] = run; run += s_rowtot[tid][row]; }
    s_tot[tid] = run;
  }
  __syncthreads();
  IPB_TICK(15);
  int* off = d.ipb_off + (size_t)slot * 3 * 64 * nch;
  for (int j = tid; j < 3 * 64 * nch; j += IPB_MT) { const int tr = j / nch; off[j] = s_rowbase[tr >> 6][tr & 63] + (int)s_cnt[j]; }
  if (tid < NS) {
    d.ring_start[slot * NS + tid] = s_rowbase[0][tid] + 5;                          // :161
    d.ring_end[slot * NS + tid] = s_rowbase[0][tid] + s_rowtot[0][tid] - 1 - 5;     // :190
  }
  if (tid == 0) { int* sc = d.scal + slot * SC_COUNT; sc[SC_M] = s_tot[0]; sc[SC_NOUT] = s_tot[1]; sc[SC_NFEAS] = s_tot[2]; }
  IPB_TICK(16);
  // the statistics entries go back to zero for the next scan (every entry ipb_band or the merge above touched belongs to a band root)
  for (int col = tid; col < H; col += IPB_MT) {
    for (u64 m = Mr[col]; m; m &= m - 1) { const int v = (__ffsll((long long)m) - 1) * H + col; csz[v] = 0; crw[v] = 0ull; }
  }
  IPB_TICK(17);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ipb_emit: grid (chunks / 4, slots), one wavefront per 64-column chunk, one lane per column.  keep bit 0: label_cnt_ numbers of the roots
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(IPB_ET) ipb_emit(DevCtx d, int ring_pos, int keep) {
  const int slot = blockIdx.y + d.slot0, tid = threadIdx.x, lane = tid & 63;
  const int H = d.H, NS = d.NS, nch = (H + 63) / 64, nblk = (nch + IPB_ET / 64 - 1) / (IPB_ET / 64);
  const int part = blockIdx.x / nblk;                                   // rows [part * rpp, (part + 1) * rpp): four times the wavefronts, a quarter of the dependent load chain each
  const int ch = (blockIdx.x - part * nblk) * (IPB_ET / 64) + (tid >> 6);
  if (ch >= nch) return;
  const int rpp = ((NS + 4 * IPB_EP - 1) / (4 * IPB_EP)) * 4, rlo = part * rpp, rhi = min(NS, rlo + rpp);
  const int col = ch * 64 + lane;
  const bool have = col < H;
  const int cc = have ? col : H - 1;
  const size_t base = (size_t)slot * d.N;
  const float4* pts = scan_pts(d, slot, ring_pos);
  const int* owner = d.owner + base;
  const u64 K = have ? ipb_mask(d, slot, 5)[cc] : 0ull, O = have ? ipb_mask(d, slot, 6)[cc] : 0ull, G = ipb_mask(d, slot, 0)[cc];
  const u64 KO = K | O;
  const double cf = d.ip_colfrac[cc];
  const int* offk = d.ipb_off + ((size_t)slot * 3 + 0) * 64 * nch + ch;
  const int* offo = d.ipb_off + ((size_t)slot * 3 + 1) * 64 * nch + ch;
  const u64 below = (1ull << lane) - 1ull;
  int obn[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) obn[u] = ((KO >> min(rlo + u, 63)) & 1ull) && rlo + u < rhi ? owner[(rlo + u) * H + cc] : 0;
  for (int row0 = rlo; row0 < rhi; row0 += 4) {
    int ob[4];
    float4 pb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ob[u] = obn[u] & ~IP_OWNER_TAG;
    if (row0 + 4 < rhi) {   // the next group's owner words, in flight while this group's points are gathered and written
#pragma unroll
      for (int u = 0; u < 4; ++u) obn[u] = ((KO >> min(row0 + 4 + u, 63)) & 1ull) && row0 + 4 + u < rhi ? owner[(row0 + 4 + u) * H + cc] : 0;
    }
    if (__ballot(((KO >> row0) & 0xFull) != 0) == 0) continue;   // (uniform)
#pragma unroll
    for (int u = 0; u < 4; ++u) pb[u] = pts[ob[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = row0 + u;
      if (row >= rhi) break;
      const bool k = (K >> row) & 1ull, o = (O >> row) & 1ull;
      const u64 bk = __ballot(k), bo = __ballot(o);
      if (k | o) {
        float4 p = pb[u];
        p.w = (float)(row + cf);   // :101
        if (k) {
          const int line = offk[row * nch] + (int)__popcll(bk & below);
          d.seg_pts[base + line] = p;
          d.seg_ground[base + line] = (uint8_t)((G >> row) & 1ull);
          d.seg_col[base + line] = col;
          d.seg_range[base + line] = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);   // = the range image's value (:99)
        } else {
          d.outlier[base + offo[row * nch] + (int)__popcll(bo & below)] = p;
        }
      }
    }
  }
  if ((keep & 1) && part == 0) {   // label_cnt_ numbering of the feasible roots in discovery (row-major) order, 0 for the others (:303-306)
    const u64 F = have ? ipb_mask(d, slot, 7)[cc] : 0ull, Rf = have ? ipb_mask(d, slot, 8)[cc] : 0ull;
    const int* offf = d.ipb_off + ((size_t)slot * 3 + 2) * 64 * nch + ch;
    for (int row = 0; row < NS; ++row) {
      const bool f = (F >> row) & 1ull;
      const u64 bf = __ballot(f);
      if ((Rf >> row) & 1ull) d.cc_label[base + row * H + col] = f ? offf[row * nch] + (int)__popcll(bf & below) + 1 : 0;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
bool ipb_eligible(const DevCtx& d) { return d.ipb_col != nullptr && d.NS > 16 && d.NS <= 64 && d.H >= 64 && (d.H + IPB_TW - 1) / IPB_TW * d.NS <= IPB_LINK_CAP; }   // (seam edges = bands x rings: the linked-root list of ipb_merge)
size_t ipb_merge_lds(const DevCtx& d) { return (size_t)3 * 64 * ((d.H + 63) / 64) * sizeof(unsigned short); }
void launch_ipb(const DevCtx& d, int ring_pos, bool keep_images, hipStream_t st) {
  const int nb = (d.H + IPB_TW - 1) / IPB_TW, nch = (d.H + 63) / 64;
  ALEGO_LAUNCH(ipb_band, dim3(nb, d.n_launch), dim3(IPB_TW), 0, st, d, ring_pos, keep_images ? 1 : 0);
  ALEGO_LAUNCH(ipb_merge, dim3(d.n_launch), dim3(IPB_MT), ipb_merge_lds(d), st, d, keep_images ? 1 : 0);
  ALEGO_LAUNCH(ipb_emit, dim3(IPB_EP * ((nch + IPB_ET / 64 - 1) / (IPB_ET / 64)), d.n_launch), dim3(IPB_ET), 0, st, d, ring_pos, keep_images ? 1 : 0);
}
