// kernels_ipf.hip — ImageProjection as ONE launch: a workgroup per stream keeps the whole range-image pipeline on chip
// (replaces src/imageProjection.cpp:49-316 for sensors of up to 16 rings and 32 768 cells; larger images and odd widths take the
// multi-kernel path of kernels_ip.hip, which also stays available as the cross-check: ALEGO_IP_FUSED=0).
//
//   phase A  a1-a3  every thread projects n / 1024 input points (ip_point_cell, shared with ip_project); last writer wins by
//                   ds_max_u32 on a 32-bit owner image in LDS (4 B / cell: 115 KB at 16 x 1800)                    (:58-104)
//   phase B  a2-a4  a thread takes two adjacent columns: owner -> point gather (the only re-read of the input, mostly
//                   coalesced), ranges in registers, ground test on consecutive rows, right- / down-edge predicates with the
//                   right neighbour's column through a lane shift (LDS only at wavefront boundaries and for the wrap-around)
//                   (:62-72,:107-143,:255-270).  The owner image shrinks to 2 B / cell in place; per-column state is four
//                   16-bit row masks (ground, active, edge->right, edge->down)
//   phase C  a5     vertical runs from the masks (no memory traffic), 16-bit lock-free union-find in LDS over the right-edges that
//                   connect different run pairs, flatten, component size and row mask accumulated PER RUN in the roots' own
//                   entries (2 B / cell of LDS for the parents: owner 2 N + parents 2 N + masks 8 H = 130 KB)      (:210-316)
//   phase D  a6     ordered compaction: per row and wavefront ballots -> one 256-entry scan -> every kept cell's output line;
//                   second (and last) gather of the kept cells' points, cloud_info arrays written once              (:158-191)
//
// Against ip_project + ip_front + cc_lds16: no owner / flag images in HBM, no tag reset protocol between workgroups, one launch
// instead of three, and the per-cell passes of cc_lds16 (29 cells x 10 passes per thread) become bit operations on 16-bit
// row masks.  HBM traffic: 16 P in, one gather of the filled cells, one of the kept cells, 25 M + 16 O out.
#include <type_traits>
#include <algorithm>
#include <cstdlib>
#include "dev_common.h"
#include "ip_common.h"
#include "prof.h"

#define IPF2_T 1024
#ifndef IPF_QUICK
#define IPF_QUICK 1   // 0: every point through ip_point_cell in the first pass (the round-3 kernel before the two-pass projection)
#endif
// (Curvature + occlusion marks as a last phase of this kernel was measured and removed: the phase re-reads cloud_info through the L2 in a
//  workgroup that owns its CU alone — 16 wavefronts, nothing to hide the round trips behind — 29 us per stream against 24 us of CU time
//  for the stand-alone fe_curv launch, which runs at full occupancy.  A fat workgroup must not wait on global memory.)
#define IPF2_NW (IPF2_T / 64)
#define IPF2_ROWS 16

struct IpfShared {
  float first_r[IPF2_NW][IPF2_ROWS];   // ranges of the first column of every wavefront (right neighbour of the previous wavefront's last column)
  unsigned first_act[IPF2_NW];
  int red[3][IPF2_NW];
  int cnt[3][IPF2_ROWS * IPF2_NW];     // per (row, wavefront): kept cells, outliers, feasible roots -> exclusive prefix in row-major order
  int wtot[3][4];
  int tot[3];
  int nlist;                            // phase A: points deferred to the exact projection
};

#ifdef ALEGO_TIMING
__device__ long long ipf_times[16];
#define IPF_TICK(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) ipf_times[k] = wall_clock64(); } while (0)
extern "C" void alego_ipf_times(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ipf_times), sizeof(long long) * 16); }
#else
#define IPF_TICK(k)
#endif

DEV_INLINE int ipf_run_start(unsigned rs, int row) { return 31 - __clz((int)(rs & ((2u << row) - 1u))); }       // highest run start at or below row
DEV_INLINE int ipf_run_end(unsigned down, int s) { return s + __ffs((int)~(down >> s)) - 1; }                    // s + number of consecutive down-edges from s

bool ipf_eligible(const DevCtx& d) { return d.NS <= IPF2_ROWS && (d.H & 1) == 0 && d.H >= 64 && d.H <= 2 * IPF2_T && d.N <= 32768; }
size_t ipf_lds_bytes(const DevCtx& d) { return (size_t)4 * d.N + (size_t)8 * d.H; }

// keep bit 0: the single-scan entry points / tests read the range, flag, root and label images back
// (128 VGPRs x 1024 threads = the CU's whole register file.  Capping the kernel at 96 / 80 VGPRs (amdgpu_waves_per_eu 5 / 6) so that other
//  streams' wavefronts could share the CU was measured: 348.5 k / 334.9 k against 349.9 k scans/s — the spills cost what the sharing gains.)
__global__ void __launch_bounds__(IPF2_T) ip_fused(DevCtx d, int ring_pos, int keep) {
  const int slot = blockIdx.x + d.slot0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = d.N, H = d.H, NS = d.NS;
  const size_t base = (size_t)slot * N;
  const alego_params& P = d.P;
  extern __shared__ __attribute__((aligned(16))) unsigned char ipf_smem[];
  unsigned* own32 = reinterpret_cast<unsigned*>(ipf_smem);                                   // [N] phases A, B
  uint16_t* own16 = reinterpret_cast<uint16_t*>(ipf_smem);                                   // [N] from phase C on: index + 1, 0 = empty
  uint16_t* par = own16 + N;                                                                 // [N]
  unsigned long long* fcol = reinterpret_cast<unsigned long long*>(ipf_smem + (size_t)4 * N);   // [H] ground | active << 16 | right << 32 | down << 48
  __shared__ IpfShared S;

  // ---------------- phase A: projection ----------------
  // Pass 1 decides every point that is not near a cell boundary from a cheap estimate of its two angles (ip_point_quick: ~70 instructions
  // against ~160 for ip_point_cell) and defers the others to a list in LDS (the column-mask area, free until phase B); pass 2 runs
  // ip_point_cell on the list with all lanes busy.  Last writer wins by atomicMax on the point index, so the order of the passes is immaterial.
  IPF_TICK(0);
  for (int v = tid; v < N; v += IPF2_T) own32[v] = 0u;
  if (tid == 0) S.nlist = 0;
  __syncthreads();
  const int n = scan_count(d, slot, ring_pos);
  const float4* pts = scan_pts(d, slot, ring_pos);
  int* s_list = reinterpret_cast<int*>(fcol);
  const int list_cap = 2 * H;   // 8 H bytes
  float qmr, qmc;
  ip_quick_margins(d, &qmr, &qmc);
  int vmin = 0x7fffffff, vmax = -1, nvalid = 0;
#pragma unroll 1
  for (int i0 = tid; i0 < n; i0 += IPF2_T * 4) {
    float4 pin[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pin[u] = pts[min(i0 + u * IPF2_T, n - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * IPF2_T;
      bool valid = false, defer = false;
      int cell = -1;
#if IPF_QUICK
      if (i < n) defer = !ip_point_quick(d, pin[u], qmr, qmc, &valid, &cell);
#else
      if (i < n) cell = ip_point_cell(d, pin[u], &valid);
#endif
      if (cell >= 0) atomicMax(&own32[cell], (unsigned)(i + 1));   // later points overwrite earlier ones (:102-103)
      if (valid) { vmin = min(vmin, i); vmax = max(vmax, i); ++nvalid; }
      const unsigned long long dm = __ballot(defer);
      if (dm) {   // wavefront-aggregated append
        int base = 0;
        if (lane == 0) base = atomicAdd(&S.nlist, (int)__popcll(dm));
        base = __shfl(base, 0, 64);
        if (defer) {
          const int pos = base + (int)__popcll(dm & ((1ull << lane) - 1ull));
          if (pos < list_cap) s_list[pos] = i;   // (a full list is dealt with below)
        }
      }
    }
  }
  __syncthreads();
  {
    // the deferred points; when the list overflowed (a scan with more than an eighth of its points on cell boundaries) every point is projected
    // again, exactly — atomicMax makes that idempotent for the ones pass 1 had decided
    const bool all = S.nlist > list_cap;
    const int nl = all ? n : S.nlist;
#pragma unroll 1
    for (int j = tid; j < nl; j += IPF2_T) {
      const int i = all ? j : s_list[j];
      bool v2;
      const int c2 = ip_point_cell(d, pts[i], &v2);
      if (c2 >= 0) atomicMax(&own32[c2], (unsigned)(i + 1));
    }
  }
  IPF_TICK(1);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    vmin = min(vmin, __shfl_xor(vmin, o, 64));
    vmax = max(vmax, __shfl_xor(vmax, o, 64));
    nvalid += __shfl_xor(nvalid, o, 64);
  }
  if (lane == 0) { S.red[0][wave] = vmin; S.red[1][wave] = vmax; S.red[2][wave] = nvalid; }
  __syncthreads();
  if (tid == IPF2_T - 1) {   // orientation block (:62-72): two dependent loads + two atan2f by ONE thread — the last one, whose wavefront has no columns
                             // in phase B at 16 x 1800, so that the other fifteen do not wait for it at their next barrier
    int first = 0x7fffffff, last = -1, pv = 0;
    for (int w = 0; w < IPF2_NW; ++w) { first = min(first, S.red[0][w]); last = max(last, S.red[1][w]); pv += S.red[2][w]; }
    d.scal[slot * SC_COUNT + SC_PVALID_OUT] = pv;
    if (last >= 0) {
      float* ori = d.ori + slot * 4;
      const float4 p0 = pts[first], p1 = pts[last];
      float so = -d_atan2f(p0.y, p0.x);
      float eo = (float)((double)(-d_atan2f(p1.y, p1.x)) + 2 * M_PI);
      if ((double)(eo - so) > 3 * M_PI) eo = (float)((double)eo - 2 * M_PI);
      else if ((double)(eo - so) < M_PI) eo = (float)((double)eo + 2 * M_PI);
      ori[0] = so; ori[1] = eo; ori[2] = eo - so;
    }
  }

  IPF_TICK(2);
  // ---------------- phase B: ranges, ground, edges (two adjacent columns per thread) ----------------
  const int c0 = 2 * tid, c1 = c0 + 1;
  const bool colv = c0 < H;
  float rng0[IPF2_ROWS], rng1[IPF2_ROWS];
  unsigned opk[IPF2_ROWS];
  unsigned filled0 = 0, filled1 = 0, ground0 = 0, ground1 = 0;
  {
    float lx0 = 0, ly0 = 0, lz0 = 0, lx1 = 0, ly1 = 0, lz1 = 0;
    bool lok0 = false, lok1 = false;
#pragma unroll
    for (int row0 = 0; row0 < IPF2_ROWS; row0 += 4) {
      uint2 ob[4];
      float4 pa[4], pb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ob[u] = (colv && row0 + u < NS) ? *reinterpret_cast<const uint2*>(&own32[(row0 + u) * H + c0]) : make_uint2(0u, 0u);
#pragma unroll
      for (int u = 0; u < 4; ++u) { pa[u] = pts[max((int)ob[u].x - 1, 0)]; pb[u] = pts[max((int)ob[u].y - 1, 0)]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = row0 + u;
        opk[row] = (ob[u].x & 0xFFFFu) | (ob[u].y << 16);
        {
          const bool ok = ob[u].x != 0u;
          const float x = pa[u].x, y = pa[u].y, z = pa[u].z;
          rng0[row] = ok ? sqrtf(x * x + y * y + z * z) : -1.0f;   // :99
          if (ok) filled0 |= 1u << row;
          if (row >= 1 && row - 1 < P.ground_scan_id && ok && lok0 && ip_is_ground(d, x - lx0, y - ly0, z - lz0)) ground0 |= 3u << (row - 1);   // :111-131
          lx0 = x; ly0 = y; lz0 = z; lok0 = ok;
        }
        {
          const bool ok = ob[u].y != 0u;
          const float x = pb[u].x, y = pb[u].y, z = pb[u].z;
          rng1[row] = ok ? sqrtf(x * x + y * y + z * z) : -1.0f;
          if (ok) filled1 |= 1u << row;
          if (row >= 1 && row - 1 < P.ground_scan_id && ok && lok1 && ip_is_ground(d, x - lx1, y - ly1, z - lz1)) ground1 |= 3u << (row - 1);
          lx1 = x; ly1 = y; lz1 = z; lok1 = ok;
        }
      }
    }
  }
  IPF_TICK(3);
  const unsigned act0 = filled0 & ~ground0, act1 = filled1 & ~ground1;
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < IPF2_ROWS; ++r) S.first_r[wave][r] = rng0[r];
    S.first_act[wave] = act0;
  }
  __syncthreads();   // every read of the 32-bit owner image has happened
  // the column to the right of c1: the next thread's first column, column 0 behind the last one (:241-248)
  const int tlast = H / 2 - 1;
  float nbr[IPF2_ROWS];
  unsigned nb_act = (unsigned)__shfl_down((int)act0, 1, 64);
#pragma unroll
  for (int r = 0; r < IPF2_ROWS; ++r) nbr[r] = __shfl_down(rng0[r], 1, 64);
  if (lane == 63 || tid == tlast) {
    const int sw = tid == tlast ? 0 : min(wave + 1, IPF2_NW - 1);
#pragma unroll
    for (int r = 0; r < IPF2_ROWS; ++r) nbr[r] = S.first_r[sw][r];
    nb_act = S.first_act[sw];
  }
  unsigned redge0 = 0, redge1 = 0, down0 = 0, down1 = 0;
  if (colv) {
#pragma unroll
    for (int row = 0; row < IPF2_ROWS; ++row) {
      if (row < NS) {
        if ((act0 >> row) & 1u) {
          const double r0 = (double)rng0[row];
          if ((act1 >> row) & 1u) {   // same row, seg_alpha_x (:258-261)
            const double r1 = (double)rng1[row], d1 = fmax(r0, r1), d2 = fmin(r0, r1);
            if (edge_angle_gt(d2 * d.sin_ax, d1 - d2 * d.cos_ax, P.seg_theta, d.tan_theta)) redge0 |= 1u << row;
          }
          if (row + 1 < NS && ((act0 >> (row + 1)) & 1u)) {   // same column, seg_alpha_y (:262-265)
            const double r1 = (double)rng0[row + 1 < IPF2_ROWS ? row + 1 : row], d1 = fmax(r0, r1), d2 = fmin(r0, r1);
            if (edge_angle_gt(d2 * d.sin_ay, d1 - d2 * d.cos_ay, P.seg_theta, d.tan_theta)) down0 |= 1u << row;
          }
        }
        if ((act1 >> row) & 1u) {
          const double r0 = (double)rng1[row];
          if ((nb_act >> row) & 1u) {
            const double r1 = (double)nbr[row], d1 = fmax(r0, r1), d2 = fmin(r0, r1);
            if (edge_angle_gt(d2 * d.sin_ax, d1 - d2 * d.cos_ax, P.seg_theta, d.tan_theta)) redge1 |= 1u << row;
          }
          if (row + 1 < NS && ((act1 >> (row + 1)) & 1u)) {
            const double r1 = (double)rng1[row + 1 < IPF2_ROWS ? row + 1 : row], d1 = fmax(r0, r1), d2 = fmin(r0, r1);
            if (edge_angle_gt(d2 * d.sin_ay, d1 - d2 * d.cos_ay, P.seg_theta, d.tan_theta)) down1 |= 1u << row;
          }
        }
      }
    }
    fcol[c0] = (unsigned long long)ground0 | ((unsigned long long)act0 << 16) | ((unsigned long long)redge0 << 32) | ((unsigned long long)down0 << 48);
    fcol[c1] = (unsigned long long)ground1 | ((unsigned long long)act1 << 16) | ((unsigned long long)redge1 << 32) | ((unsigned long long)down1 << 48);
    if (keep & 1) {
      float* rimg = d.range_img + base;
      uint8_t* fimg = d.flag_img + base;
#pragma unroll
      for (int row = 0; row < IPF2_ROWS; ++row) {
        if (row < NS) {
          rimg[row * H + c0] = rng0[row]; rimg[row * H + c1] = rng1[row];
          fimg[row * H + c0] = (uint8_t)(((ground0 >> row) & 1u) | (((act0 >> row) & 1u) << 1) | (((redge0 >> row) & 1u) << 2) | (((down0 >> row) & 1u) << 3));
          fimg[row * H + c1] = (uint8_t)(((ground1 >> row) & 1u) | (((act1 >> row) & 1u) << 1) | (((redge1 >> row) & 1u) << 2) | (((down1 >> row) & 1u) << 3));
        }
      }
    }
  }

  IPF_TICK(4);
  // ---------------- phase C: connected components over vertical runs ----------------
  // a cell starts a run when it is active and no down-edge reaches it from below; a run's representative is its first (lowest) cell
  const unsigned rs0 = act0 & ~(down0 << 1), rs1 = act1 & ~(down1 << 1);
  if (colv) {
    unsigned* own16w = reinterpret_cast<unsigned*>(own16);
    unsigned* parw0 = reinterpret_cast<unsigned*>(par);
#pragma unroll
    for (int row = 0; row < IPF2_ROWS; ++row) {
      if (row < NS) {
        const int v0 = row * H + c0;
        own16w[v0 >> 1] = opk[row];
        const int s0 = ((act0 >> row) & 1u) ? ipf_run_start(rs0, row) : row, s1 = ((act1 >> row) & 1u) ? ipf_run_start(rs1, row) : row;
        parw0[v0 >> 1] = (unsigned)(s0 * H + c0) | ((unsigned)(s1 * H + c1) << 16);
      }
    }
  }
  __syncthreads();
  IPF_TICK(5);
  const int cn = c0 + 2 == H ? 0 : c0 + 2;
  if (colv) {
    // right-edges between the runs; one is skipped when the cell below already links the same pair of runs
    const unsigned down_nb = (unsigned)(fcol[cn] >> 48) & 0xFFFFu;
    unsigned m0 = redge0 & ~((redge0 & down0 & down1) << 1);
    unsigned m1 = redge1 & ~((redge1 & down1 & down_nb) << 1);
    while (m0) { const int row = __ffs((int)m0) - 1; m0 &= m0 - 1; ccl16_union(par, row * H + c0, row * H + c1); }
    while (m1) { const int row = __ffs((int)m1) - 1; m1 &= m1 - 1; ccl16_union(par, row * H + c1, row * H + cn); }
  }
  __syncthreads();
  IPF_TICK(6);
  // flatten the run starts; root = minimum linear index of the component = BFS discovery order (:147-156)
  unsigned root0 = 0, root1 = 0;
  if (colv) {
    for (unsigned m = rs0; m; m &= m - 1) { const int row = __ffs((int)m) - 1, s = row * H + c0; int r = par[s], nx; while (r > (nx = par[r])) r = nx; par[s] = (uint16_t)r; if (r == s) root0 |= 1u << row; }
    for (unsigned m = rs1; m; m &= m - 1) { const int row = __ffs((int)m) - 1, s = row * H + c1; int r = par[s], nx; while (r > (nx = par[r])) r = nx; par[s] = (uint16_t)r; if (r == s) root1 |= 1u << row; }
  }
  __syncthreads();
  IPF_TICK(7);
  // A root's own entry is free from here on (its owner knows it through root0 / root1): it becomes the component's 16-bit
  // accumulator, first for the size (:282), then for the row mask (:283-294).  One atomic per RUN.
  unsigned* parw = reinterpret_cast<unsigned*>(par);
  auto zero_roots = [&]() {
    for (unsigned m = root0; m; m &= m - 1) par[(__ffs((int)m) - 1) * H + c0] = 0;
    for (unsigned m = root1; m; m &= m - 1) par[(__ffs((int)m) - 1) * H + c1] = 0;
    __syncthreads();
  };
  auto root_of = [&](unsigned rootm, int row, int c) -> int { const int s = row * H + c; return ((rootm >> row) & 1u) ? s : (int)par[s]; };
  zero_roots();
  if (colv) {
    for (unsigned m = rs0; m; m &= m - 1) { const int row = __ffs((int)m) - 1, r = root_of(root0, row, c0); atomicAdd(&parw[r >> 1], (unsigned)(ipf_run_end(down0, row) - row + 1) << ((r & 1) * 16)); }
    for (unsigned m = rs1; m; m &= m - 1) { const int row = __ffs((int)m) - 1, r = root_of(root1, row, c1); atomicAdd(&parw[r >> 1], (unsigned)(ipf_run_end(down1, row) - row + 1) << ((r & 1) * 16)); }
  }
  __syncthreads();
  unsigned big0 = 0, big1 = 0, mid0 = 0, mid1 = 0;   // per run (bit at its start row): size >= 30 / 5 <= size < 30
  if (colv) {
    for (unsigned m = rs0; m; m &= m - 1) { const int row = __ffs((int)m) - 1, sz = (int)par[root_of(root0, row, c0)]; if (sz >= P.seg_big_num) big0 |= 1u << row; else if (sz >= P.seg_valid_point_num) mid0 |= 1u << row; }
    for (unsigned m = rs1; m; m &= m - 1) { const int row = __ffs((int)m) - 1, sz = (int)par[root_of(root1, row, c1)]; if (sz >= P.seg_big_num) big1 |= 1u << row; else if (sz >= P.seg_valid_point_num) mid1 |= 1u << row; }
  }
  __syncthreads();
  zero_roots();
  if (colv) {   // rows touched by the mid-sized components
    for (unsigned m = mid0; m; m &= m - 1) { const int row = __ffs((int)m) - 1, r = root_of(root0, row, c0), e = ipf_run_end(down0, row); atomicOr(&parw[r >> 1], (((2u << e) - 1u) & ~((1u << row) - 1u)) << ((r & 1) * 16)); }
    for (unsigned m = mid1; m; m &= m - 1) { const int row = __ffs((int)m) - 1, r = root_of(root1, row, c1), e = ipf_run_end(down1, row); atomicOr(&parw[r >> 1], (((2u << e) - 1u) & ~((1u << row) - 1u)) << ((r & 1) * 16)); }
  }
  __syncthreads();
  unsigned feas0 = 0, feas1 = 0;   // cells of feasible components
  if (colv) {
    for (unsigned m = big0 | mid0; m; m &= m - 1) {
      const int row = __ffs((int)m) - 1, e = ipf_run_end(down0, row);
      if (((big0 >> row) & 1u) || __popc((unsigned)par[root_of(root0, row, c0)]) >= P.seg_valid_line_num) feas0 |= ((2u << e) - 1u) & ~((1u << row) - 1u);
    }
    for (unsigned m = big1 | mid1; m; m &= m - 1) {
      const int row = __ffs((int)m) - 1, e = ipf_run_end(down1, row);
      if (((big1 >> row) & 1u) || __popc((unsigned)par[root_of(root1, row, c1)]) >= P.seg_valid_line_num) feas1 |= ((2u << e) - 1u) & ~((1u << row) - 1u);
    }
  }

  IPF_TICK(8);
  // ---------------- phase D: ordered compaction (:158-191) ----------------
  const unsigned all16 = 0xFFFFu;
  const unsigned rowgt = P.ground_scan_id >= 15 ? 0u : (P.ground_scan_id < 0 ? all16 : (all16 & ~((2u << P.ground_scan_id) - 1u)));   // rows > ground_scan_id
  const bool gk0 = c0 % 5 == 0 || c0 <= 4 || c0 >= H - 5, gk1 = c1 % 5 == 0 || c1 <= 4 || c1 >= H - 5;
  const unsigned keep0 = colv ? ((gk0 ? ground0 : 0u) | feas0) : 0u, keep1 = colv ? ((gk1 ? ground1 : 0u) | feas1) : 0u;
  const unsigned outl0 = (colv && c0 % 5 == 0) ? (act0 & ~feas0 & rowgt) : 0u, outl1 = (colv && c1 % 5 == 0) ? (act1 & ~feas1 & rowgt) : 0u;
  const unsigned fr0 = colv ? (root0 & feas0) : 0u, fr1 = colv ? (root1 & feas1) : 0u;   // roots of feasible components: label_cnt_ numbering (:303-306)
#pragma unroll
  for (int row = 0; row < IPF2_ROWS; ++row) {
    const unsigned long long bk0 = __ballot((keep0 >> row) & 1u), bk1 = __ballot((keep1 >> row) & 1u);
    const unsigned long long bo0 = __ballot((outl0 >> row) & 1u), bo1 = __ballot((outl1 >> row) & 1u);
    const unsigned long long bf0 = __ballot((fr0 >> row) & 1u), bf1 = __ballot((fr1 >> row) & 1u);
    if (lane == 0) {
      S.cnt[0][row * IPF2_NW + wave] = (int)(__popcll(bk0) + __popcll(bk1));
      S.cnt[1][row * IPF2_NW + wave] = (int)(__popcll(bo0) + __popcll(bo1));
      S.cnt[2][row * IPF2_NW + wave] = (int)(__popcll(bf0) + __popcll(bf1));
    }
  }
  __syncthreads();
  {   // exclusive scan of the three 256-entry tables by the first four wavefronts
    int v3[3] = {0, 0, 0}, in3[3] = {0, 0, 0};
    if (tid < IPF2_ROWS * IPF2_NW) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        v3[a] = S.cnt[a][tid];
        int incl = v3[a];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        in3[a] = incl;
        if (lane == 63) S.wtot[a][wave] = incl;
      }
    }
    __syncthreads();
    if (tid < IPF2_ROWS * IPF2_NW) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        int woff = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) if (w < wave) woff += S.wtot[a][w];
        S.cnt[a][tid] = woff + in3[a] - v3[a];
        if (tid == IPF2_ROWS * IPF2_NW - 1) S.tot[a] = woff + in3[a];
      }
    }
  }
  __syncthreads();
  if (tid < NS) {   // startRingIndex / endRingIndex (:161,:190)
    const int row = tid;
    d.ring_start[slot * NS + row] = S.cnt[0][row * IPF2_NW] + 5;
    d.ring_end[slot * NS + row] = (row + 1 < IPF2_ROWS ? S.cnt[0][(row + 1) * IPF2_NW] : S.tot[0]) - 1 - 5;
  }
  if (tid == 0) {
    int* sc = d.scal + slot * SC_COUNT;
    sc[SC_M] = S.tot[0]; sc[SC_NOUT] = S.tot[1]; sc[SC_NFEAS] = S.tot[2];
  }
  IPF_TICK(9);
  const unsigned long long below = (1ull << lane) - 1ull;
  const unsigned* own16w = reinterpret_cast<const unsigned*>(own16);
  const double cf0 = d.ip_colfrac[min(c0, H - 1)], cf1 = d.ip_colfrac[min(c1, H - 1)];   // c / 10000.0 (host division, alego_api.hip)
#pragma unroll
  for (int row0 = 0; row0 < IPF2_ROWS; row0 += 4) {
    float4 qa[4], qb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = row0 + u;
      const unsigned ow = (colv && row < NS) ? own16w[(row * H + c0) >> 1] : 0u;
      const bool e0 = ((keep0 | outl0) >> row) & 1u, e1 = ((keep1 | outl1) >> row) & 1u;
      qa[u] = pts[e0 ? (int)(ow & 0xFFFFu) - 1 : 0];   // (clamped address instead of a branch: the eight gathers are in flight together)
      qb[u] = pts[e1 ? (int)(ow >> 16) - 1 : 0];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = row0 + u;
      const bool k0 = (keep0 >> row) & 1u, k1 = (keep1 >> row) & 1u, o0 = (outl0 >> row) & 1u, o1 = (outl1 >> row) & 1u;
      const unsigned long long bk0 = __ballot(k0), bk1 = __ballot(k1), bo0 = __ballot(o0), bo1 = __ballot(o1);
      const int lk = S.cnt[0][row * IPF2_NW + wave] + (int)(__popcll(bk0 & below) + __popcll(bk1 & below));
      const int lo = S.cnt[1][row * IPF2_NW + wave] + (int)(__popcll(bo0 & below) + __popcll(bo1 & below));
      if (k0 | o0) {
        const float4 q = qa[u];
        const float4 p = make_float4(q.x, q.y, q.z, (float)(row + cf0));   // :101
        if (k0) {
          d.seg_pts[base + lk] = p;
          d.seg_ground[base + lk] = (uint8_t)((ground0 >> row) & 1u);
          d.seg_col[base + lk] = c0;
          d.seg_range[base + lk] = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z);   // = the range image's value (:99,:184)
        } else {
          d.outlier[base + lo] = p;
        }
      }
      if (k1 | o1) {
        const float4 q = qb[u];
        const float4 p = make_float4(q.x, q.y, q.z, (float)(row + cf1));
        const int l1 = lk + (k0 ? 1 : 0), lo1 = lo + (o0 ? 1 : 0);
        if (k1) {
          d.seg_pts[base + l1] = p;
          d.seg_ground[base + l1] = (uint8_t)((ground1 >> row) & 1u);
          d.seg_col[base + l1] = c1;
          d.seg_range[base + l1] = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z);
        } else {
          d.outlier[base + lo1] = p;
        }
      }
    }
  }
  IPF_TICK(10);
  if (keep & 1) {   // root image + label_cnt_ numbers for ip_labels (:303-314)
#pragma unroll
    for (int row = 0; row < IPF2_ROWS; ++row) {
      const unsigned long long bf0 = __ballot((fr0 >> row) & 1u), bf1 = __ballot((fr1 >> row) & 1u);
      if (colv && row < NS) {
        const int v0 = row * H + c0, v1 = row * H + c1;
        const int s0 = ((act0 >> row) & 1u) ? ipf_run_start(rs0, row) : row, s1 = ((act1 >> row) & 1u) ? ipf_run_start(rs1, row) : row;
        d.parent[base + v0] = ((act0 >> row) & 1u) ? root_of(root0, s0, c0) : -1;
        d.parent[base + v1] = ((act1 >> row) & 1u) ? root_of(root1, s1, c1) : -1;
        const int nf = S.cnt[2][row * IPF2_NW + wave] + (int)(__popcll(bf0 & below) + __popcll(bf1 & below));
        if ((root0 >> row) & 1u) d.cc_label[base + v0] = ((fr0 >> row) & 1u) ? nf + 1 : 0;
        if ((root1 >> row) & 1u) d.cc_label[base + v1] = ((fr1 >> row) & 1u) ? nf + ((fr0 >> row) & 1u) + 1 : 0;
      }
    }
  }

}

// ---------------------------------------------------------------------------------------------------------------------------------
// ip_fused_h — the same pipeline in HALF a CU (round 4).  ip_fused holds 16 wavefronts x 128 VGPRs and 130 KB of LDS: it can only be placed
// on a CU that has drained completely, overlaps with nothing and — with four stream groups keeping every CU partly busy — waits five times its
// own duration for such a CU.  Here: 512 threads (8 wavefronts x 128 VGPRs: half the register file) and 2 N + 8 H bytes of LDS (72 KB at
// 16 x 1800), so that two of these workgroups share a CU, or one shares it with the other stream groups' work:
//   * a thread takes its column pairs ONE AFTER THE OTHER (pair t, then pair t + 512); nothing of a pair stays in registers between the
//     phases — the four 16-bit row masks of a column live in LDS (fcol: ground, active, right-edges -> roots, down-edges -> feasible cells);
//   * the owner image is 2 B / cell from the start (point index + 1 <= 32768): "last writer wins" is a 16-bit maximum through a
//     compare-and-swap on the aligned 32-bit word (LDS has no 16-bit atomics; adjacent cells of consecutive points retry once);
//   * the parent array of phase C takes the owner image's place; the packed owners a pair needs again for the emit of phase D wait in
//     an HBM scratch line of the stream (2 N bytes, written and read by this workgroup only: it stays in the L2).
// Results are bit-identical to ip_fused (same arithmetic per cell, same ordered compaction); ALEGO_IP_HALF=0 selects ip_fused.
#define IPH_T 512     // ip_fused_h: 8 wavefronts, two column pairs per thread: H <= 2048, N <= 32768
#define IPH_NP 2
#define IPW_T 1024    // ip_fused_w (round 4): the same kernel with 16 wavefronts for the images only a whole CU can hold — 16 x 4000, the reference's own
#define IPW_NP 2      //   geometry (utility.h:50-55): 2 N + 8 H = 160 000 B of the CU's 163 840 B of LDS, H <= 4096, N <= 65 535 (16-bit owners / prefixes)
#ifndef IPH_GB
#define IPH_GB 4   // rows gathered per batch in phase B / phase D (loads in flight against registers)
#endif
#ifndef IPH_PREFETCH
#define IPH_PREFETCH 1
#endif
#ifndef IPH_OWN_AHEAD
#define IPH_OWN_AHEAD 1
#endif
#ifndef IPH_GD
#define IPH_GD 4
#endif
// A field of the kernel argument read where it is needed instead of at the kernel's entry (round 5).  The compiler loads every member of the by-value DevCtx
// it finds used anywhere with a handful of wide s_loads in the entry block — 101 dwords for ip_fused_t — and they all stay live to their last use: more uniform
// values than there are SGPRs, parked in VGPR lanes and read back (v_readlane_b32) at their uses.  The output arrays are only touched by phase D: fetched there,
// through a pointer the optimiser cannot identify with the argument, they occupy nothing during phases A-C.
typedef const __attribute__((address_space(4))) char* IphKarg;
template <class F> DEV_INLINE F iph_karg(unsigned off) {
  IphKarg kp = (IphKarg)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  return *reinterpret_cast<const __attribute__((address_space(4))) F*>(kp + off);
}
// the whole argument struct of one phase: `d` and `P` are shadowed by references into the kernel-argument segment obtained through a pointer the optimiser cannot
// identify with the by-value parameter, so the members a phase uses are loaded at its start (scalar loads, a handful per phase) and are dead at its end
#define IPH_PHASE_CTX IphKarg iph_k_ = (IphKarg)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(iph_k_)); \
  const DevCtx& d = *(const DevCtx*)(const char*)iph_k_; const alego_params& P = d.P; (void)P
#define IPH_LATE(field) iph_karg<decltype(DevCtx::field)>((unsigned)offsetof(DevCtx, field))   // (DevCtx is the kernel's first argument: offset 0 of the segment)

// first_r / first_act (phase B) and the count tables (phase D) are never live together: one area (the wide instantiation has 16 bytes of
// LDS to spare at 16 x 4000)
template <int NW, int NP>
struct IphShared {
  static_assert(3 * NW >= IPF2_ROWS + 1, "the next-pass column shares the area of the phase-A reduction");
  union {
    struct { float first_r[NW][IPF2_ROWS]; unsigned first_act[NW]; } b;   // ranges / active mask of the first column of every wavefront of the pass in flight
    unsigned short cnt[3][IPF2_ROWS * NP * NW];   // per (row, pass, wavefront) = column-ascending inside a row: kept cells, outliers, feasible roots -> exclusive prefixes (< 65536)
  } u;
  float col0_r[IPF2_ROWS];             // column 0 (right neighbour of the last column, :241-248)
  unsigned col0_act;
  union {
    int red[3][NW];                                            // phase A: per-wavefront first / last valid point, valid count
    struct { float np_r[IPF2_ROWS]; unsigned np_act; } np;     // phase B: the first column of the NEXT pass (right neighbour of this pass's last column), one row per lane
  };
  int wtot[3][IPF2_ROWS * NP * NW / 64];
  int tot[3];
  int nlist;
};
bool iph_eligible(const DevCtx& d) { return ipf_eligible(d) && d.H <= 2 * IPH_NP * IPH_T && d.N <= 32768 && d.ipf_own != nullptr; }
bool ipw_eligible(const DevCtx& d) {
  return d.NS <= IPF2_ROWS && (d.H & 1) == 0 && d.H >= 64 && d.H <= 2 * IPW_NP * IPW_T && d.N <= 65535 && d.ipf_own != nullptr &&
         (size_t)2 * d.N + (size_t)8 * d.H + sizeof(IphShared<IPW_T / 64, IPW_NP>) <= 163840;
}
size_t iph_lds_bytes(const DevCtx& d) { return (size_t)2 * d.N + (size_t)8 * d.H; }
#ifdef ALEGO_DUP_HOOK
static size_t iph_pad() { static const char* e = std::getenv("ALEGO_IPH_PAD"); return e ? (size_t)atoi(e) : 0; }   // development: unused LDS that keeps a second ip_fused_h off the CU
#else
static size_t iph_pad() { return 0; }
#endif

// 16-bit maximum of own16[cell] and val through the aligned 32-bit word
DEV_INLINE void iph_max16(unsigned* w32, int cell, unsigned val) {
  unsigned* w = w32 + (cell >> 1);
  const int sh = (cell & 1) * 16;
  unsigned old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (((old >> sh) & 0xFFFFu) < val) {
    const unsigned nw = (old & ~(0xFFFFu << sh)) | (val << sh);
    const unsigned got = atomicCAS(w, old, nw);
    if (got == old) break;
    old = got;
  }
}
// the four row masks of a column
struct IphCol { unsigned a, b, x, y; };   // a = ground, b = active, x = right-edges (phase C: roots), y = down-edges (after phase C: feasible cells)
DEV_INLINE IphCol iph_load(const unsigned long long* fcol, int c) { const unsigned long long v = fcol[c]; IphCol m; m.a = (unsigned)v & 0xFFFFu; m.b = (unsigned)(v >> 16) & 0xFFFFu; m.x = (unsigned)(v >> 32) & 0xFFFFu; m.y = (unsigned)(v >> 48); return m; }
DEV_INLINE void iph_store(unsigned long long* fcol, int c, const IphCol& m) { fcol[c] = (unsigned long long)m.a | ((unsigned long long)m.b << 16) | ((unsigned long long)m.x << 32) | ((unsigned long long)m.y << 48); }

#ifdef IPH_STOP_AFTER   // development (instruction counts per phase, profiles/r04_phase_counts.txt): the kernel ends after phase IPH_STOP_AFTER (1 = A ... 3 = C);
#define IPH_STOP(k) do { if ((k) == IPH_STOP_AFTER) return; } while (0)   // nothing downstream may run (tools/kernel_times.py 512 1 ...: ImageProjection only)
#else
#define IPH_STOP(k)
#endif
#ifndef IPH_MINW
#define IPH_MINW 4   // wavefronts per SIMD the register budget allows (4: 128 VGPRs)
#endif
template <int T, int NP>
__global__ void __launch_bounds__(T, IPH_MINW) ip_fused_t(DevCtx d, int ring_pos, int keep) {   // 4 wavefronts per SIMD = 128 VGPRs: two workgroups of 512 threads per CU
  constexpr int NW = T / 64;
  const int slot = blockIdx.x + d.slot0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = d.N, H = d.H, NS = d.NS;
  const size_t base = (size_t)slot * N;
  const alego_params& P = d.P;
  extern __shared__ __attribute__((aligned(16))) unsigned char ipf_smem[];
  unsigned* own16w = reinterpret_cast<unsigned*>(ipf_smem);                                  // [N / 2] phases A, B: two 16-bit owners (index + 1, 0 = empty) per word
  uint16_t* par = reinterpret_cast<uint16_t*>(ipf_smem);                                     // [N] from phase C on
  unsigned long long* fcol = reinterpret_cast<unsigned long long*>(ipf_smem + (size_t)2 * N);   // [H] row masks of every column
  unsigned* own_g = d.ipf_own + (size_t)slot * (N / 2);                                      // the packed owners between phase B and phase D
  __shared__ IphShared<NW, NP> S;
  const int hpairs = H / 2;
  IPF_TICK(0);
  // ---------------- phase A: projection (as ip_fused; 16-bit owners) ----------------
  for (int v = tid; v < N / 2; v += T) own16w[v] = 0u;
  if (tid == 0) S.nlist = 0;
  __syncthreads();
  const int n = scan_count(d, slot, ring_pos);
  const float4* pts = scan_pts(d, slot, ring_pos);
  { IPH_PHASE_CTX;   // (d and P of this phase: the kernel argument read afresh — see IPH_PHASE_CTX)
  {
    int* s_list = reinterpret_cast<int*>(fcol);
    const int list_cap = 2 * H;   // 8 H bytes
    float qmr, qmc;
    ip_quick_margins(d, &qmr, &qmc);
    const IpQuickConst qc = ip_quick_const(d);
    int vmin = 0x7fffffff, vmax = -1, nvalid = 0;
#if IPH_PREFETCH
    float4 pnx[4];   // the next iteration's points are in flight while this one's are projected
#pragma unroll
    for (int u = 0; u < 4; ++u) pnx[u] = pts[max(min(tid + u * T, n - 1), 0)];
#endif
#pragma unroll 1
    for (int i0 = tid; i0 < n; i0 += T * 4) {
      float4 pin[4];
#if IPH_PREFETCH
#pragma unroll
      for (int u = 0; u < 4; ++u) pin[u] = pnx[u];
      if (i0 + T * 4 < n) {
#pragma unroll
        for (int u = 0; u < 4; ++u) pnx[u] = pts[min(i0 + T * 4 + u * T, n - 1)];
      }
#else
#pragma unroll
      for (int u = 0; u < 4; ++u) pin[u] = pts[min(i0 + u * T, n - 1)];
#endif
      // the four points of the iteration side by side: decisions without branches, the four owner words fetched together, one ballot for the (rare) deferrals
      int cellv[4];
      bool deferv[4];
      bool anydefer = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * T;
        const bool in = i < n;
        bool valid;
        int cell;
        const bool dec = ip_point_quick_bf(qc, pin[u], qmr, qmc, &valid, &cell);
        cellv[u] = in ? cell : -1;
        deferv[u] = in && !dec;
        anydefer |= deferv[u];
        const bool v = in && valid;
        vmin = min(vmin, v ? i : 0x7fffffff); vmax = max(vmax, v ? i : -1); nvalid += v ? 1 : 0;
      }
      unsigned oldw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) oldw[u] = __hip_atomic_load(own16w + (max(cellv[u], 0) >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int u = 0; u < 4; ++u) {   // later points overwrite earlier ones (:102-103): a 16-bit maximum through the aligned word (iph_max16, its first read hoisted)
        if (cellv[u] >= 0) {
          unsigned* w = own16w + (cellv[u] >> 1);
          const int sh = (cellv[u] & 1) * 16;
          const unsigned val = (unsigned)(i0 + u * T + 1);
          unsigned old = oldw[u];
          while (((old >> sh) & 0xFFFFu) < val) {
            const unsigned nw = (old & ~(0xFFFFu << sh)) | (val << sh);
            const unsigned got = atomicCAS(w, old, nw);
            if (got == old) break;
            old = got;
          }
        }
      }
      if (__ballot(anydefer)) {   // wavefront-aggregated append
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned long long dm = __ballot(deferv[u]);
          if (dm) {
            int lb = 0;
            if (lane == 0) lb = atomicAdd(&S.nlist, (int)__popcll(dm));
            lb = __shfl(lb, 0, 64);
            if (deferv[u]) {
              const int pos = lb + (int)__popcll(dm & ((1ull << lane) - 1ull));
              if (pos < list_cap) s_list[pos] = i0 + u * T;
            }
          }
        }
      }
    }
    __syncthreads();
    {
      const bool all = S.nlist > list_cap;
      const int nl = all ? n : S.nlist;
#pragma unroll 1
      for (int j = tid; j < nl; j += T) {
        const int i = all ? j : s_list[j];
        bool v2;
        const int c2 = ip_point_cell(d, pts[i], &v2);
        if (c2 >= 0) iph_max16(own16w, c2, (unsigned)(i + 1));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      vmin = min(vmin, __shfl_xor(vmin, o, 64));
      vmax = max(vmax, __shfl_xor(vmax, o, 64));
      nvalid += __shfl_xor(nvalid, o, 64);
    }
    if (lane == 0) { S.red[0][wave] = vmin; S.red[1][wave] = vmax; S.red[2][wave] = nvalid; }
  }
  __syncthreads();
  if (tid == T - 1) {   // orientation block (:62-72)
    int first = 0x7fffffff, last = -1, pv = 0;
    for (int w = 0; w < NW; ++w) { first = min(first, S.red[0][w]); last = max(last, S.red[1][w]); pv += S.red[2][w]; }
    d.scal[slot * SC_COUNT + SC_PVALID_OUT] = pv;
    if (last >= 0) {
      float* ori = d.ori + slot * 4;
      const float4 p0 = pts[first], p1 = pts[last];
      float so = -d_atan2f(p0.y, p0.x);
      float eo = (float)((double)(-d_atan2f(p1.y, p1.x)) + 2 * M_PI);
      if ((double)(eo - so) > 3 * M_PI) eo = (float)((double)eo - 2 * M_PI);
      else if ((double)(eo - so) < M_PI) eo = (float)((double)eo + 2 * M_PI);
      ori[0] = so; ori[1] = eo; ori[2] = eo - so;
    }
  }

  }
  IPH_STOP(1);
  IPF_TICK(1);
  // ---------------- phase B: ranges, ground, edges — one column pair per thread and pass ----------------
  { IPH_PHASE_CTX;   // (d and P of this phase: the kernel argument read afresh — see IPH_PHASE_CTX)
#pragma unroll 1
  for (int p = 0; p < NP; ++p) {
    const int pi = tid + p * T, c0 = 2 * pi, c1 = c0 + 1;
    const bool colv = pi < hpairs;
    float rng0[IPF2_ROWS], rng1[IPF2_ROWS];
    unsigned filled0 = 0, filled1 = 0, ground0 = 0, ground1 = 0;
    {
      float lx0 = 0, ly0 = 0, lz0 = 0, lx1 = 0, ly1 = 0, lz1 = 0;
      bool lok0 = false, lok1 = false;
#pragma unroll
      for (int row0 = 0; row0 < IPF2_ROWS; row0 += IPH_GB) {
        unsigned ob[IPH_GB];
        float4 pa[IPH_GB], pb[IPH_GB];
#pragma unroll
        for (int u = 0; u < IPH_GB; ++u) ob[u] = (colv && row0 + u < NS) ? own16w[((row0 + u) * H + c0) >> 1] : 0u;
#pragma unroll
        for (int u = 0; u < IPH_GB; ++u) { pa[u] = pts[max((int)(ob[u] & 0xFFFFu) - 1, 0)]; pb[u] = pts[max((int)(ob[u] >> 16) - 1, 0)]; }
#pragma unroll
        for (int u = 0; u < IPH_GB; ++u) {
          const int row = row0 + u;
          if (colv && row < NS) own_g[(row * H + c0) >> 1] = ob[u];
          {
            const bool ok = (ob[u] & 0xFFFFu) != 0u;
            const float x = pa[u].x, y = pa[u].y, z = pa[u].z;
            rng0[row] = ok ? sqrtf(x * x + y * y + z * z) : -1.0f;   // :99
            if (ok) filled0 |= 1u << row;
            if (row >= 1 && row - 1 < P.ground_scan_id && ok && lok0 && ip_is_ground(d, x - lx0, y - ly0, z - lz0)) ground0 |= 3u << (row - 1);   // :111-131
            lx0 = x; ly0 = y; lz0 = z; lok0 = ok;
          }
          {
            const bool ok = (ob[u] >> 16) != 0u;
            const float x = pb[u].x, y = pb[u].y, z = pb[u].z;
            rng1[row] = ok ? sqrtf(x * x + y * y + z * z) : -1.0f;
            if (ok) filled1 |= 1u << row;
            if (row >= 1 && row - 1 < P.ground_scan_id && ok && lok1 && ip_is_ground(d, x - lx1, y - ly1, z - lz1)) ground1 |= 3u << (row - 1);
            lx1 = x; ly1 = y; lz1 = z; lok1 = ok;
          }
        }
      }
    }
    const unsigned act0 = filled0 & ~ground0, act1 = filled1 & ~ground1;
    __syncthreads();   // (the previous pass has read S.u.b.first_r)
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < IPF2_ROWS; ++r) S.u.b.first_r[wave][r] = rng0[r];
      S.u.b.first_act[wave] = act0;
    }
    if (p == 0 && tid == 0) {
#pragma unroll
      for (int r = 0; r < IPF2_ROWS; ++r) S.col0_r[r] = rng0[r];
      S.col0_act = act0;
    }
    // The first column of the NEXT pass is the right neighbour of this pass's last column.  Sixteen lanes of the last wavefront take one row each — ONE gather
    // round trip — where the pass's last thread used to walk the sixteen rows on its own, every gather waiting for the one before it (~25 us with the
    // whole workgroup waiting at the next barrier; round 5, found in the ISA).  Same points, same differences, same predicates.
    {
      const int pil = (T - 1) + p * T;   // the pair of the pass's last thread
      if (p + 1 < NP && pil < hpairs - 1 && wave == NW - 1 && lane < IPF2_ROWS) {
        const int row = lane, cx = 2 * pil + 2;
        const unsigned ow = row < NS ? (own16w[(row * H + cx) >> 1] & 0xFFFFu) : 0u;
        const float4 q = pts[max((int)ow - 1, 0)];
        const bool ok = ow != 0u;
        const float lx = __shfl_up(q.x, 1, 64), ly = __shfl_up(q.y, 1, 64), lz = __shfl_up(q.z, 1, 64);
        const bool lok = __shfl_up(ok ? 1 : 0, 1, 64) != 0;
        const bool g = row >= 1 && row - 1 < P.ground_scan_id && ok && lok && ip_is_ground(d, q.x - lx, q.y - ly, q.z - lz);
        const unsigned gm = (unsigned)__ballot(g) & 0xFFFFu, fm = (unsigned)__ballot(ok) & 0xFFFFu;
        S.np.np_r[row] = ok ? sqrtf(q.x * q.x + q.y * q.y + q.z * q.z) : -1.0f;
        if (lane == 0) S.np.np_act = fm & ~(gm | (gm >> 1));   // (a ground pair (row - 1, row) marks both of its rows, :128-129)
      }
    }
    __syncthreads();
    // the column to the right of c1: the next thread's first column; column 0 behind the last one (:241-248); the first column of the
    // NEXT pass behind this pass's last thread, which gathers that one column itself
    float nbr[IPF2_ROWS];
    unsigned nb_act = (unsigned)__shfl_down((int)act0, 1, 64);
#pragma unroll
    for (int r = 0; r < IPF2_ROWS; ++r) nbr[r] = __shfl_down(rng0[r], 1, 64);
    const bool wrap = pi == hpairs - 1;                                   // right neighbour = column 0
    const bool nextpass = !wrap && tid == T - 1 && p + 1 < NP;   // right neighbour = first column of the next pass
    if ((lane == 63 && !nextpass) || wrap) {
      const int sw = min(wave + 1, NW - 1);
#pragma unroll
      for (int r = 0; r < IPF2_ROWS; ++r) nbr[r] = wrap ? S.col0_r[r] : S.u.b.first_r[sw][r];
      nb_act = wrap ? S.col0_act : S.u.b.first_act[sw];
    }
    if (nextpass) {
#pragma unroll
      for (int r = 0; r < IPF2_ROWS; ++r) nbr[r] = S.np.np_r[r];
      nb_act = S.np.np_act;
    }
    unsigned redge0 = 0, redge1 = 0, down0 = 0, down1 = 0;
    if (colv) {
      // (the constants of the edge predicate as values of their own: read straight from `d` they belong to the 16-register tuple one s_load of the kernel
      // argument brought in; the kernel uses more uniform values than there are SGPRs, the allocator parks that tuple in VGPR lanes and read all 16 lanes
      // back in front of every one of the 64 edge sites)
      double k_sax = d.sin_ax, k_cax = d.cos_ax, k_say = d.sin_ay, k_cay = d.cos_ay, k_tan = d.tan_theta, k_theta = P.seg_theta;
      asm volatile("" : "+s"(k_sax), "+s"(k_cax), "+s"(k_say), "+s"(k_cay), "+s"(k_tan), "+s"(k_theta));
#pragma unroll
      for (int row = 0; row < IPF2_ROWS; ++row) {
        if (row < NS) {
          if ((act0 >> row) & 1u) {
            const double r0 = (double)rng0[row];
            if ((act1 >> row) & 1u) {   // same row, seg_alpha_x (:258-261)
              const double r1 = (double)rng1[row], d1 = fmax(r0, r1), d2 = fmin(r0, r1);
              if (edge_angle_gt(d2 * k_sax, d1 - d2 * k_cax, k_theta, k_tan)) redge0 |= 1u << row;
            }
            if (row + 1 < NS && ((act0 >> (row + 1)) & 1u)) {   // same column, seg_alpha_y (:262-265)
              const double r1 = (double)rng0[row + 1 < IPF2_ROWS ? row + 1 : row], d1 = fmax(r0, r1), d2 = fmin(r0, r1);
              if (edge_angle_gt(d2 * k_say, d1 - d2 * k_cay, k_theta, k_tan)) down0 |= 1u << row;
            }
          }
          if ((act1 >> row) & 1u) {
            const double r0 = (double)rng1[row];
            if ((nb_act >> row) & 1u) {
              const double r1 = (double)nbr[row], d1 = fmax(r0, r1), d2 = fmin(r0, r1);
              if (edge_angle_gt(d2 * k_sax, d1 - d2 * k_cax, k_theta, k_tan)) redge1 |= 1u << row;
            }
            if (row + 1 < NS && ((act1 >> (row + 1)) & 1u)) {
              const double r1 = (double)rng1[row + 1 < IPF2_ROWS ? row + 1 : row], d1 = fmax(r0, r1), d2 = fmin(r0, r1);
              if (edge_angle_gt(d2 * k_say, d1 - d2 * k_cay, k_theta, k_tan)) down1 |= 1u << row;
            }
          }
        }
      }
      fcol[c0] = (unsigned long long)ground0 | ((unsigned long long)act0 << 16) | ((unsigned long long)redge0 << 32) | ((unsigned long long)down0 << 48);
      fcol[c1] = (unsigned long long)ground1 | ((unsigned long long)act1 << 16) | ((unsigned long long)redge1 << 32) | ((unsigned long long)down1 << 48);
      if (keep & 1) {
        float* rimg = d.range_img + base;
        uint8_t* fimg = d.flag_img + base;
#pragma unroll
        for (int row = 0; row < IPF2_ROWS; ++row) {
          if (row < NS) {
            rimg[row * H + c0] = rng0[row]; rimg[row * H + c1] = rng1[row];
            fimg[row * H + c0] = (uint8_t)(((ground0 >> row) & 1u) | (((act0 >> row) & 1u) << 1) | (((redge0 >> row) & 1u) << 2) | (((down0 >> row) & 1u) << 3));
            fimg[row * H + c1] = (uint8_t)(((ground1 >> row) & 1u) | (((act1 >> row) & 1u) << 1) | (((redge1 >> row) & 1u) << 2) | (((down1 >> row) & 1u) << 3));
          }
        }
      }
    }
  }
  __syncthreads();   // every read of the owner image has happened: its LDS becomes the parent array

  }
  IPH_STOP(2);
  IPF_TICK(3);
  // ---------------- phase C: connected components over vertical runs (ip_fused's steps, column by column from the masks in LDS) ----------------
  { IPH_PHASE_CTX;   // (d and P of this phase: the kernel argument read afresh — see IPH_PHASE_CTX)
  // a cell starts a run when it is active and no down-edge reaches it from below; a run's representative is its first (lowest) cell
  const int ncol = NP * 2;   // columns of a thread: 2 (tid + p T) + k
  auto col_of = [&](int q) -> int { return 2 * (tid + (q >> 1) * T) + (q & 1); };
  unsigned* parw = reinterpret_cast<unsigned*>(par);
#pragma unroll 1
  for (int p = 0; p < NP; ++p) {
    const int pi = tid + p * T, c0 = 2 * pi, c1 = c0 + 1;
    if (pi < hpairs) {
      const IphCol m0 = iph_load(fcol, c0), m1 = iph_load(fcol, c1);
      const unsigned rs0 = m0.b & ~(m0.y << 1), rs1 = m1.b & ~(m1.y << 1);
#pragma unroll
      for (int row = 0; row < IPF2_ROWS; ++row) {
        if (row < NS) {
          const int s0 = ((m0.b >> row) & 1u) ? ipf_run_start(rs0, row) : row, s1 = ((m1.b >> row) & 1u) ? ipf_run_start(rs1, row) : row;
          parw[(row * H + c0) >> 1] = (unsigned)(s0 * H + c0) | ((unsigned)(s1 * H + c1) << 16);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int p = 0; p < NP; ++p) {
    const int pi = tid + p * T, c0 = 2 * pi, c1 = c0 + 1;
    if (pi < hpairs) {
      const int cn = c0 + 2 == H ? 0 : c0 + 2;
      const IphCol a0 = iph_load(fcol, c0), a1 = iph_load(fcol, c1);
      // right-edges between the runs; one is skipped when the cell below already links the same pair of runs
      const unsigned down_nb = (unsigned)(fcol[cn] >> 48) & 0xFFFFu;
      unsigned m0 = a0.x & ~((a0.x & a0.y & a1.y) << 1);
      unsigned m1 = a1.x & ~((a1.x & a1.y & down_nb) << 1);
      while (m0) { const int row = __ffs((int)m0) - 1; m0 &= m0 - 1; ccl16_union(par, row * H + c0, row * H + c1); }
      while (m1) { const int row = __ffs((int)m1) - 1; m1 &= m1 - 1; ccl16_union(par, row * H + c1, row * H + cn); }
    }
  }
  __syncthreads();
  // flatten the run starts; root = minimum linear index of the component = BFS discovery order (:147-156).  The roots take the right-edges' place in fcol.
#pragma unroll 1
  for (int q = 0; q < ncol; ++q) {
    const int c = col_of(q);
    if (c < H) {
      IphCol m = iph_load(fcol, c);
      unsigned root = 0;
      for (unsigned rm = m.b & ~(m.y << 1); rm; rm &= rm - 1) { const int row = __ffs((int)rm) - 1, s = row * H + c; int r = par[s], nx; while (r > (nx = par[r])) r = nx; par[s] = (uint16_t)r; if (r == s) root |= 1u << row; }
      m.x = root;
      iph_store(fcol, c, m);
    }
  }
  __syncthreads();
  // A root's own entry is free from here on: the component's 16-bit accumulator, first for the size (:282), then for the row mask (:283-294)
  auto zero_roots = [&]() {
#pragma unroll 1
    for (int q = 0; q < ncol; ++q) {
      const int c = col_of(q);
      if (c < H) for (unsigned m = (unsigned)(fcol[c] >> 32) & 0xFFFFu; m; m &= m - 1) par[(__ffs((int)m) - 1) * H + c] = 0;
    }
    __syncthreads();
  };
  auto root_of = [&](unsigned rootm, int row, int c) -> int { const int s = row * H + c; return ((rootm >> row) & 1u) ? s : (int)par[s]; };
  zero_roots();
#pragma unroll 1
  for (int q = 0; q < ncol; ++q) {
    const int c = col_of(q);
    if (c < H) {
      const IphCol m = iph_load(fcol, c);
      for (unsigned rm = m.b & ~(m.y << 1); rm; rm &= rm - 1) { const int row = __ffs((int)rm) - 1, r = root_of(m.x, row, c); atomicAdd(&parw[r >> 1], (unsigned)(ipf_run_end(m.y, row) - row + 1) << ((r & 1) * 16)); }
    }
  }
  __syncthreads();
  unsigned bm[NP * 2];   // per column: runs of big components | runs of mid-sized ones << 16 (bit at the run's start row)
#pragma unroll
  for (int q = 0; q < NP * 2; ++q) {
    const int c = col_of(q);
    unsigned big = 0, mid = 0;
    if (c < H) {
      const IphCol m = iph_load(fcol, c);
      for (unsigned rm = m.b & ~(m.y << 1); rm; rm &= rm - 1) { const int row = __ffs((int)rm) - 1, sz = (int)par[root_of(m.x, row, c)]; if (sz >= P.seg_big_num) big |= 1u << row; else if (sz >= P.seg_valid_point_num) mid |= 1u << row; }
    }
    bm[q] = big | (mid << 16);
  }
  __syncthreads();
  zero_roots();
#pragma unroll
  for (int q = 0; q < NP * 2; ++q) {   // rows touched by the mid-sized components
    const int c = col_of(q);
    if (c < H && (bm[q] >> 16)) {
      const IphCol m = iph_load(fcol, c);
      for (unsigned rm = bm[q] >> 16; rm; rm &= rm - 1) { const int row = __ffs((int)rm) - 1, r = root_of(m.x, row, c), e = ipf_run_end(m.y, row); atomicOr(&parw[r >> 1], (((2u << e) - 1u) & ~((1u << row) - 1u)) << ((r & 1) * 16)); }
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NP * 2; ++q) {   // cells of feasible components: they take the down-edges' place in fcol
    const int c = col_of(q);
    if (c < H) {
      IphCol m = iph_load(fcol, c);
      const unsigned big = bm[q] & 0xFFFFu, mid = bm[q] >> 16;
      unsigned feas = 0;
      for (unsigned rm = big | mid; rm; rm &= rm - 1) {
        const int row = __ffs((int)rm) - 1, e = ipf_run_end(m.y, row);
        if (((big >> row) & 1u) || __popc((unsigned)par[root_of(m.x, row, c)]) >= P.seg_valid_line_num) feas |= ((2u << e) - 1u) & ~((1u << row) - 1u);
      }
      if (keep & 1) {   // root image for ip_labels / the tests (:303-314), while the down-edges are still there
        const unsigned rs = m.b & ~(m.y << 1);
#pragma unroll
        for (int row = 0; row < IPF2_ROWS; ++row)
          if (row < NS) d.parent[base + row * H + c] = ((m.b >> row) & 1u) ? root_of(m.x, ipf_run_start(rs, row), c) : -1;
      }
      m.y = feas;
      iph_store(fcol, c, m);
    }
  }
  // (fcol entries are read and written by their own thread only from here on: no barrier)

  }
  IPH_STOP(3);
  IPF_TICK(8);
  // ---------------- phase D: ordered compaction (:158-191) ----------------
  { IPH_PHASE_CTX;   // (d and P of this phase: the kernel argument read afresh — see IPH_PHASE_CTX)
  const unsigned all16 = 0xFFFFu;
  const unsigned rowgt = P.ground_scan_id >= 15 ? 0u : (P.ground_scan_id < 0 ? all16 : (all16 & ~((2u << P.ground_scan_id) - 1u)));   // rows > ground_scan_id
  // keep | outliers << 16 of column c; feasible roots (label_cnt_ numbering, :303-306)
  auto col_sets = [&](int c, const IphCol& m, unsigned& keepm, unsigned& outlm, unsigned& frm) {
    const bool gk = c % 5 == 0 || c <= 4 || c >= H - 5;
    keepm = (gk ? m.a : 0u) | m.y;
    outlm = c % 5 == 0 ? (m.b & ~m.y & rowgt) : 0u;
    frm = m.x & m.y;
  };
#pragma unroll 1
  for (int p = 0; p < NP; ++p) {
    const int pi = tid + p * T, c0 = 2 * pi, c1 = c0 + 1;
    unsigned k0 = 0, k1 = 0, o0 = 0, o1 = 0, f0 = 0, f1 = 0;
    if (pi < hpairs) { col_sets(c0, iph_load(fcol, c0), k0, o0, f0); col_sets(c1, iph_load(fcol, c1), k1, o1, f1); }
#pragma unroll
    for (int row = 0; row < IPF2_ROWS; ++row) {
      const unsigned long long bk0 = __ballot((k0 >> row) & 1u), bk1 = __ballot((k1 >> row) & 1u);
      const unsigned long long bo0 = __ballot((o0 >> row) & 1u), bo1 = __ballot((o1 >> row) & 1u);
      const unsigned long long bf0 = __ballot((f0 >> row) & 1u), bf1 = __ballot((f1 >> row) & 1u);
      if (lane == 0) {
        const int e = (row * NP + p) * NW + wave;
        S.u.cnt[0][e] = (unsigned short)(__popcll(bk0) + __popcll(bk1));
        S.u.cnt[1][e] = (unsigned short)(__popcll(bo0) + __popcll(bo1));
        S.u.cnt[2][e] = (unsigned short)(__popcll(bf0) + __popcll(bf1));
      }
    }
  }
  __syncthreads();
  constexpr int NCNT = IPF2_ROWS * NP * NW;   // 256 (ip_fused_h) / 512 (ip_fused_w) <= T
  {   // exclusive scan of the three tables by the first NCNT threads
    int v3[3] = {0, 0, 0}, in3[3] = {0, 0, 0};
    if (tid < NCNT) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        v3[a] = S.u.cnt[a][tid];
        int incl = v3[a];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        in3[a] = incl;
        if (lane == 63) S.wtot[a][wave] = incl;
      }
    }
    __syncthreads();
    if (tid < NCNT) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        int woff = 0;
#pragma unroll
        for (int w = 0; w < NCNT / 64; ++w) if (w < wave) woff += S.wtot[a][w];
        S.u.cnt[a][tid] = (unsigned short)(woff + in3[a] - v3[a]);
        if (tid == NCNT - 1) S.tot[a] = woff + in3[a];
      }
    }
  }
  __syncthreads();
  int* const o_rs = IPH_LATE(ring_start) + slot * NS; int* const o_re = IPH_LATE(ring_end) + slot * NS; int* const o_sc = IPH_LATE(scal) + slot * SC_COUNT;   // (fetched in uniform code)
  if (tid < NS) {   // startRingIndex / endRingIndex (:161,:190)
    const int row = tid;
    o_rs[row] = (int)S.u.cnt[0][row * NP * NW] + 5;
    o_re[row] = (row + 1 < IPF2_ROWS ? (int)S.u.cnt[0][(row + 1) * NP * NW] : S.tot[0]) - 1 - 5;
  }
  if (tid == 0) {
    int* sc = o_sc;
    sc[SC_M] = S.tot[0]; sc[SC_NOUT] = S.tot[1]; sc[SC_NFEAS] = S.tot[2];
  }
  IPF_TICK(9);
  const unsigned long long below = (1ull << lane) - 1ull;
  float4* const o_pts = IPH_LATE(seg_pts) + base; uint8_t* const o_gnd = IPH_LATE(seg_ground) + base; int* const o_col = IPH_LATE(seg_col) + base;
  float* const o_rng = IPH_LATE(seg_range) + base; float4* const o_out = IPH_LATE(outlier) + base; int* const o_lab = IPH_LATE(cc_label) + base;
#pragma unroll 1
  for (int p = 0; p < NP; ++p) {
    const int pi = tid + p * T, c0 = 2 * pi, c1 = c0 + 1;
    const bool colv = pi < hpairs;
    unsigned keep0 = 0, keep1 = 0, outl0 = 0, outl1 = 0, fr0 = 0, fr1 = 0, ground0 = 0, ground1 = 0, root0 = 0, root1 = 0;
    if (colv) {
      const IphCol m0 = iph_load(fcol, c0), m1 = iph_load(fcol, c1);
      col_sets(c0, m0, keep0, outl0, fr0); col_sets(c1, m1, keep1, outl1, fr1);
      ground0 = m0.a; ground1 = m1.a; root0 = m0.x; root1 = m1.x;
    }
    const double cf0 = d.ip_colfrac[min(c0, H - 1)], cf1 = d.ip_colfrac[min(c1, H - 1)];   // c / 10000.0 (host division: two fp64 divisions per pass and thread less)
#if IPH_OWN_AHEAD
    unsigned owa[IPF2_ROWS];   // the packed owners of all sixteen rows at once: one round trip in front of the gathers instead of one per group of rows
#pragma unroll
    for (int r = 0; r < IPF2_ROWS; ++r) owa[r] = (colv && r < NS) ? own_g[(r * H + c0) >> 1] : 0u;
#endif
#pragma unroll
    for (int row0 = 0; row0 < IPF2_ROWS; row0 += IPH_GD) {
      unsigned ow[IPH_GD];
      float4 qa[IPH_GD], qb[IPH_GD];
#pragma unroll
#if IPH_OWN_AHEAD
      for (int u = 0; u < IPH_GD; ++u) ow[u] = owa[row0 + u];
#else
      for (int u = 0; u < IPH_GD; ++u) ow[u] = (colv && row0 + u < NS) ? own_g[((row0 + u) * H + c0) >> 1] : 0u;
#endif
#pragma unroll
      for (int u = 0; u < IPH_GD; ++u) {
        const int row = row0 + u;
        const bool e0 = ((keep0 | outl0) >> row) & 1u, e1 = ((keep1 | outl1) >> row) & 1u;
        qa[u] = pts[e0 ? (int)(ow[u] & 0xFFFFu) - 1 : 0];   // (clamped address instead of a branch: the eight gathers are in flight together)
        qb[u] = pts[e1 ? (int)(ow[u] >> 16) - 1 : 0];
      }
#ifdef IPH_EXTRA_GATHER
      // development (what does the gathers' read amplification cost?): the same gather once more on the NEXT scan's points —
      // other cache lines, the same pattern: one more pass of 128-byte requests over a scan's 0.41 MB — folded into a value that is never stored
      {
        const float4* pts2 = scan_pts(d, slot, ring_pos + 1);
        float sink = 0.f;
#pragma unroll
        for (int u = 0; u < IPH_GD; ++u) {
          const int row = row0 + u;
          const bool e0 = ((keep0 | outl0) >> row) & 1u, e1 = ((keep1 | outl1) >> row) & 1u;
          sink += pts2[e0 ? (int)(ow[u] & 0xFFFFu) - 1 : 0].x + pts2[e1 ? (int)(ow[u] >> 16) - 1 : 0].y;
        }
        if (sink == 1.2345e38f) d.seg_range[base] = sink;
      }
#endif
#pragma unroll
      for (int u = 0; u < IPH_GD; ++u) {
        const int row = row0 + u;
        const bool k0 = (keep0 >> row) & 1u, k1 = (keep1 >> row) & 1u, o0 = (outl0 >> row) & 1u, o1 = (outl1 >> row) & 1u;
        const unsigned long long bk0 = __ballot(k0), bk1 = __ballot(k1), bo0 = __ballot(o0), bo1 = __ballot(o1);
        const int e = (row * NP + p) * NW + wave;
        const int lk = (int)S.u.cnt[0][e] + (int)(__popcll(bk0 & below) + __popcll(bk1 & below));
        const int lo = (int)S.u.cnt[1][e] + (int)(__popcll(bo0 & below) + __popcll(bo1 & below));
        if (k0 | o0) {
          const float4 q = qa[u];
          const float4 pp = make_float4(q.x, q.y, q.z, (float)(row + cf0));   // :101
          if (k0) {
            o_pts[lk] = pp;
            o_gnd[lk] = (uint8_t)((ground0 >> row) & 1u);
            o_col[lk] = c0;
            o_rng[lk] = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z);   // = the range image's value (:99,:184)
          } else {
            o_out[lo] = pp;
          }
        }
        if (k1 | o1) {
          const float4 q = qb[u];
          const float4 pp = make_float4(q.x, q.y, q.z, (float)(row + cf1));
          const int l1 = lk + (k0 ? 1 : 0), lo1 = lo + (o0 ? 1 : 0);
          if (k1) {
            o_pts[l1] = pp;
            o_gnd[l1] = (uint8_t)((ground1 >> row) & 1u);
            o_col[l1] = c1;
            o_rng[l1] = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z);
          } else {
            o_out[lo1] = pp;
          }
        }
      }
    }
    if (keep & 1) {   // label_cnt_ numbers of the feasible roots for ip_labels (:303-314)
#pragma unroll
      for (int row = 0; row < IPF2_ROWS; ++row) {
        const unsigned long long bf0 = __ballot((fr0 >> row) & 1u), bf1 = __ballot((fr1 >> row) & 1u);
        if (colv && row < NS) {
          const int nf = (int)S.u.cnt[2][(row * NP + p) * NW + wave] + (int)(__popcll(bf0 & below) + __popcll(bf1 & below));
          if ((root0 >> row) & 1u) o_lab[row * H + c0] = ((fr0 >> row) & 1u) ? nf + 1 : 0;
          if ((root1 >> row) & 1u) o_lab[row * H + c1] = ((fr1 >> row) & 1u) ? nf + ((fr0 >> row) & 1u) + 1 : 0;
        }
      }
    }
  }
  }
  IPF_TICK(10);
}

static constexpr auto ip_fused_h = ip_fused_t<IPH_T, IPH_NP>;
static constexpr auto ip_fused_w = ip_fused_t<IPW_T, IPW_NP>;
// IPH_PHASE_CTX / IPH_LATE read DevCtx members at offsetof(DevCtx, member) of the kernel-argument segment: that is only the by-value argument `d` while DevCtx is the
// FIRST explicit parameter of ip_fused_t (offset 0 of the segment, HSA code-object ABI) and a plain-layout struct.  Enforced here, not by a comment:
template <class F> struct iph_first_arg;
template <class A0, class... R> struct iph_first_arg<void (*)(A0, R...)> { typedef A0 type; };
static_assert(std::is_same<iph_first_arg<decltype(&ip_fused_t<IPH_T, IPH_NP>)>::type, DevCtx>::value && std::is_same<iph_first_arg<decltype(&ip_fused_t<IPW_T, IPW_NP>)>::type, DevCtx>::value,
              "ip_fused_t: DevCtx must stay the first kernel parameter (IPH_PHASE_CTX / IPH_LATE read it at offset 0 of the kernel-argument segment)");
static_assert(std::is_standard_layout<DevCtx>::value && std::is_trivially_copyable<DevCtx>::value, "DevCtx is passed by value and re-read through offsetof()");

void launch_ip_fused(const DevCtx& d, int ring_pos, bool keep_images, hipStream_t st) {
  if (d.opt_ip_half && iph_eligible(d)) { ALEGO_LAUNCH(ip_fused_h, dim3(d.n_launch), dim3(IPH_T), iph_lds_bytes(d) + iph_pad(), st, d, ring_pos, keep_images ? 1 : 0); return; }
  if (!ipf_eligible(d)) { ALEGO_LAUNCH(ip_fused_w, dim3(d.n_launch), dim3(IPW_T), iph_lds_bytes(d), st, d, ring_pos, keep_images ? 1 : 0); return; }   // (the caller checked ipw_eligible)

  ALEGO_LAUNCH(ip_fused, dim3(d.n_launch), dim3(IPF2_T), ipf_lds_bytes(d), st, d, ring_pos, keep_images ? 1 : 0);
}

// dynamic LDS above 64 KB has to be requested explicitly
int ipf_configure(const DevCtx& d) {
  if (!ipf_eligible(d)) {
    if (ipw_eligible(d) && hipFuncSetAttribute(reinterpret_cast<const void*>(ip_fused_w), hipFuncAttributeMaxDynamicSharedMemorySize, (int)iph_lds_bytes(d)) != hipSuccess) return -1;
    return 0;
  }
  if (iph_eligible(d) && hipFuncSetAttribute(reinterpret_cast<const void*>(ip_fused_h), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(iph_lds_bytes(d) + iph_pad())) != hipSuccess) return -1;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(ip_fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ipf_lds_bytes(d)) == hipSuccess ? 0 : -1;
}
