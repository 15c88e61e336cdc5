// kernels_ip.hip — ImageProjection on gfx950 (replaces src/imageProjection.cpp:49-316).
//
//   ip_project    a1-a3: NaN/near filter, row/col from the shared fdlibm atan2f, last-writer-wins
//                 scatter resolved by atomicMax on the input index          (:58-59,:76-104)
//   ip_front      a2,a3,a4 + a5's edges: orientation, range gather, per-column ground test, 4-neighbour edge predicate
//                 atan2(d2 sin a, d1 - d2 cos a) > theta, one launch (:62-72,:107-143,:255-270)
//   cc_lds16      a5 + a6 for <= 16 rings (16x1800, 16x4000): union-find with 2 B/cell in LDS, statistics, feasibility and the
//                 ordered compaction in one workgroup per stream (:147-191,:210-316)
//   cc_lds        a5 for small images with more than 16 rings (4 B/cell in LDS)
//   cc_tile / cc_seam   a5 for larger images (64x2048): column bands labelled in LDS, seams linked in global memory
//   cc_runs / cc_link   a5, the global-memory union-find they replaced (ALEGO_CC_TILE=0): root = minimum linear index == BFS discovery order (:147-156)
//   cc_stats      a5: per-component size and row mask -> feasibility (:282-301)
//   ip_rowcount / ip_compact  a6: per-row ballot compaction with the +5/-6 ring convention (:158-191)
//   ip_labels     label_mat_ numbering 1,2,.. in discovery order / 999999 / -1 (:303-314)
//
// All kernels are launched with blockIdx.y = slot (independent stream).
// HBM-bound: one scan moves 16 B/point in and 25 B/segmented point out; the images
// in between (owner, range, flags, parent: 13 B/cell) stay L2-resident.
#include <cstdlib>
#include "dev_common.h"
#include "ip_common.h"
#include "prof.h"

#define IP_BLOCK 256
#define IP_PW 4   // points / cells per thread in the streaming kernels

__global__ void __launch_bounds__(IP_BLOCK) ip_project(DevCtx d, int ring_pos) {
  const int slot = blockIdx.y + d.slot0;
  const int i0 = blockIdx.x * IP_BLOCK * IP_PW + threadIdx.x;   // IP_PW points per thread, their loads in flight together
  const int n = scan_count(d, slot, ring_pos);
  const float4* in = scan_pts(d, slot, ring_pos);
  float4 pin[IP_PW];
#pragma unroll
  for (int u = 0; u < IP_PW; ++u) pin[u] = in[min(i0 + u * IP_BLOCK, max(n - 1, 0))];
  int vmin = 0x7fffffff, vmax = -1, nvalid = 0;
  // Two passes, as in ip_fused: ip_point_quick decides the points away from cell boundaries from an estimate of their angles, the others go
  // onto the workgroup's list in LDS and through ip_point_cell afterwards, all lanes busy (last writer wins by atomicMax: order immaterial).
  __shared__ int s_def[IP_BLOCK * IP_PW];
  __shared__ int s_ndef;
  if (threadIdx.x == 0) s_ndef = 0;
  __syncthreads();
  float qmr, qmc;
  ip_quick_margins(d, &qmr, &qmc);
  const IpQuickConst qc = ip_quick_const(d);
  const int lane = lane_id();
  // the IP_PW points side by side (round 6, as ip_fused_t's point loop since round 5): decisions without branches (ip_point_quick_bf), one test for the (rare) deferrals
  bool deferv[IP_PW];
  bool anydefer = false;
#pragma unroll
  for (int u = 0; u < IP_PW; ++u) {
    const int i = i0 + u * IP_BLOCK;
    const bool in = i < n;
    bool valid;
    int cell;
    const bool dec = ip_point_quick_bf(qc, pin[u], qmr, qmc, &valid, &cell);
    deferv[u] = in && !dec;
    anydefer |= deferv[u];
    if (in && cell >= 0)
      atomicMax(&d.owner[(size_t)slot * d.N + cell], IP_OWNER_TAG | i);  // later points overwrite earlier ones (:102-103);
        // whatever the previous scan left in the cell (a plain index or -1, see ip_front) loses against a tagged entry: no reset pass
    const bool v = in && valid;
    vmin = min(vmin, v ? i : 0x7fffffff); vmax = max(vmax, v ? i : -1); nvalid += v ? 1 : 0;
  }
  if (__ballot(anydefer)) {
#pragma unroll
    for (int u = 0; u < IP_PW; ++u) {
      const unsigned long long dm = __ballot(deferv[u]);
      if (dm) {
        int b0 = 0;
        if (lane == 0) b0 = atomicAdd(&s_ndef, (int)__popcll(dm));
        b0 = __shfl(b0, 0, 64);
        if (deferv[u]) s_def[b0 + (int)__popcll(dm & ((1ull << lane) - 1ull))] = i0 + u * IP_BLOCK;   // (at most IP_BLOCK * IP_PW entries: one per point of the workgroup)
      }
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < s_ndef; j += IP_BLOCK) {
    const int i = s_def[j];
    bool v2;
    const int cell = ip_point_cell(d, in[i], &v2);
    if (cell >= 0) atomicMax(&d.owner[(size_t)slot * d.N + cell], IP_OWNER_TAG | i);
  }
  // first / last valid point for the orientation block (:62-63): one atomic per wavefront
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    vmin = min(vmin, __shfl_xor(vmin, o, 64));
    vmax = max(vmax, __shfl_xor(vmax, o, 64));
    nvalid += __shfl_xor(nvalid, o, 64);
  }
  if (lane_id() == 0 && nvalid) {
    int* sc = d.scal + slot * SC_COUNT;
    atomicMin(&sc[SC_FIRST], vmin);
    atomicMax(&sc[SC_LAST], vmax);
    atomicAdd(&sc[SC_PVALID], nvalid);
  }
}

DEV_INLINE int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV_INLINE void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ip_front = range image + ground labelling + edge predicates in one launch (imageProjection.cpp:62-72,:107-143,:255-270).
// A workgroup takes IPF_W - 1 consecutive columns plus the column to their right (halo, with wrap-around): one thread per
// column walks its rows bottom-up (owner -> point gather -> range, ground test on consecutive rows), keeps the column's
// ranges in LDS and its filled / ground masks in registers; after one barrier every thread knows its right neighbour's
// column and evaluates the right- and down-edge predicates of its own cells.  The images that used to make an HBM / L2
// round trip between ip_image and cc_edges (range 4 B/cell written + read three times, flags written + read + rewritten)
// stay on chip: written are the flag byte per cell (bit0 ground, bit1 active, bit2 edge->right, bit3 edge->down), the plain
// owner image (= the reset for the next scan), and — only for the single-scan entry points / tests — the range image.
//   init: bit 0 parent (global union-find path), bit 1 component statistics (cc_stats path), bit 2 labels (ip_compact path),
//         bit 3 write the range image
#define IPF_W 128
__global__ void __launch_bounds__(IPF_W) ip_front(DevCtx d, int ring_pos, int init) {
  const int slot = blockIdx.y + d.slot0;
  const int tid = threadIdx.x;
  const int col_raw = blockIdx.x * (IPF_W - 1) + tid;          // thread IPF_W - 1 computes the halo column
  const bool mine = tid < IPF_W - 1 && col_raw < d.H;           // this thread's column is written by this workgroup
  const int col = col_raw < d.H ? col_raw : col_raw - d.H;      // wrap-around (:241-248); H >= IPF_W so one subtraction is enough
  const bool have = col_raw <= d.H;                             // column exists (col_raw == H: the halo of the last workgroup is column 0)
  const alego_params& P = d.P;
  const float4* pts = scan_pts(d, slot, ring_pos);
  extern __shared__ __attribute__((aligned(16))) unsigned char ipf_smem[];
  float* s_r = reinterpret_cast<float*>(ipf_smem);                                   // [NS][IPF_W] ranges (-1 empty)
  unsigned long long* s_act = reinterpret_cast<unsigned long long*>(s_r + (size_t)d.NS * IPF_W);   // [IPF_W] active mask of a column
  if (blockIdx.x == 0 && tid == 0) {  // orientation, :62-72
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
  int* owner = d.owner + (size_t)slot * d.N;
  float* rimg = d.range_img + (size_t)slot * d.N;
  uint8_t* fimg = d.flag_img + (size_t)slot * d.N;
  unsigned long long filled = 0, ground = 0;
  float lx = 0, ly = 0, lz = 0;
  bool lower_ok = false;
  // rows in batches of IM_U: the owner indices of a batch, then its point gathers, are independent loads
  constexpr int IM_U = 8;
  for (int row0 = 0; row0 < d.NS; row0 += IM_U) {
    int ob[IM_U];
    float4 pb[IM_U];
#pragma unroll
    for (int u = 0; u < IM_U; ++u) ob[u] = (have && row0 + u < d.NS) ? owner[(row0 + u) * d.H + col] : -1;
    // entries of this scan carry the tag; everything else is stale.  The plain form is written back (by the column's own
    // workgroup) for the later readers (compaction) and doubles as the reset for the next scan — except for the workgroup's
    // FIRST column: the workgroup to the left reads that column as its halo, possibly much later (workgroups of one launch are
    // not co-scheduled when other streams keep the CUs busy), and would find the tag gone = an empty column = missing
    // right-edges.  Those columns stay tagged here; ip_strip_halo_columns (next kernel in the stream) finishes them.
#pragma unroll
    for (int u = 0; u < IM_U; ++u) {
      ob[u] = (ob[u] >= 0 && (ob[u] & IP_OWNER_TAG)) ? (ob[u] & ~IP_OWNER_TAG) : -1;
      if (mine && tid != 0 && row0 + u < d.NS) owner[(row0 + u) * d.H + col] = ob[u];
    }
#pragma unroll
    for (int u = 0; u < IM_U; ++u) pb[u] = ob[u] >= 0 ? pts[ob[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < IM_U; ++u) {
      const int row = row0 + u;
      if (row >= d.NS) break;
      const int o = ob[u];
      float r = -1.0f;
      const bool ok = o >= 0;
      float x = 0, y = 0, z = 0;
      if (ok) {
        const float4 p = pb[u];
        x = p.x; y = p.y; z = p.z;
        r = sqrtf(x * x + y * y + z * z);  // :99
        filled |= 1ull << row;
      }
      s_r[row * IPF_W + tid] = r;
      if (mine && (init & 8)) rimg[row * d.H + col] = r;
      if (row >= 1 && row - 1 < P.ground_scan_id && ok && lower_ok) {  // :111-131, pair (row-1,row)
        if (ip_is_ground(d, x - lx, y - ly, z - lz)) ground |= 3ull << (row - 1);
      }
      lx = x; ly = y; lz = z; lower_ok = ok;
    }
  }
  const unsigned long long act = filled & ~ground;
  s_act[tid] = act;
  __syncthreads();
  if (!mine) return;
  const unsigned long long act_r = s_act[tid + 1];
  const size_t base = (size_t)slot * d.N;
  for (int row = 0; row < d.NS; ++row) {
    const bool g = (ground >> row) & 1, a = (act >> row) & 1;
    uint8_t e = 0;
    if (a) {
      const double r0 = (double)s_r[row * IPF_W + tid];
      if (((act_r >> row) & 1) && d.H > 1) {  // same row, seg_alpha_x (:258-261)
        const double r1 = (double)s_r[row * IPF_W + tid + 1];
        const double d1 = fmax(r0, r1), d2 = fmin(r0, r1);
        if (edge_angle_gt(d2 * d.sin_ax, d1 - d2 * d.cos_ax, d.P.seg_theta, d.tan_theta)) e |= 4;
      }
      if (row + 1 < d.NS && ((act >> (row + 1)) & 1)) {  // same column, seg_alpha_y (:262-265)
        const double r1 = (double)s_r[(row + 1) * IPF_W + tid];
        const double d1 = fmax(r0, r1), d2 = fmin(r0, r1);
        if (edge_angle_gt(d2 * d.sin_ay, d1 - d2 * d.cos_ay, d.P.seg_theta, d.tan_theta)) e |= 8;
      }
    }
    const int v = row * d.H + col;
    fimg[v] = (uint8_t)((g ? 1 : 0) | (a ? 2 : 0) | e);
    if (init & 1) d.parent[base + v] = a ? v : -1;
    if (init & 2) { d.cc_size[base + v] = 0; d.cc_rows[base + v] = 0ull; }
    if (init & 4) d.cc_label[base + v] = 0;
  }
}

// The owner entries ip_front left tagged (the first column of each of its workgroups) -> plain form / -1.  Called by exactly one
// kernel per scan after ip_front (cc_lds16, cc_lds or cc_runs); readers of the owner image in the same kernel mask the tag off
// (ip_owner_index), so they may see either form.
DEV_INLINE void ip_strip_halo_columns(const DevCtx& d, int slot, int t, int nt) {
  int* owner = d.owner + (size_t)slot * d.N;
  const int nb = (d.H + IPF_W - 2) / (IPF_W - 1);
  for (int j = t; j < nb * d.NS; j += nt) {
    const int row = j / nb, col = (j - row * nb) * (IPF_W - 1);
    const int v = owner[row * d.H + col];
    owner[row * d.H + col] = (v >= 0 && (v & IP_OWNER_TAG)) ? (v & ~IP_OWNER_TAG) : -1;
  }
}
DEV_INLINE int ip_owner_index(int v) { return v & ~IP_OWNER_TAG; }   // of a FILLED cell (never -1)

// ECL-CC style find with intermediate pointer jumping; parents only ever decrease.
DEV_INLINE int cc_find(int* parent, int v) {
  int curr = ld_agent(parent + v);
  if (curr != v) {
    int prev = v, next;
    while (curr > (next = ld_agent(parent + curr))) {
      st_agent(parent + prev, next);
      prev = curr;
      curr = next;
    }
  }
  return curr;
}
// read-only find (no pointer jumping): used once linking is complete, so that the
// final roots written by cc_stats cannot be overwritten by another thread's path halving
DEV_INLINE int cc_find_ro(const int* parent, int v) {
  int curr = ld_agent(parent + v), next;
  while (curr > (next = ld_agent(parent + curr))) curr = next;
  return curr;
}
DEV_INLINE void cc_union(int* parent, int a, int b) {
  int ra = cc_find(parent, a), rb = cc_find(parent, b);
  bool repeat;
  do {
    repeat = false;
    if (ra != rb) {
      int ret;
      if (ra < rb) { if ((ret = atomicCAS(parent + rb, rb, ra)) != rb) { rb = ret; repeat = true; } }
      else { if ((ret = atomicCAS(parent + ra, ra, rb)) != ra) { ra = ret; repeat = true; } }
    }
  } while (repeat);
}

// Vertical runs without atomics.  With alpha_x << alpha_y (0.09-0.2 deg vs 2 deg) a range step of
// ~0.2 % already cuts a horizontal edge while vertical neighbours tolerate ~2 %: measured on the synthetic
// scene 15 k down-edges vs 7.7 k right-edges per scan.  So the runs are taken along columns: one thread per
// column walks the rows; the representative of a run is its first (lowest) row = its minimum linear index.
__global__ void __launch_bounds__(128) cc_runs(DevCtx d) {
  const int slot = blockIdx.y + d.slot0;
  const int col = blockIdx.x * 128 + threadIdx.x;
  if (col >= d.H) return;
  const size_t base = (size_t)slot * d.N;
  const uint8_t* f = d.flag_img + base;
  int* parent = d.parent + base;
  if (col % (IPF_W - 1) == 0) {   // ip_strip_halo_columns, one column per thread
    int* owner = d.owner + base;
    for (int row = 0; row < d.NS; ++row) { const int v = owner[row * d.H + col]; owner[row * d.H + col] = (v >= 0 && (v & IP_OWNER_TAG)) ? (v & ~IP_OWNER_TAG) : -1; }
  }
  int start = 0;
  uint8_t prev = 0;
  for (int row = 0; row < d.NS; ++row) {
    const uint8_t fl = f[row * d.H + col];
    if (!(prev & 8)) start = row;  // no down-edge from the row below: a new run begins here
    parent[row * d.H + col] = (fl & 2) ? start * d.H + col : -1;
    prev = fl;
  }
}

// horizontal (incl. wrap-around) edges between the vertical runs: lock-free union of the run representatives.
// A right-edge is skipped when the cell below already links the same pair of runs.
__global__ void __launch_bounds__(IP_BLOCK) cc_link(DevCtx d) {
  const int slot = blockIdx.y + d.slot0;
  const int v = blockIdx.x * IP_BLOCK + threadIdx.x;
  if (v >= d.N) return;
  const size_t base = (size_t)slot * d.N;
  const uint8_t* fi = d.flag_img + base;
  const uint8_t f = fi[v];
  if (!(f & 4)) return;
  int* parent = d.parent + base;
  const int row = v / d.H, col = v - row * d.H;
  const int u = row * d.H + ((col + 1 == d.H) ? 0 : col + 1);  // right neighbour with column wrap-around (:241-248)
  if (row > 0) {
    const uint8_t fb = fi[v - d.H];
    // cell below me is in my run, has a right-edge, and its right neighbour is in my right neighbour's run
    if ((fb & 8) && (fb & 4) && (fi[u - d.H] & 8)) return;
  }
  cc_union(parent, v, u);
}

// Whole-image union-find in LDS for images of up to CC_LDS_MAXN cells (16x1800): one workgroup per stream keeps
// the parent array on chip (4 B/cell, 115 KB at 16x1800), so every find hop is an LDS access (~64 cycles) instead
// of an L2 round trip, and the compare-and-swap of a union is an LDS atomic.  Same algorithm as cc_link
// (ECL-CC: pointer jumping find, link the larger root under the smaller), so the roots are identical.
// Larger images (16x4000, 64x2048) use the global-memory path cc_runs + cc_link.
#define CC_LDS_THREADS 1024
#define CC_LDS_MAXN 36864
#define CC_LDS16_MAXN 64512   // cc_lds16 (2 B/cell, <= 16 rings): 63 cells per thread; covers the reference geometry 16 x 4000 (126 KB of LDS)
DEV_INLINE int ccl_find(int* parent, int v) {
  int curr = parent[v];
  if (curr != v) {
    int prev = v, next;
    while (curr > (next = parent[curr])) { parent[prev] = next; prev = curr; curr = next; }
  }
  return curr;
}
DEV_INLINE void ccl_union(int* parent, int a, int b) {
  int ra = ccl_find(parent, a), rb = ccl_find(parent, b);
  bool repeat;
  do {
    repeat = false;
    if (ra != rb) {
      int ret;
      if (ra < rb) { if ((ret = atomicCAS(parent + rb, rb, ra)) != rb) { rb = ret; repeat = true; } }
      else { if ((ret = atomicCAS(parent + ra, ra, rb)) != ra) { ra = ret; repeat = true; } }
    }
  } while (repeat);
}
__global__ void __launch_bounds__(CC_LDS_THREADS) cc_lds(DevCtx d, int ring_pos, int fused) {
  const int slot = blockIdx.x + d.slot0;
  const size_t base = (size_t)slot * d.N;
  extern __shared__ __attribute__((aligned(16))) unsigned char cc_smem[];
  int* parent = reinterpret_cast<int*>(cc_smem);
  const uint8_t* fi = d.flag_img + base;
  const int N = d.N, H = d.H;
  for (int v = threadIdx.x; v < N; v += CC_LDS_THREADS) parent[v] = v;
  ip_strip_halo_columns(d, slot, threadIdx.x, CC_LDS_THREADS);
  __syncthreads();
  for (int v = threadIdx.x; v < N; v += CC_LDS_THREADS) {
    const uint8_t f = fi[v];
    if (f & 4) { const int row = v / H, col = v - row * H; ccl_union(parent, v, row * H + (col + 1 == H ? 0 : col + 1)); }
    if (f & 8) ccl_union(parent, v, v + H);
  }
  __syncthreads();
  // roots into registers, then the LDS array is reused for the per-root statistics of :282-301
  // (low 16 bits = size < 65536, high 16 bits = row mask: needs n_scan <= 16, otherwise cc_stats does it)
  constexpr int PER = (CC_LDS_MAXN + CC_LDS_THREADS - 1) / CC_LDS_THREADS;
  int rt[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int v = threadIdx.x + k * CC_LDS_THREADS;
    rt[k] = -1;
    if (v < N && (fi[v] & 2)) { int r = parent[v], nx; while (r > (nx = parent[r])) r = nx; rt[k] = r; }
  }
  __syncthreads();
  const bool stats = d.NS <= 16;
  if (stats) {
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int v = threadIdx.x + k * CC_LDS_THREADS; if (v < N) parent[v] = 0; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int v = threadIdx.x + k * CC_LDS_THREADS;
      if (rt[k] >= 0) {  // LDS atomics resolve same-address lanes in hardware; no wave aggregation needed here
        atomicAdd(&parent[rt[k]], 1);
        atomicOr((unsigned*)&parent[rt[k]], 1u << (16 + v / H));
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int v = threadIdx.x + k * CC_LDS_THREADS;
    if (v < N) {
      d.parent[base + v] = rt[k];
      if (stats && rt[k] == v) { const unsigned w = (unsigned)parent[v]; d.cc_size[base + v] = (int)(w & 0xFFFFu); d.cc_rows[base + v] = (unsigned long long)(w >> 16); }
    }
  }
  if (!stats || !fused) return;  // ip_rowcount / ip_compact follow (launch_ip)
  // ---- fused a6: ordered compaction of the whole image (replaces ip_rowcount + ip_compact for this geometry).
  // Chunk k = cells [1024 k, 1024 k + 1024) in row-major order, one cell per thread: per-(chunk, wavefront) counts of
  // kept cells / outliers / feasible roots, one exclusive scan over the (chunk, wavefront) table, then every cell
  // knows its output line.  Classification as ip_classify, with the component statistics still in LDS.
  constexpr int NW = CC_LDS_THREADS / 64;
  __shared__ int s_cnt[3][PER * NW];
  __shared__ int s_wtot[3][NW];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const alego_params& P = d.P;
  unsigned long long keep_m = 0, outl_m = 0, root_m = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int v = threadIdx.x + k * CC_LDS_THREADS;
    int c = 0;
    bool fr = false;
    if (v < N) {
      const uint8_t f = fi[v];
      const int row = v / H, col = v - row * H;
      if (f & 1) c = (col % 5 == 0 || col <= 4 || col >= H - 5) ? 1 : 0;
      else if (f & 2) {
        const unsigned w = (unsigned)parent[rt[k]];
        const int sz = (int)(w & 0xFFFFu);
        const bool feas = sz >= P.seg_big_num || (sz >= P.seg_valid_point_num && __popc(w >> 16) >= P.seg_valid_line_num);
        fr = feas && rt[k] == v;
        c = feas ? 1 : ((row > P.ground_scan_id && col % 5 == 0) ? 2 : 0);
      }
    }
    if (c == 1) keep_m |= 1ull << k;
    if (c == 2) outl_m |= 1ull << k;
    if (fr) root_m |= 1ull << k;
    const unsigned long long bk = __ballot(c == 1), bo = __ballot(c == 2), bf = __ballot(fr);
    if (lane == 0) { s_cnt[0][k * NW + wave] = (int)__popcll(bk); s_cnt[1][k * NW + wave] = (int)__popcll(bo); s_cnt[2][k * NW + wave] = (int)__popcll(bf); }
  }
  __syncthreads();
  {  // exclusive scan of the three count tables (PER * NW <= 1024 entries: one per thread)
    const int e = threadIdx.x;
    int v3[3], in3[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      v3[a] = e < PER * NW ? s_cnt[a][e] : 0;
      int incl = v3[a];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      in3[a] = incl;
      if (lane == 63) s_wtot[a][wave] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int woff = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) if (w < wave) woff += s_wtot[a][w];
      if (e < PER * NW) s_cnt[a][e] = woff + in3[a] - v3[a];
    }
    if (threadIdx.x == 0) {
      int t3[3] = {0, 0, 0};
#pragma unroll
      for (int a = 0; a < 3; ++a) for (int w = 0; w < NW; ++w) t3[a] += s_wtot[a][w];
      int* sc = d.scal + slot * SC_COUNT;
      sc[SC_M] = t3[0]; sc[SC_NOUT] = t3[1]; sc[SC_NFEAS] = t3[2];
      d.ring_end[slot * d.NS + d.NS - 1] = t3[0] - 1 - 5;   // :190 for the last row
    }
  }
  __syncthreads();
  const float4* pts = scan_pts(d, slot, ring_pos);
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll 4
  for (int k = 0; k < PER; ++k) {
    const int v = threadIdx.x + k * CC_LDS_THREADS;
    const bool kp = (keep_m >> k) & 1ull, ol = (outl_m >> k) & 1ull, fr = (root_m >> k) & 1ull;
    const unsigned long long bk = __ballot(kp), bo = __ballot(ol), bf = __ballot(fr);
    if (v >= N) continue;
    const int row = v / H, col = v - row * H;
    const int line = s_cnt[0][k * NW + wave] + (int)__popcll(bk & below);   // kept cells before this one
    if (col == 0) {   // ring convention of :161,:190
      d.ring_start[slot * d.NS + row] = line + 5;
      if (row > 0) d.ring_end[slot * d.NS + row - 1] = line - 1 - 5;
    }
    if (kp || ol) {
      float4 p = pts[ip_owner_index(d.owner[base + v])];
      p.w = (float)(row + d.ip_colfrac[col]);  // :101
      if (kp) {
        d.seg_pts[base + line] = p;
        d.seg_ground[base + line] = fi[v] & 1;
        d.seg_col[base + line] = col;
        d.seg_range[base + line] = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);   // = the range image's value (:99), recomputed from the point
      } else {
        d.outlier[base + s_cnt[1][k * NW + wave] + (int)__popcll(bo & below)] = p;
      }
    }
    // label_cnt_ numbering (:303-306); 0 for the root of an infeasible component (ip_labels turns it into 999999)
    if (rt[k] == v) d.cc_label[base + v] = fr ? s_cnt[2][k * NW + wave] + (int)__popcll(bf & below) + 1 : 0;
  }
}

// Images too large for one workgroup's LDS (more than 16 rings with more than CC_LDS_MAXN cells: 64x2048): two-level labelling.
// cc_tile: a workgroup takes a band of columns that fits LDS (<= CC_TILE_CELLS cells, all rows), runs the same union-find as
// cc_lds on it — right-edges that stay inside the band, all down-edges — and writes every cell's band root (as a global linear index;
// row-major order inside a band is the global order restricted to it, so the band root is the minimum global index of the piece).
// cc_seam: the right-edges that cross a band boundary (incl. the wrap-around column) are linked with the global-memory union
// (cc_union, as cc_link did for EVERY right-edge: NS x bands edges instead of ~N / 4).  cc_stats then resolves every cell to
// its final root with a read-only find whose chains are at most a few band roots long.
#define CC_TILE_CELLS 16384
#define CC_TILE_T 512
DEV_INLINE int cc_tile_width(const DevCtx& d) { return max(1, CC_TILE_CELLS / d.NS); }
__global__ void __launch_bounds__(CC_TILE_T) cc_tile(DevCtx d) {
  const int slot = blockIdx.y + d.slot0, tid = threadIdx.x;
  const size_t base = (size_t)slot * d.N;
  const int H = d.H, TW0 = cc_tile_width(d);
  const int c0 = blockIdx.x * TW0, TW = min(TW0, H - c0), n = TW * d.NS;
  __shared__ uint16_t parent[CC_TILE_CELLS];   // 16-bit parents (a band has <= 16384 cells): 32 KB, four of these workgroups per CU
  const uint8_t* fi = d.flag_img + base;
  if (blockIdx.x == 0) ip_strip_halo_columns(d, slot, tid, CC_TILE_T);
  // vertical runs without atomics (most edges are vertical, see cc_runs): one thread per column, a run's representative is its lowest row
  for (int lc = tid; lc < TW; lc += CC_TILE_T) {
    int start = 0;
    uint8_t prev = 0;
    for (int row = 0; row < d.NS; ++row) {
      const uint8_t fl = fi[row * H + c0 + lc];
      if (!(prev & 8)) start = row;
      parent[row * TW + lc] = (uint16_t)(start * TW + lc);
      prev = fl;
    }
  }
  __syncthreads();
  // right-edges inside the band between the runs; skipped when the cell below already links the same pair of runs
  for (int l = tid; l < n; l += CC_TILE_T) {
    const int row = l / TW, lc = l - row * TW;
    if (lc + 1 >= TW) continue;
    const int v = row * H + c0 + lc;
    if (!(fi[v] & 4)) continue;
    if (row > 0) {
      const uint8_t fb = fi[v - H];
      if ((fb & 8) && (fb & 4) && (fi[v + 1 - H] & 8)) continue;
    }
    ccl16_union(parent, l, l + 1);
  }
  __syncthreads();
  int* gp = d.parent + base;
  for (int l = tid; l < n; l += CC_TILE_T) {
    const int row = l / TW, lc = l - row * TW;
    const int v = row * H + c0 + lc;
    int out = -1;
    if (fi[v] & 2) {
      int r = parent[l], nx;
      while (r > (nx = parent[r])) r = nx;   // read-only find: nobody writes any more
      const int rr = r / TW;
      out = rr * H + c0 + (r - rr * TW);
    }
    gp[v] = out;
  }
}
__global__ void __launch_bounds__(256) cc_seam(DevCtx d) {
  const int slot = blockIdx.x + d.slot0;
  const size_t base = (size_t)slot * d.N;
  const int H = d.H, TW0 = cc_tile_width(d), nt = (H + TW0 - 1) / TW0;
  const uint8_t* fi = d.flag_img + base;
  int* parent = d.parent + base;
  for (int e = threadIdx.x; e < nt * d.NS; e += 256) {
    const int t = e / d.NS, row = e - t * d.NS;
    const int col = min((t + 1) * TW0, H) - 1;           // last column of band t
    const int v = row * H + col;
    if (fi[v] & 4) cc_union(parent, v, row * H + (col + 1 == H ? 0 : col + 1));
  }
}

#ifdef ALEGO_TIMING
__device__ long long cc_times[16];
#define CC_TICK(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) cc_times[k] = wall_clock64(); } while (0)
extern "C" void alego_cc_times(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(cc_times), sizeof(long long) * 16); }
#else
#define CC_TICK(k)
#endif
#define CC_T CC_LDS_THREADS   // (512-thread workgroups were measured: 188 k vs 203 k scans/s)
// One workgroup per stream: column runs, union-find over the right-edges, component statistics, feasibility, and the
// ordered compaction of the whole image (a6: replaces ip_rowcount + ip_compact for this geometry).
// Register-light on purpose.  A first version kept the root of each of a thread's 36 cells in registers and unrolled every
// pass: 128 VGPRs x 1024 threads is the whole register file of a CU, so a workgroup could only start on an EMPTY CU and
// nothing ran next to it (the pipeline gained 3 % when it was replaced although the kernel itself takes as long).  Here
// the flattened roots stay in the LDS parent array (the statistics reuse the roots' entries: 2 B/cell of LDS in total), the
// passes are plain loops over the thread's cells and the per-cell state is a handful of 64-bit masks: 64 VGPRs, other
// streams' wavefronts share the CU.
// Compaction: chunk k = cells [1024 k, 1024 k + 1024) in row-major order, one cell per thread: per-(chunk, wavefront)
// counts of kept cells / outliers / feasible roots, one exclusive scan over the (chunk, wavefront) table, then every cell
// knows its output line.  Classification as ip_classify.
__global__ void __attribute__((amdgpu_waves_per_eu(8, 8))) __launch_bounds__(CC_T) cc_lds16(DevCtx d, int ring_pos, int fused) {
  const int slot = blockIdx.x + d.slot0;
  const size_t base = (size_t)slot * d.N;
  const int N = d.N, H = d.H;
  extern __shared__ __attribute__((aligned(16))) unsigned char cc_smem[];
  uint16_t* par = reinterpret_cast<uint16_t*>(cc_smem);
  const uint8_t* fi = d.flag_img + base;
  constexpr int PER = (CC_LDS16_MAXN + CC_T - 1) / CC_T;
  static_assert(PER <= 64, "four 64-bit words of 4-bit flags, one bit per cell in the 64-bit masks");
  const int per = (N + CC_T - 1) / CC_T;   // cells per thread of THIS image (29 at 16x1800, 63 at 16x4000; PER = 63 is the capacity)
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  CC_TICK(0);
  ip_strip_halo_columns(d, slot, threadIdx.x, CC_T);
  unsigned long long f0 = 0, f1 = 0, f2 = 0, f3 = 0;   // the 4 flag bits of this thread's cells, 16 cells per word
#pragma unroll 1
  for (int w = 0; w < (per + 15) / 16; ++w) {   // (one word at a time: 16 loads in flight, not 36 64-bit addresses in registers)
    unsigned long long acc = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = w * 16 + q, v = threadIdx.x + k * CC_T;
      const unsigned long long f = (unsigned long long)(fi[min(v, N - 1)] & 15u) * (unsigned long long)(k < PER && v < N);   // (unconditional load: behind a branch each of the 16 waited for the previous one)
      acc |= f << (q * 4);
    }
    if (w == 0) f0 = acc; else if (w == 1) f1 = acc; else if (w == 2) f2 = acc; else f3 = acc;
  }
  auto flag_of = [&](int k) -> unsigned { const unsigned long long w = k < 16 ? f0 : (k < 32 ? f1 : (k < 48 ? f2 : f3)); return (unsigned)(w >> ((k & 15) * 4)) & 15u; };
  // Vertical neighbours (2 deg apart) pass the angle test far more often than horizontal ones (0.2 deg apart: a few cm of
  // range difference already fail), so the components are mostly column strips.  One thread per column walks its rows
  // bottom-up and gives every cell the start of its vertical run as parent (as cc_runs does on the global path): no
  // atomics, and the union-find proper only has to process the right-edges.
  for (int c = threadIdx.x; c < H; c += CC_T) {
    unsigned fcol[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) fcol[r] = r < d.NS ? (unsigned)fi[r * H + c] : 0u;
    int start = 0;
    unsigned prev = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < d.NS) {
        if (!(prev & 8)) start = r;
        par[r * H + c] = (uint16_t)(((fcol[r] & 2) ? start : r) * H + c);
        prev = fcol[r];
      }
    }
  }
  __syncthreads();
  CC_TICK(1);
#pragma unroll 1
  for (int k = 0; k < per; ++k) {
    const int v = threadIdx.x + k * CC_T;
    if (flag_of(k) & 4) { const int row = cell_row(d, v), col = v - row * H; ccl16_union(par, v, row * H + (col + 1 == H ? 0 : col + 1)); }
  }
  __syncthreads();
  CC_TICK(2);
  // flatten: par[v] = root.  Concurrent walkers see either the old parent or the root of a cell, both are ancestors.
  unsigned long long self_m = 0;   // this thread's cells that are roots
#pragma unroll 1
  for (int k = 0; k < per; ++k) {
    const int v = threadIdx.x + k * CC_T;
    if (flag_of(k) & 2) { int r = par[v], nx; while (r > (nx = par[r])) r = nx; par[v] = (uint16_t)r; if (r == v) self_m |= 1ull << k; }
  }
  __syncthreads();
  CC_TICK(3);
  // From here on a root's own entry (par[r] == r, known to its owner through self_m) is free: it becomes the component's
  // 16-bit accumulator, first for the size, then for the row mask.  No second array: 2 B/cell of LDS in total.
  unsigned* parw = reinterpret_cast<unsigned*>(par);
  auto root_of = [&](int k, int v) -> int { return ((self_m >> k) & 1ull) ? v : (int)par[v]; };
  auto zero_roots = [&]() {
#pragma unroll 1
    for (int k = 0; k < per; ++k) if ((self_m >> k) & 1ull) par[threadIdx.x + k * CC_T] = 0;
    __syncthreads();
  };
  zero_roots();
  const alego_params& P = d.P;
  unsigned long long head_m = 0;
#pragma unroll 1
  for (int k = 0; k < per; ++k) {   // component sizes: one atomic per run of equal roots in the wavefront
    const int v = threadIdx.x + k * CC_T;
    const int r = (flag_of(k) & 2) ? root_of(k, v) : -1;
    const int prev_r = __shfl_up(r, 1, 64);
    const bool brk = lane == 0 || prev_r != r || (v - cell_row(d, v) * H) == 0;
    const unsigned long long starts = __ballot(brk);
    if (r >= 0 && brk) {
      head_m |= 1ull << k;
      const unsigned long long after = lane == 63 ? 0ull : (starts >> (lane + 1));
      const int len = after ? __ffsll((long long)after) : 64 - lane;
      atomicAdd(&parw[r >> 1], (unsigned)len << ((r & 1) * 16));   // sizes < 65536: no carry into the neighbour's half
    }
  }
  __syncthreads();
  unsigned long long big_m = 0, mid_m = 0;
#pragma unroll 1
  for (int k = 0; k < per; ++k) {
    const int v = threadIdx.x + k * CC_T;
    if (flag_of(k) & 2) {
      const int r = root_of(k, v);
      const int sz = (int)par[r];
      if (sz >= P.seg_big_num) big_m |= 1ull << k;
      else if (sz >= P.seg_valid_point_num) mid_m |= 1ull << k;
      if (!(fused & 1) && r == v) d.cc_size[base + v] = sz;
    }
  }
  __syncthreads();
  CC_TICK(4);
  zero_roots();
#pragma unroll 1
  for (int k = 0; k < per; ++k) {   // rows touched by every component (a run lies in one row)
    const int v = threadIdx.x + k * CC_T;
    if ((head_m >> k) & 1ull) { const int r = root_of(k, v); atomicOr(&parw[r >> 1], (1u << cell_row(d, v)) << ((r & 1) * 16)); }
  }
  __syncthreads();
  unsigned long long feas_m = big_m;
#pragma unroll 1
  for (int k = 0; k < per; ++k) {
    const int v = threadIdx.x + k * CC_T;
    int r = -1;
    if (flag_of(k) & 2) {
      r = root_of(k, v);
      const unsigned rows = par[r];
      if (((mid_m >> k) & 1ull) && __popc(rows) >= P.seg_valid_line_num) feas_m |= 1ull << k;
      if (!(fused & 1) && r == v) d.cc_rows[base + v] = (unsigned long long)rows;
    }
    if ((fused & 2) && v < N) d.parent[base + v] = r;
  }
  CC_TICK(5);
  if (!(fused & 1)) return;
  // ---- ordered compaction
  constexpr int NW = CC_T / 64;
  __shared__ int s_cnt[3][PER * NW];
  __shared__ int s_wtot[3][NW];
  unsigned long long keep_m = 0, outl_m = 0, root_m = 0;
#pragma unroll 1
  for (int k = 0; k < per; ++k) {
    const int v = threadIdx.x + k * CC_T;
    int c = 0;
    bool fr = false;
    if (v < N) {
      const unsigned f = flag_of(k);
      const int row = cell_row(d, v), col = v - row * H;
      if (f & 1) c = (col % 5 == 0 || col <= 4 || col >= H - 5) ? 1 : 0;
      else if (f & 2) {
        const bool feas = (feas_m >> k) & 1ull;
        fr = feas && ((self_m >> k) & 1ull);
        c = feas ? 1 : ((row > P.ground_scan_id && col % 5 == 0) ? 2 : 0);
      }
    }
    if (c == 1) keep_m |= 1ull << k;
    if (c == 2) outl_m |= 1ull << k;
    if (fr) root_m |= 1ull << k;
    const unsigned long long bk = __ballot(c == 1), bo = __ballot(c == 2), bf = __ballot(fr);
    if (lane == 0) { s_cnt[0][k * NW + wave] = (int)__popcll(bk); s_cnt[1][k * NW + wave] = (int)__popcll(bo); s_cnt[2][k * NW + wave] = (int)__popcll(bf); }
  }
  __syncthreads();
  CC_TICK(6);
  {
    const int e = threadIdx.x;
    int v3[3], in3[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      v3[a] = e < per * NW ? s_cnt[a][e] : 0;
      int incl = v3[a];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      in3[a] = incl;
      if (lane == 63) s_wtot[a][wave] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int woff = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) if (w < wave) woff += s_wtot[a][w];
      if (e < per * NW) s_cnt[a][e] = woff + in3[a] - v3[a];
    }
    if (threadIdx.x == 0) {
      int t3[3] = {0, 0, 0};
#pragma unroll
      for (int a = 0; a < 3; ++a) for (int w = 0; w < NW; ++w) t3[a] += s_wtot[a][w];
      int* sc = d.scal + slot * SC_COUNT;
      sc[SC_M] = t3[0]; sc[SC_NOUT] = t3[1]; sc[SC_NFEAS] = t3[2];
      d.ring_end[slot * d.NS + d.NS - 1] = t3[0] - 1 - 5;
    }
  }
  __syncthreads();
  CC_TICK(7);
  const float4* pts = scan_pts(d, slot, ring_pos);
  const unsigned long long below = (1ull << lane) - 1ull;
  constexpr int CB = 3;   // cells per batch (emit() below is called CB times)
  static_assert(PER % CB == 0, "cells per thread must be a multiple of the batch");
  const unsigned long long out_m = keep_m | outl_m;
  // (scalars and a lambda instead of a float4 pt[CB] array: hipcc kept the array in scratch memory — a store and a load
  //  through global memory per output cell in the longest phase of the kernel)
  auto emit = [&](int k, const float4& q, float rg) {
    const int v = threadIdx.x + k * CC_T;
    const bool kp = (keep_m >> k) & 1ull, ol = (outl_m >> k) & 1ull, fr = (root_m >> k) & 1ull;
    const unsigned long long bk = __ballot(kp), bo = __ballot(ol), bf = __ballot(fr);
    if (v >= N) return;
    const int row = cell_row(d, v), col = v - row * H;
    const int line = s_cnt[0][k * NW + wave] + (int)__popcll(bk & below);
    if (col == 0) {
      d.ring_start[slot * d.NS + row] = line + 5;
      if (row > 0) d.ring_end[slot * d.NS + row - 1] = line - 1 - 5;
    }
    if (kp || ol) {
      const float4 p = make_float4(q.x, q.y, q.z, (float)(row + d.ip_colfrac[col]));
      if (kp) {
        d.seg_pts[base + line] = p;
        d.seg_ground[base + line] = (uint8_t)(flag_of(k) & 1);
        d.seg_col[base + line] = col;
        d.seg_range[base + line] = rg;
      } else {
        d.outlier[base + s_cnt[1][k * NW + wave] + (int)__popcll(bo & below)] = p;
      }
    }
    if ((self_m >> k) & 1ull) d.cc_label[base + v] = fr ? s_cnt[2][k * NW + wave] + (int)__popcll(bf & below) + 1 : 0;
  };
  // clamped addresses instead of branches: the three owner loads, then the three point gathers and ranges, are in flight together
  auto cell_of = [&](int k) -> int { return ((out_m >> k) & 1ull) ? threadIdx.x + k * CC_T : min((int)threadIdx.x, N - 1); };   // (images of fewer than CC_T cells exist: 1 x 720)
#pragma unroll 1
  for (int k0 = 0; k0 < per; k0 += CB) {
    const int c0 = cell_of(k0), c1 = cell_of(k0 + 1), c2 = cell_of(k0 + 2);
    const int o0 = d.owner[base + c0], o1 = d.owner[base + c1], o2 = d.owner[base + c2];
    const float4 q0 = pts[ip_owner_index(max(o0, 0))], q1 = pts[ip_owner_index(max(o1, 0))], q2 = pts[ip_owner_index(max(o2, 0))];
    // segmentedCloudRange = the range image's value (:99,:184), recomputed from the point instead of read back
    emit(k0, q0, sqrtf(q0.x * q0.x + q0.y * q0.y + q0.z * q0.z)); emit(k0 + 1, q1, sqrtf(q1.x * q1.x + q1.y * q1.y + q1.z * q1.z));
    emit(k0 + 2, q2, sqrtf(q2.x * q2.x + q2.y * q2.y + q2.z * q2.z));
  }
  __syncthreads();
  CC_TICK(8);
}

__global__ void __launch_bounds__(IP_BLOCK) cc_stats(DevCtx d) {
  const int slot = blockIdx.y + d.slot0;
  const int v = blockIdx.x * IP_BLOCK + threadIdx.x;
  const size_t base = (size_t)slot * d.N;
  int* parent = d.parent + base;
  bool active = false;
  int root = -1, row = 0;
  if (v < d.N) {
    active = d.flag_img[base + v] & 2;
    if (active) {
      root = cc_find_ro(parent, v);
      // cc_link has completed (kernel boundary), so `root` is final; concurrent readers see either
      // the old ancestor or the root, both valid
      st_agent(parent + v, root);
      row = v / d.H;
    }
  }
  // wavefront aggregation: one atomic pair per distinct root in the wave
  unsigned long long todo = __ballot(active);
  const int lane = lane_id();
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int lroot = __shfl(root, leader, 64);
    const bool same = active && root == lroot;
    const unsigned long long m = __ballot(same);
    unsigned long long rb = same ? (1ull << row) : 0ull;
    if (d.H & 63) {   // (a wavefront's 64 consecutive cells share a row when the width is a multiple of 64 — 64 x 2048 —: nothing to combine)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) rb |= __shfl_xor(rb, o, 64);
    }
    if (lane == leader) {
      atomicAdd(&d.cc_size[base + lroot], (int)__popcll(m));
      atomicOr(&d.cc_rows[base + lroot], rb);
    }
    todo &= ~m;
  }
}

DEV_INLINE bool cc_feasible(const DevCtx& d, size_t base, int root) {
  const int sz = d.cc_size[base + root];
  if (sz >= d.P.seg_big_num) return true;
  if (sz >= d.P.seg_valid_point_num) return __popcll(d.cc_rows[base + root]) >= d.P.seg_valid_line_num;
  return false;
}

// classification of a cell for the compaction sweep (:164-188): 1 keep, 2 outlier, 0 drop
DEV_INLINE int ip_classify(const DevCtx& d, size_t base, int v, int row, int col, bool* is_feas_root) {
  const uint8_t f = d.flag_img[base + v];
  *is_feas_root = false;
  if (f & 1) return (col % 5 == 0 || col <= 4 || col >= d.H - 5) ? 1 : 0;
  if (f & 2) {
    const int root = d.parent[base + v];
    const bool feas = cc_feasible(d, base, root);
    *is_feas_root = feas && root == v;
    if (feas) return 1;
    return (row > d.P.ground_scan_id && col % 5 == 0) ? 2 : 0;
  }
  return 0;
}

// per-row counts: kept cells, outliers, feasible roots.  grid (NS, slots)
__global__ void __launch_bounds__(IP_BLOCK) ip_rowcount(DevCtx d) {
  const int slot = blockIdx.y + d.slot0, row = blockIdx.x;
  const size_t base = (size_t)slot * d.N;
  int nk = 0, no = 0, nf = 0;
  for (int col = threadIdx.x; col < d.H; col += IP_BLOCK) {
    bool fr;
    const int v = row * d.H + col;
    const int c = ip_classify(d, base, v, row, col, &fr);
    nk += c == 1; no += c == 2; nf += fr;
    // the class of the cell for ip_compact (bits 4-5, bit 6: feasible root): the sweep that follows asks the flag byte instead of walking
    // flag -> root -> size -> row mask again (every other reader of the flag image masks the low bits; ip_front rewrites the image every scan)
    d.flag_img[base + v] = (uint8_t)((d.flag_img[base + v] & 0x0F) | (c << 4) | (fr ? 0x40 : 0));
  }
  __shared__ int s[3][IP_BLOCK / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { nk += __shfl_xor(nk, o, 64); no += __shfl_xor(no, o, 64); nf += __shfl_xor(nf, o, 64); }
  if (lane_id() == 0) { s[0][threadIdx.x >> 6] = nk; s[1][threadIdx.x >> 6] = no; s[2][threadIdx.x >> 6] = nf; }
  __syncthreads();
  if (threadIdx.x < 3) {
    int t = 0;
    for (int w = 0; w < IP_BLOCK / 64; ++w) t += s[threadIdx.x][w];
    d.row_cnt[((size_t)slot * d.NS + row) * 4 + threadIdx.x] = t;
  }
}

// per-row ordered compaction.  grid (NS, slots)
__global__ void __launch_bounds__(IP_BLOCK) ip_compact(DevCtx d, int ring_pos) {
  const int slot = blockIdx.y + d.slot0, row = blockIdx.x;
  const size_t base = (size_t)slot * d.N;
  const float4* pts = scan_pts(d, slot, ring_pos);
  __shared__ int s_off[3];
  __shared__ int s_wave[3][IP_BLOCK / 64];
  // exclusive prefix over rows (NS <= 64 rows, recomputed by every block)
  if (threadIdx.x < 64) {
    const int* rc = d.row_cnt + (size_t)slot * d.NS * 4;
    const int r = threadIdx.x;
    int k = r < d.NS ? rc[r * 4 + 0] : 0, o = r < d.NS ? rc[r * 4 + 1] : 0, f = r < d.NS ? rc[r * 4 + 2] : 0;
    int pk = (r < row) ? k : 0, po = (r < row) ? o : 0, pf = (r < row) ? f : 0;
    int tk = k, to = o, tf = f;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      pk += __shfl_xor(pk, s, 64); po += __shfl_xor(po, s, 64); pf += __shfl_xor(pf, s, 64);
      tk += __shfl_xor(tk, s, 64); to += __shfl_xor(to, s, 64); tf += __shfl_xor(tf, s, 64);
    }
    if (threadIdx.x == 0) {
      s_off[0] = pk; s_off[1] = po; s_off[2] = pf;
      const int mykeep = rc[row * 4 + 0];
      d.ring_start[slot * d.NS + row] = pk + 5;                // :161
      d.ring_end[slot * d.NS + row] = pk + mykeep - 1 - 5;     // :190
      if (row == 0) {
        int* sc = d.scal + slot * SC_COUNT;
        sc[SC_M] = tk; sc[SC_NOUT] = to; sc[SC_NFEAS] = tf;
      }
    }
  }
  __syncthreads();
  int run_k = s_off[0], run_o = s_off[1], run_f = s_off[2];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  for (int c0 = 0; c0 < d.H; c0 += IP_BLOCK) {
    const int col = c0 + threadIdx.x;
    int c = 0;
    bool fr = false;
    const int v = row * d.H + col;
    uint8_t fl = 0;
    if (col < d.H) { fl = d.flag_img[base + v]; c = (fl >> 4) & 3; fr = (fl & 0x40) != 0; }   // (classified by ip_rowcount)
    const unsigned long long bk = __ballot(c == 1), bo = __ballot(c == 2), bf = __ballot(fr);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) { s_wave[0][wave] = (int)__popcll(bk); s_wave[1][wave] = (int)__popcll(bo); s_wave[2][wave] = (int)__popcll(bf); }
    __syncthreads();
    int wk = 0, wo = 0, wf = 0, tk = 0, to = 0, tf = 0;
#pragma unroll
    for (int w = 0; w < IP_BLOCK / 64; ++w) {
      const int a = s_wave[0][w], b = s_wave[1][w], e = s_wave[2][w];
      if (w < wave) { wk += a; wo += b; wf += e; }
      tk += a; to += b; tf += e;
    }
    if (c == 1 || c == 2) {
      const int o = ip_owner_index(d.owner[base + v]);
      float4 p = pts[o];
      p.w = (float)(row + d.ip_colfrac[col]);  // :101
      if (c == 1) {
        const int line = run_k + wk + (int)__popcll(bk & below);
        d.seg_pts[base + line] = p;
        d.seg_ground[base + line] = fl & 1;
        d.seg_col[base + line] = col;
        d.seg_range[base + line] = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);   // = the range image's value (:99)
      } else {
        d.outlier[base + run_o + wo + (int)__popcll(bo & below)] = p;
      }
    }
    if (fr) d.cc_label[base + v] = run_f + wf + (int)__popcll(bf & below) + 1;  // label_cnt_ numbering (:303-306)
    run_k += tk; run_o += to; run_f += tf;
    __syncthreads();
  }
}

// label_mat_: -1 ground/empty, 999999 infeasible, else the discovery-order counter
__global__ void __launch_bounds__(IP_BLOCK) ip_labels(DevCtx d) {
  const int slot = blockIdx.y + d.slot0;
  const int v = blockIdx.x * IP_BLOCK + threadIdx.x;
  if (v >= d.N) return;
  const size_t base = (size_t)slot * d.N;
  const uint8_t f = d.flag_img[base + v];
  int lab = -1;
  if (f & 2) {
    const int root = d.parent[base + v];
    const int l = d.cc_label[base + root];
    lab = l > 0 ? l : 999999;
  }
  d.label_img[base + v] = lab;
}

__global__ void atan2f_probe(const float* y, const float* x, float* out, int n, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = mode == 0 ? d_atan2f(y[i], x[i]) : mode == 1 ? d_hypotf(x[i], y[i]) : mode == 2 ? d_sinf(y[i]) : d_cosf(y[i]);
}

// ---- host-side launchers -------------------------------------------------------
bool ipf_eligible(const DevCtx& d);
bool ipw_eligible(const DevCtx& d);
void launch_ip_fused(const DevCtx& d, int ring_pos, bool keep_images, hipStream_t st);
int ipf_configure(const DevCtx& d);
bool ipb_eligible(const DevCtx& d);
void launch_ipb(const DevCtx& d, int ring_pos, bool keep_images, hipStream_t st);

void launch_ip(const DevCtx& d, int ring_pos, bool want_labels, hipStream_t st) {
  if (d.opt_ip_fused && (ipf_eligible(d) || (d.opt_ip_half && ipw_eligible(d)))) {   // one workgroup per stream, everything between the input points and cloud_info on chip (kernels_ipf.hip)
    launch_ip_fused(d, ring_pos, want_labels || d.n_launch == 1, st);
    if (want_labels) hipLaunchKernelGGL(ip_labels, dim3((d.N + IP_BLOCK - 1) / IP_BLOCK, d.n_launch), dim3(IP_BLOCK), 0, st, d);
    return;
  }
  const bool lds16 = d.NS <= 16 && d.N <= CC_LDS16_MAXN;
  const bool fused = d.opt_cc_fused && lds16;   // cc_lds16 also does the compaction
  const dim3 gN((d.N + IP_BLOCK - 1) / IP_BLOCK, d.n_launch);
  const dim3 gN4((d.N + IP_BLOCK * IP_PW - 1) / (IP_BLOCK * IP_PW), d.n_launch), gP4((d.Pcap + IP_BLOCK * IP_PW - 1) / (IP_BLOCK * IP_PW), d.n_launch);
  ALEGO_LAUNCH(ip_project, gP4, dim3(IP_BLOCK), 0, st, d, ring_pos);
  if (d.opt_ip_band && ipb_eligible(d)) {   // 17 - 64 rings: column bands + row masks (kernels_ipb.hip)
    launch_ipb(d, ring_pos, want_labels || d.n_launch == 1, st);
    if (want_labels) hipLaunchKernelGGL(ip_labels, gN, dim3(IP_BLOCK), 0, st, d);
    return;
  }
  const bool lds_cc = lds16 || d.N <= CC_LDS_MAXN, lds_stats = lds16;
  const bool keep_images = want_labels || d.n_launch == 1;   // the single-scan entry points / tests read the range and root images back
  ALEGO_LAUNCH(ip_front, dim3((d.H + IPF_W - 2) / (IPF_W - 1), d.n_launch), dim3(IPF_W), (size_t)d.NS * IPF_W * 4 + IPF_W * 8, st, d, ring_pos,
               (lds_cc ? 0 : 1) | (lds_stats ? 0 : 2) | (fused ? 0 : 4) | (keep_images ? 8 : 0));
  if (lds_stats) {
    // bit 0: fused compaction; bit 1: write the root image to HBM (only ip_classify, ip_labels and alego_debug_get read it:
    // the single-scan entry points keep it, the batch path does not)
    const int cc_flags = (fused ? 1 : 0) | ((!fused || keep_images) ? 2 : 0);
    ALEGO_LAUNCH(cc_lds16, dim3(d.n_launch), dim3(CC_LDS_THREADS), (size_t)4 * ((d.N + 1) / 2), st, d, ring_pos, cc_flags);
  } else if (lds_cc) {
    ALEGO_LAUNCH(cc_lds, dim3(d.n_launch), dim3(CC_LDS_THREADS), (size_t)4 * d.N, st, d, ring_pos, 0);
  } else if (d.opt_cc_tile && d.NS <= CC_TILE_CELLS / 16) {
    const int tw = std::max(1, CC_TILE_CELLS / d.NS);
    ALEGO_LAUNCH(cc_tile, dim3((d.H + tw - 1) / tw, d.n_launch), dim3(CC_TILE_T), 0, st, d);
    ALEGO_LAUNCH(cc_seam, dim3(d.n_launch), dim3(256), 0, st, d);
  } else {
    ALEGO_LAUNCH(cc_runs, dim3((d.H + 127) / 128, d.n_launch), dim3(128), 0, st, d);
    ALEGO_LAUNCH(cc_link, gN, dim3(IP_BLOCK), 0, st, d);
  }
  if (!lds16) ALEGO_LAUNCH(cc_stats, gN, dim3(IP_BLOCK), 0, st, d);  // otherwise cc_lds16 produced the statistics
  if (!fused) {
    ALEGO_LAUNCH(ip_rowcount, dim3(d.NS, d.n_launch), dim3(IP_BLOCK), 0, st, d);
    ALEGO_LAUNCH(ip_compact, dim3(d.NS, d.n_launch), dim3(IP_BLOCK), 0, st, d, ring_pos);
  }
  if (want_labels) hipLaunchKernelGGL(ip_labels, gN, dim3(IP_BLOCK), 0, st, d);
}

void launch_atan2f_probe(const float* y, const float* x, float* out, int n, int mode, hipStream_t st) {
  ALEGO_LAUNCH(atan2f_probe, dim3((n + 255) / 256), dim3(256), 0, st, y, x, out, n, mode);
}

// dynamic LDS above 64 KB has to be requested explicitly
int ip_configure(const DevCtx& d) {
  if (ipf_configure(d) != 0) return -1;
  if (d.N <= CC_LDS_MAXN && hipFuncSetAttribute(reinterpret_cast<const void*>(cc_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * d.N) != hipSuccess) return -1;
  if (d.NS <= 16 && d.N <= CC_LDS16_MAXN &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(cc_lds16), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * ((d.N + 1) / 2)) != hipSuccess) return -1;   // <= 126 KB
  return 0;
}
