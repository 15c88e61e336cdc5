// kernels_map.hip — the local map of LaserMapping from PRE-SORTED key frames
// (replaces extractSurroundingKeyFrames' concatenation + pcl::VoxelGrid, src/laserMapping.cpp:238-243,:315-319).
//
// pcl::VoxelGrid orders its output by idx = i + j dx + k dx dy with (i, j, k) the voxel coordinates relative to the cloud's bounding
// box: lexicographically by (floor(z/leaf), floor(y/leaf), floor(x/leaf)), whatever the box.  A key frame's transformed clouds never
// change, so they are sorted by that key ONCE when the frame enters the ring (VoxelGrid kernels, mode 1), and the filtered map of
// a window is the stable K-way merge of its sorted runs: one output point per occupied voxel in ascending key order, each the f32
// sum of the voxel's points in window order (run after run, input order inside a run — exactly the order a stable sort of the
// concatenation gives, which is what the radix-sort path and the oracle sum in) divided by the count.
//
//   map_update  (2 maps x slots) keeps, per map, the sorted list U of occupied voxel keys with their point counts.  A window
//               changes by one run out / one run in, so U is updated incrementally: binary-search decrement for the run that
//               left, binary-search increment + ordered insertion of the new voxels for the run that entered, empty voxels dropped.
//               (Any other change — first use, pose corrections, a cleared window — rebuilds U by inserting the runs one by one.)
//               The last workgroup of the launch writes the work list of map_accum.
//   map_accum   one workgroup per 512 consecutive voxels of a map: a run's points of the chunk are one contiguous piece (found by
//               binary search); the pieces are staged in LDS batch by batch in window order, each point finds its voxel by binary
//               search in the chunk's keys (LDS), and one thread per voxel adds its points in order (registers).
//
// HBM traffic per rebuild: the window's points once (16 B each) + the new / old run twice + the output, against ~21x that for the
// radix sort of the concatenation (rocprofv3 FETCH/WRITE counters, round 1).
#include <algorithm>
#include <cstdlib>
#include "dev_common.h"
#include "lm_ctx.h"
#include "prof.h"

typedef unsigned long long u64;

#define MU_T 512          // threads of map_update
#ifndef MA_T
#define MA_T 512          // threads of map_accum = voxels per work item
#endif
#ifndef MAP_R
#define MAP_R 512         // voxels per map_accum work item (= MA_T: one thread per voxel in its accumulation phase)
#endif
#define MAP_KMAX 512      // window entries held in LDS (alego_create refuses a larger recent_keyframe_num)

DEV_INLINE int* lipm(const LmCtx& L, int slot) { return L.li + (size_t)slot * LI_COUNT; }
DEV_INLINE const float4* run_pts(const LmCtx& L, int slot, int m, int entry) {
  return m == 0 ? L.kfs_c + ((size_t)slot * L.KR + entry) * L.kf_cap_c : L.kfs_s + ((size_t)slot * L.KR + entry) * L.total_cap;
}
DEV_INLINE int run_n(const LmCtx& L, int slot, int m, int entry) { return L.kfs_n[((size_t)slot * 2 + m) * L.KR + entry]; }
DEV_INLINE const float* run_box(const LmCtx& L, int slot, int m, int entry) { return L.kfs_box + (((size_t)slot * 2 + m) * L.KR + entry) * 8; }
DEV_INLINE u64* map_U(const LmCtx& L, int slot, int m) { return m == 0 ? L.U_c + (size_t)slot * L.map_cap_c : L.U_s + (size_t)slot * L.map_cap_s; }
DEV_INLINE int* map_Ucnt(const LmCtx& L, int slot, int m) { return m == 0 ? L.Ucnt_c + (size_t)slot * L.map_cap_c : L.Ucnt_s + (size_t)slot * L.map_cap_s; }
DEV_INLINE float4* map_out(const LmCtx& L, int slot, int m) { return m == 0 ? L.map_corner_ds + (size_t)slot * L.map_cap_c : L.map_surf_ds + (size_t)slot * L.map_cap_s; }

DEV_INLINE int lower_bound_U(const u64* U, int n, u64 key) {   // first index with U[i] >= key
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (U[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}
// first index of a sorted run whose key is >= key (strict = false) or > key (strict = true)
DEV_INLINE int bound_run(const float4* pts, int n, float inv, u64 key, bool strict) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const u64 k = vkey_of(pts[mid], inv);
    if (strict ? k <= key : k < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// both ends of the piece of a sorted run that falls into a chunk's key range [key_lo, key_hi]: first index with key >= key_lo and first index with key > key_hi,
// found TOGETHER by 4-ary search — every round issues the six probes of both searches at once, so the dependent chain is ~log4(n) global round trips instead of
// the 2 log2(n) of two binary searches one after the other (one wavefront does this in front of every work item of map_accum: it is latency, not work)
DEV_INLINE void bound_run_pair(const float4* pts, int n, float inv, u64 key_lo, u64 key_hi, int* out_lo, int* out_hi) {
  int lo0 = 0, hi0 = n, lo1 = 0, hi1 = n;   // invariants: [0, lo0) < key_lo <= [hi0, n);  [0, lo1) <= key_hi < [hi1, n)
  while (lo0 < hi0 || lo1 < hi1) {
    int m0[3], m1[3];
    u64 k0[3], k1[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      m0[q] = lo0 + (int)(((long long)(hi0 - lo0) * (q + 1)) >> 2); m1[q] = lo1 + (int)(((long long)(hi1 - lo1) * (q + 1)) >> 2);
      m0[q] = min(m0[q], max(hi0 - 1, 0)); m1[q] = min(m1[q], max(hi1 - 1, 0));
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) { k0[q] = vkey_of(pts[max(min(m0[q], n - 1), 0)], inv); k1[q] = vkey_of(pts[max(min(m1[q], n - 1), 0)], inv); }
    if (lo0 < hi0) {
      int nl = lo0, nh = hi0;
#pragma unroll
      for (int q = 0; q < 3; ++q) { if (k0[q] < key_lo) nl = max(nl, m0[q] + 1); else nh = min(nh, m0[q]); }
      lo0 = nl; hi0 = nh;
    }
    if (lo1 < hi1) {
      int nl = lo1, nh = hi1;
#pragma unroll
      for (int q = 0; q < 3; ++q) { if (k1[q] <= key_hi) nl = max(nl, m1[q] + 1); else nh = min(nh, m1[q]); }
      lo1 = nl; hi1 = nh;
    }
  }
  *out_lo = lo0; *out_hi = lo1;
}

struct MapWork {
  int* items;      // [cap] packed (slot - slot0) << 12 | m << 11 | chunk   (chunk < 2048)
  int* count;      // [2]: items, ticket
  int cap;
};

// block-wide exclusive scan of one int per thread (MU_T threads); returns the exclusive prefix, *total = sum
DEV_INLINE int block_excl_scan(int v, int* s_w /*[MU_T/64 + 1]*/, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
  __syncthreads();
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < MU_T / 64; ++w) { const int c = s_w[w]; if (w < wave) woff += c; tot += c; }
  *total = tot;
  return woff + incl - v;
}

#define MU_FCAP 4096      // points of a run whose voxel keys the fast path of map_update stages in LDS (2 x 32 KB)
#define MU_E 16           // consecutive entries of the voxel list per thread in the fast path
#ifndef MU_WORKLIST
#define MU_WORKLIST 1
#endif
#define MU_LIST 2048      // slots whose rebuild flags are compacted into one work list (a launch of more slots takes them MU_LIST at a time)

// grid (2, slots); dynamic LDS: 2 * MU_FCAP * 8 bytes
__global__ void __launch_bounds__(MU_T) map_update(DevCtx d, LmCtx L, MapWork W) {
  const int m = blockIdx.x, tid = threadIdx.x;
  __shared__ int s_rem[MAP_KMAX], s_add[MAP_KMAX], s_nrem, s_nadd, s_nU, s_nnew, s_err;
  __shared__ int s_w[MU_T / 64 + 1];
  __shared__ int s_last;
  __shared__ float s_box[6][MU_T / 64];
  __shared__ int s_kr_tot[MU_T / 64];
  extern __shared__ __attribute__((aligned(16))) unsigned char mu_smem[];
  // Only every sixth mapping frame changes its window, and a workgroup that has nothing to do still has to wait for 68 KB of LDS before it can start —
  // 1024 of them per launch held the back stream of their group for 1.2 ms and took LDS from everybody else: 367 k -> 409 k scans/s with 32 slots per
  // workgroup (round 3, launch_map_update).  Round 5: the few workgroups of a launch no longer walk fixed slot ranges (32 slots of which ~9 % have work: the
  // unluckiest of 16 workgroups got six or seven rebuilds and set the kernel's duration) but a WORK LIST — every workgroup compacts the launch's rebuild
  // flags itself (one load per thread and a block scan: the same list in every workgroup, slot-ascending, no atomics) and takes the items blockIdx.y,
  // blockIdx.y + gridDim.y, ...: the rebuilds are dealt out evenly whatever slots they fall on.
  __shared__ unsigned short s_items[MU_LIST];
  for (int c0 = 0; c0 < d.n_launch; c0 += MU_LIST) {
  int n_items = 0;
#if !MU_WORKLIST   // (development: round 4's fixed slot ranges — every slot of the launch is an "item", most of them with nothing to do)
  n_items = min(MU_LIST, d.n_launch - c0);
  for (int i = tid; i < n_items; i += MU_T) s_items[i] = (unsigned short)i;
#else
  for (int s0 = c0; s0 < min(c0 + MU_LIST, d.n_launch); s0 += MU_T) {
    const int sx = s0 + tid;
    const int flag = (sx < min(c0 + MU_LIST, d.n_launch) && lipm(L, sx + d.slot0)[LI_REBUILD] && d.opt_map_merge) ? 1 : 0;
    int tot;
    const int ex = block_excl_scan(flag, s_w, &tot);
    if (flag) s_items[n_items + ex] = (unsigned short)(sx - c0);
    n_items += tot;
  }
#endif
  __syncthreads();
  for (int it = blockIdx.y; it < n_items; it += gridDim.y) {
  const int slot = c0 + (int)s_items[it] + d.slot0;
  int* li = lipm(L, slot);
  __syncthreads();   // LDS of the previous slot is reused
  const bool active = li[LI_REBUILD] && d.opt_map_merge;
  if (active) {
    const float inv = 1.0f / (m == 0 ? d.P.lm_leaf_corner : d.P.lm_leaf_surf);
    const int cap = m == 0 ? L.map_cap_c : L.map_cap_s;
    u64* U = map_U(L, slot, m);
    int* cnt = map_Ucnt(L, slot, m);
    const int* rec = L.rec + (size_t)slot * L.K;
    const int ncur = li[LI_REC_CNT];
    {
      // the two windows into LDS (s_add / s_rem double as staging), then their multiset difference by one thread:
      // both are non-decreasing lists of frame ids
      const int* prev = L.rec_prev + (size_t)slot * L.K;
      const int np = li[LI_PREV_CNT], nkf = li[LI_NKF];
      int* s_cur = reinterpret_cast<int*>(mu_smem);
      int* s_prev = s_cur + MAP_KMAX;
      for (int j = tid; j < ncur; j += MU_T) s_cur[j] = rec[j];
      for (int j = tid; j < np; j += MU_T) s_prev[j] = prev[j];
      __syncthreads();
      if (tid == 0) {
        int nr = 0, na = 0;
        bool valid = li[LI_UVALID] != 0;
        if (valid) {
          int a = 0, b = 0;
          while (a < np || b < ncur) {
            if (b >= ncur || (a < np && s_prev[a] < s_cur[b])) { if (s_prev[a] < nkf - L.KR) valid = false; s_rem[nr++] = s_prev[a++] % L.KR; }   // its ring entry must still hold it
            else if (a >= np || s_cur[b] < s_prev[a]) s_add[na++] = s_cur[b++] % L.KR;
            else { ++a; ++b; }
          }
        }
        if (!valid) {   // rebuild from nothing: every run of the window is inserted
          nr = 0; na = 0;
          for (int b = 0; b < ncur; ++b) s_add[na++] = s_cur[b] % L.KR;
        }
        s_nrem = nr; s_nadd = na; s_nU = valid ? li[LI_NU_C + m] : 0; s_err = 0;
      }
    }
    __syncthreads();
    int nU = s_nU;
    u64* S_key = reinterpret_cast<u64*>(map_out(L, slot, m));   // scratch of the merges: the map's output buffer (rewritten by map_accum afterwards)
    int* S_cnt = reinterpret_cast<int*>(S_key + cap);
    const int nR = s_nrem == 1 ? run_n(L, slot, m, s_rem[0]) : 0, nA = s_nadd == 1 ? run_n(L, slot, m, s_add[0]) : 0;
    const bool fast = nU > 0 && s_nrem <= 1 && s_nadd <= 1 && nR <= MU_FCAP && nA <= MU_FCAP;
    if (fast) {
      // ---- steady state: one run out, one run in.  The voxel keys of both runs are staged in LDS (sorted, as the runs are);
      // every thread owns MU_E consecutive entries of the list, finds the pieces of both runs that fall into its key interval by
      // binary search and merges the three sorted sequences: counts go down / up, new voxels are inserted in order, empty ones
      // disappear.  No atomics; two passes (count, write) around one scan.
      u64* s_kr = reinterpret_cast<u64*>(mu_smem);
      u64* s_ka = s_kr + MU_FCAP;
      __syncthreads();   // (the window lists staged in the same LDS are dead)
      if (nR) { const float4* pts = run_pts(L, slot, m, s_rem[0]); for (int i = tid; i < nR; i += MU_T) s_kr[i] = vkey_of(pts[i], inv); }
      if (nA) { const float4* pts = run_pts(L, slot, m, s_add[0]); for (int i = tid; i < nA; i += MU_T) s_ka[i] = vkey_of(pts[i], inv); }
      __syncthreads();
      auto lb = [](const u64* k, int n, u64 key) { int lo = 0, hi = n; while (lo < hi) { const int mid = (lo + hi) >> 1; if (k[mid] < key) lo = mid + 1; else hi = mid; } return lo; };
      int out_off = 0;
      for (int t0 = 0; t0 < nU; t0 += MU_T * MU_E) {
        const int i0 = t0 + tid * MU_E, ne = max(0, min(MU_E, nU - i0));
        u64 uk[MU_E];
        int uc[MU_E];
#pragma unroll
        for (int e = 0; e < MU_E; ++e) { const int i = min(i0 + e, nU - 1); uk[e] = U[i]; uc[e] = cnt[i]; }
        const u64 klo = i0 == 0 ? 0ull : uk[0];
        const u64 khi = (ne > 0 && i0 + ne < nU) ? U[i0 + ne] : ~0ull;
        const int pr0 = ne > 0 ? lb(s_kr, nR, klo) : 0, pa0 = ne > 0 ? lb(s_ka, nA, klo) : 0, pa1 = ne > 0 ? (khi == ~0ull ? nA : lb(s_ka, nA, khi)) : 0;
        // the merge of this thread's interval; emit(key, count) is called for every voxel of the new list, in order
        auto merge = [&](auto&& emit) {
          int r = pr0, a = pa0;
#pragma unroll
          for (int e = 0; e < MU_E; ++e) {
            if (e < ne) {
              const u64 key = uk[e];
              while (a < pa1 && s_ka[a] < key) { const u64 k2 = s_ka[a]; int c = 0; while (a < pa1 && s_ka[a] == k2) { ++c; ++a; } emit(k2, c); }
              while (r < nR && s_kr[r] < key) { ++r; s_err = 1; }   // a point that left was never counted: list out of sync
              int c = uc[e];
              while (r < nR && s_kr[r] == key) { --c; ++r; }
              while (a < pa1 && s_ka[a] == key) { ++c; ++a; }
              if (c > 0) emit(key, c);
            }
          }
          while (a < pa1) { const u64 k2 = s_ka[a]; int c = 0; while (a < pa1 && s_ka[a] == k2) { ++c; ++a; } emit(k2, c); }
        };
        int nout = 0;
        merge([&](u64, int) { ++nout; });
        int tot;
        int pos = out_off + block_excl_scan(nout, s_w, &tot);
        merge([&](u64 k, int c) { if (pos < cap) { S_key[pos] = k; S_cnt[pos] = c; } ++pos; });
        out_off += tot;
      }
      __threadfence_block();
      __syncthreads();
      int nU2 = out_off;
      if (nU2 > cap) { nU2 = cap; if (tid == 0) s_err = 2; }
      for (int i = tid; i < nU2; i += MU_T) { U[i] = S_key[i]; cnt[i] = S_cnt[i]; }
      __threadfence_block();
      __syncthreads();
      nU = nU2;
    } else {
    // ---- the runs that left: one decrement per point
    for (int r = 0; r < s_nrem; ++r) {
      const float4* pts = run_pts(L, slot, m, s_rem[r]);
      const int n = run_n(L, slot, m, s_rem[r]);
      for (int i = tid; i < n; i += MU_T) {
        const u64 key = vkey_of(pts[i], inv);
        const int pos = lower_bound_U(U, nU, key);
        if (pos < nU && U[pos] == key) atomicSub(&cnt[pos], 1); else s_err = 1;
      }
    }
    __threadfence_block();
    __syncthreads();
    // ---- the runs that entered, one after the other
    float4* nk = L.newkeys + ((size_t)slot * 2 + m) * L.total_cap;   // (key lo, key hi, lower bound, count) of the voxels this run adds
    int* S_E = S_cnt + cap;   // third part of the scratch: exclusive keep-scan
    const int nadd = s_nadd;
    for (int a = 0; a <= nadd; ++a) {
      // (iteration nadd is the purge-only pass when nothing was added but something left)
      if (a == nadd && !(nadd == 0 && s_nrem > 0)) break;
      const bool purge_only = a == nadd;
      const float4* pts = purge_only ? nullptr : run_pts(L, slot, m, s_add[a]);
      const int n = purge_only ? 0 : run_n(L, slot, m, s_add[a]);
      if (tid == 0) s_nnew = 0;
      __syncthreads();
      // B1: known voxels count the point; the first point of every unknown voxel is collected in run order
      for (int i0 = 0; i0 < n; i0 += MU_T) {
        const int i = i0 + tid;
        bool newhead = false;
        u64 key = 0;
        int pos = 0, c = 0;
        if (i < n) {
          key = vkey_of(pts[i], inv);
          pos = lower_bound_U(U, nU, key);
          const bool found = pos < nU && U[pos] == key;
          if (found) atomicAdd(&cnt[pos], 1);
          else if (i == 0 || vkey_of(pts[i - 1], inv) != key) {
            newhead = true;
            int i2 = i + 1;
            while (i2 < n && vkey_of(pts[i2], inv) == key) ++i2;
            c = i2 - i;
          }
        }
        int tot;
        const int ex = block_excl_scan(newhead ? 1 : 0, s_w, &tot);
        const int base = s_nnew;
        if (newhead) nk[base + ex] = make_float4(__uint_as_float((unsigned)key), __uint_as_float((unsigned)(key >> 32)), __int_as_float(pos), __int_as_float(c));
        __syncthreads();
        if (tid == 0) s_nnew = base + tot;
        __syncthreads();
      }
      __threadfence_block();
      __syncthreads();
      const int nnew = s_nnew;
      // B2: merge.  Old entry i (kept iff its count is > 0) goes to E(i) + #{new keys with lower bound <= i}; new key j to E(lb_j) + j,
      // E = exclusive scan of the keep flags.
      int kept_total = 0;
      for (int i0 = 0; i0 < nU; i0 += MU_T) {
        const int i = i0 + tid;
        const int c = i < nU ? __hip_atomic_load(&cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const int keep = c > 0 ? 1 : 0;
        int tot;
        const int ex = kept_total + block_excl_scan(keep, s_w, &tot);
        if (i < nU) {
          S_E[i] = ex;
          if (keep) {
            int lo = 0, hi = nnew;   // new keys with lower bound <= i
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (__float_as_int(nk[mid].z) <= i) lo = mid + 1; else hi = mid; }
            const int p = ex + lo;
            if (p < cap) { S_key[p] = U[i]; S_cnt[p] = c; }
          }
        }
        kept_total += tot;
      }
      __threadfence_block();
      __syncthreads();
      for (int j = tid; j < nnew; j += MU_T) {
        const float4 e = nk[j];
        const int lb = __float_as_int(e.z);
        const int p = (lb < nU ? S_E[lb] : kept_total) + j;
        if (p < cap) { S_key[p] = ((u64)__float_as_uint(e.y) << 32) | (u64)__float_as_uint(e.x); S_cnt[p] = __float_as_int(e.w); }
      }
      __threadfence_block();
      __syncthreads();
      int nU2 = kept_total + nnew;
      if (nU2 > cap) { nU2 = cap; if (tid == 0) s_err = 2; }
      for (int i = tid; i < nU2; i += MU_T) { U[i] = S_key[i]; cnt[i] = S_cnt[i]; }
      __threadfence_block();
      __syncthreads();
      nU = nU2;
    }
    }   // general path
    // ---- window totals, bounding box (for PCL's leaf-size check and for the k-NN grid), outputs
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    int kraw = 0;
    for (int j = tid; j < ncur; j += MU_T) {
      const int e = rec[j] % L.KR;
      const int n = run_n(L, slot, m, e);
      kraw += n;
      if (n > 0) {
        const float* b = run_box(L, slot, m, e);
        mn[0] = fminf(mn[0], b[0]); mn[1] = fminf(mn[1], b[1]); mn[2] = fminf(mn[2], b[2]);
        mx[0] = fmaxf(mx[0], b[4]); mx[1] = fmaxf(mx[1], b[5]); mx[2] = fmaxf(mx[2], b[6]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      kraw += __shfl_xor(kraw, o, 64);
#pragma unroll
      for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64)); }
    }
    __syncthreads();
    if ((tid & 63) == 0) { s_kr_tot[tid >> 6] = kraw; for (int a = 0; a < 3; ++a) { s_box[a][tid >> 6] = mn[a]; s_box[3 + a][tid >> 6] = mx[a]; } }
    __syncthreads();
    kraw = 0;
    for (int w = 0; w < MU_T / 64; ++w) {
      kraw += s_kr_tot[w];
      for (int a = 0; a < 3; ++a) { mn[a] = fminf(w ? mn[a] : s_box[a][0], s_box[a][w]); mx[a] = fmaxf(w ? mx[a] : s_box[3 + a][0], s_box[3 + a][w]); }
    }
    const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    const bool pass = kraw > 0 && dx * dy * dz > 2147483647LL;   // PCL: "leaf size too small" -> output = input (in the reference's order)
    if (pass) {
      float4* out = map_out(L, slot, m);
      const int* kc = L.kf_cnt + (size_t)slot * L.KR * 4;
      int off = 0;
      for (int j = 0; j < ncur; ++j) {
        const int e = rec[j] % L.KR;
        const size_t rs = (size_t)slot * L.KR + e;
        // (the sorted runs lost the input order: the raw clouds are transformed again, in the reference's order)
        float mm[3][4];
        keypose_matrix(L.kf_pose + rs * 8, mm);
        const int n0 = m == 0 ? kc[e * 4 + 0] : kc[e * 4 + 1], n1 = m == 0 ? 0 : kc[e * 4 + 2];
        const float4* a0 = m == 0 ? L.kf_raw_c + rs * L.kf_cap_c : L.kf_raw_s + rs * L.kf_cap_s;
        const float4* a1 = L.kf_raw_o + rs * L.kf_cap_o;
        for (int i = tid; i < n0 && off + i < cap; i += MU_T) out[off + i] = kf_transform(mm, a0[i]);
        for (int i = tid; i < n1 && off + n0 + i < cap; i += MU_T) out[off + n0 + i] = kf_transform(mm, a1[i]);
        const int n = n0 + n1;
        off += n;
      }
    }
    if (tid == 0) {
      li[LI_NU_C + m] = nU;
      li[LI_KRAW_C + m] = kraw;
      li[LI_KDS_C + m] = pass ? min(kraw, cap) : nU;
      if (pass) atomicOr(&li[LI_MAP_PASS], 1 << m); else atomicAnd(&li[LI_MAP_PASS], ~(1 << m));
      if (s_err) li[LI_OVERFLOW] = s_err == 2 ? 1 : 4;   // 4: voxel list out of sync with the window (internal error)
      unsigned* bb = L.map_bbox + ((size_t)slot * 2 + m) * 8;
      for (int a = 0; a < 3; ++a) { bb[a] = vbox_enc(mn[a]); bb[4 + a] = ~vbox_enc(mx[a]); }
    }
  }
  }   // items of this workgroup
  __syncthreads();   // (s_items is rewritten for the next MU_LIST slots)
  }
  // ---- the last workgroup of the launch plans map_accum and closes the bookkeeping of every slot
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(&W.count[1], 1) == (int)(gridDim.x * gridDim.y) - 1;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // (every thread takes slots tid, tid + MU_T, ...: a single thread walking a few hundred slots' counters in global memory took a millisecond)
  int run_total = 0;
  for (int s0 = 0; s0 < d.n_launch; s0 += MU_T) {
    const int s = s0 + tid;
    int nmine = 0, nch[2] = {0, 0};
    int* l2 = s < d.n_launch ? lipm(L, s + d.slot0) : nullptr;
    bool rebuilt = false;
    if (l2) {
      l2[LI_KF_PENDING] = 0;   // the VoxelGrid round before this kernel has sorted the pending key frame
      rebuilt = l2[LI_REBUILD] && d.opt_map_merge;
      if (rebuilt) {
        for (int mm = 0; mm < 2; ++mm) nch[mm] = ((l2[LI_MAP_PASS] >> mm) & 1) ? 0 : (l2[LI_NU_C + mm] + MAP_R - 1) / MAP_R;
        nmine = nch[0] + nch[1];
        const int* rec = L.rec + (size_t)(s + d.slot0) * L.K;
        int* prev = L.rec_prev + (size_t)(s + d.slot0) * L.K;
        const int nc = l2[LI_REC_CNT];
        for (int j = 0; j < nc; ++j) prev[j] = rec[j];
        l2[LI_PREV_CNT] = nc; l2[LI_UVALID] = 1;
      }
    }
    int tot;
    int n = run_total + block_excl_scan(nmine, s_w, &tot);
    if (rebuilt) {
      for (int mm = 0; mm < 2; ++mm)
        for (int c = 0; c < nch[mm]; ++c) {
          if (n < W.cap && c < 2048) W.items[n] = (s << 12) | (mm << 11) | c; else l2[LI_OVERFLOW] = 1;
          ++n;
        }
    }
    run_total += tot;
  }
  if (tid == 0) {
    W.count[0] = min(run_total, W.cap);
    W.count[1] = 0;
  }
}

// persistent workgroups over the work list.
// Round 3 rewrite.  The first version visited the window's runs one after the other — a barrier per run, a run's piece of the chunk
// (a few hundred points) on 512 threads, the followers of a voxel-run added by up to 63 dependent wave shuffles: 50 (K = 200: 200) serial
// rounds of ~6 us per work item, 23 % VALU-busy, 8.7 % of the whole pipeline's time for a kernel that runs on every sixth mapping
// frame.  Now the pieces of several runs form a BATCH (<= MA_JB sub-pieces, <= MA_BCAP points) that is staged in LDS in one sweep:
//   1   every point of the batch is loaded once into LDS (s_pts); the first point of every (sub-piece, voxel) pair finds its voxel by
//       binary search in the chunk's keys (LDS) and writes its position to s_start[sub][voxel]; followers are marked in s_rank
//   2   one THREAD PER VOXEL walks the sub-pieces in window order and adds its points in order, accumulator in registers
// Same additions in the same order as before (run after run, input order inside a run, starting from 0): bit-identical sums.
#ifndef MA_JB
#define MA_JB 8           // sub-pieces per batch.  (12 x 2048 points = 62 KB of LDS: 351 k scans/s; 8 x 1024 = 38 KB: 362 k; 6 x 512: the same)
#endif
#ifndef MA_PT
#define MA_PT 2           // consecutive points of a batch per thread
#endif
#define MA_BCAP (MA_PT * MA_T)   // points per batch (a longer piece is cut into sub-pieces; they stay in order)
#ifndef MA_HASH
#define MA_HASH 0                // 1: a staged point finds its voxel through a hash table of the chunk's keys instead of a binary search (measured, round 5: slower — see DESIGN.md)
#endif
#ifndef MA_PAIR
#define MA_PAIR 0                // 1: both ends of a run's piece by one interleaved 4-ary search (bound_run_pair)
#endif
#define MA_HT 1024               // hash slots for the chunk's MAP_R keys (load factor <= 0.5)
DEV_INLINE unsigned ma_hash(u64 key) {   // voxel keys of a chunk differ in their low (x) bits first, then y (bit 21), rarely z (bit 42): fold, then a multiplicative hash
  const unsigned f = (unsigned)key ^ (unsigned)(key >> 21) * 0x9E3779B1u ^ (unsigned)(key >> 42) * 0x85EBCA6Bu;
  return (f * 0x9E3779B1u) >> 22;
}
static_assert(MA_HT == 1024 && MAP_R <= MA_HT / 2, "ma_hash returns 10 bits; the table is at most half full");
static_assert(MAP_R == MA_T, "phase 2: one thread per voxel of the chunk");
__global__ void __launch_bounds__(MA_T, 8) map_accum(DevCtx d, LmCtx L, MapWork W) {   // 8 wavefronts per SIMD = 64 VGPRs: four of these workgroups per CU (what the 38 KB of LDS allow); a variant at 67 VGPRs ran three per CU and cost the bench 1.5 %
  __shared__ u64 s_key[MAP_R];
  __shared__ float4 s_pts[MA_BCAP];
  __shared__ unsigned short s_rank[MA_BCAP];
  __shared__ unsigned short s_start[MA_JB * MAP_R];   // [sub-piece][voxel]: position + 1 of the pair's first point in the batch (valid where s_mask has the bit)
  __shared__ unsigned s_mask[MAP_R];                  // per voxel: the sub-pieces of the batch that hold points of it
  __shared__ int s_lo[MAP_KMAX], s_hi[MAP_KMAX];
  __shared__ unsigned short s_ent[MAP_KMAX];   // (ring entries < KR <= MAP_KMAX + 1; 16 bits keep the kernel under 40 KB: four workgroups per CU)
  __shared__ int s_sub_ent[MA_JB], s_sub_lo[MA_JB], s_sub_off[MA_JB + 1];
  __shared__ int s_nsub, s_next_j, s_next_pos;
  // the chunk's voxels by key: open addressing, MA_HT slots for <= MAP_R keys (entry = voxel + 1, 0 = empty; the key itself is compared in s_key).  A staged
  // point finds its voxel with ~1.5 probes instead of the nine dependent LDS reads of a binary search over 64-bit keys (round 5; ~2/3 of the staging loop's
  // instructions were that search: nearly every staged point is the first of its voxel in its run)
#if MA_HASH
  __shared__ unsigned short s_tab[MA_HT];
#endif
  const int tid = threadIdx.x;
  const int nitems = W.count[0];
  for (int i = tid; i < MAP_R; i += MA_T) s_mask[i] = 0u;   // phase 2 leaves every entry it used at 0 again
  for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
    const int item = W.items[it];
    const int slot = (item >> 12) + d.slot0, m = (item >> 11) & 1, chunk = item & 2047;
    int* li = lipm(L, slot);
    const float inv = 1.0f / (m == 0 ? d.P.lm_leaf_corner : d.P.lm_leaf_surf);
    const u64* U = map_U(L, slot, m);
    const int nU = li[LI_NU_C + m];
    const int r0 = chunk * MAP_R, nr = min(MAP_R, nU - r0);
    const int* rec = L.rec + (size_t)slot * L.K;
    const int nwin = li[LI_REC_CNT];
    __syncthreads();   // LDS of the previous item is reused
    const u64 mykey = tid < nr ? U[r0 + tid] : 0ull;
    if (tid < nr) s_key[tid] = mykey;
#if MA_HASH
    for (int i = tid; i < MA_HT; i += MA_T) s_tab[i] = 0;
    __syncthreads();
    if (tid < nr) {   // (16-bit entries: the insert is a compare-and-swap on the aligned 32-bit word)
      unsigned h = ma_hash(mykey);
      while (true) {
        unsigned* w = reinterpret_cast<unsigned*>(s_tab) + (h >> 1);
        const int sh = (h & 1u) * 16;
        const unsigned old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (((old >> sh) & 0xFFFFu) == 0u) { if (atomicCAS(w, old, old | ((unsigned)(tid + 1) << sh)) == old) break; }
        else h = (h + 1) & (MA_HT - 1);
      }
    }
#endif
    __syncthreads();
    const u64 key_lo = s_key[0], key_hi = s_key[nr - 1];
    for (int j = tid; j < nwin; j += MA_T) {   // the piece of every run that falls into this chunk's key range
      const int e = rec[j] % L.KR;
      const float4* pts = run_pts(L, slot, m, e);
      const int n = run_n(L, slot, m, e);
      s_ent[j] = (unsigned short)e;
#if MA_PAIR
      int blo, bhi;
      bound_run_pair(pts, n, inv, key_lo, key_hi, &blo, &bhi);
      s_lo[j] = blo; s_hi[j] = bhi;
#else
      s_lo[j] = bound_run(pts, n, inv, key_lo, false);
      s_hi[j] = bound_run(pts, n, inv, key_hi, true);
#endif
    }
    if (tid == 0) { s_next_j = 0; s_next_pos = -1; }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;
    while (true) {
      __syncthreads();
      if (tid == 0) {   // the next batch: sub-pieces in window order
        int j = s_next_j, pos = s_next_pos, ns = 0, tot = 0;
        while (j < nwin && ns < MA_JB && tot < MA_BCAP) {
          const int lo = pos >= 0 ? pos : s_lo[j], hi = s_hi[j];
          if (hi <= lo) { ++j; pos = -1; continue; }
          const int take = min(hi - lo, MA_BCAP - tot);
          s_sub_ent[ns] = s_ent[j]; s_sub_lo[ns] = lo; s_sub_off[ns] = tot;
          tot += take; ++ns;
          if (lo + take < hi) pos = lo + take; else { ++j; pos = -1; }
        }
        s_sub_off[ns] = tot;
        s_nsub = ns; s_next_j = j; s_next_pos = pos;
      }
      __syncthreads();
      const int nsub = s_nsub;
      if (nsub == 0) break;
      const int btot = s_sub_off[nsub];
      // ---- 1: stage the batch; thread t takes the MA_PT consecutive points from t * MA_PT (their loads are issued together).  Only the
      // first point of a voxel-run looks its voxel up (the runs are sorted: the followers have the same key)
      {
        const int idx0 = tid * MA_PT;
        int jj = 0;
#pragma unroll
        for (int u = 1; u < MA_JB; ++u) jj += (u < nsub && s_sub_off[u] <= idx0) ? 1 : 0;
        float4 p[MA_PT], pprev;
        int sj[MA_PT];
        {
          int j2 = jj;
#pragma unroll
          for (int u = 0; u < MA_PT; ++u) {
            const int idx = min(idx0 + u, btot - 1);
            while (j2 + 1 < nsub && idx >= s_sub_off[j2 + 1]) ++j2;
            sj[u] = j2;
            p[u] = run_pts(L, slot, m, s_sub_ent[j2])[s_sub_lo[j2] + (idx - s_sub_off[j2])];
          }
          const int ip = min(idx0, btot - 1);
          pprev = run_pts(L, slot, m, s_sub_ent[jj])[s_sub_lo[jj] + max(ip - s_sub_off[jj] - 1, 0)];
        }
        u64 kprev = vkey_of(pprev, inv);
        int jprev = idx0 > s_sub_off[jj] ? jj : -1;   // -1: the thread's first point opens its sub-piece
#pragma unroll
        for (int u = 0; u < MA_PT; ++u) {
          const int idx = idx0 + u;
          if (idx < btot) {
            const u64 key = vkey_of(p[u], inv);
            const bool head = sj[u] != jprev || key != kprev;
            int a = 0xFFFF;
            if (head) {
#if MA_HASH
              unsigned h = ma_hash(key);
              int lo2 = -1;
              for (int probe = 0; probe < MA_HT; ++probe) {
                const int e = s_tab[h];
                if (e == 0) break;
                if (s_key[e - 1] == key) { lo2 = e - 1; break; }
                h = (h + 1) & (MA_HT - 1);
              }
              if (lo2 < 0) { li[LI_OVERFLOW] = 4; lo2 = 0; }   // voxel list out of sync (internal error)
#else
              int lo2 = 0, hi2 = nr;
              while (lo2 < hi2) { const int mid = (lo2 + hi2) >> 1; if (s_key[mid] < key) lo2 = mid + 1; else hi2 = mid; }
              if (lo2 >= nr || s_key[lo2] != key) { li[LI_OVERFLOW] = 4; lo2 = min(lo2, nr - 1); }   // voxel list out of sync (internal error)
#endif
              a = lo2;
              s_start[sj[u] * MAP_R + a] = (unsigned short)(idx + 1);
              atomicOr(&s_mask[a], 1u << sj[u]);
            }
            s_pts[idx] = p[u];
            s_rank[idx] = (unsigned short)a;
            kprev = key; jprev = sj[u];
          }
        }
      }
      __syncthreads();
      // ---- 2: thread tid = voxel tid of the chunk.  ONE loop, one point per iteration; a thread moves on to its next sub-piece (the set bits
      // of s_mask) inside the same iteration, so a wavefront runs max-over-lanes(points of the voxel in this batch) iterations — not the sum
      // over the sub-pieces of the longest voxel-run in each.
      if (tid < nr) {
        unsigned msk = s_mask[tid];
        s_mask[tid] = 0u;
        int k = -1, end = 0;
        while (true) {
          if (k < 0) {
            if (!msk) break;
            const int jj = __ffs((int)msk) - 1;
            msk &= msk - 1;
            k = (int)s_start[jj * MAP_R + tid] - 1;
            end = s_sub_off[jj + 1];
          }
          const float4 q = s_pts[k];
          acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; ++cnt; ++k;
          if (!(k < end && s_rank[k] == 0xFFFFu)) k = -1;
        }
      }
    }
    float4* out = map_out(L, slot, m);
    const int cap = m == 0 ? L.map_cap_c : L.map_cap_s;
    if (tid < nr && r0 + tid < cap) {
      const float fn = (float)cnt;
      out[r0 + tid] = make_float4(acc.x / fn, acc.y / fn, acc.z / fn, acc.w / fn);
    }
  }
}

void launch_map_update(const DevCtx& d, const LmCtx& L, const MapWork& W, hipStream_t st) {
  static const bool cfg = hipFuncSetAttribute(reinterpret_cast<const void*>(map_update), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MU_FCAP * 8) == hipSuccess;
  (void)cfg;
  // slots per workgroup (2048-stream bench, scans/s): 1: 367 k, 2: 388 k, 4: 396 k, 8: 403 k, 16: 405-408 k, 32: 409 k, 64: 409 k; at least 16 workgroups
  // per map where there are that many slots, so that a small handle's rebuilds are not serialised
  static const int div = []() { const char* e = getenv("ALEGO_MU_DIV"); return e && atoi(e) > 0 ? atoi(e) : 32; }();
  const int gy = std::max(std::min(d.n_launch, 16), (d.n_launch + div - 1) / div);
  ALEGO_LAUNCH(map_update, dim3(2, gy), dim3(MU_T), (size_t)2 * MU_FCAP * 8, st, d, L, W);
}
void launch_map_accum(const DevCtx& d, const LmCtx& L, const MapWork& W, hipStream_t st) {
  static const int gmax = []() { const char* e = getenv("ALEGO_MA_GRID"); return e && atoi(e) > 0 ? atoi(e) : 1024; }();
  const int grid = std::min(gmax, std::max(16, 4 * d.n_launch));
  ALEGO_LAUNCH(map_accum, dim3(grid), dim3(MA_T), 0, st, d, L, W);
}
