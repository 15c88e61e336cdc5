// kernels_voxel.hip — batched pcl::VoxelGrid<PointXYZI>::applyFilter (see voxel.h).
// Replaces the VoxelGrid calls of laserMapping.cpp:316-319 (map; the reference redoes it every mapping
// frame, here only when the key-frame set changed) and :329-342 (current scan).
//
// Round 1 first used rocprim::segmented_radix_sort_pairs for the (voxel id, position) sort; a 45-75 k point
// map segment is sorted there by ONE workgroup (~1.2 ms, tools/micro/segsort_bench.hip).  This version is a
// two-level bucket sort written for the job: voxel ids are bounded by the grid size, so
//   vox_keys      key = voxel id; bucket = key >> shift (<= 4096 buckets/job); bucket histogram (atomics)
//   vox_bscan     exclusive scan of the bucket histogram (one workgroup per job)
//   vox_bscatter  (key << 32 | position) into its bucket (atomic cursor: order inside a bucket is arbitrary)
//   vox_bsort     one wavefront per bucket: rank-by-counting sort of the bucket (registers for <= 64
//                 elements, LDS above), counts the voxels of the bucket
//   vox_vscan     exclusive scan of the per-bucket voxel counts -> output ranks (ascending voxel id)
//   vox_bcentroid one wavefront per bucket: voxel heads; the head lane accumulates its run in sorted
//                 (= original) order in f32 and divides by the count (pcl::CentroidPoint)
// Jobs differ by three orders of magnitude in size (a 45-75 k point map vs a 50 point outlier cloud) and most
// of the time the big ones are disabled, so the point- and bucket-parallel kernels run on a fixed pool of
// workgroups that walk a device-built work list (vox_plan: job -> number of 1024-point / 16-bucket items).
#include <cstring>

#include <hip/hip_runtime.h>

#include <vector>

#include "voxel.h"
#include "prof.h"

#define VB 256

typedef unsigned long long u64;

__device__ __forceinline__ unsigned vx_enc(float f) {
  const unsigned b = (unsigned)__float_as_int(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float vx_dec(unsigned e) {
  const unsigned b = (e >> 31) ? (e ^ 0x80000000u) : ~e;
  return __int_as_float((int)b);
}
__device__ __forceinline__ bool vx_enabled(const VoxJob& J) { return J.enable == nullptr || *J.enable != 0; }

#define VX_SMALL_MAX_ 8192     // jobs up to this many points are done by vox_small in one launch
#define VX_POOL 1024           // workgroups of the point / bucket parallel kernels
#define VX_PT_ITEM 1024        // points per work item
#define VX_BK_ITEM 64          // buckets per work item (= one wavefront; bsort / bcentroid run 64-thread workgroups)
#define VX_BPOOL 4096          // single-wave workgroups of the bucket kernels
#define VX_TINY 8              // buckets up to this size are handled by a single thread

// one workgroup: exclusive scan over jobs of their item counts.  which 0: ceil(n/VX_PT_ITEM) from the job
// inputs; which 1: ceil(nb/VX_BK_ITEM) from the geometry written by vox_keys.
__global__ void __launch_bounds__(VB) vox_plan(VoxCtx V, int which) {
  __shared__ int s[VB / 64];
  __shared__ int s_run;
  int* off = which == 0 ? V.pt_items : V.bk_items;
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  for (int j0 = 0; j0 < V.njobs; j0 += VB) {
    const int j = j0 + threadIdx.x;
    int v = 0;
    if (j < V.njobs) {
      const VoxJob J = V.jobs[j];
      if (vx_enabled(J)) {
        if (which == 0) { const int n = min(*J.n_in, J.cap); v = n > VX_SMALL_MAX_ ? (n + VX_PT_ITEM - 1) / VX_PT_ITEM : 0; }
        else v = (V.geom[j * VX_GEOM + 8] + VX_BK_ITEM - 1) / VX_BK_ITEM;
      }
    }
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((int)(threadIdx.x & 63) >= o) incl += t; }
    if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < VB / 64; ++w) { if (w < (int)(threadIdx.x >> 6)) woff += s[w]; tot += s[w]; }
    const int run = s_run;
    if (j < V.njobs) off[j] = run + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) s_run = run + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) off[V.njobs] = s_run;
}

// work item -> (job, chunk): largest job with off[job] <= item (jobs without items are skipped by the search)
__device__ __forceinline__ int vx_item_job(const int* off, int njobs, int item, int* chunk) {
  int lo = 0, hi = njobs;  // invariant: off[lo] <= item < off[hi]
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= item) lo = mid; else hi = mid; }
  *chunk = item - off[lo];
  return lo;
}

__global__ void __launch_bounds__(VB) vox_bbox(VoxCtx V) {
  __shared__ float s[6][VB / 64];
  const int total = V.pt_items[V.njobs];
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    int chunk;
    const int job = vx_item_job(V.pt_items, V.njobs, item, &chunk);
    const VoxJob J = V.jobs[job];
    const int n = min(*J.n_in, J.cap);
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (int i = chunk * VX_PT_ITEM + threadIdx.x; i < min(n, (chunk + 1) * VX_PT_ITEM); i += VB) {
      const float4 p = J.in[i];
      mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
      mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64)); }
      if ((threadIdx.x & 63) == 0) { s[a][threadIdx.x >> 6] = mn[a]; s[3 + a][threadIdx.x >> 6] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      const int a = threadIdx.x;
      float v = s[a][0];
      for (int w = 1; w < VB / 64; ++w) v = a < 3 ? fminf(v, s[a][w]) : fmaxf(v, s[a][w]);
      unsigned* bb = V.bbox + job * 8;
      if (a < 3) atomicMin(&bb[a], vx_enc(v)); else atomicMin(&bb[4 + a - 3], ~vx_enc(v));
    }
    __syncthreads();
  }
}

// one thread per job: voxel-grid geometry and bucket layout from the bounding box
__global__ void __launch_bounds__(VB) vox_geom(VoxCtx V) {
  const int job = blockIdx.x * VB + threadIdx.x;
  if (job >= V.njobs) return;
  const VoxJob J = V.jobs[job];
  int* g = V.geom + job * VX_GEOM;
  if (!vx_enabled(J)) { g[5] = 0; g[8] = 0; g[9] = 1; return; }
  const int n = min(*J.n_in, J.cap);
  if (n <= VX_SMALL_MAX_) { g[5] = 0; g[8] = 0; g[9] = 1; return; }  // done by vox_small
  g[9] = 0;
  const float inv = 1.0f / J.leaf;
  const unsigned* bb = V.bbox + job * 8;
  int minb[3] = {0, 0, 0}, mul1 = 1, mul2 = 1, pass = 0;
  unsigned T = 1;
  if (n > 0) {
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) { mn[a] = vx_dec(bb[a]); mx[a] = vx_dec(~bb[4 + a]); }
    const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    pass = (dx * dy * dz > 2147483647LL) ? 1 : 0;  // PCL: "leaf size too small" -> output = input
    int divb[3];
    for (int a = 0; a < 3; ++a) { minb[a] = (int)floorf(mn[a] * inv); divb[a] = (int)floorf(mx[a] * inv) - minb[a] + 1; }
    mul1 = divb[0]; mul2 = divb[0] * divb[1];
    T = pass ? (unsigned)n : (unsigned)divb[0] * (unsigned)divb[1] * (unsigned)divb[2];
    if (T == 0) T = 1;
  }
  // bucket = key >> shift, about 1 point per bucket (few multi-voxel buckets), at most J.nbcap buckets
  int target = 64;
  while (target < J.nbcap && target < n) target <<= 1;
  int shift = 0;
  while (((T - 1) >> shift) >= (unsigned)target) ++shift;
  g[0] = minb[0]; g[1] = minb[1]; g[2] = minb[2]; g[3] = mul1; g[4] = mul2; g[5] = n; g[6] = pass; g[7] = shift;
  g[8] = n > 0 ? (int)((T - 1) >> shift) + 1 : 0;
}

__global__ void __launch_bounds__(VB) vox_keys(VoxCtx V) {
  const int total = V.pt_items[V.njobs];
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    int chunk;
    const int job = vx_item_job(V.pt_items, V.njobs, item, &chunk);
    const VoxJob J = V.jobs[job];
    const int* g = V.geom + job * VX_GEOM;
    const int n = g[5], pass = g[6], shift = g[7], nb = g[8];
    const int m0 = g[0], m1 = g[1], m2 = g[2], mul1 = g[3], mul2 = g[4];
    const float inv = 1.0f / J.leaf;
    int* bcnt = V.bcnt + J.boff0;
    for (int i = chunk * VX_PT_ITEM + threadIdx.x; i < min(n, (chunk + 1) * VX_PT_ITEM); i += VB) {
      unsigned key;
      if (pass) key = (unsigned)i;
      else {
        const float4 p = J.in[i];
        const int i0 = (int)(floorf(p.x * inv) - (float)m0);
        const int i1 = (int)(floorf(p.y * inv) - (float)m1);
        const int i2 = (int)(floorf(p.z * inv) - (float)m2);
        key = (unsigned)(i0 + i1 * mul1 + i2 * mul2);
      }
      V.keys[J.off + i] = key;
      atomicAdd(&bcnt[min(key >> shift, (unsigned)(nb - 1))], 1);
    }
  }
}

// exclusive scan of cnt[0..nb) by one workgroup; writes `off` and optionally a copy `cur`, the total to *total
__device__ void vx_block_scan(const int* cnt, int* off, int* cur, int nb, int* total) {
  __shared__ int s[VB / 64];
  __shared__ int s_run;
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += VB) {
    const int b = b0 + threadIdx.x;
    const int v = b < nb ? cnt[b] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((int)(threadIdx.x & 63) >= o) incl += t; }
    if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < VB / 64; ++w) { if (w < (int)(threadIdx.x >> 6)) woff += s[w]; tot += s[w]; }
    const int run = s_run;
    if (b < nb) { const int e = run + woff + incl - v; off[b] = e; if (cur) cur[b] = e; }
    __syncthreads();
    if (threadIdx.x == 0) s_run = run + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total) *total = s_run;
}

__global__ void __launch_bounds__(VB) vox_bscan(VoxCtx V) {
  const int job = blockIdx.x;
  const VoxJob J = V.jobs[job];
  if (!vx_enabled(J) || V.geom[job * VX_GEOM + 9]) return;
  const int nb = V.geom[job * VX_GEOM + 8];
  vx_block_scan(V.bcnt + J.boff0, V.boff + J.boff0, V.bcur + J.boff0, nb, nullptr);
}

__global__ void __launch_bounds__(VB) vox_bscatter(VoxCtx V) {
  const int total = V.pt_items[V.njobs];
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    int chunk;
    const int job = vx_item_job(V.pt_items, V.njobs, item, &chunk);
    const VoxJob J = V.jobs[job];
    const int* g = V.geom + job * VX_GEOM;
    const int n = g[5], shift = g[7], nb = g[8];
    int* bcur = V.bcur + J.boff0;
    for (int i = chunk * VX_PT_ITEM + threadIdx.x; i < min(n, (chunk + 1) * VX_PT_ITEM); i += VB) {
      const unsigned key = V.keys[J.off + i];
      const int pos = atomicAdd(&bcur[min(key >> shift, (unsigned)(nb - 1))], 1);
      V.pairs_a[J.off + pos] = ((u64)key << 32) | (unsigned)i;
    }
  }
}

__device__ __forceinline__ u64 vx_readlane64(u64 v, int lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return ((u64)hi << 32) | lo;
}

// sort every bucket ascending by (voxel id, position) and count its voxels: rank by counting
// (keys are unique: the position is in the low word).  One wavefront per item of 64 buckets:
//   m <= VX_TINY : one lane per bucket (the 64 buckets of the item in parallel)
//   m <= 64      : the wavefront takes the bucket, elements in registers, ranks via v_readlane
//   larger       : the wavefront through LDS (straight from memory beyond VX_WAVE_LDS elements)
#define VX_WAVE_LDS 1024   // 8 KB of LDS per single-wave workgroup keeps ~20 of them resident per CU
__global__ void __launch_bounds__(64) vox_bsort(VoxCtx V) {
  const int lane = threadIdx.x;
  __shared__ u64 s_buf[VX_WAVE_LDS];
  const int total = V.bk_items[V.njobs];
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    int chunk;
    const int job = vx_item_job(V.bk_items, V.njobs, item, &chunk);
    const VoxJob J = V.jobs[job];
    const int nb = V.geom[job * VX_GEOM + 8];
    const int* bcnt = V.bcnt + J.boff0;
    const int* boff = V.boff + J.boff0;
    int* bvox = V.bvox + J.boff0;
    // buckets are spatially ordered and dense regions cluster: lane l of item c takes bucket l*nitems + c
    const int nitems = (nb + VX_BK_ITEM - 1) / VX_BK_ITEM;
    const int b = lane * nitems + chunk;
    const int m = b < nb ? bcnt[b] : 0;
    if (m > 0 && m <= VX_TINY) {
      const u64* src = V.pairs_a + J.off + boff[b];
      u64* dst = V.pairs_b + J.off + boff[b];
      u64 e[VX_TINY];
#pragma unroll
      for (int i = 0; i < VX_TINY; ++i) e[i] = i < m ? src[i] : ~0ull;  // all loads in flight together
      int heads = 0;
#pragma unroll
      for (int i = 0; i < VX_TINY; ++i) {
        if (i < m) {
          int rank = 0;
          bool head = true;
#pragma unroll
          for (int j = 0; j < VX_TINY; ++j) { rank += e[j] < e[i]; if ((e[j] >> 32) == (e[i] >> 32) && e[j] < e[i]) head = false; }
          dst[rank] = e[i];
          heads += head;
        }
      }
      bvox[b] = heads;
    } else if (m == 0 && b < nb) {
      bvox[b] = 0;
    }
    u64 bigger = __ballot(m > VX_TINY);
    while (bigger) {
      const int src_lane = __ffsll((long long)bigger) - 1;
      bigger &= bigger - 1;
      const int bb = __shfl(b, src_lane, 64), mm = __shfl(m, src_lane, 64);
      const u64* src = V.pairs_a + J.off + boff[bb];
      u64* dst = V.pairs_b + J.off + boff[bb];
      int heads;
      if (mm <= 64) {
        const u64 e = lane < mm ? src[lane] : ~0ull;
        int rank = 0;
        bool head = lane < mm;  // head of a voxel: no element with the same voxel id and a smaller position
        for (int j = 0; j < mm; ++j) {
          const u64 o = vx_readlane64(e, j);
          rank += o < e;
          if ((o >> 32) == (e >> 32) && o < e) head = false;
        }
        if (lane < mm) dst[rank] = e;
        heads = (int)__popcll(__ballot(head));
      } else {
        const bool in_lds = mm <= VX_WAVE_LDS;
        __syncthreads();
        if (in_lds) for (int t = lane; t < mm; t += 64) s_buf[t] = src[t];
        __syncthreads();
        heads = 0;
        for (int t = lane; t < mm; t += 64) {
          const u64 e = in_lds ? s_buf[t] : src[t];
          int rank = 0;
          bool head = true;
          if (in_lds) { for (int j = 0; j < mm; ++j) { const u64 o = s_buf[j]; rank += o < e; if ((o >> 32) == (e >> 32) && o < e) head = false; } }
          else { for (int j = 0; j < mm; ++j) { const u64 o = src[j]; rank += o < e; if ((o >> 32) == (e >> 32) && o < e) head = false; } }
          dst[rank] = e;
          heads += head;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) heads += __shfl_xor(heads, o, 64);
      }
      if (lane == 0) bvox[bb] = heads;
    }
  }
}

__global__ void __launch_bounds__(VB) vox_vscan(VoxCtx V) {
  const int job = blockIdx.x;
  const VoxJob J = V.jobs[job];
  if (!vx_enabled(J) || V.geom[job * VX_GEOM + 9]) return;
  const int nb = V.geom[job * VX_GEOM + 8];
  vx_block_scan(V.bvox + J.boff0, V.voff + J.boff0, nullptr, nb, J.n_out);
}

// Centroids: f32 sums in sorted (= original) order, one output per voxel in ascending voxel id
// (pcl::CentroidPoint).  Tiny buckets: one lane walks its bucket.  Larger buckets: the wavefront gathers the
// points 64 at a time (parallel loads, one point per lane) and accumulates them in lock-step through
// v_readlane broadcasts — the additions stay strictly sequential, only the loads are parallel.
__device__ __forceinline__ float vx_bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__global__ void __launch_bounds__(64) vox_bcentroid(VoxCtx V) {
  const int lane = threadIdx.x;
  const int total = V.bk_items[V.njobs];
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    int chunk;
    const int job = vx_item_job(V.bk_items, V.njobs, item, &chunk);
    const VoxJob J = V.jobs[job];
    const int nb = V.geom[job * VX_GEOM + 8];
    int* bcnt = V.bcnt + J.boff0;
    const int* boff = V.boff + J.boff0;
    const int* voff = V.voff + J.boff0;
    // buckets are spatially ordered and dense regions cluster: lane l of item c takes bucket l*nitems + c
    const int nitems = (nb + VX_BK_ITEM - 1) / VX_BK_ITEM;
    const int b = lane * nitems + chunk;
    const int m = b < nb ? bcnt[b] : 0;
    if (m > 0 && m <= VX_TINY) {
      const u64* srt = V.pairs_b + J.off + boff[b];
      int rank = voff[b];
      u64 e[VX_TINY];
      float4 pt[VX_TINY];
#pragma unroll
      for (int k = 0; k < VX_TINY; ++k) e[k] = k < m ? srt[k] : 0ull;
#pragma unroll
      for (int k = 0; k < VX_TINY; ++k) pt[k] = k < m ? J.in[(unsigned)e[k]] : make_float4(0.f, 0.f, 0.f, 0.f);  // gathers in flight together
      float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
      int c = 0;
      unsigned cur = 0;
#pragma unroll
      for (int k = 0; k < VX_TINY; ++k) {
        if (k < m) {
          const unsigned vid = (unsigned)(e[k] >> 32);
          if (c > 0 && vid != cur) {
            const float fn = (float)c;
            if (rank < J.cap) J.out[rank] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
            ++rank; sx = sy = sz = si = 0.f; c = 0;
          }
          sx += pt[k].x; sy += pt[k].y; sz += pt[k].z; si += pt[k].w;
          ++c; cur = vid;
        }
      }
      const float fn = (float)c;
      if (rank < J.cap) J.out[rank] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
      bcnt[b] = 0;  // keep the histogram zeroed between rounds
    }
    u64 bigger = __ballot(m > VX_TINY);
    while (bigger) {
      const int src_lane = __ffsll((long long)bigger) - 1;
      bigger &= bigger - 1;
      const int bb = __shfl(b, src_lane, 64), mm = __shfl(m, src_lane, 64);
      const u64* srt = V.pairs_b + J.off + boff[bb];
      int rank = voff[bb];
      float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;  // wave-uniform accumulators
      int c = 0;
      unsigned cur = 0;
      for (int t0 = 0; t0 < mm; t0 += 64) {
        const int t = t0 + lane;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned myvid = 0;
        if (t < mm) { const u64 e = srt[t]; myvid = (unsigned)(e >> 32); p = J.in[(unsigned)e]; }
        const int cnt = min(64, mm - t0);
        for (int k = 0; k < cnt; ++k) {
          const unsigned vid = (unsigned)__builtin_amdgcn_readlane((int)myvid, k);
          if (c > 0 && vid != cur) {
            const float fn = (float)c;
            if (lane == 0 && rank < J.cap) J.out[rank] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
            ++rank; sx = sy = sz = si = 0.f; c = 0;
          }
          sx += vx_bcast(p.x, k); sy += vx_bcast(p.y, k); sz += vx_bcast(p.z, k); si += vx_bcast(p.w, k);
          ++c; cur = vid;
        }
      }
      const float fn = (float)c;
      if (lane == 0) {
        if (rank < J.cap) J.out[rank] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
        bcnt[bb] = 0;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Small jobs (n <= VX_SMALL_MAX points: the VoxelGrid calls on the current scan, laserMapping.cpp:329-342) run the
// whole filter in ONE launch, one workgroup per job, with every intermediate in LDS: bounding box, voxel ids,
// bucket histogram (LDS atomics), scan, scatter, per-bucket rank sort, voxel ranks, centroids.  The multi-kernel
// path above is left for the map (45-75 k points).  LDS: 4 B key + 2 B index per point, 6 B per bucket.
#define VX_SMALL_MAX 8192
#define VX_SMALL_NB 4096
#define VX_SB 512
__device__ __forceinline__ bool vx_less(const unsigned* key, unsigned a, unsigned b) { return key[a] < key[b] || (key[a] == key[b] && a < b); }

__global__ void __launch_bounds__(VX_SB) vox_small(VoxCtx V) {
  const int job = blockIdx.x;
  const VoxJob J = V.jobs[job];
  if (!vx_enabled(J)) return;
  const int n = min(*J.n_in, J.cap);
  if (n > VX_SMALL_MAX) return;  // left to the multi-kernel path (vox_geom reads the same condition)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  extern __shared__ __attribute__((aligned(16))) unsigned char vs_smem[];
  unsigned* s_key = reinterpret_cast<unsigned*>(vs_smem);                                             // [VX_SMALL_MAX]
  int* s_cnt = reinterpret_cast<int*>(vs_smem + 4 * VX_SMALL_MAX);                                    // [VX_SMALL_NB] histogram -> starts -> ends
  unsigned short* s_idx = reinterpret_cast<unsigned short*>(vs_smem + 4 * VX_SMALL_MAX + 4 * VX_SMALL_NB);          // [VX_SMALL_MAX]
  unsigned short* s_vox = reinterpret_cast<unsigned short*>(vs_smem + 6 * VX_SMALL_MAX + 4 * VX_SMALL_NB);          // [VX_SMALL_NB]
  __shared__ float s_red[6][VX_SB / 64];
  __shared__ int s_scan[VX_SB / 64];
  __shared__ int s_run;
  if (n == 0) { if (tid == 0) *J.n_out = 0; return; }
  const float inv = 1.0f / J.leaf;
  // getMinMax3D
  float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  for (int i = tid; i < n; i += VX_SB) {
    const float4 p = J.in[i];
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64)); }
    if (lane == 0) { s_red[a][wave] = mn[a]; s_red[3 + a][wave] = mx[a]; }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mn[a] = s_red[a][0]; mx[a] = s_red[3 + a][0];
#pragma unroll
    for (int w = 1; w < VX_SB / 64; ++w) { mn[a] = fminf(mn[a], s_red[a][w]); mx[a] = fmaxf(mx[a], s_red[3 + a][w]); }
  }
  const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > 2147483647LL) {  // PCL: "leaf size too small" -> output = input
    for (int i = tid; i < n; i += VX_SB) J.out[i] = J.in[i];
    if (tid == 0) *J.n_out = n;
    return;
  }
  int minb[3], divb[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { minb[a] = (int)floorf(mn[a] * inv); divb[a] = (int)floorf(mx[a] * inv) - minb[a] + 1; }
  const int mul1 = divb[0], mul2 = divb[0] * divb[1];
  unsigned T = (unsigned)divb[0] * (unsigned)divb[1] * (unsigned)divb[2];
  if (T == 0) T = 1;
  int target = 64;
  while (target < VX_SMALL_NB && target < n) target <<= 1;
  int shift = 0;
  while (((T - 1) >> shift) >= (unsigned)target) ++shift;
  const int nb = (int)((T - 1) >> shift) + 1;
  for (int b = tid; b < nb; b += VX_SB) s_cnt[b] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += VX_SB) {
    const float4 p = J.in[i];
    const int i0 = (int)(floorf(p.x * inv) - (float)minb[0]);
    const int i1 = (int)(floorf(p.y * inv) - (float)minb[1]);
    const int i2 = (int)(floorf(p.z * inv) - (float)minb[2]);
    const unsigned key = (unsigned)(i0 + i1 * mul1 + i2 * mul2);
    s_key[i] = key;
    atomicAdd(&s_cnt[min(key >> shift, (unsigned)(nb - 1))], 1);
  }
  __syncthreads();
  // exclusive scan of the histogram in place
  auto block_scan = [&](auto get, auto put, int cnt) -> int {
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int b0 = 0; b0 < cnt; b0 += VX_SB) {
      const int b = b0 + tid;
      const int v = b < cnt ? get(b) : 0;
      int incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      if (lane == 63) s_scan[wave] = incl;
      __syncthreads();
      int woff = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < VX_SB / 64; ++w) { if (w < wave) woff += s_scan[w]; tot += s_scan[w]; }
      const int run = s_run;
      if (b < cnt) put(b, run + woff + incl - v);
      __syncthreads();
      if (tid == 0) s_run = run + tot;
      __syncthreads();
    }
    return s_run;
  };
  block_scan([&](int b) { return s_cnt[b]; }, [&](int b, int v) { s_cnt[b] = v; }, nb);
  // scatter: s_cnt[b] turns from bucket start into bucket end
  for (int i = tid; i < n; i += VX_SB) {
    const int pos = atomicAdd(&s_cnt[min(s_key[i] >> shift, (unsigned)(nb - 1))], 1);
    s_idx[pos] = (unsigned short)i;
  }
  __syncthreads();
  // per-bucket rank sort by (voxel id, position) + voxel count.  thread per bucket; buckets > VX_TINY by the wave
  for (int b0 = 0; b0 < nb; b0 += VX_SB) {
    const int b = b0 + tid;
    const int bs = b < nb ? (b == 0 ? 0 : s_cnt[b - 1]) : 0;
    const int m = b < nb ? s_cnt[b] - bs : 0;
    if (m > 0 && m <= VX_TINY) {
      unsigned short e[VX_TINY];
#pragma unroll
      for (int i = 0; i < VX_TINY; ++i) e[i] = i < m ? s_idx[bs + i] : (unsigned short)0;
      int heads = 0;
#pragma unroll
      for (int i = 0; i < VX_TINY; ++i) {
        if (i < m) {
          int rank = 0;
          bool head = true;
#pragma unroll
          for (int j = 0; j < VX_TINY; ++j) if (j < m && j != i) { const bool lt = vx_less(s_key, e[j], e[i]); rank += lt; if (lt && s_key[e[j]] == s_key[e[i]]) head = false; }
          s_idx[bs + rank] = e[i];
          heads += head;
        }
      }
      s_vox[b] = (unsigned short)heads;
    } else if (b < nb && m == 0) {
      s_vox[b] = 0;
    }
    unsigned long long bigger = __ballot(m > VX_TINY);
    while (bigger) {
      const int src_lane = __ffsll((long long)bigger) - 1;
      bigger &= bigger - 1;
      const int bb = __shfl(b, src_lane, 64), mm = __shfl(m, src_lane, 64), bbs = __shfl(bs, src_lane, 64);
      int heads = 0;
      // ranks first (all reads), then the writes: in place is safe because the wave runs in lock-step
      unsigned short mine[8];  // up to 512 elements per bucket through registers; beyond: serial fallback below
      int myrank[8];
      if (mm <= 512) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int t = lane + 64 * q;
          mine[q] = 0; myrank[q] = -1;
          if (t < mm) {
            const unsigned short e = s_idx[bbs + t];
            int rank = 0;
            bool head = true;
            for (int j = 0; j < mm; ++j) { const unsigned short o = s_idx[bbs + j]; if (o != e) { const bool lt = vx_less(s_key, o, e); rank += lt; if (lt && s_key[o] == s_key[e]) head = false; } }
            mine[q] = e; myrank[q] = rank; heads += head;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 8; ++q) if (myrank[q] >= 0) s_idx[bbs + myrank[q]] = mine[q];
      } else if (lane == 0) {
        // very large bucket (never seen on scan clouds): insertion sort by one lane
        for (int i = 1; i < mm; ++i) {
          const unsigned short e = s_idx[bbs + i];
          int j = i - 1;
          while (j >= 0 && vx_less(s_key, e, s_idx[bbs + j])) { s_idx[bbs + j + 1] = s_idx[bbs + j]; --j; }
          s_idx[bbs + j + 1] = e;
        }
        for (int i = 0; i < mm; ++i) heads += (i == 0 || s_key[s_idx[bbs + i]] != s_key[s_idx[bbs + i - 1]]);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) heads += __shfl_xor(heads, o, 64);
      if (lane == 0) s_vox[bb] = (unsigned short)heads;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  __syncthreads();
  const int nvox = block_scan([&](int b) { return (int)s_vox[b]; }, [&](int b, int v) { s_vox[b] = (unsigned short)v; }, nb);
  // centroids: one thread per bucket walks its sorted points; f32 sums in sorted (= original) order
  for (int b = tid; b < nb; b += VX_SB) {
    const int bs = b == 0 ? 0 : s_cnt[b - 1], m = s_cnt[b] - bs;
    if (m == 0) continue;
    int rank = s_vox[b];
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    int c = 0;
    unsigned cur = 0;
    for (int k = 0; k < m; ++k) {
      const unsigned short e = s_idx[bs + k];
      const unsigned vid = s_key[e];
      if (c > 0 && vid != cur) {
        const float fn = (float)c;
        if (rank < J.cap) J.out[rank] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
        ++rank; sx = sy = sz = si = 0.f; c = 0;
      }
      const float4 p = J.in[e];
      sx += p.x; sy += p.y; sz += p.z; si += p.w;
      ++c; cur = vid;
    }
    const float fn = (float)c;
    if (rank < J.cap) J.out[rank] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
  }
  if (tid == 0) *J.n_out = nvox;
}

// ---- host ------------------------------------------------------------------------
#define VX_SMALL_LDS (6 * VX_SMALL_MAX + 6 * VX_SMALL_NB)
int vox_create(VoxCtx* V, const VoxJob* jobs, int njobs, std::string* err) {
  std::memset(V, 0, sizeof(*V));
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(vox_small), hipFuncAttributeMaxDynamicSharedMemorySize, VX_SMALL_LDS) != hipSuccess) { *err = "vox_create: hipFuncSetAttribute"; return -2; }
  std::vector<VoxJob> h(jobs, jobs + njobs);
  size_t total = 0;
  int max_cap = 0;
  size_t nbtot = 0;
  for (auto& j : h) {
    j.off = (int)total; total += (size_t)j.cap; max_cap = j.cap > max_cap ? j.cap : max_cap;
    int nbc = 64;
    while (nbc < 65536 && nbc < j.cap) nbc <<= 1;   // ~1 point per bucket at capacity, 64 .. 65536 buckets
    j.nbcap = nbc; j.boff0 = (int)nbtot; nbtot += (size_t)nbc;
  }
  if (total > 0x7fffffffull) { *err = "vox_create: scratch exceeds 2^31 elements"; return -3; }
  V->njobs = njobs; V->max_cap = max_cap; V->total = (unsigned)total;
  V->gx = 2048 / (njobs > 0 ? njobs : 1);
  if (V->gx < 4) V->gx = 4;
  if (V->gx > 64) V->gx = 64;
  hipError_t e = hipSuccess;
  auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) { e = hipMalloc(p, bytes ? bytes : 16); if (e == hipSuccess) e = hipMemset(*p, 0, bytes ? bytes : 16); } };
  A((void**)&V->jobs, sizeof(VoxJob) * njobs);
  A((void**)&V->bbox, (size_t)njobs * 8 * 4); A((void**)&V->geom, (size_t)njobs * VX_GEOM * 4);
  A((void**)&V->keys, total * 4); A((void**)&V->pairs_a, total * 8); A((void**)&V->pairs_b, total * 8);
  A((void**)&V->bcnt, nbtot * 4); A((void**)&V->boff, nbtot * 4); A((void**)&V->bcur, nbtot * 4);
  A((void**)&V->bvox, nbtot * 4); A((void**)&V->voff, nbtot * 4);
  A((void**)&V->pt_items, (size_t)(njobs + 1) * 4); A((void**)&V->bk_items, (size_t)(njobs + 1) * 4);
  if (e == hipSuccess) e = hipMemcpy(V->jobs, h.data(), sizeof(VoxJob) * njobs, hipMemcpyHostToDevice);
  if (e != hipSuccess) { *err = std::string("vox_create: ") + hipGetErrorString(e); return -2; }
  return 0;
}

void vox_destroy(VoxCtx* V) {
  void* ps[] = {V->jobs, V->bbox, V->geom, V->keys, V->pairs_a, V->pairs_b, V->bcnt, V->boff, V->bcur, V->bvox, V->voff, V->pt_items, V->bk_items};
  for (void* p : ps) if (p) (void)hipFree(p);
  std::memset(V, 0, sizeof(*V));
}

int vox_run(const VoxCtx& V, hipStream_t st, std::string* err) {
  (void)err;
  if (V.njobs == 0) return 0;
  { ProfScope ms_("memset_bbox", st); (void)hipMemsetAsync(V.bbox, 0xFF, (size_t)V.njobs * 8 * 4, st); }
  const dim3 pool(VX_POOL), blk(VB), jobs1((V.njobs + VB - 1) / VB), perjob(V.njobs);
  ALEGO_LAUNCH(vox_small, perjob, dim3(VX_SB), (size_t)VX_SMALL_LDS, st, V);
  ALEGO_LAUNCH(vox_plan, dim3(1), blk, 0, st, V, 0);
  ALEGO_LAUNCH(vox_bbox, pool, blk, 0, st, V);
  ALEGO_LAUNCH(vox_geom, jobs1, blk, 0, st, V);
  ALEGO_LAUNCH(vox_keys, pool, blk, 0, st, V);
  ALEGO_LAUNCH(vox_bscan, perjob, blk, 0, st, V);
  ALEGO_LAUNCH(vox_plan, dim3(1), blk, 0, st, V, 1);
  ALEGO_LAUNCH(vox_bscatter, pool, blk, 0, st, V);
  ALEGO_LAUNCH(vox_bsort, dim3(VX_BPOOL), dim3(64), 0, st, V);
  ALEGO_LAUNCH(vox_vscan, perjob, blk, 0, st, V);
  ALEGO_LAUNCH(vox_bcentroid, dim3(VX_BPOOL), dim3(64), 0, st, V);
  return 0;
}
