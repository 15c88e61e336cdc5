// kernels_voxel.hip — batched pcl::VoxelGrid<PointXYZI>::applyFilter (see voxel.h).
// Replaces the VoxelGrid calls of laserMapping.cpp:316-319 (map; the reference redoes it every mapping
// frame, here only when the key-frame set changed) and :329-342 (current scan).
//
// One workgroup per job (= one VoxelGrid::filter call), two kernels per round:
//   vox_small  jobs of <= 8192 points (the current-scan clouds): every intermediate in LDS
//   vox_big    larger jobs (the 45-75 k point local maps): a stable LSD radix sort of (voxel id, position) that
//              streams through HBM/L2, run by the 16 wavefronts of one workgroup
// History: round 1 first used rocprim::segmented_radix_sort_pairs (one workgroup per segment, ~1.2 ms), then a
// multi-kernel two-level bucket sort spread over the whole chip (9 launches, ~2 ms per mapping frame at 256 streams
// because a 50-key-frame map has ~17 points per voxel and every bucket needed a rank-by-counting sort).  The stable
// radix sort needs no in-bucket sort at all, and since streams run concurrently (stream groups) one CU per job is
// the right granularity: the other CUs are busy with other streams.
#include <algorithm>
#include <cstring>

#include <hip/hip_runtime.h>

#include <vector>

#include "voxel.h"
#include "guard_alloc.h"
#include "prof.h"

typedef unsigned long long u64;

__device__ __forceinline__ unsigned vx_enc(float f) {
  const unsigned b = (unsigned)__float_as_int(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ bool vx_enabled(const VoxJob& J) { return J.enable == nullptr || *J.enable != 0; }

#ifndef VX_SMALL_MAX
#define VX_SMALL_MAX 8192      // jobs up to this many points are done by vox_small
#endif

// ---------------------------------------------------------------------------------------------------------
// vox_big: one 1024-thread workgroup per job.
//   1 getMinMax3D, grid geometry (identical arithmetic to vox_small / PCL)
//   2 voxel id per point -> keys[]; every wavefront owns a contiguous segment of the cloud and histograms the
//     first radix digit of its segment in its own LDS row
//   3 P = ceil(bits/9) stable LSD passes over (voxel id << 32 | position): digit-major / wave-minor exclusive
//     scan of the 16 x 2^D histogram (D <= 9), then every wavefront streams its segment in order; within a round of 64
//     the rank among equal digits comes from D ballots (lanes are in index order), so the scatter is stable and
//     the final order is (voxel id, original position) without any comparison sort.  The histogram of the next
//     digit is accumulated while scattering (the destination index tells the owning wavefront).
//   4 voxel heads -> output ranks (ascending voxel id), list of run starts
//   5 one thread per voxel: f32 sums in sorted (= original) order, divided by the count (pcl::CentroidPoint)
#ifndef VG_T
#define VG_T 512   // = VX_SB: large jobs are taken by the same workgroups as the small ones (one launch per round)
#endif
#define VG_W (VG_T / 64)
#define VG_DMAX 9               // radix digit width: 2 x 16 x 512 counters = 64 KB of LDS
#define VG_ND (1 << VG_DMAX)
#define VG_U 4                 // independent loads kept in flight per lane in the streaming loops

__device__ __forceinline__ int vg_div(int pos, int seglen, double inv_seglen) {  // floor(pos / seglen) for any int32 pair
  int q = (int)((double)pos * inv_seglen);
  if (q * seglen > pos) --q;
  if ((q + 1) * seglen <= pos) ++q;
  return q;
}

#ifdef ALEGO_TIMING
__device__ long long vg_times[16];
#define VG_TICK(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) vg_times[k] = wall_clock64(); } while (0)
#else
#define VG_TICK(k)
#endif
// LDS (carved from the launch's dynamic buffer, which vox_small_job uses for its own arrays): 2 x VG_W x VG_ND counters (32 KB),
// VG_ND digit totals, a few per-wavefront scalars
#define VG_LDS_BYTES ((2 * VG_W * VG_ND + VG_ND + 6 * VG_W + VG_W + 2 * (VG_W + 1)) * 4)
__device__ void vox_big_job(const VoxCtx& V, int job, unsigned char* smem) {
  const VoxJob J = V.jobs[job];
  const int n = min(*J.n_in, J.cap);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  typedef int cnt_t[VG_W][VG_ND];
  cnt_t* s_cnt = reinterpret_cast<cnt_t*>(smem);                                  // [2][VG_W][VG_ND]
  int* s_tot = reinterpret_cast<int*>(smem) + 2 * VG_W * VG_ND;                   // [VG_ND]
  typedef float red_t[VG_W];
  red_t* s_red = reinterpret_cast<red_t*>(s_tot + VG_ND);                         // [6][VG_W]
  int* s_w = reinterpret_cast<int*>(s_red + 6);                                   // [VG_W]
  int* s_wbase = s_w + VG_W;                                                      // [VG_W + 1]
  int* s_nextfirst = s_wbase + VG_W + 1;                                          // [VG_W + 1]
  const float inv = 1.0f / J.leaf;
  VG_TICK(0);
  // ---- 1. getMinMax3D
  float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  // one workgroup streams the whole cloud: keep VG_U loads per lane in flight (a single CU is latency bound otherwise)
  for (int i0 = tid; i0 < n; i0 += VG_T * VG_U) {
    float4 p[VG_U];
#pragma unroll
    for (int k = 0; k < VG_U; ++k) p[k] = J.in[min(i0 + k * VG_T, n - 1)];
#pragma unroll
    for (int k = 0; k < VG_U; ++k) {
      mn[0] = fminf(mn[0], p[k].x); mn[1] = fminf(mn[1], p[k].y); mn[2] = fminf(mn[2], p[k].z);
      mx[0] = fmaxf(mx[0], p[k].x); mx[1] = fmaxf(mx[1], p[k].y); mx[2] = fmaxf(mx[2], p[k].z);
    }
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
    for (int w = 1; w < VG_W; ++w) { mn[a] = fminf(mn[a], s_red[a][w]); mx[a] = fmaxf(mx[a], s_red[3 + a][w]); }
  }
  if (tid < 3) {   // (selects, not mn[tid]: a dynamically indexed local array lives in scratch memory)
    unsigned* bb = V.bbox + job * 8;
    bb[tid] = vx_enc(tid == 0 ? mn[0] : tid == 1 ? mn[1] : mn[2]);
    bb[4 + tid] = ~vx_enc(tid == 0 ? mx[0] : tid == 1 ? mx[1] : mx[2]);
  }
  const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
  const int sel = (J.mode == 1 && J.out_sel) ? *J.out_sel : 0;
  float4* const outp = J.out + (size_t)sel * (J.mode == 1 ? J.out_stride : 0);
  if (J.mode == 1 && tid < 6 && J.box_out) J.box_out[(size_t)sel * 8 + (tid < 3 ? tid : tid + 1)] = tid == 0 ? mn[0] : tid == 1 ? mn[1] : tid == 2 ? mn[2] : tid == 3 ? mx[0] : tid == 4 ? mx[1] : mx[2];
  if (dx * dy * dz > 2147483647LL) {  // PCL: "leaf size too small" -> output = input
    // (mode 1: a cloud that cannot be ordered by a 31-bit voxel id is reported as a capacity error)
    for (int i = tid; i < min(n, J.out_cap); i += VG_T) outp[i] = J.in[i];
    if (tid == 0) { *J.n_out = min(n, J.out_cap); if ((n > J.out_cap || J.mode == 1) && J.overflow) *J.overflow = J.mode == 1 ? 3 : 1;
                    if (J.mode == 1 && J.n_sel_out) J.n_sel_out[(size_t)sel * J.n_sel_stride] = min(n, J.out_cap); }
    return;
  }
  int minb[3], divb[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { minb[a] = (int)floorf(mn[a] * inv); divb[a] = (int)floorf(mx[a] * inv) - minb[a] + 1; }
  const int mul1 = divb[0], mul2 = divb[0] * divb[1];
  unsigned T = (unsigned)divb[0] * (unsigned)divb[1] * (unsigned)divb[2];
  if (T < 2) T = 2;
  const int bits = 32 - __clz((int)(T - 1));
  const int P = (bits + VG_DMAX - 1) / VG_DMAX, D = (bits + P - 1) / P, nd = 1 << D;
  const unsigned dmask = (unsigned)(nd - 1);
  // every wavefront owns [seg0, seg1): whole rounds of 64 so that lanes are in index order
  const int seglen = ((n + VG_T - 1) / VG_T) * 64;
  const double inv_seglen = 1.0 / (double)seglen;
  const int seg0 = min(n, wave * seglen), seg1 = min(n, seg0 + seglen);
  unsigned* keys = V.keys + J.off;
  u64* bufA = V.pairs_a + J.off;
  u64* bufB = V.pairs_b + J.off;
  VG_TICK(1);
  // ---- 2. voxel ids + first-digit histogram of the own segment
  for (int q = lane; q < VG_ND; q += 64) { s_cnt[0][wave][q] = 0; }
  for (int r0 = seg0 + lane; r0 < seg1; r0 += 64 * VG_U) {
    float4 pt[VG_U];
#pragma unroll
    for (int k = 0; k < VG_U; ++k) pt[k] = J.in[min(r0 + 64 * k, n - 1)];
#pragma unroll
    for (int k = 0; k < VG_U; ++k) {
      const int i = r0 + 64 * k;
      if (i < seg1) {
        const int i0 = (int)(floorf(pt[k].x * inv) - (float)minb[0]);
        const int i1 = (int)(floorf(pt[k].y * inv) - (float)minb[1]);
        const int i2 = (int)(floorf(pt[k].z * inv) - (float)minb[2]);
        const unsigned key = (unsigned)(i0 + i1 * mul1 + i2 * mul2);
        keys[i] = key;
        atomicAdd(&s_cnt[0][wave][key & dmask], 1);
      }
    }
  }
  // ---- 3. stable LSD radix passes
  VG_TICK(2);
  for (int p = 0; p < P; ++p) {
    const int cur = p & 1, nxt = cur ^ 1;
    const int sh = p * D, shn = (p + 1) * D;
    const u64* src = (p & 1) ? bufA : bufB;   // pass 0 reads keys[], writes A; pass 1 reads A writes B; ...
    u64* dst = (p & 1) ? bufB : bufA;
    __threadfence_block();
    __syncthreads();
    if (tid < nd) {  // exclusive prefix over the wavefronts of every digit, digit totals
      int run = 0;
#pragma unroll
      for (int w = 0; w < VG_W; ++w) { const int c = s_cnt[cur][w][tid]; s_cnt[cur][w][tid] = run; run += c; }
      s_tot[tid] = run;
    }
    for (int q = tid; q < VG_W * VG_ND; q += VG_T) (&s_cnt[nxt][0][0])[q] = 0;
    __syncthreads();
    if (wave == 0) {  // exclusive scan of the digit totals (VG_ND / 64 per lane)
      constexpr int PER = VG_ND / 64;
      int v[PER], sum = 0;
#pragma unroll
      for (int k = 0; k < PER; ++k) { const int dgt = lane * PER + k; v[k] = dgt < nd ? s_tot[dgt] : 0; sum += v[k]; }
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      int run = incl - sum;
#pragma unroll
      for (int k = 0; k < PER; ++k) { const int dgt = lane * PER + k; if (dgt < nd) s_tot[dgt] = run; run += v[k]; }
    }
    __syncthreads();
    int* my = s_cnt[cur][wave];
    VG_TICK(3 + 2 * p);
    for (int r0 = seg0; r0 < seg1; r0 += 256) {
      u64 e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // four rounds of loads in flight
        const int i = r0 + 64 * k + lane;
        e[k] = 0;
        if (i < seg1) e[k] = p == 0 ? (((u64)keys[i] << 32) | (unsigned)i) : src[i];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = r0 + 64 * k + lane;
        const bool valid = i < seg1;
        if (r0 + 64 * k >= seg1) break;
        const unsigned key = (unsigned)(e[k] >> 32);
        const unsigned dg = (key >> sh) & dmask;
        u64 m = __ballot(valid);
        for (int bit = 0; bit < D; ++bit) {
          const bool one = (dg >> bit) & 1u;
          const u64 bal = __ballot(one);
          m &= one ? bal : ~bal;
        }
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (valid) {
          const int pos = my[dg] + s_tot[dg] + rank;
          dst[pos] = e[k];
          if (p + 1 < P) atomicAdd(&s_cnt[nxt][vg_div(pos, seglen, inv_seglen)][(key >> shn) & dmask], 1);
        }
        // LDS operations of one wavefront execute in order: every lane has read my[dg] before the leaders add
        if (valid && rank == 0) my[dg] += (int)__popcll(m);
      }
    }
    VG_TICK(4 + 2 * p);
  }
  const u64* srt = (P & 1) ? bufA : bufB;
  __threadfence_block();
  __syncthreads();
  VG_TICK(11);
  if (J.mode == 1) {   // sort only: the points in sorted order
    for (int i = tid; i < min(n, J.out_cap); i += VG_T) outp[i] = J.in[(unsigned)srt[i]];
    if (tid == 0) { *J.n_out = min(n, J.out_cap); if (n > J.out_cap && J.overflow) *J.overflow = 1;
                    if (J.n_sel_out) J.n_sel_out[(size_t)sel * J.n_sel_stride] = min(n, J.out_cap); }
    __syncthreads();
    return;
  }
  // ---- 4. voxel heads: count per segment, prefix over the wavefronts, list of run starts (reuses keys[])
  // head(i) = voxel id differs from the predecessor; bit k of `hm` = lane's element of round k is a head
  auto head_rounds = [&](int r0, bool* head) {
    u64 cur[VG_U], prv[VG_U];
#pragma unroll
    for (int k = 0; k < VG_U; ++k) {
      const int i = r0 + 64 * k + lane;
      cur[k] = srt[min(i, n - 1)]; prv[k] = srt[max(min(i, n - 1) - 1, 0)];
    }
#pragma unroll
    for (int k = 0; k < VG_U; ++k) {
      const int i = r0 + 64 * k + lane;
      head[k] = i < seg1 && (i == 0 || (unsigned)(cur[k] >> 32) != (unsigned)(prv[k] >> 32));
    }
  };
  // One pass: every wavefront compacts the run starts of its own segment into the front of that segment of hl[] (a segment has
  // at most as many heads as elements), then the per-wavefront counts give every voxel its (wavefront, local rank).
  int* hl = reinterpret_cast<int*>(keys);
  int heads = 0;
  for (int r0 = seg0; r0 < seg1; r0 += 64 * VG_U) {
    bool head[VG_U];
    head_rounds(r0, head);
#pragma unroll
    for (int k = 0; k < VG_U; ++k) {
      const u64 hb = __ballot(head[k]);
      if (head[k]) hl[seg0 + heads + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hb, 0u))] = r0 + 64 * k + lane;
      heads += (int)__popcll(hb);
    }
  }
  if (lane == 0) s_w[wave] = heads;
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int w = 0; w < VG_W; ++w) { s_wbase[w] = run; run += s_w[w]; }
    s_wbase[VG_W] = run;
    int nf = n;   // start of the first run at or after wavefront w's segment (n behind the last one)
    s_nextfirst[VG_W] = n;
    for (int w = VG_W - 1; w >= 0; --w) { if (s_w[w] > 0) nf = hl[min(n, w * seglen)]; s_nextfirst[w] = nf; }
  }
  __syncthreads();
  const int nvox = s_wbase[VG_W];
  VG_TICK(12);
  // ---- 5. centroids
  for (int r = tid; r < min(nvox, J.out_cap); r += VG_T) {
    int w = 0;
#pragma unroll
    for (int q = 1; q < VG_W; ++q) w += (r >= s_wbase[q]) ? 1 : 0;   // wavefront whose segment holds the start of voxel r
    const int loc = r - s_wbase[w], ws0 = min(n, w * seglen);
    const int a = hl[ws0 + loc], bn = hl[ws0 + min(loc + 1, s_w[w] - 1)];
    const int b = loc + 1 < s_w[w] ? bn : s_nextfirst[w + 1];
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    for (int j = a; j < b; j += 8) {
      unsigned q[8];
      float4 pt[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) q[k] = (unsigned)srt[min(j + k, b - 1)];
#pragma unroll
      for (int k = 0; k < 8; ++k) pt[k] = J.in[q[k]];  // gathers in flight together
#pragma unroll
      for (int k = 0; k < 8; ++k) if (j + k < b) { sx += pt[k].x; sy += pt[k].y; sz += pt[k].z; si += pt[k].w; }
    }
    const float fn = (float)(b - a);
    J.out[r] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
  }
  if (tid == 0) { *J.n_out = min(nvox, J.out_cap); if (nvox > J.out_cap && J.overflow) *J.overflow = 1; }
  __syncthreads();
  VG_TICK(13);
}
#ifdef ALEGO_TIMING
extern "C" void alego_vg_times(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(vg_times), sizeof(long long) * 16); }
#endif

// ---------------------------------------------------------------------------------------------------------
// Small jobs (n <= VX_SMALL_MAX points: the VoxelGrid calls on the current scan, laserMapping.cpp:329-342) run the
// whole filter with every intermediate in LDS: bounding box, voxel ids, bucket histogram (LDS atomics), scan,
// scatter, per-bucket rank sort, voxel ranks, centroids.  LDS: 4 B key + 2 B index per point, 6 B per bucket.
#define VX_SB 512
#define VS_W (VX_SB / 64)
#define VS_DMAX 9          // radix digit width: 8 x 512 16-bit counters = 8 KB of LDS
#define VS_ND (1 << VS_DMAX)
__device__ __forceinline__ bool vx_less(const unsigned* key, unsigned a, unsigned b) { return key[a] < key[b] || (key[a] == key[b] && a < b); }

__device__ void vox_small_job(const VoxCtx& V, int job) {
  const VoxJob J = V.jobs[job];
  const int n = min(*J.n_in, J.cap);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  extern __shared__ __attribute__((aligned(16))) unsigned char vs_smem[];
  unsigned* s_key = reinterpret_cast<unsigned*>(vs_smem);                                              // voxel id per point [VX_SMALL_MAX]
  unsigned short* s_ia = reinterpret_cast<unsigned short*>(vs_smem + 4 * VX_SMALL_MAX);                // point order, ping  [VX_SMALL_MAX]
  unsigned short* s_ib = reinterpret_cast<unsigned short*>(vs_smem + 6 * VX_SMALL_MAX);                // point order, pong  [VX_SMALL_MAX]
  unsigned short* s_cnt = reinterpret_cast<unsigned short*>(vs_smem + 8 * VX_SMALL_MAX);               // [VS_W][VS_ND] digit counters per wavefront
  __shared__ float s_red[6][VX_SB / 64];
  __shared__ int s_scan[VX_SB / 64];
  __shared__ int s_tot[VS_ND];
  const int sel = (J.mode == 1 && J.out_sel) ? *J.out_sel : 0;
  float4* const outp = J.out + (size_t)sel * (J.mode == 1 ? J.out_stride : 0);
  if (n == 0) { if (tid == 0) { *J.n_out = 0; if (J.mode == 1 && J.n_sel_out) J.n_sel_out[(size_t)sel * J.n_sel_stride] = 0; } return; }
  const float inv = 1.0f / J.leaf;
  VG_TICK(0);
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
  if (tid < 3) {   // (selects, not mn[tid]: a dynamically indexed local array lives in scratch memory)
    unsigned* bb = V.bbox + job * 8;
    bb[tid] = vx_enc(tid == 0 ? mn[0] : tid == 1 ? mn[1] : mn[2]);
    bb[4 + tid] = ~vx_enc(tid == 0 ? mx[0] : tid == 1 ? mx[1] : mx[2]);
  }
  const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
  if (J.mode == 1 && tid < 6 && J.box_out) J.box_out[(size_t)sel * 8 + (tid < 3 ? tid : tid + 1)] = tid == 0 ? mn[0] : tid == 1 ? mn[1] : tid == 2 ? mn[2] : tid == 3 ? mx[0] : tid == 4 ? mx[1] : mx[2];
  if (dx * dy * dz > 2147483647LL) {  // PCL: "leaf size too small" -> output = input
    for (int i = tid; i < min(n, J.out_cap); i += VX_SB) outp[i] = J.in[i];
    if (tid == 0) { *J.n_out = min(n, J.out_cap); if ((n > J.out_cap || J.mode == 1) && J.overflow) *J.overflow = J.mode == 1 ? 3 : 1;
                    if (J.mode == 1 && J.n_sel_out) J.n_sel_out[(size_t)sel * J.n_sel_stride] = min(n, J.out_cap); }
    return;
  }
  VG_TICK(1);
  int minb[3], divb[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { minb[a] = (int)floorf(mn[a] * inv); divb[a] = (int)floorf(mx[a] * inv) - minb[a] + 1; }
  const int mul1 = divb[0], mul2 = divb[0] * divb[1];
  unsigned T = (unsigned)divb[0] * (unsigned)divb[1] * (unsigned)divb[2];
  if (T == 0) T = 1;
  // ---- voxel ids
  for (int i = tid; i < n; i += VX_SB) {
    const float4 p = J.in[i];
    const int i0 = (int)(floorf(p.x * inv) - (float)minb[0]);
    const int i1 = (int)(floorf(p.y * inv) - (float)minb[1]);
    const int i2 = (int)(floorf(p.z * inv) - (float)minb[2]);
    s_key[i] = (unsigned)(i0 + i1 * mul1 + i2 * mul2);
  }
  VG_TICK(2);
  // ---- stable LSD radix sort of the point order by voxel id (as vox_big, but every array in LDS): every wavefront owns a
  // contiguous segment, counts its digits, and after the digit-major / wave-minor prefix scatters its segment in order; the
  // rank among equal digits inside a round of 64 comes from D ballots.  The order inside a voxel stays the original one.
  if (T < 2) T = 2;
  const int bits = 32 - __clz((int)(T - 1));
  const int P = (bits + VS_DMAX - 1) / VS_DMAX, D = (bits + P - 1) / P, nd = 1 << D;
  const unsigned dmask = (unsigned)(nd - 1);
  const int seglen = ((n + VX_SB - 1) / VX_SB) * 64;
  const int seg0 = min(n, wave * seglen), seg1 = min(n, seg0 + seglen);
  unsigned short* my = s_cnt + wave * VS_ND;
  for (int p = 0; p < P; ++p) {
    const unsigned short* src = (p & 1) ? s_ia : s_ib;   // pass 0 reads the identity order, writes ia; pass 1 ia -> ib; ...
    unsigned short* dst = (p & 1) ? s_ib : s_ia;
    const int sh = p * D;
    __syncthreads();
    for (int q = lane; q < nd; q += 64) my[q] = 0;
    // two sweeps over the own segment: count, then (after the prefix) scatter
    for (int sweep = 0; sweep < 2; ++sweep) {
      if (sweep == 1) {
        __syncthreads();
        if (tid < nd) {  // exclusive prefix over the wavefronts of every digit, digit totals
          int run = 0;
#pragma unroll
          for (int w = 0; w < VS_W; ++w) { const int c = s_cnt[w * VS_ND + tid]; s_cnt[w * VS_ND + tid] = (unsigned short)run; run += c; }
          s_tot[tid] = run;
        }
        __syncthreads();
        if (wave == 0) {  // exclusive scan of the digit totals (VS_ND / 64 per lane)
          constexpr int PER = VS_ND / 64;
          int v[PER], sum = 0;
#pragma unroll
          for (int k = 0; k < PER; ++k) { const int dgt = lane * PER + k; v[k] = dgt < nd ? s_tot[dgt] : 0; sum += v[k]; }
          int incl = sum;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
          int run = incl - sum;
#pragma unroll
          for (int k = 0; k < PER; ++k) { const int dgt = lane * PER + k; if (dgt < nd) s_tot[dgt] = run; run += v[k]; }
        }
        __syncthreads();
      }
      for (int r0 = seg0; r0 < seg1; r0 += 64) {
        const int i = r0 + lane;
        const bool valid = i < seg1;
        const unsigned idx = valid ? (p == 0 ? (unsigned)i : (unsigned)src[i]) : 0u;
        const unsigned dg = valid ? (s_key[idx] >> sh) & dmask : 0u;
        unsigned long long m = __ballot(valid);
        for (int bit = 0; bit < D; ++bit) {
          const bool one = (dg >> bit) & 1u;
          const unsigned long long bal = __ballot(one);
          m &= one ? bal : ~bal;
        }
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (sweep == 1 && valid) dst[(int)my[dg] + s_tot[dg] + rank] = (unsigned short)idx;
        // LDS operations of one wavefront execute in order: every lane has read my[dg] before the leaders add
        if (valid && rank == 0) my[dg] = (unsigned short)(my[dg] + (int)__popcll(m));
      }
    }
  }
  __syncthreads();
  const unsigned short* srt = (P & 1) ? s_ia : s_ib;
  unsigned short* hl = (P & 1) ? s_ib : s_ia;   // the other buffer: list of voxel run starts
  VG_TICK(3);
  if (J.mode == 1) {   // sort only: the points in sorted order
    for (int i = tid; i < min(n, J.out_cap); i += VX_SB) outp[i] = J.in[srt[i]];
    if (tid == 0) { *J.n_out = min(n, J.out_cap); if (n > J.out_cap && J.overflow) *J.overflow = 1;
                    if (J.n_sel_out) J.n_sel_out[(size_t)sel * J.n_sel_stride] = min(n, J.out_cap); }
    __syncthreads();
    return;
  }
  // ---- voxel heads: count per segment, prefix over the wavefronts, list of run starts
  int heads = 0;
  for (int r0 = seg0; r0 < seg1; r0 += 64) {
    const int i = r0 + lane;
    const bool head = i < seg1 && (i == 0 || s_key[srt[i]] != s_key[srt[i - 1]]);
    heads += (int)__popcll(__ballot(head));
  }
  if (lane == 0) s_scan[wave] = heads;
  __syncthreads();
  int vbase = 0, nvox = 0;
#pragma unroll
  for (int w = 0; w < VX_SB / 64; ++w) { const int c = s_scan[w]; if (w < wave) vbase += c; nvox += c; }
  for (int r0 = seg0; r0 < seg1; r0 += 64) {
    const int i = r0 + lane;
    const bool head = i < seg1 && (i == 0 || s_key[srt[i]] != s_key[srt[i - 1]]);
    const unsigned long long hb = __ballot(head);
    if (head) hl[vbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hb, 0u))] = (unsigned short)i;
    vbase += (int)__popcll(hb);
  }
  __syncthreads();
  VG_TICK(4);
  VG_TICK(5);
  VG_TICK(6);
  // ---- centroids: one thread per voxel, f32 sums in sorted (= original) order, four gathers in flight
  for (int r = tid; r < min(nvox, J.out_cap); r += VX_SB) {
    const int a = hl[r], b = r + 1 < nvox ? (int)hl[r + 1] : n;
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    for (int j = a; j < b; j += 4) {
      float4 pt[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) pt[k] = J.in[srt[min(j + k, b - 1)]];
#pragma unroll
      for (int k = 0; k < 4; ++k) if (j + k < b) { sx += pt[k].x; sy += pt[k].y; sz += pt[k].z; si += pt[k].w; }
    }
    const float fn = (float)(b - a);
    J.out[r] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
  }
  if (tid == 0) { *J.n_out = min(nvox, J.out_cap); if (nvox > J.out_cap && J.overflow) *J.overflow = 1; }
  __syncthreads();
  VG_TICK(7);
}
// One launch per round: persistent workgroups over the list of small jobs, then (the first grid_big of them) over the list of
// large ones.  (The large jobs used to have a kernel of their own with 1024-thread workgroups: its launch cost ~150 us of
// queueing under load even when the list was empty, which it is on every round once the maps come from the sorted key frames.)
__global__ void __launch_bounds__(VX_SB) vox_small(VoxCtx V) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vs_smem_k[];
  const int cnt = V.cnt[0];
  for (int j = blockIdx.x; j < cnt; j += gridDim.x) {
    vox_small_job(V, V.list_small[j]);
    __syncthreads();
  }
  const int nbig = V.cnt[1];
  if ((int)blockIdx.x < V.grid_big) {
    for (int j = blockIdx.x; j < nbig; j += min((int)gridDim.x, V.grid_big)) {
      vox_big_job(V, V.list_big[j], vs_smem_k);
      __syncthreads();   // LDS of the job just finished is reused by the next one
    }
  }
}

// One workgroup sorts the jobs of a round into the two work lists.  Disabled jobs keep their previous output, empty
// clouds are finished right here.  The lists let vox_small / vox_big run as a few persistent workgroups where little
// work is expected (a context's launch sizes are fixed by the host), instead of one 72 KB-LDS workgroup per job that
// only finds out on the CU that it has nothing to do.
__global__ void __launch_bounds__(256) vox_plan(VoxCtx V) {
  __shared__ int s_n[2];
  if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < V.njobs; j += 256) {
    const VoxJob J = V.jobs[j];
    if (!vx_enabled(J)) continue;
    const int n = min(*J.n_in, J.cap);
    if (n == 0) { *J.n_out = 0; if (J.mode == 1 && J.n_sel_out) J.n_sel_out[(size_t)(J.out_sel ? *J.out_sel : 0) * J.n_sel_stride] = 0; continue; }
    if (n <= VX_SMALL_MAX) V.list_small[atomicAdd(&s_n[0], 1)] = j;
    else V.list_big[atomicAdd(&s_n[1], 1)] = j;
  }
  __syncthreads();
  if (threadIdx.x < 2) V.cnt[threadIdx.x] = s_n[threadIdx.x];
}

// ---- host ------------------------------------------------------------------------
#define VX_SMALL_LDS (8 * VX_SMALL_MAX + 2 * VS_W * VS_ND)
int vox_create(VoxCtx* V, const VoxJob* jobs, int njobs, std::string* err) {
  std::memset(V, 0, sizeof(*V));
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(vox_small), hipFuncAttributeMaxDynamicSharedMemorySize, VX_SMALL_LDS) != hipSuccess) { *err = "vox_create: hipFuncSetAttribute"; return -2; }
  std::vector<VoxJob> h(jobs, jobs + njobs);
  size_t total = 0;
  for (auto& j : h) {
    j.off = (int)total;
    if (j.cap > VX_SMALL_MAX) total += (size_t)j.cap;  // only jobs that can reach vox_big need sort scratch
  }
  if (total > 0x7fffffffull) { *err = "vox_create: scratch exceeds 2^31 elements"; return -3; }
  V->njobs = njobs; V->total = (unsigned)total;
  hipError_t e = hipSuccess;
  auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) { e = guard_malloc(p, bytes ? bytes : 16); if (e == hipSuccess) e = hipMemset(*p, 0, bytes ? bytes : 16); } };
  A((void**)&V->jobs, sizeof(VoxJob) * njobs);
  A((void**)&V->bbox, (size_t)njobs * 8 * 4);
  A((void**)&V->keys, total * 4); A((void**)&V->pairs_a, total * 8); A((void**)&V->pairs_b, total * 8);
  A((void**)&V->list_small, (size_t)njobs * 4); A((void**)&V->list_big, (size_t)njobs * 4); A((void**)&V->cnt, 8);
  V->grid_small = V->grid_big = njobs;
  if (e == hipSuccess) e = hipMemcpy(V->jobs, h.data(), sizeof(VoxJob) * njobs, hipMemcpyHostToDevice);
  if (e != hipSuccess) { *err = std::string("vox_create: ") + hipGetErrorString(e); return -2; }
  return 0;
}

void vox_destroy(VoxCtx* V) {
  void* ps[] = {V->jobs, V->bbox, V->keys, V->pairs_a, V->pairs_b, V->list_small, V->list_big, V->cnt};
  for (void* p : ps) if (p) (void)guard_free(p);
  std::memset(V, 0, sizeof(*V));
}

int vox_run(const VoxCtx& V, hipStream_t st, std::string* err) {
  (void)err;
  if (V.njobs == 0) return 0;
  ALEGO_LAUNCH(vox_plan, dim3(1), dim3(256), 0, st, V);
  static_assert(VG_T == VX_SB && VG_LDS_BYTES <= VX_SMALL_LDS, "large jobs run in vox_small's workgroups and LDS");
  ALEGO_LAUNCH(vox_small, dim3(std::max(V.grid_small, V.grid_big)), dim3(VX_SB), (size_t)VX_SMALL_LDS, st, V);
  return 0;
}
