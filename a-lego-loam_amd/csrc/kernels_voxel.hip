// kernels_voxel.hip — batched pcl::VoxelGrid<PointXYZI>::applyFilter (see voxel.h).
// Replaces the VoxelGrid calls of laserMapping.cpp:316-319 (map, every mapping frame in the
// reference; here only when the key-frame set changed) and :329-342 (current scan).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <vector>

#include "voxel.h"
#include "prof.h"

#define VB 256

__device__ __forceinline__ unsigned vx_enc(float f) {
  const unsigned b = (unsigned)__float_as_int(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float vx_dec(unsigned e) {
  const unsigned b = (e >> 31) ? (e ^ 0x80000000u) : ~e;
  return __int_as_float((int)b);
}
__device__ __forceinline__ bool vx_enabled(const VoxJob& J) { return J.enable == nullptr || *J.enable != 0; }

__global__ void __launch_bounds__(VB) vox_bbox(VoxCtx V) {
  const VoxJob J = V.jobs[blockIdx.y];
  if (!vx_enabled(J)) return;
  const int n = min(*J.n_in, J.cap);
  if ((int)blockIdx.x * VB >= n) return;
  float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  for (int i = blockIdx.x * VB + threadIdx.x; i < n; i += gridDim.x * VB) {
    const float4 p = J.in[i];
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
  __shared__ float s[6][VB / 64];
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
    unsigned* bb = V.bbox + blockIdx.y * 8;
    if (a < 3) atomicMin(&bb[a], vx_enc(v)); else atomicMin(&bb[4 + a - 3], ~vx_enc(v));
  }
}

__global__ void __launch_bounds__(VB) vox_keys(VoxCtx V) {
  const int job = blockIdx.y;
  const VoxJob J = V.jobs[job];
  if (!vx_enabled(J)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { V.seg_begin[job] = J.off; V.seg_end[job] = J.off; }
    return;
  }
  const int n = min(*J.n_in, J.cap);
  const float inv = 1.0f / J.leaf;
  const unsigned* bb = V.bbox + job * 8;
  int minb[3] = {0, 0, 0}, mul1 = 1, mul2 = 1, pass = 0;
  if (n > 0) {
    float mn[3], mx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = vx_dec(bb[a]); mx[a] = vx_dec(~bb[4 + a]); }
    const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    pass = (dx * dy * dz > 2147483647LL) ? 1 : 0;  // PCL: "leaf size too small" -> output = input
    int divb[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { minb[a] = (int)floorf(mn[a] * inv); divb[a] = (int)floorf(mx[a] * inv) - minb[a] + 1; }
    mul1 = divb[0]; mul2 = divb[0] * divb[1];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int* g = V.geom + job * 8;
    g[0] = minb[0]; g[1] = minb[1]; g[2] = minb[2]; g[3] = mul1; g[4] = mul2; g[5] = n; g[6] = pass;
    V.seg_begin[job] = J.off; V.seg_end[job] = J.off + n;
  }
  for (int i = blockIdx.x * VB + threadIdx.x; i < n; i += gridDim.x * VB) {
    unsigned key;
    if (pass) key = (unsigned)i;
    else {
      const float4 p = J.in[i];
      const int i0 = (int)(floorf(p.x * inv) - (float)minb[0]);
      const int i1 = (int)(floorf(p.y * inv) - (float)minb[1]);
      const int i2 = (int)(floorf(p.z * inv) - (float)minb[2]);
      key = (unsigned)(i0 + i1 * mul1 + i2 * mul2);
    }
    V.keys_a[J.off + i] = key;
    V.vals_a[J.off + i] = i;
  }
}

__global__ void __launch_bounds__(VB) vox_heads(VoxCtx V) {
  const int job = blockIdx.y;
  const VoxJob J = V.jobs[job];
  if (!vx_enabled(J)) return;
  const int n = V.geom[job * 8 + 5];
  const unsigned* keys = V.keys_b + J.off;
  __shared__ int s[VB / 64];
  for (int b = blockIdx.x; b * VB < n; b += gridDim.x) {
    const int i = b * VB + threadIdx.x;
    const bool head = i < n && (i == 0 || keys[i] != keys[i - 1]);
    const unsigned long long m = __ballot(head);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = (int)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < VB / 64; ++w) t += s[w];
      V.blk_cnt[job * V.blk_stride + b] = t;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(VB) vox_scan(VoxCtx V) {
  const int job = blockIdx.x;
  const VoxJob J = V.jobs[job];
  if (!vx_enabled(J)) return;
  const int n = V.geom[job * 8 + 5];
  const int nblk = (n + VB - 1) / VB;
  int* bc = V.blk_cnt + job * V.blk_stride;
  __shared__ int s[VB / 64];
  __shared__ int s_run;
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nblk; b0 += VB) {
    const int b = b0 + threadIdx.x;
    const int v = b < nblk ? bc[b] : 0;
    int incl = v;  // inclusive scan within the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((threadIdx.x & 63) >= o) incl += t; }
    if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < VB / 64; ++w) { if (w < (threadIdx.x >> 6)) woff += s[w]; tot += s[w]; }
    const int run = s_run;
    if (b < nblk) bc[b] = run + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) s_run = run + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *J.n_out = s_run;
}

__global__ void __launch_bounds__(VB) vox_centroid(VoxCtx V) {
  const int job = blockIdx.y;
  const VoxJob J = V.jobs[job];
  if (!vx_enabled(J)) return;
  const int n = V.geom[job * 8 + 5];
  const unsigned* keys = V.keys_b + J.off;
  const int* vals = V.vals_b + J.off;
  __shared__ int s[VB / 64];
  for (int b = blockIdx.x; b * VB < n; b += gridDim.x) {
    const int i = b * VB + threadIdx.x;
    const bool head = i < n && (i == 0 || keys[i] != keys[i - 1]);
    const unsigned long long m = __ballot(head);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = (int)__popcll(m);
    __syncthreads();
    if (head) {
      int woff = 0;
      for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += s[w];
      const int rank = V.blk_cnt[job * V.blk_stride + b] + woff + (int)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
      const unsigned vid = keys[i];
      float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;  // pcl::CentroidPoint: f32 accumulators
      int c = 0;
      for (int li = i; li < n && keys[li] == vid; ++li) {
        const float4 p = J.in[vals[li]];
        sx += p.x; sy += p.y; sz += p.z; si += p.w;
        ++c;
      }
      const float fn = (float)c;
      if (rank < J.cap) J.out[rank] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
    }
    __syncthreads();
  }
}

// ---- host ------------------------------------------------------------------------
int vox_create(VoxCtx* V, const VoxJob* jobs, int njobs, std::string* err) {
  std::memset(V, 0, sizeof(*V));
  std::vector<VoxJob> h(jobs, jobs + njobs);
  unsigned total = 0;
  int max_cap = 0;
  for (auto& j : h) { j.off = (int)total; total += (unsigned)j.cap; max_cap = j.cap > max_cap ? j.cap : max_cap; }
  V->njobs = njobs; V->max_cap = max_cap; V->total = total; V->gx = 32;
  V->blk_stride = (max_cap + VB - 1) / VB + 1;
  hipError_t e = hipSuccess;
  auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 16); };
  A((void**)&V->jobs, sizeof(VoxJob) * njobs);
  A((void**)&V->bbox, (size_t)njobs * 8 * 4); A((void**)&V->geom, (size_t)njobs * 8 * 4);
  A((void**)&V->keys_a, (size_t)total * 4); A((void**)&V->keys_b, (size_t)total * 4);
  A((void**)&V->vals_a, (size_t)total * 4); A((void**)&V->vals_b, (size_t)total * 4);
  A((void**)&V->seg_begin, (size_t)njobs * 4); A((void**)&V->seg_end, (size_t)njobs * 4);
  A((void**)&V->blk_cnt, (size_t)njobs * V->blk_stride * 4);
  if (e == hipSuccess) e = hipMemcpy(V->jobs, h.data(), sizeof(VoxJob) * njobs, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(V->geom, 0, (size_t)njobs * 8 * 4);
  size_t bytes = 0;
  if (e == hipSuccess)
    e = rocprim::segmented_radix_sort_pairs(nullptr, bytes, V->keys_a, V->keys_b, V->vals_a, V->vals_b, total, (unsigned)njobs,
                                            V->seg_begin, V->seg_end, 0, 32, (hipStream_t)0);
  V->sort_tmp_bytes = bytes;
  A(&V->sort_tmp, bytes);
  if (e != hipSuccess) { *err = std::string("vox_create: ") + hipGetErrorString(e); return -2; }
  return 0;
}

void vox_destroy(VoxCtx* V) {
  void* ps[] = {V->jobs, V->bbox, V->geom, V->keys_a, V->keys_b, V->vals_a, V->vals_b, V->seg_begin, V->seg_end, V->blk_cnt, V->sort_tmp};
  for (void* p : ps) if (p) (void)hipFree(p);
  std::memset(V, 0, sizeof(*V));
}

int vox_run(const VoxCtx& V, hipStream_t st, std::string* err) {
  if (V.njobs == 0) return 0;
  { ProfScope ms_("memset_bbox", st); (void)hipMemsetAsync(V.bbox, 0xFF, (size_t)V.njobs * 8 * 4, st); }
  int nb = (V.max_cap + VB - 1) / VB;
  if (nb > V.gx) nb = V.gx;  // grid-stride inside the kernels: bounded block count per job
  ALEGO_LAUNCH(vox_bbox, dim3(nb, V.njobs), dim3(VB), 0, st, V);
  ALEGO_LAUNCH(vox_keys, dim3(nb, V.njobs), dim3(VB), 0, st, V);
  size_t bytes = V.sort_tmp_bytes;
  hipError_t e;
  {
    ProfScope sort_scope_("rocprim_segmented_radix_sort", st);
    e = rocprim::segmented_radix_sort_pairs(V.sort_tmp, bytes, V.keys_a, V.keys_b, V.vals_a, V.vals_b, V.total, (unsigned)V.njobs,
                                            V.seg_begin, V.seg_end, 0, 32, st);
  }
  if (e != hipSuccess) { *err = std::string("segmented_radix_sort_pairs: ") + hipGetErrorString(e); return -2; }
  ALEGO_LAUNCH(vox_heads, dim3(nb, V.njobs), dim3(VB), 0, st, V);
  ALEGO_LAUNCH(vox_scan, dim3(V.njobs), dim3(VB), 0, st, V);
  ALEGO_LAUNCH(vox_centroid, dim3(nb, V.njobs), dim3(VB), 0, st, V);
  return 0;
}
