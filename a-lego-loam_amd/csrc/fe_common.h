// fe_common.h — curvature + occlusion marks of a chunk of the segmented cloud (src/laserOdometry.cpp:122-159): the body of fe_curv.
#ifndef ALEGO_FE_COMMON_H_
#define ALEGO_FE_COMMON_H_
#include "dev_common.h"

#define FE_HALO 6

// points [t0, t0 + CW) of slot `slot` by a workgroup of BLOCK threads; s_r / s_c / s_f hold CW + 2 FE_HALO entries.
//   cd[i]       f32 11-tap sum, strictly left to right (:124); curvature = (double)cd * cd (:125)
//   fe_flag[i]  bit 0 cloud_neighbor_picked_ after markOccludedPoints (:131-159), bit 1 ground, bit 2 curvature > edge_thres,
//               bit 3 curvature < surf_thres, bit 4 = 1 (cloud_label_ 0 + 1), bit 6 |col[i + 1] - col[i]| > suppress_col_diff
//   picked0[i]  bit 0 alone, for the single-scan entry points / tests
// The caller synchronises before s_r / s_c / s_f are reused.
template <int BLOCK, int CW>
DEV_INLINE void fe_curv_chunk(const DevCtx& d, int slot, int M, int t0, float* s_r, int* s_c, uint8_t* s_f) {
  const size_t base = (size_t)slot * d.N;
  const float* rng = d.seg_range + base;
  const int* colv = d.seg_col + base;
  const alego_params& P = d.P;
#pragma unroll 4
  for (int j = threadIdx.x; j < CW + 2 * FE_HALO; j += BLOCK) {
    const int i = t0 - FE_HALO + j;
    const bool in = i >= 0 && i < M;
    s_r[j] = in ? rng[i] : 0.f;
    s_c[j] = in ? colv[i] : 0;
  }
  __syncthreads();
  // per-point conditions of markOccludedPoints
  for (int j = threadIdx.x + 1; j < CW + 2 * FE_HALO - 1; j += BLOCK) {
    const int i = t0 - FE_HALO + j;
    uint8_t f = 0;
    if (i >= 5 && i < M - 5) {
      const float r0 = s_r[j], r1 = s_r[j + 1], rm = s_r[j - 1];
      int cdiff = s_c[j] - s_c[j + 1];
      cdiff = cdiff < 0 ? -cdiff : cdiff;
      bool c1, c2;
      double diff1, diff2;
      if (P.occl_f32) {  // LO.cpp:203-204
        c1 = (double)(r0 - r1) > P.occl_depth; c2 = (double)(r1 - r0) > P.occl_depth;
        diff1 = (double)fabsf(rm - r0); diff2 = (double)fabsf(r1 - r0);
      } else {           // laserOdometry.cpp:134-135
        const double d1 = (double)r0, d2 = (double)r1;
        c1 = d1 - d2 > P.occl_depth; c2 = d2 - d1 > P.occl_depth;
        diff1 = fabs((double)rm - d1); diff2 = fabs(d2 - d1);
      }
      const bool near = cdiff < P.occl_col_diff;
      const bool A = near && c1;            // marks i-5..i and skips the rest (:142-144)
      const bool B = near && !c1 && c2;     // marks i+1..i+5 (:148)
      const bool C = !A && diff1 > P.parallel_ratio * (double)r0 && diff2 > P.parallel_ratio * (double)r0;  // (:154-157)
      f = (uint8_t)((A ? 1 : 0) | (B ? 2 : 0) | (C ? 4 : 0));
    }
    s_f[j] = f;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < CW / BLOCK; ++u) {
    const int i = t0 + u * BLOCK + threadIdx.x;
    if (i >= M) break;
    const int j = u * BLOCK + threadIdx.x + FE_HALO;
    float cdv = 0.f;
    if (i >= 5 && i < M - 5) {
      // strictly left-to-right f32 sum (:124); built with -ffp-contract=off
      cdv = s_r[j - 5] + s_r[j - 4] + s_r[j - 3] + s_r[j - 2] + s_r[j - 1] - s_r[j] * 10 + s_r[j + 1] + s_r[j + 2] + s_r[j + 3] + s_r[j + 4] + s_r[j + 5];
    }
    uint8_t pk = (s_f[j] & 4) ? 1 : 0;
#pragma unroll
    for (int l = 0; l <= 5; ++l) pk |= (s_f[j + l] & 1);       // A(i'), i' in [i, i+5]
#pragma unroll
    for (int l = 1; l <= 5; ++l) pk |= (s_f[j - l] & 2) >> 1;  // B(i'), i' in [i-5, i-1]
    const double ad = (double)fabsf(cdv), curv = ad * ad;        // (double)diff_range * diff_range, exact (:125)
    int dc = i + 1 < M ? s_c[j + 1] - s_c[j] : 0;
    dc = dc < 0 ? -dc : dc;
    d.cd[base + i] = cdv;
    d.fe_flag[base + i] = (uint8_t)(pk | (d.seg_ground[base + i] ? 2 : 0) | (curv > P.edge_thres ? 4 : 0) | (curv < P.surf_thres ? 8 : 0) | (1 << 4) |
                                    (dc > P.suppress_col_diff ? 64 : 0));
    if (d.n_launch == 1) d.picked0[base + i] = pk;
  }
}
#endif
