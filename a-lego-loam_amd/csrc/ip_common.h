// ip_common.h — device pieces shared by the ImageProjection kernels (kernels_ip.hip: the multi-kernel path; kernels_ipf.hip: the
// one-workgroup-per-stream fused path): the cell of an input point, the ground test, the segmentation's edge predicate and the
// 16-bit LDS union-find.
#ifndef ALEGO_IP_COMMON_H_
#define ALEGO_IP_COMMON_H_
#include "dev_common.h"

// The cell of one input point (imageProjection.cpp:76-97; IP.cpp:91,142-172): row * H + col, or -1 when the point is rejected (row /
// column out of range, non-finite, near filter).  *valid_out = the point survives pcl::removeNaNFromPointCloud / removeClosedPointCloud,
// i.e. it counts as first / last point of the scan for the orientation block (:62-63).  Shared by ip_project and ip_fused.
DEV_INLINE int ip_point_cell(const DevCtx& d, const float4 p, bool* valid_out) {
  const alego_params& P = d.P;
  bool valid;
  int cell = -1;
    // pcl::removeNaNFromPointCloud only filters when the message is not dense (alego_params.input_is_dense); an unfiltered
    // non-finite point still counts as first / last point of the scan and is rejected by the row test (int(NaN) < 0)
    const bool finite = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
    valid = finite || P.input_is_dense != 0;
    if (valid && P.near_filter) {
      const float th = (float)P.near_thres;
      if (p.x * p.x + p.y * p.y + p.z * p.z < th * th) valid = false;  // IP.cpp:91
    }
    if (valid && finite) {
      // ---- fast path: cell from the boundary tables.  The reference rounds atan2f / hypotf to f32 before it divides by the
      // angular resolution, so its row / column is the cell of the TRUE angle unless that angle lies within ~7e-5 cells of a
      // boundary.  A polynomial estimate (good to 0.03 cells) picks a boundary, the exact offset to it comes from one
      // cross / dot product against the boundary's (cos, sin) (fp64 products, f32 quotient: error < 1e-6 cells), and a point
      // closer than 2.5e-4 columns / 1e-4 rows to any boundary takes the reference expressions below.  Every comparison is
      // written so that a NaN or an out-of-range estimate also lands there.
      int row = -1, col = -1;
      bool slow = true;
      {
        const float h2 = p.x * p.x + p.y * p.y, hf = __builtin_amdgcn_sqrtf(h2);   // (v_sqrt_f32 / v_rcp_f32: 1 ulp is plenty here)
        bool okr = (d.ip_fast & 1) && hf > 1e-3f && hf < 1e6f && fabsf(p.z) < 1e6f;
        const float t = p.z * __builtin_amdgcn_rcpf(hf), t2 = t * t;
        okr = okr && fabsf(t) < 0.6f;
        const float a = t * (0.99997726f + t2 * (-0.33262347f + t2 * (0.19354346f + t2 * (-0.11643287f + t2 * (0.05265332f + t2 * -0.01172120f)))));
        const float r0 = (a * 57.29577951f + (float)P.ang_bottom) * (float)d.inv_res_y + 0.5f;
        const float kf = floorf(r0);
        okr = okr && kf >= -3.0f && kf <= (float)(d.NS + 1);
        const int kk = okr ? (int)kf : 0;
        const double2 cs = d.ip_rowtab[kk + 3];
        const float cr = (float)((double)p.z * cs.x - (double)hf * cs.y), dt = (float)((double)hf * cs.x + (double)p.z * cs.y);
        const float q = cr * __builtin_amdgcn_rcpf(dt);
        const float frac = q * (1.0f - q * q * 0.33333333f) * (57.29577951f * (float)d.inv_res_y);
        const float fl = floorf(frac), dlo = frac - fl, dhi = 1.0f - dlo;
        const int rfl = kk + (int)fl;   // floor of the reference's r; (int)r truncates: r in (-1, 1) is row 0, so 0 is no boundary
        okr = okr && dt > 0.0f && frac > -1.0f && frac < 2.0f && (dlo >= 1e-4f || rfl == 0) && (dhi >= 1e-4f || rfl + 1 == 0);
        // columns
        const float ax = fabsf(p.x), ay = fabsf(p.y), mn = fminf(ax, ay), mx = fmaxf(ax, ay);
        bool okc = (d.ip_fast & 2) && mx > 1e-3f && mx < 1e6f;
        const float u = mn * __builtin_amdgcn_rcpf(mx), u2 = u * u;
        float b = u * (0.99997726f + u2 * (-0.33262347f + u2 * (0.19354346f + u2 * (-0.11643287f + u2 * (0.05265332f + u2 * -0.01172120f)))));
        b = ay > ax ? 1.57079633f - b : b;
        b = p.x < 0.0f ? 3.14159265f - b : b;
        b = p.y < 0.0f ? -b : b;
        const float cf = floorf((6.28318531f - b) * (57.29577951f * (float)d.inv_res_x)) - (float)d.ip_cmin;
        okc = okc && cf >= 0.0f && cf <= (float)(d.ip_ncb - 1);
        const int ci = okc ? (int)cf : 0;
        const double2 cc = d.ip_coltab[ci];
        const float crc = (float)(-(double)p.y * cc.x - (double)p.x * cc.y), dtc = (float)((double)p.x * cc.x - (double)p.y * cc.y);
        const float qc = crc * __builtin_amdgcn_rcpf(dtc);
        const float fc = qc * (1.0f - qc * qc * 0.33333333f) * (57.29577951f * (float)d.inv_res_x);
        const float flc = floorf(fc), dloc = fc - flc;
        okc = okc && dtc > 0.0f && fc > -1.0f && fc < 2.0f && dloc >= 2.5e-4f && 1.0f - dloc >= 2.5e-4f;
        if (okr && okc) {
          slow = false;
          row = rfl >= 0 ? rfl : (rfl == -1 ? 0 : -1);
          col = d.ip_cmin + ci + (int)flc;
          if (col >= d.H) col -= d.H;
        }
      }
      if (slow) {
      // imageProjection.cpp:79-80 (IP.cpp:142-172 for the RFANS table)
      // (a * 180.0) / M_PI and x / ang_res are IEEE fp64 divisions in the reference (~35 instructions each here).  The
      // integer part of the result only depends on the exact quotient when it lies within rounding error of an
      // integer: multiply by the reciprocals first and redo the reference expression only in that case.
      const double va = (double)d_atan2f(p.z, d_hypotf(p.x, p.y)) * 180.0;
      if (P.laser_type == ALEGO_LASER_UNIFORM) {
        const double r = (va * (1.0 / M_PI) + P.ang_bottom) * d.inv_res_y + 0.5;
        row = (int)r;
        if (fabs(r - rint(r)) < 1e-9 * (1.0 + fabs(r))) row = (int)((va / M_PI + P.ang_bottom) / P.ang_res_y + 0.5);
      } else {  // RFANS ring table (IP.cpp:142-172)
        const double vertical_ang = va / M_PI;
        if (vertical_ang > 4.5) row = (int)(13 + (vertical_ang - 5.) / 3 + 0.5);
        else if (vertical_ang > 0.5) row = (int)(11 + (vertical_ang - 1.0) / 2 + 0.5);
        else if (vertical_ang > -7.) row = (int)(10.5 + vertical_ang);
        else if (vertical_ang > -8.5) row = 3;
        else if (vertical_ang > -10.5) row = 2;
        else if (vertical_ang > -13.5) row = 1;
        else row = 0;
      }
      // :87-97
      const double ha = ((double)(-d_atan2f(p.y, p.x)) + 2 * M_PI) * 180.0;
      const double cfast = ha * (1.0 / M_PI) * d.inv_res_x;
      col = (int)cfast;
      if (fabs(cfast - rint(cfast)) < 1e-9 * (1.0 + fabs(cfast))) col = (int)((ha / M_PI) / P.ang_res_x);
      if (col >= d.H) col -= d.H;
      }
      if (row >= 0 && row < d.NS && col >= 0 && col < d.H) cell = col + row * d.H;
    }
  *valid_out = valid;
  return cell;
}

// Quick decision for ip_fused's first pass.  The row / column of a point is the cell of its TRUE elevation / azimuth unless that angle lies
// within the reference's own rounding (~3e-5 cells) of a cell boundary, so a cheap estimate of both angles whose error is bounded decides every
// point that is not NEAR a boundary; the others (a few per cent of a real scan) are deferred to ip_point_cell (table refinement or the
// reference expressions).  Error budget of the estimate: the odd degree-11 polynomial for atan on [0, 1] is within 1.8e-6 rad of atan in f32
// (tests/test_oracle.py::test_quick_projection_polynomial_bound evaluates it on 2M arguments), v_rcp / v_sqrt (1 ulp) and the quadrant fix-ups
// add < 1.5e-6 rad, the product with 180 / (pi res) two roundings of a value <= 1.5 H: 4e-6 rad and 4e-7 H cells in total are charged, twice
// over, in the margins below (at least 0.02 columns / 0.005 rows).  Nothing here has to be bit-exact, so FMA contraction is allowed.
// Returns true when decided: *cell_out = row * H + col, or -1 when the point has no cell (filtered, non-finite, outside the image).
// the uniform values of ip_point_quick, converted once (round 5: read from the kernel argument inside ip_fused_t's point loop they were 18 scalar loads and
// 8 v_cvt_f32_f64 per iteration of four points)
struct IpQuickConst {
  float th2;          // near_thres^2 as the reference squares it (f32), < 0: no near filter
  float ang_bottom, inv_res_y, col_scale;   // (float)P.ang_bottom, (float)(1 / ang_res_y), 57.29577951f * (float)(1 / ang_res_x)
  float two_h;        // (float)(2 H)
  int H, NS, dense, fast;
};
DEV_INLINE IpQuickConst ip_quick_const(const DevCtx& d) {
  IpQuickConst c;
  const float th = (float)d.P.near_thres;
  c.th2 = d.P.near_filter ? th * th : -1.0f;
  c.ang_bottom = (float)d.P.ang_bottom; c.inv_res_y = (float)d.inv_res_y; c.col_scale = 57.29577951f * (float)d.inv_res_x;
  c.two_h = (float)(2 * d.H);
  c.H = d.H; c.NS = d.NS; c.dense = d.P.input_is_dense != 0; c.fast = (d.ip_fast & 3) == 3;
  return c;
}
DEV_INLINE bool ip_point_quick(const IpQuickConst& c, const float4 p, float mr, float mc, bool* valid_out, int* cell_out) {
  const bool finite = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
  bool valid = finite || c.dense;
  if (valid && p.x * p.x + p.y * p.y + p.z * p.z < c.th2) valid = false;  // IP.cpp:91 (same expression as ip_point_cell; th2 < 0: filter off, NaN compares false)
  *valid_out = valid;
  *cell_out = -1;
  if (!(valid && finite)) return true;
  if (!c.fast) return false;
  bool ok;
  int rfl, col;
  {
#pragma clang fp contract(fast)
    const float h2 = p.x * p.x + p.y * p.y, hf = __builtin_amdgcn_sqrtf(h2);
    const float t = p.z * __builtin_amdgcn_rcpf(hf), t2 = t * t;
    ok = hf > 1e-3f && hf < 1e6f && fabsf(t) < 0.6f;
    const float a = t * (0.99997726f + t2 * (-0.33262347f + t2 * (0.19354346f + t2 * (-0.11643287f + t2 * (0.05265332f + t2 * -0.01172120f)))));
    const float r0 = (a * 57.29577951f + c.ang_bottom) * c.inv_res_y + 0.5f;
    const float rf = floorf(r0), fr = r0 - rf;
    ok = ok && fr >= mr && fr <= 1.0f - mr && rf >= -64.0f && rf <= 4096.0f;
    rfl = (int)rf;
    const float ax = fabsf(p.x), ay = fabsf(p.y), mn = fminf(ax, ay), mx = fmaxf(ax, ay);
    ok = ok && mx > 1e-3f && mx < 1e6f;
    const float u = mn * __builtin_amdgcn_rcpf(mx), u2 = u * u;
    float b = u * (0.99997726f + u2 * (-0.33262347f + u2 * (0.19354346f + u2 * (-0.11643287f + u2 * (0.05265332f + u2 * -0.01172120f)))));
    b = ay > ax ? 1.57079633f - b : b;
    b = p.x < 0.0f ? 3.14159265f - b : b;
    b = p.y < 0.0f ? -b : b;
    const float c0 = (6.28318531f - b) * c.col_scale;   // (-atan2f(y, x) + 2 pi) * 180 / pi / ang_res_x (:87-95)
    const float cf = floorf(c0), fc = c0 - cf;
    ok = ok && fc >= mc && fc <= 1.0f - mc && cf >= 0.0f && cf < c.two_h;
    col = (int)cf;
  }
  if (!ok) return false;   // (a NaN anywhere fails a comparison above)
  const int row = rfl >= 0 ? rfl : (rfl == -1 ? 0 : -1);   // (int)r truncates: r in (-1, 1) is row 0
  if (col >= c.H) col -= c.H;
  if (row >= 0 && row < c.NS && col >= 0 && col < c.H) *cell_out = col + row * c.H;
  return true;
}
DEV_INLINE bool ip_point_quick(const DevCtx& d, const float4 p, float mr, float mc, bool* valid_out, int* cell_out) {
  return ip_point_quick(ip_quick_const(d), p, mr, mc, valid_out, cell_out);
}
// the same decision without a branch (ip_fused_t's point loop evaluates four points per iteration: the early returns above were four nests of exec-mask
// bookkeeping): every expression is evaluated for every point — a non-finite or filtered point only produces values nobody looks at — and the three
// outcomes are selected at the end.  Returns "decided"; *cell_out as above.
DEV_INLINE bool ip_point_quick_bf(const IpQuickConst& c, const float4 p, float mr, float mc, bool* valid_out, int* cell_out) {
  const bool finite = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
  const bool valid = (finite || c.dense) && !(p.x * p.x + p.y * p.y + p.z * p.z < c.th2);
  bool ok;
  int rfl, col;
  {
#pragma clang fp contract(fast)
    const float h2 = p.x * p.x + p.y * p.y, hf = __builtin_amdgcn_sqrtf(h2);
    const float t = p.z * __builtin_amdgcn_rcpf(hf), t2 = t * t;
    ok = hf > 1e-3f && hf < 1e6f && fabsf(t) < 0.6f;
    const float a = t * (0.99997726f + t2 * (-0.33262347f + t2 * (0.19354346f + t2 * (-0.11643287f + t2 * (0.05265332f + t2 * -0.01172120f)))));
    const float r0 = (a * 57.29577951f + c.ang_bottom) * c.inv_res_y + 0.5f;
    const float rf = floorf(r0), fr = r0 - rf;
    ok = ok && fr >= mr && fr <= 1.0f - mr && rf >= -64.0f && rf <= 4096.0f;
    rfl = (int)(ok ? rf : 0.0f);   // (a float -> int conversion of NaN / inf / out-of-range values is undefined: convert only what the checks above admitted)
    const float ax = fabsf(p.x), ay = fabsf(p.y), mn = fminf(ax, ay), mx = fmaxf(ax, ay);
    ok = ok && mx > 1e-3f && mx < 1e6f;
    const float u = mn * __builtin_amdgcn_rcpf(mx), u2 = u * u;
    float b = u * (0.99997726f + u2 * (-0.33262347f + u2 * (0.19354346f + u2 * (-0.11643287f + u2 * (0.05265332f + u2 * -0.01172120f)))));
    b = ay > ax ? 1.57079633f - b : b;
    b = p.x < 0.0f ? 3.14159265f - b : b;
    b = p.y < 0.0f ? -b : b;
    const float c0 = (6.28318531f - b) * c.col_scale;
    const float cf = floorf(c0), fc = c0 - cf;
    ok = ok && fc >= mc && fc <= 1.0f - mc && cf >= 0.0f && cf < c.two_h;
    col = (int)(ok ? cf : 0.0f);
  }
  const bool live = valid && finite;            // the point can have a cell at all
  const bool dec = live && c.fast && ok;        // ... and the estimate decides it
  const int row = rfl >= 0 ? rfl : (rfl == -1 ? 0 : -1);
  col = col >= c.H ? col - c.H : col;
  const bool in = row >= 0 && row < c.NS && col >= 0 && col < c.H;
  *valid_out = valid;
  *cell_out = (dec && in) ? col + row * c.H : -1;
  return !live || dec;
}
// the margins of ip_point_quick (cells) for this sensor
DEV_INLINE void ip_quick_margins(const DevCtx& d, float* mr, float* mc) {
  *mr = fmaxf(0.005f, 8e-6f * 57.29577951f * (float)d.inv_res_y + 1e-5f);
  *mc = fmaxf(0.02f, 8e-6f * 57.29577951f * (float)d.inv_res_x + 8e-7f * (float)d.H);
}

// ground test of two vertically adjacent cells (imageProjection.cpp:111-131): |deg(atan2(dz, hypot(dx, dy))) - mount| < thres
// <=> tan(lo) h < dz < tan(hi) h; decided by the two signed margins unless one of them is within 1e-9 (relative) of zero, where
// the reference expression decides.  fx, fy, fz = the f32 differences upper - lower.
// (the reference expressions themselves are real calls, not inlined: they are taken by a vanishing fraction of the cells, and inlined into
//  the unrolled per-row code of ip_fused / ip_front their fp64 atan2 / hypot bodies were 90 % of those kernels' instructions and the
//  source of their register spills)
__device__ __attribute__((noinline)) inline bool ip_is_ground_ref(double dx, double dy, double dz, double mount, double thres) {
  const double angle = (atan2(dz, hypot(dx, dy)) * 180.0) / M_PI;
  return fabs(angle - mount) < thres;
}
__device__ __attribute__((noinline)) inline bool edge_angle_ref(double y, double x, double theta) { return atan2(y, x) > theta; }
// (round 5: the two margins are taken in "signed square" space — f(x) = sign(x) x^2 is strictly increasing, so f(dz) - f(tan h) has the sign of dz - tan h —
//  which needs h^2 = dx^2 + dy^2 instead of the fp64 square root (~30 instructions of the ~48 this test took; it runs for 20 cell pairs per column pair).
//  dx^2, dy^2, dz^2 are exact in fp64 (24-bit factors), the other products round at 1e-16: the 1e-9 band around zero still belongs to the reference expression.)
// 1 = ground, 0 = not ground, -1 = the margins cannot decide (within 1e-9 of a bound, or the shortcut is disabled: NaN tans)
DEV_INLINE int ip_ground_margins(double tan_lo, double tan_hi, float fx, float fy, float fz) {
  const double dx = (double)fx, dy = (double)fy, dz = (double)fz;
  const double h2 = dx * dx + dy * dy, z2 = dz * dz;
  const double l2 = (tan_lo * tan_lo) * h2, u2 = (tan_hi * tan_hi) * h2;   // (tan h)^2 of the two bounds
  const double sz = copysign(z2, dz);
  const double m1 = sz - copysign(l2, tan_lo), m2 = copysign(u2, tan_hi) - sz;
  const double t1 = 1e-9 * (z2 + l2), t2 = 1e-9 * (z2 + u2);
  const bool yes = m1 > t1 && m2 > t2, no = m1 < -t1 || m2 < -t2;
  return yes ? 1 : (no ? 0 : -1);
}
DEV_INLINE bool ip_is_ground(const DevCtx& d, float fx, float fy, float fz) {
  const int g = ip_ground_margins(d.tan_g_lo, d.tan_g_hi, fx, fy, fz);
  if (g >= 0) return g != 0;
  return ip_is_ground_ref((double)fx, (double)fy, (double)fz, d.P.sensor_mount_ang, d.P.ground_angle_thres);
}

// atan2(y, x) > theta for y > 0, x > 0 (y = d2 sin a, x = d1 - d2 cos a with d1 >= d2 > 0 and 0 < a < pi/2).
// tan is monotone on (0, pi/2): the comparison is decided by the sign of y - x tan(theta) whenever that difference is
// far (1e-9 relative) from zero — rounding errors of either form are ~1e-16 — and by the reference expression itself
// otherwise.  Saves two fp64 atan2 per cell on all but a vanishing fraction of the edges.
// the same decision without the reference expression: 1 / 0, or -1 when the margin cannot decide
DEV_INLINE int edge_angle_margin(double y, double x, double tan_theta) {
  const double xt = x * tan_theta, m = y - xt;
  return fabs(m) > 1e-9 * (y + fabs(xt)) ? (m > 0.0 ? 1 : 0) : -1;
}
DEV_INLINE bool edge_angle_gt(double y, double x, double theta, double tan_theta) {
  const double xt = x * tan_theta, m = y - xt;
  if (fabs(m) > 1e-9 * (y + fabs(xt))) return m > 0.0;
  return edge_angle_ref(y, x, theta);
}


// 16-bit variant for n_scan <= 16 (every such image has < 65536 cells): the parent array takes 2 B/cell (58 KB at 16x1800
// instead of 115 KB), so that two of these workgroups — or one of them and the other stream groups' workgroups — fit a CU;
// under load the kernel's duration is decided by that, not by its instruction count.  LDS has no 16-bit atomics: the union's
// compare-and-swap goes through the aligned 32-bit word (a concurrent change of the other half makes it retry), the
// path-halving stores are plain 16-bit stores (parents only ever decrease, any ancestor is a valid parent).  Component size
// and row mask are accumulated one after the other in the ROOTS' OWN parent entries with 32-bit atomics on the packed
// halves (sizes < 65536 cannot carry into the neighbour; OR is bit-local): 2 B/cell of LDS in total.
DEV_INLINE int ccl16_find(uint16_t* par, int v) {
  int curr = par[v];
  if (curr != v) {
    int prev = v, next;
    while (curr > (next = par[curr])) { par[prev] = (uint16_t)next; prev = curr; curr = next; }
  }
  return curr;
}
// compare-and-swap of entry idx (expect -> val); returns the entry's previous value
DEV_INLINE int ccl16_cas(uint16_t* par, int idx, int expect, int val) {
  unsigned* w = reinterpret_cast<unsigned*>(par) + (idx >> 1);
  const int sh = (idx & 1) * 16;
  unsigned old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (true) {
    const int cur = (int)((old >> sh) & 0xFFFFu);
    if (cur != expect) return cur;
    const unsigned nw = (old & ~(0xFFFFu << sh)) | ((unsigned)val << sh);
    const unsigned got = atomicCAS(w, old, nw);
    if (got == old) return expect;
    old = got;
  }
}
DEV_INLINE void ccl16_union(uint16_t* par, int a, int b) {
  int ra = ccl16_find(par, a), rb = ccl16_find(par, b);
  bool repeat;
  do {
    repeat = false;
    if (ra != rb) {
      int ret;
      if (ra < rb) { if ((ret = ccl16_cas(par, rb, rb, ra)) != rb) { rb = ret; repeat = true; } }
      else { if ((ret = ccl16_cas(par, ra, ra, rb)) != ra) { ra = ret; repeat = true; } }
    }
  } while (repeat);
}

#endif
