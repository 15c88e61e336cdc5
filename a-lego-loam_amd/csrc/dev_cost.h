// dev_cost.h — the four analytic-Jacobian cost functors of include/alego/utility.h:122-349
// and the small Eigen-equivalent rotation helpers, for the device.
#ifndef ALEGO_DEV_COST_H_
#define ALEGO_DEV_COST_H_

#include "dev_common.h"

enum { BLK_SURF = 0, BLK_CORNER = 1, BLK_EDGE = 2, BLK_PLANE = 3 };

struct DQuat { double w, x, y, z; };

DEV_INLINE DQuat dq_mul(const DQuat& a, const DQuat& b) {
  DQuat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
// AngleAxisd(yaw,Z) * AngleAxisd(pitch,Y) * AngleAxisd(roll,X)  (utility.h:128, laserOdometry.cpp:731)
DEV_INLINE DQuat dq_zyx(double yaw, double pitch, double roll) {
  const double hz = 0.5 * yaw, hy = 0.5 * pitch, hx = 0.5 * roll;
  DQuat qz{cos(hz), 0, 0, sin(hz)}, qy{cos(hy), 0, sin(hy), 0}, qx{cos(hx), sin(hx), 0, 0};
  return dq_mul(dq_mul(qz, qy), qx);
}
DEV_INLINE void dq_rotate(const DQuat& q, const double v[3], double out[3]) {
  const double u0 = 2 * (q.y * v[2] - q.z * v[1]), u1 = 2 * (q.z * v[0] - q.x * v[2]), u2 = 2 * (q.x * v[1] - q.y * v[0]);
  out[0] = v[0] + q.w * u0 + (q.y * u2 - q.z * u1);
  out[1] = v[1] + q.w * u1 + (q.z * u0 - q.x * u2);
  out[2] = v[2] + q.w * u2 + (q.x * u1 - q.y * u0);
}
DEV_INLINE void dq_to_mat(const DQuat& q, double R[9]) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
DEV_INLINE DQuat dq_from_mat(const double m[9]) {
  DQuat q;
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0); q.w = 0.5 * t; t = 0.5 / t;
    q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    double v[3];
    v[i] = 0.5 * t; t = 0.5 / t;
    q.w = (m[k * 3 + j] - m[j * 3 + k]) * t;
    v[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    v[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}
DEV_INLINE DQuat dq_inverse(const DQuat& q) {
  const double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  return DQuat{q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}

// Pose-dependent terms shared by every residual of one evaluation.
struct PoseTerms {
  DQuat q;
  double t[3];
  double sr, cr, sp, cp, sy, cy;
};
DEV_INLINE PoseTerms pose_terms(const double* p) {
  PoseTerms T;
  T.q = dq_zyx(p[5], p[4], p[3]);
  T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2];
  T.sr = sin(p[3]); T.cr = cos(p[3]); T.sp = sin(p[4]); T.cp = cos(p[4]); T.sy = sin(p[5]); T.cy = cos(p[5]);
  return T;
}

// The same terms computed cooperatively by a workgroup: six lanes take one angle each (the three half angles of the
// quaternion, the three full angles of the Jacobian terms), everybody assembles the result from LDS.  Evaluating the
// nine sin/cos pairs in every thread cost ~2000 fp64 instructions per wavefront and solver evaluation — most of what
// lo_solve / lm_solve issued.  Same sin()/cos() calls on the same arguments: bit-identical to pose_terms().
// Ends with a barrier; the caller must pass another barrier before the next call (block_reduce28_lds does).
DEV_INLINE PoseTerms pose_terms_coop(const double* p, double* s_trig /*[12] LDS*/) {
  const int tid = threadIdx.x;
  if (tid < 6) {
    const double a = tid == 0 ? 0.5 * p[5] : tid == 1 ? 0.5 * p[4] : tid == 2 ? 0.5 * p[3] : tid == 3 ? p[3] : tid == 4 ? p[4] : p[5];
    s_trig[2 * tid] = cos(a);
    s_trig[2 * tid + 1] = sin(a);
  }
  __syncthreads();
  PoseTerms T;
  const DQuat qz{s_trig[0], 0, 0, s_trig[1]}, qy{s_trig[2], 0, s_trig[3], 0}, qx{s_trig[4], s_trig[5], 0, 0};
  T.q = dq_mul(dq_mul(qz, qy), qx);
  T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2];
  T.cr = s_trig[6]; T.sr = s_trig[7]; T.cp = s_trig[8]; T.sp = s_trig[9]; T.cy = s_trig[10]; T.sy = s_trig[11];
  return T;
}

// residual + 1x6 Jacobian (uncorrected).  a = lpj | plane normal, b = lpl, c = lpm, dd = negative_OA_dot_norm.
DEV_INLINE void eval_block(int type, const double cp_[3], const double a[3], const double b[3], const double c[3], double dd,
                           const PoseTerms& T, double* res, double J[6]) {
  double lp[3];
  dq_rotate(T.q, cp_, lp);
  lp[0] += T.t[0]; lp[1] += T.t[1]; lp[2] += T.t[2];
  const double X = cp_[0], Y = cp_[1], Z = cp_[2];
  const double sr = T.sr, cr = T.cr, sp = T.sp, cp = T.cp, sy = T.sy, cy = T.cy;
  if (type == BLK_SURF) {  // utility.h:185-235
    double ca = (a[1] - b[1]) * (a[2] - c[2]) - (a[2] - b[2]) * (a[1] - c[1]);
    double cb = (a[2] - b[2]) * (a[0] - c[0]) - (a[0] - b[0]) * (a[2] - c[2]);
    double cc = (a[0] - b[0]) * (a[1] - c[1]) - (a[1] - b[1]) * (a[0] - c[0]);
    ca *= ca; cb *= cb; cc *= cc;
    const double ex = lp[0] - a[0], ey = lp[1] - a[1], ez = lp[2] - a[2];
    const double m = sqrt(ex * ex * ca + ey * ey * cb + ez * ez * cc);
    const double k = sqrt(ca + cb + cc);
    *res = m / k;
    const double tmp = m * k;
    J[0] = 0.; J[1] = 0.; J[2] = ((ez * cc) / tmp) / k; J[3] = 0.; J[4] = 0.; J[5] = 0.;
    return;
  }
  // shared derivative table utility.h:148-158 (dy_dp keeps the reference's cr*sr*cp term)
  const double dx_dr = (cy * sp * cr + sr * sy) * Y + (sy * cr - cy * sr * sp) * Z;
  const double dy_dr = (-cy * sr + sy * sp * cr) * Y + (-sr * sy * sp - cy * cr) * Z;
  const double dz_dr = cp * cr * Y - cp * sr * Z;
  const double dx_dp = -cy * sp * X + cy * cp * sr * Y + cy * cr * cp * Z;
  const double dy_dp = -sp * sy * X + sy * cp * sr * Y + cr * sr * cp * Z;
  const double dz_dp = -cp * X - sp * sr * Y - sp * cr * Z;
  const double dx_dy = -sy * cp * X - (sy * sp * sr + cr * cy) * Y + (cy * sr - sy * cr * sp) * Z;
  const double dy_dy = cp * cy * X + (-sy * cr + cy * sp * sr) * Y + (cy * cr * sp + sy * sr) * Z;
  const double dz_dy = 0.;
  if (type == BLK_PLANE) {  // utility.h:307-343
    *res = (a[0] * lp[0] + a[1] * lp[1] + a[2] * lp[2]) + dd;
    J[0] = a[0]; J[1] = a[1]; J[2] = a[2];
    J[3] = a[0] * dx_dr + a[1] * dy_dr + a[2] * dz_dr;
    J[4] = a[0] * dx_dp + a[1] * dy_dp + a[2] * dz_dp;
    J[5] = a[0] * dx_dy + a[1] * dy_dy + a[2] * dz_dy;
    return;
  }
  // BLK_CORNER utility.h:126-174 / BLK_EDGE utility.h:246-294
  const double e0 = a[0] - b[0], e1 = a[1] - b[1], e2 = a[2] - b[2];
  const double k = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
  const double ca = (lp[1] - a[1]) * (lp[2] - b[2]) - (lp[2] - a[2]) * (lp[1] - b[1]);
  const double cb = (lp[2] - a[2]) * (lp[0] - b[0]) - (lp[0] - a[0]) * (lp[2] - b[2]);
  const double cc = (lp[0] - a[0]) * (lp[1] - b[1]) - (lp[1] - a[1]) * (lp[0] - b[0]);
  const double m = sqrt(ca * ca + cb * cb + cc * cc);
  *res = m / k;
  const double dm_dx = (cb * (b[2] - a[2]) + cc * (a[1] - b[1])) / m;
  const double dm_dy = (ca * (a[2] - b[2]) - cc * (a[0] - b[0])) / m;
  const double dm_dz = (-ca * (a[1] - b[1]) + cb * (a[0] - b[0])) / m;
  if (type == BLK_CORNER) {
    J[0] = dm_dx / k; J[1] = dm_dy / k; J[2] = 0.; J[3] = 0.; J[4] = 0.;
    J[5] = (dm_dx * dx_dy + dm_dy * dy_dy + dm_dz * dz_dy) / k;
  } else {
    J[0] = dm_dx / k; J[1] = dm_dy / k; J[2] = dm_dz / k;
    J[3] = (dm_dx * dx_dr + dm_dy * dy_dr + dm_dz * dz_dr) / k;
    J[4] = (dm_dx * dx_dp + dm_dy * dy_dp + dm_dz * dz_dp) / k;
    J[5] = (dm_dx * dx_dy + dm_dy * dy_dy + dm_dz * dz_dy) / k;
  }
}

// ResidualBlock::Evaluate with HuberLoss(a) + Corrector (rho'' <= 0): accumulates the
// upper triangle of J^T J (21), J^T r (6) and the cost (1) into acc[28].
DEV_INLINE void accumulate_block(double r, const double J[6], double huber_a, double acc[28]) {
  const double s = r * r, b2 = huber_a * huber_a;
  double rho0, rho1;
  if (s > b2) { const double rr = sqrt(s); rho0 = 2.0 * huber_a * rr - b2; rho1 = fmax(2.2250738585072014e-308, huber_a / rr); }
  else { rho0 = s; rho1 = 1.0; }
  const double sq = s > b2 ? sqrt(rho1) : 1.0;  // inliers: rho' = 1, no correction
  double Jc[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) Jc[k] = J[k] * sq;
  const double rc = r * sq;
  int t = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) acc[t++] += Jc[i] * Jc[j];
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[21 + k] += Jc[k] * rc;
  acc[27] += 0.5 * rho0;
}

// Deterministic workgroup reduction of the 28 normal-equation scalars.  Each quad of lanes first adds its four
// partials with two DPP quad_perm steps (no LDS), one lane per quad stores the sums k-major (conflict-free),
// then a group of 8 / 16 lanes per scalar finishes (below).  The quad step keeps the LDS footprint at 28 x T/4 doubles
// (28 KB for 512 threads): these solver workgroups live for hundreds of microseconds, and their LDS is what keeps other
// streams' workgroups off the CU.  (An earlier version let 4 threads per scalar add 32 entries each and a third stage add
// the 4 sums: 30 % of lm_solve's time.)
template <int CTRL>
DEV_INLINE double dpp_add_f64(double v) {   // v + (v of the lane selected by the DPP control CTRL)
  const long long b = __double_as_longlong(v);
  const int lo = (int)(unsigned)(unsigned long long)b, hi = (int)(unsigned)((unsigned long long)b >> 32);
  const int plo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false), phi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return v + __longlong_as_double((long long)(((unsigned long long)(unsigned)phi << 32) | (unsigned)plo));
}
// T threads -> Q = T/4 partials per scalar after the quad step; G = Q/8 lanes per scalar (16 for 512 threads ... 2 for 64)
// add 8 strided partials each and finish with log2(G) DPP steps inside their row: two barriers per reduction.
template <int T>
DEV_INLINE void block_reduce28_lds(const double acc[28], double* s_acc /*[28*T/4]*/, double* /*unused*/, double* s_out /*[28]*/) {
  constexpr int Q = T / 4, G = Q / 8;
  static_assert(G == 2 || G == 4 || G == 8 || G == 16, "64, 128, 256 or 512 threads");
  static_assert(28 * G <= T, "one lane group per scalar");
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 28; ++k) {
    double v = dpp_add_f64<0xB1>(acc[k]);   // lane ^ 1: both lanes of a pair hold a + b (commutative: identical bits)
    v = dpp_add_f64<0x4E>(v);               // lane ^ 2: all four lanes hold (a + b) + (c + d)
    if ((tid & 3) == 0) s_acc[k * Q + (tid >> 2)] = v;
  }
  __syncthreads();
  if (tid < 28 * G) {   // whole lane groups: the DPP steps below never leave a group
    const int k = tid / G, g = tid - k * G;
    double t = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += s_acc[k * Q + j * G + g];
    t = dpp_add_f64<0xB1>(t);
    if (G >= 4) t = dpp_add_f64<0x4E>(t);
    if (G >= 8) t = dpp_add_f64<0x141>(t);   // row_half_mirror: the two quads of an 8-lane group
    if (G == 16) t = dpp_add_f64<0x140>(t);  // row_mirror: the two halves of a row
    if (g == 0) s_out[k] = t;
  }
  __syncthreads();
}

// The same with a third DPP step (row_half_mirror: the two quads of eight lanes) before LDS: 28 x T/8 doubles of scratch (14 KB for
// 512 threads).  lm_solve keeps its residual rows in LDS, so every KB of reduction scratch is 20 rows streamed from the L2 instead.
template <int T>
DEV_INLINE void block_reduce28_oct(const double acc[28], double* s_acc /*[28*T/8]*/, double* s_out /*[28]*/) {
  constexpr int Q = T / 8, G = Q / 8;
  static_assert(G == 2 || G == 4 || G == 8 || G == 16, "128, 256, 512 or 1024 threads");
  static_assert(28 * G <= T, "one lane group per scalar");
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 28; ++k) {
    double v = dpp_add_f64<0xB1>(acc[k]);
    v = dpp_add_f64<0x4E>(v);
    v = dpp_add_f64<0x141>(v);              // all eight lanes hold ((a + b) + (c + d)) + ((e + f) + (g + h))
    if ((tid & 7) == 0) s_acc[k * Q + (tid >> 3)] = v;
  }
  __syncthreads();
  if (tid < 28 * G) {
    const int k = tid / G, g = tid - k * G;
    double t = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += s_acc[k * Q + j * G + g];
    t = dpp_add_f64<0xB1>(t);
    if (G >= 4) t = dpp_add_f64<0x4E>(t);
    if (G >= 8) t = dpp_add_f64<0x141>(t);
    if (G == 16) t = dpp_add_f64<0x140>(t);
    if (g == 0) s_out[k] = t;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
// Trust-region Levenberg-Marquardt control (Ceres TrustRegionMinimizer +
// LevenbergMarquardtStrategy defaults) driven by one thread; the
// residual evaluations are done by the whole workgroup between its steps.
// The DENSE_QR solve of [J; D] y = [r; 0] is replaced by the equivalent normal
// equations (J^T J + D^2) y = J^T r with a 6x6 Cholesky factorisation.
// ---------------------------------------------------------------------------
struct LmState {
  double x[6], cand[6], scale[6], x_cost, x_norm, gmax, radius, dec, mcc;
  double H[21], g[6];   // unscaled J^T J (upper) and J^T r at x
  int iter, max_iter, num_invalid, step_successful, successful, termination;
  double initial_cost;
};
enum { LM_STOP = 0, LM_EVAL = 1, LM_AGAIN = 2 };

DEV_INLINE int tri(int i, int j) { return i <= j ? i * 6 - i * (i - 1) / 2 + (j - i) : j * 6 - j * (j - 1) / 2 + (i - j); }

DEV_INLINE void lm_load_eval(LmState& S, const double* acc) {
#pragma unroll
  for (int k = 0; k < 21; ++k) S.H[k] = acc[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) S.g[k] = acc[21 + k];
  double gm = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) gm = fmax(gm, fabs(S.x[k] - (S.x[k] - S.g[k])));
  S.gmax = gm;
}

DEV_INLINE void lm_begin(LmState& S, const double* x0, const double* acc, int max_iter) {
  double xn = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) { S.x[k] = x0[k]; xn += x0[k] * x0[k]; }
  S.x_norm = sqrt(xn);
  S.x_cost = acc[27]; S.initial_cost = acc[27];
  lm_load_eval(S, acc);
#pragma unroll
  for (int k = 0; k < 6; ++k) S.scale[k] = 1.0 / (1.0 + sqrt(S.H[tri(k, k)]));  // jacobi scaling, iteration 0 only
  S.radius = 1e4; S.dec = 2.0; S.iter = 0; S.max_iter = max_iter; S.num_invalid = 0;
  S.step_successful = 1; S.successful = 0; S.termination = 0;
  if (!isfinite(S.x_cost)) S.termination = 4;
}

// decide whether to stop, and if not compute the next candidate into S.cand
DEV_INLINE int lm_propose(LmState& S) {
  if (S.termination == 4) return LM_STOP;
  if (S.iter >= S.max_iter) { S.termination = 0; return LM_STOP; }
  if (S.step_successful && S.gmax <= 1e-10) { S.termination = 1; return LM_STOP; }
  if (S.radius <= 1e-32) { S.termination = 5; return LM_STOP; }
  ++S.iter;
  double A[6][6], gs[6], y[6], Hl[21], sc[6], gl[6];
#pragma unroll
  for (int k = 0; k < 21; ++k) Hl[k] = S.H[k];  // LDS -> registers once
#pragma unroll
  for (int k = 0; k < 6; ++k) { sc[k] = S.scale[k]; gl[k] = S.g[k]; }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    gs[i] = gl[i] * sc[i];
#pragma unroll
    for (int j = 0; j < 6; ++j) A[i][j] = Hl[tri(i, j)] * sc[i] * sc[j];
  }
  double Hs[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) Hs[i][j] = A[i][j];
#pragma unroll
  for (int i = 0; i < 6; ++i) A[i][i] += fmin(fmax(Hs[i][i], 1e-6), 1e32) / S.radius;
  // Cholesky A = L L^T; the diagonal is stored inverted: one division per column instead of one per entry (this thread
  // is the serial part of every solver iteration, and an fp64 division is ~30 instructions)
  bool ok = true;
  double dinv[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double s = A[j][j];
    for (int k = 0; k < j; ++k) s -= A[j][k] * A[j][k];
    if (!(s > 0)) { ok = false; s = 1; }
    const double l = sqrt(s);
    dinv[j] = 1.0 / l;
    for (int i = j + 1; i < 6; ++i) {
      double t = A[i][j];
      for (int k = 0; k < j; ++k) t -= A[i][k] * A[j][k];
      A[i][j] = t * dinv[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) { double t = gs[i]; for (int k = 0; k < i; ++k) t -= A[i][k] * y[k]; y[i] = t * dinv[i]; }
#pragma unroll
  for (int i = 5; i >= 0; --i) { double t = y[i]; for (int k = i + 1; k < 6; ++k) t -= A[k][i] * y[k]; y[i] = t * dinv[i]; }
  double step[6], mcc = 0;
  bool finite = ok;
#pragma unroll
  for (int i = 0; i < 6; ++i) { step[i] = -y[i]; finite = finite && isfinite(step[i]); }
  if (finite) {
    double lin = 0, quad = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      lin += step[i] * gs[i];
      double t = 0;
#pragma unroll
      for (int j = 0; j < 6; ++j) t += Hs[i][j] * step[j];
      quad += step[i] * t;
    }
    mcc = -lin - 0.5 * quad;
  }
  if (!finite || !(mcc > 0.0)) {  // HandleInvalidStep
    if (++S.num_invalid >= 5) { S.termination = 4; return LM_STOP; }
    S.radius /= S.dec; S.dec *= 2.0; S.step_successful = 0;
    return LM_AGAIN;
  }
  S.num_invalid = 0;
  S.mcc = mcc;
#pragma unroll
  for (int i = 0; i < 6; ++i) S.cand[i] = S.x[i] + step[i] * sc[i];
  return LM_EVAL;
}

// consume the evaluation at S.cand; returns LM_STOP on convergence
DEV_INLINE int lm_consume(LmState& S, const double* acc) {
  double cand_cost = acc[27];
  if (!isfinite(cand_cost)) cand_cost = 1.7976931348623157e308;
  double sn = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) sn += (S.x[k] - S.cand[k]) * (S.x[k] - S.cand[k]);
  if (sqrt(sn) <= 1e-8 * (S.x_norm + 1e-8)) { S.termination = 2; return LM_STOP; }
  const double cost_change = S.x_cost - cand_cost;
  if (fabs(cost_change) <= 1e-6 * S.x_cost) { S.termination = 3; return LM_STOP; }
  const double rd = cost_change / S.mcc;
  if (rd > 1e-3) {
    double xn = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { S.x[k] = S.cand[k]; xn += S.x[k] * S.x[k]; }
    S.x_norm = sqrt(xn);
    S.x_cost = cand_cost;
    lm_load_eval(S, acc);
    S.step_successful = 1; ++S.successful;
    const double t = 2.0 * rd - 1.0;
    S.radius = S.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
    S.radius = fmin(1e16, S.radius);
    S.dec = 2.0;
  } else {
    S.step_successful = 0;
    S.radius /= S.dec; S.dec *= 2.0;
  }
  return LM_AGAIN;
}

#endif
