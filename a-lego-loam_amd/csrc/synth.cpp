// synth.cpp — deterministic synthetic LiDAR stream (scene "S0", trajectory "T0").
//
// There is no bag file in the reference repository (README.md:33 links to Google
// Drive) and no network here, so every config of BASELINE.json runs on this
// generator.  It is data infrastructure, not part of the hot
// path: plain host C++, built with g++ into libalego_synth.so and used by the
// tests (to feed oracle and HIP path the same scans) and by bench.py.
//
// Scene S0: ground plane z=0; rectangular room x in [-20,30], y in [-15,12] with
// walls of height kWallH; 6 axis-aligned boxes 1x1x2.5 m; 4 vertical cylinders
// r=0.3 m, h=3 m.  Trajectory T0: stadium lap, 0.10 m/scan, 560 scans per lap.
// Ray (ring i, column j): elevation -ang_bottom + i*ang_res_y, azimuth
// -(j+0.5)*ang_res_x (cell-centred for imageProjection.cpp:80,87-88).
// Range noise N(0, 0.02^2) from SplitMix64 + Box-Muller in double.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#include "../../include/alego_params.h"

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kWallH = 6.0;
constexpr double kMaxRange = 100.0;
constexpr double kSigma = 0.02;
constexpr double kArcR = 0.1 / (kPi / 180.0);  // 0.1 m per 1 degree

struct Box { double cx, cy; };
struct Cyl { double cx, cy; };
const Box kBoxes[6] = {{0.0, -9.5}, {9.0, -10.0}, {14.0, 0.0}, {0.0, -0.3}, {-14.0, 3.0}, {3.0, 9.5}};
const Cyl kCyls[4] = {{-3.0, -9.0}, {12.0, -6.0}, {-12.0, -7.0}, {6.0, 9.0}};
constexpr double kBoxHalf = 0.5, kBoxH = 2.5, kCylR = 0.3, kCylH = 3.0;

struct SplitMix64 {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return ((double)(next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
};

void pose_t0(long k, double* x, double* y, double* z, double* yaw) {
  long idx = ((k % 560) + 560) % 560;
  const double y0 = -6.0;
  if (idx < 100) {
    *x = -5.0 + 0.1 * idx; *y = y0; *yaw = 0.0;
  } else if (idx < 280) {
    double a = (idx - 100) * (kPi / 180.0);
    *x = 5.0 + kArcR * std::sin(a); *y = y0 + kArcR - kArcR * std::cos(a); *yaw = a;
  } else if (idx < 380) {
    *x = 5.0 - 0.1 * (idx - 280); *y = y0 + 2.0 * kArcR; *yaw = kPi;
  } else {
    double a = (idx - 380) * (kPi / 180.0);
    *x = -5.0 - kArcR * std::sin(a); *y = y0 + kArcR + kArcR * std::cos(a); *yaw = kPi + a;
  }
  *z = 1.8 + 0.01 * std::sin(0.1 * (double)k);
}

// nearest hit distance along (o + t d), or +inf
double cast(const double o[3], const double d[3]) {
  double best = std::numeric_limits<double>::infinity();
  // ground
  if (d[2] < 0.0) { double t = -o[2] / d[2]; if (t > 1e-9 && t < best) best = t; }
  // walls: exit point of the convex room
  {
    double tw = std::numeric_limits<double>::infinity();
    if (d[0] > 0) tw = std::fmin(tw, (30.0 - o[0]) / d[0]); else if (d[0] < 0) tw = std::fmin(tw, (-20.0 - o[0]) / d[0]);
    if (d[1] > 0) tw = std::fmin(tw, (12.0 - o[1]) / d[1]); else if (d[1] < 0) tw = std::fmin(tw, (-15.0 - o[1]) / d[1]);
    if (std::isfinite(tw)) { double zh = o[2] + tw * d[2]; if (zh >= 0.0 && zh <= kWallH && tw < best) best = tw; }
  }
  // boxes (slab method)
  for (const Box& b : kBoxes) {
    double lo[3] = {b.cx - kBoxHalf, b.cy - kBoxHalf, 0.0}, hi[3] = {b.cx + kBoxHalf, b.cy + kBoxHalf, kBoxH};
    double t0 = 0.0, t1 = best; bool ok = true;
    for (int a = 0; a < 3 && ok; ++a) {
      if (std::fabs(d[a]) < 1e-15) { if (o[a] < lo[a] || o[a] > hi[a]) ok = false; }
      else {
        double ta = (lo[a] - o[a]) / d[a], tb = (hi[a] - o[a]) / d[a];
        if (ta > tb) { double s = ta; ta = tb; tb = s; }
        if (ta > t0) t0 = ta; if (tb < t1) t1 = tb; if (t0 > t1) ok = false;
      }
    }
    if (ok && t0 > 1e-9 && t0 < best) best = t0;
  }
  // cylinders (side surface only)
  for (const Cyl& c : kCyls) {
    double ox = o[0] - c.cx, oy = o[1] - c.cy;
    double A = d[0] * d[0] + d[1] * d[1]; if (A < 1e-18) continue;
    double B = ox * d[0] + oy * d[1], C = ox * ox + oy * oy - kCylR * kCylR;
    double disc = B * B - A * C; if (disc < 0) continue;
    double t = (-B - std::sqrt(disc)) / A;
    if (t > 1e-9 && t < best) { double zh = o[2] + t * d[2]; if (zh >= 0.0 && zh <= kCylH) best = t; }
  }
  return best;
}

}  // namespace

extern "C" {

// Ground-truth pose of scan `scan_index` of stream `stream` (x, y, z, yaw).
void alego_synth_pose(int stream, long scan_index, double* pose4) {
  pose_t0(scan_index + 70L * stream, &pose4[0], &pose4[1], &pose4[2], &pose4[3]);
}

// flags: bit0 = azimuth jitter U(-0.45,0.45)*ang_res_x (stresses the atan2f cell
// assignment); bit1 = emit NaN points for rays without a return (exercises the
// NaN filter, imageProjection.cpp:58-59) instead of dropping them; bit2 = azimuth uniform over the whole column (what a real
// spinning sensor delivers: ip_fused's quick projection defers the points near a column boundary to the exact one).
// Output order: ring-major (ring 0 all columns, ring 1 ...).  Returns the point count.
int alego_synth_scan(const alego_params* P, int stream, long scan_index, int flags,
                     alego_point* out, int cap) {
  double px, py, pz, yaw;
  pose_t0(scan_index + 70L * stream, &px, &py, &pz, &yaw);
  SplitMix64 rng{0xA1E60000ull + ((uint64_t)stream << 20) + (uint64_t)scan_index};
  const double cyaw = std::cos(yaw), syaw = std::sin(yaw);
  const double o[3] = {px, py, pz};
  int n = 0;
  for (int i = 0; i < P->n_scan; ++i) {
    const double phi = (-P->ang_bottom + i * P->ang_res_y) * (kPi / 180.0);
    const double cphi = std::cos(phi), sphi = std::sin(phi);
    for (int j = 0; j < P->horizon_scan; ++j) {
      double u1 = rng.uniform(), u2 = rng.uniform(), u3 = rng.uniform();
      double az = -(j + 0.5) * P->ang_res_x;
      if (flags & 1) az += (u3 - 0.5) * 0.9 * P->ang_res_x;
      if (flags & 4) az += (u3 - 0.5) * P->ang_res_x;   // azimuth anywhere in the column: ~4 % of the points within 0.02 columns of a boundary
      az *= (kPi / 180.0);
      const double ds[3] = {cphi * std::cos(az), cphi * std::sin(az), sphi};  // sensor frame
      const double dw[3] = {cyaw * ds[0] - syaw * ds[1], syaw * ds[0] + cyaw * ds[1], ds[2]};
      double t = cast(o, dw);
      if (!(t < kMaxRange)) {
        if ((flags & 2) && n < cap) {
          float q = std::numeric_limits<float>::quiet_NaN();
          out[n++] = alego_point{q, q, q, 0.f};
        }
        continue;
      }
      double g = std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * kPi * u2);
      double r = t + kSigma * g;
      if (n < cap) {
        out[n++] = alego_point{(float)(r * ds[0]), (float)(r * ds[1]), (float)(r * ds[2]),
                               (float)((i * 37 + j) % 256)};
      }
    }
  }
  return n;
}

}  // extern "C"

extern "C" void alego_synth_default_params(alego_params* p, int n_scan, int horizon) { alego_default_params(p, n_scan, horizon); }
extern "C" int alego_synth_params_sizeof() { return (int)sizeof(alego_params); }
