// oracle_math.h — TEST INFRASTRUCTURE (parity oracle). Not part of the product.
//
// Single-precision transcendental functions the reference's projection loop
// resolves to (imageProjection.cpp:79,87,99: `using namespace std` + float
// arguments select std::atan2(float,float) / std::hypot(float,float) /
// std::sqrt(float)).  On the reference's class of host (x86-64 glibc 2.35)
// atan2f is the Sun fdlibm algorithm below and hypotf is
// evaluated in double.  tests/test_oracle_math.py pins both against this
// container's libm.
//
// The atanf/atan2f algorithm and constants restate Sun Microsystems' fdlibm
// (e_atan2f.c / s_atanf.c): "Copyright (C) 1993 by Sun Microsystems, Inc. All
// rights reserved.  Developed at SunPro, a Sun Microsystems, Inc. business.
// Permission to use, copy, modify, and distribute this software is freely
// granted, provided that this notice is preserved."
#ifndef ORACLE_MATH_H_
#define ORACLE_MATH_H_

#include <cmath>
#include <cstdint>
#include <cstring>

namespace omath {

inline int32_t f2i(float f) { int32_t i; std::memcpy(&i, &f, 4); return i; }
inline float i2f(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }

inline float o_atanf(float x) {
  static const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  static const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  static const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                               9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                               4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
  int32_t hx = f2i(x), ix = hx & 0x7fffffff, id;
  if (ix >= 0x4c800000) {
    if (ix > 0x7f800000) return x + x;
    return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {
    if (ix < 0x31000000) return x;
    id = -1;
  } else {
    x = std::fabs(x);
    if (ix < 0x3f980000) {
      if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
      else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
      else { id = 3; x = -1.0f / x; }
    }
  }
  float z = x * x, w = z * z;
  float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return hx < 0 ? -z : z;
}

inline float o_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
              pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  int32_t hx = f2i(x), ix = hx & 0x7fffffff, hy = f2i(y), iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return o_atanf(y);
  int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) {
    switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; }
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny;
                   case 2: return 3.0f * pi_o_4 + tiny; default: return -3.0f * pi_o_4 - tiny; }
    } else {
      switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi + tiny; default: return -pi - tiny; }
    }
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  int32_t k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = o_atanf(std::fabs(y / x));
  switch (m) {
    case 0: return z;
    case 1: return i2f(f2i(z) ^ (int32_t)0x80000000);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

// glibc 2.35 hypotf evaluates in double and rounds once (validated).
inline float o_hypotf(float x, float y) { return (float)std::sqrt((double)x * (double)x + (double)y * (double)y); }

// ---------------------------------------------------------------------------
// sinf / cosf as Eigen's AngleAxisf -> Quaternionf conversion calls them in
// transformPointCloud (include/alego/laserMapping.h:166-173: std::sin / std::cos on
// float half angles).  glibc >= 2.28 (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c,
// sincosf.h) evaluates an fp64 minimax polynomial after a fast quadrant reduction and
// rounds once; it is NOT the correctly rounded value ((float)sin((double)x) differs on
// 1.3 % of the inputs in [-pi/2, pi/2], measured here).  Restated below for |x| < 120
// (every half angle of a key pose lies in [-pi/2, pi/2]); the coefficient table was read
// out of this container's libm.so.6 (.rodata, __sincosf_table) and
// tests/test_oracle.py::test_sinf_cosf_equal_this_hosts_libm pins the result bit for bit.
// Outside that range (never reached by a key pose) the correctly rounded value is used.
// ---------------------------------------------------------------------------
struct SinCosTab { double sign[4], hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };
inline const SinCosTab& sincos_tab(int neg) {
  static const SinCosTab T[2] = {
      {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
       0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16,
       -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
      {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
       -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16,
       -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
  return T[neg];
}
inline uint32_t abstop12(float x) { return ((uint32_t)f2i(x) >> 20) & 0x7ff; }
inline float sincos_poly(double x, double x2, const SinCosTab& p, int n) {
  if ((n & 1) == 0) {
    const double x3 = x * x2, s1 = p.s2 + x2 * p.s3, x7 = x3 * x2, s = x + x3 * p.s1;
    return (float)(s + x7 * s1);
  }
  const double x4 = x2 * x2, c2 = p.c3 + x2 * p.c4, c1 = p.c0 + x2 * p.c1, x6 = x4 * x2, c = c1 + x4 * p.c2;
  return (float)(c + x6 * c2);
}
inline double sincos_reduce_fast(double x, const SinCosTab& p, int* np) {
  const double r = x * p.hpi_inv;                     // hpi_inv is prescaled by 2^24: the quadrant ends up in bits 24..31
  const int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return x - n * p.hpi;
}
inline float o_sinf(float y) {
  double x = y;
  const SinCosTab* p = &sincos_tab(0);
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    if (abstop12(y) < abstop12(0x1p-12f)) return y;
    return sincos_poly(x, x * x, *p, 0);
  }
  if (abstop12(y) < abstop12(120.0f)) {
    int n;
    x = sincos_reduce_fast(x, *p, &n);
    const double s = p->sign[n & 3];
    if (n & 2) p = &sincos_tab(1);
    return sincos_poly(x * s, x * x, *p, n);
  }
  return (float)std::sin((double)y);
}
inline float o_cosf(float y) {
  double x = y;
  const SinCosTab* p = &sincos_tab(0);
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
    return sincos_poly(x, x * x, *p, 1);
  }
  if (abstop12(y) < abstop12(120.0f)) {
    int n;
    x = sincos_reduce_fast(x, *p, &n);
    const double s = p->sign[(n + 1) & 3];
    if ((n + 1) & 2) p = &sincos_tab(1);
    return sincos_poly(x * s, x * x, *p, n ^ 1);
  }
  return (float)std::cos((double)y);
}

}  // namespace omath
#endif
