// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Deterministic float cos/sin used for the rBRIEF steering (reference orbExtractor.cpp:424-425
// calls libm cos/sin on a float, i.e. cosf/sinf through libstdc++'s <math.h> overloads).
// libm's cosf/sinf are platform code (glibc ifunc variants, <1 ULP but not correctly rounded),
// so "what the reference computes" is not reproducible off-platform.  The contract used by
// the oracle AND restated independently in the HIP kernels is the fixed sequence of IEEE-754
// double operations below (Cody-Waite reduction by pi/2 + fdlibm-style minimax kernels, no
// FMA contraction), rounded once to float.  It agrees with correctly rounded cosf/sinf except
// on ~2^-29 of inputs; tests/test_oracle_primitives.py counts the disagreements with this
// container's libm over the whole angle domain and checks the descriptors are unaffected.
// Valid for |x| <= 1e3 (angles here are in [0, 2*pi]).
#pragma once
#include <cmath>

namespace yd_trig {

inline void sincos_core(float xf, double* s, double* c, int* quad) {
  const double TWO_OVER_PI = 6.36619772367581382433e-01;
  const double PIO2_HI = 1.57079632673412561417e+00;  // first 33 bits of pi/2
  const double PIO2_LO = 6.07710050650619224932e-11;  // pi/2 - PIO2_HI
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  volatile double x = (double)xf;  // volatile temporaries: forbid contraction/reassociation
  volatile double t = x * TWO_OVER_PI;
  double n = nearbyint(t);
  volatile double a = n * PIO2_HI;
  volatile double b = x - a;
  volatile double d = n * PIO2_LO;
  volatile double r = b - d;
  volatile double z = r * r;
  // sin kernel: r + r*z*(S1 + z*(S2 + z*(S3 + z*(S4 + z*(S5 + z*S6)))))
  volatile double p = z * S6; p = p + S5; p = p * z; p = p + S4; p = p * z; p = p + S3;
  p = p * z; p = p + S2; p = p * z; p = p + S1; p = p * z; p = p * r; p = p + r;
  // cos kernel: 1 - (z/2 - z*z*(C1 + z*(C2 + z*(C3 + z*(C4 + z*(C5 + z*C6))))))
  volatile double q = z * C6; q = q + C5; q = q * z; q = q + C4; q = q * z; q = q + C3;
  q = q * z; q = q + C2; q = q * z; q = q + C1; q = q * z; q = q * z;
  volatile double h = z * 0.5; h = h - q; h = 1.0 - h;
  *s = p;
  *c = h;
  *quad = ((int)n) & 3;
}

inline float cosf_det(float x) {
  double s, c;
  int q;
  sincos_core(x, &s, &c, &q);
  double v = q == 0 ? c : q == 1 ? -s : q == 2 ? -c : s;
  return (float)v;
}

inline float sinf_det(float x) {
  double s, c;
  int q;
  sincos_core(x, &s, &c, &q);
  double v = q == 0 ? s : q == 1 ? c : q == 2 ? -s : -c;
  return (float)v;
}

}  // namespace yd_trig
