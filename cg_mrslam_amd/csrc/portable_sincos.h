// Portable double-precision sin / cos for device code (gfx950).
//
// Same routine as oracle/matcher_oracle.c (Cody-Waite reduction + classic minimax kernels): every sine / cosine
// that decides a grid cell (matcher search angle, occupancy-map beam end points) must be bit-identical on host
// and device, and libm and ocml are not.  Translation units that include this header must be compiled with
// -ffp-contract=off (see Makefile).
//  The polynomial kernels k_sin / k_cos and the three-stage pi/2 reduction below follow FreeBSD/Sun fdlibm
// (__kernel_sin, __kernel_cos, __ieee754_rem_pio2: same coefficients, same evaluation order), whose licence asks
// that this notice be preserved:
// ====================================================
// Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
// 
// Developed at SunPro, a Sun Microsystems, Inc. business.
// Permission to use, copy, modify, and distribute this
// software is freely granted, provided that this notice
// is preserved.
// ====================================================
#ifndef CGMR_PORTABLE_SINCOS_H
#define CGMR_PORTABLE_SINCOS_H
#include <hip/hip_runtime.h>

namespace cgmr {
namespace psc {

__device__ inline double k_sin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

__device__ inline double k_cos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double ax = fabs(x);
  if (ax < 0.3) return 1.0 - (0.5 * z - (z * r - x * y));
  double qx;
  if (ax > 0.78125) qx = 0.28125;
  else {
    unsigned long long u = (unsigned long long)__double_as_longlong(ax * 0.25);
    u &= 0xffffffff00000000ULL;
    qx = __longlong_as_double((long long)u);
  }
  double hz = 0.5 * z - qx;
  double a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}

__device__ inline int rem_pio2(double x, double* y0, double* y1) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
               pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
               pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
               pio2_3t = 8.47842766036889956997e-32;
  double ax = fabs(x);
  int n = (int)(ax * invpio2 + 0.5);
  double fn = (double)n;
  double r = ax - fn * pio2_1;
  double w = fn * pio2_1t;
  double a0 = r - w;
  int ex = (int)(((unsigned long long)__double_as_longlong(ax) >> 52) & 0x7ff);
  int ea = (int)(((unsigned long long)__double_as_longlong(a0) >> 52) & 0x7ff);
  if (ex - ea > 16) {
    double t = r;
    w = fn * pio2_2;
    r = t - w;
    w = fn * pio2_2t - ((t - r) - w);
    a0 = r - w;
    ea = (int)(((unsigned long long)__double_as_longlong(a0) >> 52) & 0x7ff);
    if (ex - ea > 49) {
      t = r;
      w = fn * pio2_3;
      r = t - w;
      w = fn * pio2_3t - ((t - r) - w);
      a0 = r - w;
    }
  }
  double a1 = (r - a0) - w;
  if (x < 0) { *y0 = -a0; *y1 = -a1; return -n; }
  *y0 = a0; *y1 = a1;
  return n;
}

__device__ inline void portable_sincos(double x, double* s, double* c) {
  if (fabs(x) <= 0.78539816339744830962) { *s = k_sin(x, 0.0, 0); *c = k_cos(x, 0.0); return; }
  double y0, y1;
  int n = rem_pio2(x, &y0, &y1);
  double sn = k_sin(y0, y1, 1), cs = k_cos(y0, y1);
  switch (n & 3) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
  }
}

}  // namespace psc
}  // namespace cgmr
#endif
