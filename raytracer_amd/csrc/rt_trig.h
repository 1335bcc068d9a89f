// sin / cos for the two working precisions.
//   float : device library sinf / cosf (the fp32 path is held to a tolerance, not to bits).
//   double: the fdlibm 5.3 algorithm (Sun Microsystems' k_sin.c / k_cos.c / e_rem_pio2.c), which
//           is what the reference's JavaScript engine evaluates Math.sin / Math.cos with; the
//           fp64 parity build has to land on the same bits as the reference for photon
//           directions (rayphoton.js:134-136) and light samples (:1923-1925).  Arguments on the
//           hot path are in [0, 2*pi) or scene coordinates; beyond 2^19*pi/2 it falls back to
//           the device library.
#pragma once
#include "rt_math.h"

namespace fd {
RT_D int hi_word(double x) { return __double2hiint(x); }
RT_D double k_sin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  int ix = hi_word(x) & 0x7fffffff;
  if (ix < 0x3e400000) { if ((int)x == 0) return x; }
  double z = x * x, v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
RT_D double k_cos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  int ix = hi_word(x) & 0x7fffffff;
  if (ix < 0x3e400000) { if ((int)x == 0) return 1.0; }
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  if (ix < 0x3FD33333) return 1.0 - (0.5 * z - (z * r - x * y));
  double qx = (ix > 0x3fe90000) ? 0.28125 : __hiloint2double(ix - 0x00200000, 0);
  double hz = 0.5 * z - qx, a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}
// n (quadrant) and reduced argument y0 + y1; returns -999 outside the medium range
RT_D int rem_pio2(double x, double &y0, double &y1) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
               pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
               pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
               pio2_3t = 8.47842766036889956997e-32;
  int hx = hi_word(x), ix = hx & 0x7fffffff;
  if (ix <= 0x3fe921fb) { y0 = x; y1 = 0; return 0; }
  if (ix < 0x4002d97c) {
    double z;
    if (hx > 0) {
      z = x - pio2_1;
      if (ix != 0x3ff921fb) { y0 = z - pio2_1t; y1 = (z - y0) - pio2_1t; }
      else { z -= pio2_2; y0 = z - pio2_2t; y1 = (z - y0) - pio2_2t; }
      return 1;
    }
    z = x + pio2_1;
    if (ix != 0x3ff921fb) { y0 = z + pio2_1t; y1 = (z - y0) + pio2_1t; }
    else { z += pio2_2; y0 = z + pio2_2t; y1 = (z - y0) + pio2_2t; }
    return -1;
  }
  if (ix > 0x413921fb) return -999;
  double t = fabs(x);
  int n = (int)(t * invpio2 + 0.5);
  double fn = (double)n;
  double r = t - fn * pio2_1, w = fn * pio2_1t;
  // high words of n*pi/2, n = 1..32: when x's high word equals one of them, r - w may cancel
  const int npio2_hw[32] = {
    0x3FF921FB, 0x400921FB, 0x4012D97C, 0x401921FB, 0x401F6A7A, 0x4022D97C, 0x4025FDBB, 0x402921FB,
    0x402C463A, 0x402F6A7A, 0x4031475C, 0x4032D97C, 0x40346B9C, 0x4035FDBB, 0x40378FDB, 0x403921FB,
    0x403AB41B, 0x403C463A, 0x403DD85A, 0x403F6A7A, 0x40407E4C, 0x4041475C, 0x4042106C, 0x4042D97C,
    0x4043A28C, 0x40446B9C, 0x404534AC, 0x4045FDBB, 0x4046C6CB, 0x40478FDB, 0x404858EB, 0x404921FB };
  bool near_multiple = (n < 32) && (ix == npio2_hw[n - 1]);
  y0 = r - w;
  if (!(n < 32 && !near_multiple)) {
    int j = ix >> 20;
    int i = j - ((hi_word(y0) >> 20) & 0x7ff);
    if (i > 16) {
      t = r; w = fn * pio2_2; r = t - w; w = fn * pio2_2t - ((t - r) - w); y0 = r - w;
      i = j - ((hi_word(y0) >> 20) & 0x7ff);
      if (i > 49) { t = r; w = fn * pio2_3; r = t - w; w = fn * pio2_3t - ((t - r) - w); y0 = r - w; }
    }
  }
  y1 = (r - y0) - w;
  if (hx < 0) { y0 = -y0; y1 = -y1; return -n; }
  return n;
}
}  // namespace fd

RT_D double rt_sin(double x) {
  int ix = fd::hi_word(x) & 0x7fffffff;
  if (ix <= 0x3fe921fb) return fd::k_sin(x, 0.0, 0);
  if (ix >= 0x7ff00000) return x - x;
  double y0, y1; int n = fd::rem_pio2(x, y0, y1);
  if (n == -999) return sin(x);
  switch (n & 3) {
    case 0: return fd::k_sin(y0, y1, 1);
    case 1: return fd::k_cos(y0, y1);
    case 2: return -fd::k_sin(y0, y1, 1);
    default: return -fd::k_cos(y0, y1);
  }
}
RT_D double rt_cos(double x) {
  int ix = fd::hi_word(x) & 0x7fffffff;
  if (ix <= 0x3fe921fb) return fd::k_cos(x, 0.0);
  if (ix >= 0x7ff00000) return x - x;
  double y0, y1; int n = fd::rem_pio2(x, y0, y1);
  if (n == -999) return cos(x);
  switch (n & 3) {
    case 0: return fd::k_cos(y0, y1);
    case 1: return -fd::k_sin(y0, y1, 1);
    case 2: return -fd::k_cos(y0, y1);
    default: return fd::k_sin(y0, y1, 1);
  }
}
RT_D float rt_sin(float x) { return sinf(x); }
RT_D float rt_cos(float x) { return cosf(x); }
