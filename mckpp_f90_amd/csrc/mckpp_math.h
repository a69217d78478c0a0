// mckpp_math.h - arithmetic shared by host set-up code and device kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

// exp(x) built from +,-,*,/ and integer operations only (Cody-Waite argument
// reduction, rational kernel on |r| <= ln2/2), |error| < 1 ulp.  The library
// uses this one implementation everywhere an EXP appears on the path -
// host-built Jerlov tables (src/mckpp_physics_swfrac_mod.F90:36-41,
// src/mckpp_fluxes_mod.F90:121-137) and the in-kernel swfrac at -hbl
// (src/mckpp_physics_verticalmixing_bldepth_mod.F90:193) - so host and device
// agree to the bit and results do not depend on which libm/ocml is linked.
__host__ __device__ inline double mckpp_exp(double x)
{
  const double ln2hi = 6.93147180369123816490e-01;
  const double ln2lo = 1.90821492927058770002e-10;
  const double invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01;
  const double P2 = -2.77777777770155933842e-03;
  const double P3 = 6.61375632143793436117e-05;
  const double P4 = -1.65339022054652515390e-06;
  const double P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 709.0) return __builtin_huge_val();
  if (x < -745.0) return 0.0;
  double t = invln2 * x;
  int k = (int)(t + (x < 0.0 ? -0.5 : 0.5));
  double fk = (double)k;
  double hi = x - fk * ln2hi;
  double lo = fk * ln2lo;
  double r = hi - lo;
  double tt = r * r;
  double c = r - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
  double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  uint64_t b;
  double s;
  if (k >= -1021) {
    b = (uint64_t)(1023 + k) << 52;
    memcpy(&s, &b, 8);
    return y * s;
  }
  b = (uint64_t)(1023 + k + 1000) << 52;
  memcpy(&s, &b, 8);
  y = y * s;
  b = (uint64_t)(1023 - 1000) << 52;
  memcpy(&s, &b, 8);
  return y * s;
}
