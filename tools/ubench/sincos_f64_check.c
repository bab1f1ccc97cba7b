/* CPU check of the fp64 sin/cos of rbdreference_amd/csrc/rbd_spatial.h (same arithmetic, gcc -O2 -mfma ... -lm): max abs error against long double. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static void sincos_core_d(double q, double* s, double* c) {
  const double kf = __builtin_rint(q * 6.36619772367581382433e-01);
  double r = __builtin_fma(-kf, 1.57079632673412561417e+00, q);
  r = __builtin_fma(-kf, 6.07710050630396597660e-11, r);
  r = __builtin_fma(-kf, 2.02226624879595063154e-21, r);
  const long long k = (long long)kf;
  const double z = r * r;
  double sp = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  sp = __builtin_fma(sp, z, 2.75573137070700676789e-06);
  sp = __builtin_fma(sp, z, -1.98412698298579493134e-04);
  sp = __builtin_fma(sp, z, 8.33333333332248946124e-03);
  sp = __builtin_fma(sp, z, -1.66666666666666324348e-01);
  sp = __builtin_fma(sp * z, r, r);
  double cp = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  cp = __builtin_fma(cp, z, -2.75573143513906633035e-07);
  cp = __builtin_fma(cp, z, 2.48015872894767294178e-05);
  cp = __builtin_fma(cp, z, -1.38888888888741095749e-03);
  cp = __builtin_fma(cp, z, 4.16666666666666019037e-02);
  cp = __builtin_fma(cp * z, z, __builtin_fma(-0.5, z, 1.0));
  uint64_t us, uc; memcpy(&us, &sp, 8); memcpy(&uc, &cp, 8);
  const uint64_t m = (uint64_t)-(k & 1);
  uint64_t ss = (uc & m) | (us & ~m), cc = (us & m) | (uc & ~m);
  ss ^= ((uint64_t)k << 62) & 0x8000000000000000ull;
  cc ^= ((uint64_t)(k + 1) << 62) & 0x8000000000000000ull;
  memcpy(s, &ss, 8); memcpy(c, &cc, 8);
}
int main() {
  double worst = 0, worstq = 0; srand48(1);
  const double ranges[] = {1e-3, 3.2, 100.0, 8192.0, 1e6};
  for (int ri = 0; ri < 5; ++ri) {
    double w = 0, wq = 0;
    for (long i = 0; i < 4000000; ++i) {
      double q = (drand48() * 2 - 1) * ranges[ri];
      double s, c; sincos_core_d(q, &s, &c);
      long double sr = sinl((long double)q), cr = cosl((long double)q);
      double e = fmax(fabs((double)(s - sr)), fabs((double)(c - cr)));
      if (e > w) { w = e; wq = q; }
    }
    printf("|q| <= %-8g max abs err %.3e at q=%.17g\n", ranges[ri], w, wq);
  }
  /* exact multiples / edge cases */
  double s, c; double qs[] = {0.0, -0.0, M_PI/2, M_PI, -M_PI, 3*M_PI/2, 2*M_PI, 1e6, -1e6, 0.78539816339744828, 999999.99999};
  for (int i = 0; i < 11; ++i) { sincos_core_d(qs[i], &s, &c); printf("q=%.17g s err %.2e c err %.2e\n", qs[i], (double)(s - sinl(qs[i])), (double)(c - cosl(qs[i]))); }
  return 0;
}
