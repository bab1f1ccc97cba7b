/* CPU check of rbdreference_amd/csrc/rbd_sincos.h (the same header, compiled for the host):
     g++ -O2 -mfma -DRBD_SINCOS_HOST -x c++ tools/ubench/sincos_wide_check.c -o /tmp/sincos_wide_check -lm && /tmp/sincos_wide_check
   sincos_wide_ (branch-free Payne-Hanek + fast path + NaN/Inf) against glibc sinl / cosl (x87 long double, argument
   reduction with the full 2/pi) over every binade 2^-30 .. 2^1023, the classic worst cases of the reduction, and the
   special values. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../rbdreference_amd/csrc/rbd_sincos.h"
using namespace rbdsc;
int main() {
  srand48(7);
  double worst = 0, worstq = 0;
  for (int e = -30; e <= 1023; ++e) {
    double w = 0, wq = 0;
    const int n = (e >= 15 && e <= 60) ? 200000 : 20000;
    for (int i = 0; i < n; ++i) {
      double q = ldexp(1.0 + drand48(), e) * (drand48() < 0.5 ? -1 : 1);
      double s, c; sincos_wide_(q, &s, &c);
      long double sr = sinl((long double)q), cr = cosl((long double)q);
      double er = fmax(fabs((double)(s - sr)), fabs((double)(c - cr)));
      if (!(er <= w)) { w = er; wq = q; }
    }
    if (w > worst) { worst = w; worstq = wq; }
    if (e % 64 == 0 || e == 1023 || e == 19 || e == 20) printf("binade 2^%-5d max abs err %.3e at q=%.17g\n", e, w, wq);
  }
  printf("ALL binades: max abs err %.3e at q=%.17g\n", worst, worstq);
  /* worst cases for double argument reduction (Muller et al.), multiples of pi/2, fast-range edge */
  const double hard[] = {0x1.6ac5b262ca1ffp+849 /* 6381956970095103 * 2^797 */, 0x1.921fb54442d18p+0, 0x1.921fb54442d18p+1, 1e6, 1.0000001e6, -1e6,
                         -1.0000001e6, 3e6, -3e6, 1e22, 1e300, 0x1.fffffffffffffp+1023, 5e5, 524288.0, 1e-300, 0.0, -0.0, 4.9e-324,
                         0x1.d130f68f3bb7ap+600, 0x1.2a5f3e8p+40};
  for (unsigned i = 0; i < sizeof(hard) / sizeof(hard[0]); ++i) {
    double s, c; sincos_wide_(hard[i], &s, &c);
    printf("q=%-24.17g s=%+.17e (err %.2e) c=%+.17e (err %.2e)\n", hard[i], s, (double)(s - sinl(hard[i])), c, (double)(c - cosl(hard[i])));
  }
  const double bad[] = {NAN, INFINITY, -INFINITY};
  int ok = 1;
  for (int i = 0; i < 3; ++i) { double s, c; sincos_wide_(bad[i], &s, &c); printf("q=%g -> s=%g c=%g\n", bad[i], s, c); ok &= isnan(s) && isnan(c); }
  /* fp32 wrapper */
  float wf = 0, wfq = 0;
  for (int e = -20; e <= 127; ++e)
    for (int i = 0; i < 20000; ++i) {
      float q = (float)ldexp(1.0 + drand48(), e) * (drand48() < 0.5 ? -1 : 1);
      if (!isfinite(q)) continue;
      float s, c; sincos_wide_(q, &s, &c);
      float er = fmaxf(fabsf(s - (float)sinl(q)), fabsf(c - (float)cosl(q)));
      if (!(er <= wf)) { wf = er; wfq = q; }
    }
  printf("fp32 wrapper: max abs err %.3e at q=%.9g\n", wf, wfq);
  { float s, c; sincos_wide_(NAN, &s, &c); ok &= isnan(s) && isnan(c); sincos_wide_(INFINITY, &s, &c); ok &= isnan(s) && isnan(c); }
  printf(ok && worst < 4e-16 && wf < 2e-7 ? "PASS\n" : "FAIL\n");
  return !(ok && worst < 4e-16 && wf < 2e-7);
}
