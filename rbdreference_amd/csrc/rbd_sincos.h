// rbd_sincos.h -- sin / cos of a joint angle for every kernel of the package, WITHOUT lane-divergent control flow and
// without any libm / ocml routine (the reference evaluates Xmat(q) = f(sin q, cos q) for any q in fp64,
// /root/reference/RBDReference.py:562-564, :574).
//
// Why there is no library call in here (round 3 fault, round 4 root cause; docs/NOTEBOOK.md 3.1 a' (xv)):
// `sincos(double)` / `sincosf(float)` of the device library carry a lane-masked `if (|x| large) {Payne-Hanek} else {..}`.
// In register-starved fp64 kernels (AGPRs in use) the register allocator put a live-range copy of a CALLER value
// (`v_accvgpr_write_b32 a1, v73`, the high half of a 64-bit address offset) at the head of that ELSE block, BEFORE the
// instruction that restores EXEC there -- when no lane takes the IF side the copy runs with EXEC = 0, does nothing,
// and the kernel later reads through a garbage pointer (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION).  Round 3 routed
// |q| <= 1e6 around the library; the library body stayed inlined for larger / non-finite angles.  This file removes
// the cause: every path below is straight-line code (selects are v_cndmask / bit arithmetic), so no if / ELSE lowering
// exists in any kernel on account of sin / cos; tools/isa_exec_audit.py checks the ISA for exactly that.
//
//   sincos_core_(T)   |q| <= 8192 (fp32) / 1e6 (fp64): Cody-Waite reduction + minimax polynomials (the fast path).
//   sincos_wide_(T)   ANY q: finite angles of any magnitude through a branch-free Payne-Hanek reduction (1216 bits of
//                     2/pi, 192-bit window selected by the exponent), NaN / +-Inf -> NaN; every lane computes both
//                     reductions and selects.  fp32 goes through the fp64 routine.
//   sincos_(T)        WAVE-UNIFORM dispatch (ballot): the wide routine runs only in waves where some lane needs it.
//
// The same header compiles on the host (gcc / clang, -DRBD_SINCOS_HOST) for tools/ubench/sincos_wide_check.c, which
// holds it to glibc's sinl / cosl over 40 exponent ranges and the known worst cases of the reduction.
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef RBD_SINCOS_HOST
#define RBD_SC_FN static inline
#define RBD_SC_TABLE static const
#else
#include <hip/hip_runtime.h>
#define RBD_SC_FN __device__ __forceinline__
#define RBD_SC_TABLE __device__ static const
#endif

namespace rbdsc {

RBD_SC_FN uint64_t bits_(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
RBD_SC_FN double dbl_(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
RBD_SC_FN uint32_t bits_(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
RBD_SC_FN float flt_(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

// ---- fp32 fast path: |q| <= 8192 ------------------------------------------------------------------------------------
// two-constant Cody-Waite reduction by pi/2 (exact under FMA) and the classic degree-7 / degree-8 minimax polynomials on
// [-pi/4, pi/4]: max abs error 9.2e-8 over the range; quadrant fix-up in bit arithmetic (tools/ubench/pk_issue.hip: a
// v_cndmask_b32 costs a wave 6.4-16.7 cycles, a v_cmp 9, a plain VOP2 instruction 5.2-5.7).
RBD_SC_FN void sincos_core_(float q, float* s, float* c) {
  const float kf = __builtin_rintf(q * 0.63661977236758134f);                 // q * 2 / pi
  float r = __builtin_fmaf(-kf, 1.5707963705062866f, q);                      // pi/2 = hi + mid (+ 1.8e-15)
  r = __builtin_fmaf(-kf, -4.371138828673793e-08f, r);
  const int k = (int)kf;
  const float z = r * r;
  float sp = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = __builtin_fmaf(sp, z, -1.6666654611e-1f);
  sp = __builtin_fmaf(sp * z, r, r);
  float cp = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = __builtin_fmaf(cp, z, 4.166664568298827e-2f);
  cp = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
  const uint32_t us = bits_(sp), uc = bits_(cp);
  const uint32_t m = (uint32_t)-(k & 1);                                      // all ones when k is odd: sin and cos trade places
  uint32_t ss = (uc & m) | (us & ~m);
  uint32_t cc = (us & m) | (uc & ~m);
  ss ^= ((uint32_t)k << 30) & 0x80000000u;                                    // sin changes sign in quadrants 2, 3
  cc ^= ((uint32_t)(k + 1) << 30) & 0x80000000u;                              // cos in quadrants 1, 2
  *s = flt_(ss);
  *c = flt_(cc);
}

// ---- fp64 kernels on [-pi/4, pi/4] (the classic libm minimax polynomials, degree 13 / 14) ----------------------------
RBD_SC_FN void poly_sincos_(double r, double* sp_, double* cp_) {
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
  *sp_ = sp;
  *cp_ = cp;
}
// (sin, cos) of k * pi/2 + r from the kernels' values at r: swap for odd k, signs by quadrant; bit arithmetic only.
RBD_SC_FN void quadrant_(double sp, double cp, unsigned k, double* s, double* c) {
  const uint64_t us = bits_(sp), uc = bits_(cp);
  const uint64_t m = (uint64_t)-(int64_t)(k & 1u);
  uint64_t ss = (uc & m) | (us & ~m);
  uint64_t cc = (us & m) | (uc & ~m);
  ss ^= ((uint64_t)k << 62) & 0x8000000000000000ull;
  cc ^= ((uint64_t)(k + 1u) << 62) & 0x8000000000000000ull;
  *s = dbl_(ss);
  *c = dbl_(cc);
}

// ---- fp64 fast path: |q| <= 1e6 -------------------------------------------------------------------------------------
// three-constant Cody-Waite reduction (pi/2 = 33 + 33 + 53 bits: k * P1 and k * P2 are exact for |k| < 2^20); max abs
// error 2.1e-16 against long double (tools/ubench/sincos_f64_check.c).
RBD_SC_FN void sincos_core_(double q, double* s, double* c) {
  const double kf = __builtin_rint(q * 6.36619772367581382433e-01);           // q * 2 / pi
  double r = __builtin_fma(-kf, 1.57079632673412561417e+00, q);
  r = __builtin_fma(-kf, 6.07710050630396597660e-11, r);
  r = __builtin_fma(-kf, 2.02226624879595063154e-21, r);
  const int k = (int)kf;
  double sp, cp;
  poly_sincos_(r, &sp, &cp);
  quadrant_(sp, cp, (unsigned)k, s, c);
}

// ---- any q: branch-free Payne-Hanek ----------------------------------------------------------------------------------
// TWO_OVER_PI[0] = 0 (64 pad bits), then 1216 bits of 2/pi, most significant first (computed with integer Machin
// arithmetic; the first words are the familiar A2F9836E 4E441529 FC2757D1 ...).
RBD_SC_TABLE uint64_t TWO_OVER_PI[20] = {
    0x0000000000000000ull, 0xa2f9836e4e441529ull, 0xfc2757d1f534ddc0ull, 0xdb6295993c439041ull, 0xfe5163abdebbc561ull,
    0xb7246e3a424dd2e0ull, 0x06492eea09d1921cull, 0xfe1deb1cb129a73eull, 0xe88235f52ebb4484ull, 0xe99c7026b45f7e41ull,
    0x3991d639835339f4ull, 0x9c845f8bbdf9283bull, 0x1ff897ffde05980full, 0xef2f118b5a0a6d1full, 0x6d367ecf27cb09b7ull,
    0x4f463f669e5fea2dull, 0x7527bac7ebe5f17bull, 0x3d0739f78a5292eaull, 0x6bfb5fb11f8d5d08ull, 0x56033046fc7b6babull};

RBD_SC_FN uint64_t mulhi64_(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

// |x| = m64 * 2^(e - 63) with m64 = mantissa << 11.  With the window V = bits [e + 10, e + 202) of the padded stream
// (every bit in front of it contributes a multiple of 4 quadrants: dropped), |x| * 2/pi = (m64 * V) * 2^-201 (mod 4),
// truncation error < 2^-137 quadrants -- the closest a double comes to a multiple of pi/2 is 2^-61.
// Returns the quadrant and r = hi + lo, |r| <= pi/4.  Meaningful for |x| >= 2^19; smaller / non-finite x give finite
// garbage (their lanes are discarded by the caller), never a trap or an out-of-range table index.
RBD_SC_FN unsigned payne_hanek_(uint64_t ax, double* r_hi, double* r_lo) {
  int e = (int)(ax >> 52) - 1023;
  e = e < 0 ? 0 : (e > 1023 ? 1023 : e);
  const uint64_t m64 = ((ax & 0x000fffffffffffffull) | 0x0010000000000000ull) << 11;
  const int sft = e + 10, j = sft >> 6, r = sft & 63;
  const uint64_t w0 = TWO_OVER_PI[j], w1 = TWO_OVER_PI[j + 1], w2 = TWO_OVER_PI[j + 2], w3 = TWO_OVER_PI[j + 3];
  const uint64_t v0 = (w0 << r) | ((w1 >> 1) >> (63 - r));
  const uint64_t v1 = (w1 << r) | ((w2 >> 1) >> (63 - r));
  const uint64_t v2 = (w2 << r) | ((w3 >> 1) >> (63 - r));
  // words 1..3 of the 256-bit product m64 * (v0 : v1 : v2)  (word 0 never carries into them)
  const uint64_t hi0 = mulhi64_(m64, v2);
  const uint64_t lo1 = m64 * v1, hi1 = mulhi64_(m64, v1);
  const uint64_t lo2 = m64 * v0, hi2 = mulhi64_(m64, v0);
  const uint64_t word1 = hi0 + lo1;
  const uint64_t c1 = word1 < hi0 ? 1u : 0u;
  const uint64_t t2 = hi1 + lo2;
  const uint64_t word2 = t2 + c1;
  const uint64_t c2 = (t2 < hi1 ? 1u : 0u) + (word2 < t2 ? 1u : 0u);
  const uint64_t word3 = hi2 + c2;
  // bits 202..201 = quadrant, 200..0 = fraction: shift left by 53 so that they sit at the top
  const uint64_t H = (word3 << 53) | (word2 >> 11);
  const uint64_t M = (word2 << 53) | (word1 >> 11);
  const unsigned k = (unsigned)(H >> 62) + (unsigned)((H >> 61) & 1u);       // round to the nearest quadrant
  const int64_t Fh = (int64_t)((H << 2) | (M >> 62));                          // signed 128-bit fraction in [-1/2, 1/2) * 2^128
  const uint64_t Fl = M << 2;
  const uint64_t neg = (uint64_t)(Fh >> 63);                                   // all ones when negative
  const uint64_t ml = (Fl ^ neg) + (neg & 1u);
  const uint64_t mh = ((uint64_t)Fh ^ neg) + ((neg & 1u) & (ml == 0 ? 1u : 0u));
  // magnitude -> double-double
  const double dh = (double)mh;                                                // <= 2^63
  const double dl = (double)(int64_t)(mh - (uint64_t)dh) + (double)ml * 5.42101086242752217004e-20;   // 2^-64
  const double t = dh * 5.42101086242752217004e-20, tl = dl * 5.42101086242752217004e-20;
  const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
  double rh = t * PIO2_HI;
  double rl = __builtin_fma(t, PIO2_LO, __builtin_fma(tl, PIO2_HI, __builtin_fma(t, PIO2_HI, -rh)));
  const uint64_t sgn = neg & 0x8000000000000000ull;
  *r_hi = dbl_(bits_(rh) ^ sgn);
  *r_lo = dbl_(bits_(rl) ^ sgn);
  return k;
}

RBD_SC_FN void sincos_wide_(double q, double* s, double* c) {
  const uint64_t uq = bits_(q), ax = uq & 0x7fffffffffffffffull;
  // (a) the fast reduction, right for |q| <= 1e6 (finite garbage elsewhere)
  double s1, c1;
  sincos_core_(dbl_(ax <= 0x412e848000000000ull ? uq : 0ull), &s1, &c1);
  // (b) Payne-Hanek on |q|, right for |q| >= 2^19
  double rh, rl, sp, cp, s2, c2;
  const unsigned k = payne_hanek_(ax, &rh, &rl);
  poly_sincos_(rh, &sp, &cp);
  const double sp2 = __builtin_fma(rl, cp, sp), cp2 = __builtin_fma(-rl, sp, cp);     // first order in the low word
  quadrant_(sp2, cp2, k, &s2, &c2);
  s2 = dbl_(bits_(s2) ^ (uq & 0x8000000000000000ull));                          // sin(-x) = -sin(x)
  const bool small = ax <= 0x412e848000000000ull;                               // |q| <= 1e6 (NaN / Inf compare false)
  const bool finite = ax < 0x7ff0000000000000ull;
  const double nan = dbl_(0x7ff8000000000000ull);
  const double so = small ? s1 : s2, co = small ? c1 : c2;
  *s = finite ? so : nan;
  *c = finite ? co : nan;
}
RBD_SC_FN void sincos_wide_(float q, float* s, float* c) {
  // lanes inside the fast range keep the fast routine's value, so a row's result does not depend on its neighbours
  const bool small = (bits_(q) & 0x7fffffffu) <= 0x46000000u;                   // |q| <= 8192
  float s1, c1;
  sincos_core_(small ? q : 0.0f, &s1, &c1);
  double sd, cd;
  sincos_wide_((double)q, &sd, &cd);
  *s = small ? s1 : (float)sd;
  *c = small ? c1 : (float)cd;
}

#ifndef RBD_SINCOS_HOST
// WAVE-UNIFORM dispatch: a wave leaves the fast path only when one of its active lanes holds an angle beyond the
// fast range or a non-finite one; then ALL its lanes run the wide routine (which is right for every q).
__device__ __forceinline__ void sincos_(float q, float* s, float* c) {
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(__builtin_fabsf(q) <= 8192.0f)) != 0, 0)) { sincos_wide_(q, s, c); return; }
  sincos_core_(q, s, c);
}
__device__ __forceinline__ void sincos_(double q, double* s, double* c) {
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(__builtin_fabs(q) <= 1.0e6)) != 0, 0)) { sincos_wide_(q, s, c); return; }
  sincos_core_(q, s, c);
}
#endif

}  // namespace rbdsc
