// rbd_spatial.h -- compile-time-specialised spatial algebra for one robot (gfx950 device code).
//
// Included by rbd_kernels.hip AFTER the generated model header (namespace rbdm: N, PARENT[],
// DEPTH[], JTYPE[], AXIS[], XT[][36], IM[][36], DAMPING[]).  Every loop over bodies, matrix rows
// or columns is a template-unrolled `sfor`, so that
//   * the kinematic tree is resolved at compile time (no runtime parent[] walks, all per-body state
//     is statically indexed and lives in VGPRs),
//   * structural zeros / +-1 of X_tree and of the spatial inertias cost nothing,
//   * the model constants become literal / SGPR operands of the FMAs.
// This replaces the reference's getter calls and dense 6x6 numpy products
// (/root/reference/RBDReference.py:570-596) and its sparse-S special cases mx1..mx6 (:77-147).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include "rbd_sincos.h"

namespace rbdk {
using namespace rbdm;

#define RBD_DEV __device__ __forceinline__

template <int I, int E, class F>
RBD_DEV void sfor(F&& f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, E>(static_cast<F&&>(f));
  }
}
// descending: E-1 ... I
template <int I, int E, class F>
RBD_DEV void sfor_down(F&& f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, E - 1>{});
    sfor_down<I, E - 1>(static_cast<F&&>(f));
  }
}

// Opaque copy: the compiler cannot see that launder(x) == x, so values recomputed from laundered
// inputs are NOT merged (CSE) with their first computation.  rnea_grad_kernel uses it to trade ~400
// cheap instructions (the v/a recursion) for ~170 VGPRs that would otherwise stay live.
RBD_DEV float launder(float x) { asm volatile("" : "+v"(x)); return x; }
RBD_DEV double launder(double x) { asm volatile("" : "+v"(x)); return x; }

// by-value select: `c ? x[i] : y[i]` on two lvalues is an lvalue conditional, which clang lowers to a
// select of ADDRESSES and thereby forces the arrays into scratch memory.
template <class T>
RBD_DEV T sel(bool c, T a, T b) { return c ? a : b; }

RBD_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RBD_DEV double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// 1 / d for a joint-space inertia D = S^T IA S (positive, well scaled: no denormal / overflow handling needed).  fp32: v_rcp_f32
// (1 ulp) + one Newton step, three dependent instructions instead of the ten of an IEEE division (v_div_scale x 2, v_rcp,
// four fma, v_div_fmas, v_div_fixup) on the critical path of every body step of the eight-lane recursions; NaN stays NaN.
// fp64 keeps the division.
RBD_DEV float rcp_inertia(float d) {
#ifdef RBD_EXP_TRUE_DIV
  return 1.0f / d;
#else
  const float r0 = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
#endif
}
RBD_DEV double rcp_inertia(double d) { return 1.0 / d; }

// ---- compile-time tree queries ---------------------------------------------------------------
constexpr bool is_anc_or_self(int a, int j) {
  while (j != -1) {
    if (j == a) return true;
    j = PARENT[j];
  }
  return false;
}
constexpr bool related(int i, int j) { return is_anc_or_self(i, j) || is_anc_or_self(j, i); }
// ancestor-or-self of body j that sits at depth d (d <= DEPTH[j])
constexpr int anc_at(int j, int d) {
  while (DEPTH[j] > d) j = PARENT[j];
  return j;
}
constexpr int subtree_size(int i) {
  int k = 0;
  for (int j = 0; j < N; ++j) k += is_anc_or_self(i, j) ? 1 : 0;
  return k;
}
constexpr bool has_child(int i) {
  for (int j = 0; j < N; ++j)
    if (PARENT[j] == i) return true;
  return false;
}

// ---- compile-time body constants ---------------------------------------------------------------
constexpr double cabs_(double x) { return x < 0 ? -x : x; }
constexpr double Et_(int j, int r, int c) { return XT[j][r * 6 + c]; }              // E_tree
constexpr double rx_(int j, int r, int c) {                                         // (r_tree)^x = -E^T B
  double s = 0;
  for (int m = 0; m < 3; ++m) s -= XT[j][m * 6 + r] * XT[j][(3 + m) * 6 + c];
  return s;
}
constexpr double rt_(int j, int k) { return k == 0 ? rx_(j, 2, 1) : k == 1 ? rx_(j, 0, 2) : rx_(j, 1, 0); }
constexpr double mass_(int j) { return IM[j][3 * 6 + 3]; }
constexpr double hb_(int j, int k) {   // h = m c from the top-right block H = h^x
  return k == 0 ? IM[j][2 * 6 + 3 + 1] : k == 1 ? IM[j][0 * 6 + 3 + 2] : IM[j][1 * 6 + 3 + 0];
}
constexpr double com_(int j, int k) { return hb_(j, k) / mass_(j); }
constexpr double Ic_(int j, int r, int c) {   // inertia about the centre of mass: Ibar + m (c c^T - |c|^2 1)
  const double cc = com_(j, 0) * com_(j, 0) + com_(j, 1) * com_(j, 1) + com_(j, 2) * com_(j, 2);
  return IM[j][r * 6 + c] + mass_(j) * (com_(j, r) * com_(j, c) - (r == c ? cc : 0.0));
}
constexpr bool rigid_inertia_(int j) {
  const double m = mass_(j);
  if (!(m > 0)) return false;
  const double tol = 1e-12 * (m > 1 ? m : 1);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      if (cabs_(IM[j][(3 + r) * 6 + 3 + c] - (r == c ? m : 0.0)) > tol) return false;          // m 1
      if (cabs_(IM[j][r * 6 + 3 + c] + IM[j][c * 6 + 3 + r]) > tol) return false;              // H skew
      if (cabs_(IM[j][(3 + r) * 6 + c] - IM[j][c * 6 + 3 + r]) > tol) return false;            // lower-left = H^T
    }
  return true;
}
constexpr int n_children_(int i) {
  int k = 0;
  for (int j = 0; j < N; ++j) k += (PARENT[j] == i) ? 1 : 0;
  return k;
}
// ---- constant 6x6 operators (X_tree, X_tree^T, I) ---------------------------------------------
struct MatXT {
  static constexpr double at(int j, int r, int c) { return XT[j][r * 6 + c]; }
};
struct MatXTt {
  static constexpr double at(int j, int r, int c) { return XT[j][c * 6 + r]; }
};
struct MatI {
  static constexpr double at(int j, int r, int c) { return IM[j][r * 6 + c]; }
};
template <class M, int J>
constexpr int nz_before(int r, int c) {
  int k = 0;
  for (int i = 0; i < c; ++i) k += (M::at(J, r, i) != 0.0) ? 1 : 0;
  return k;
}

// y = M_J x with compile-time sparsity; y must not alias x.
template <class M, int J, class T>
RBD_DEV void cmatvec(const T (&x)[6], T (&y)[6]) {
  sfor<0, 6>([&](auto R) {
    constexpr int r = decltype(R)::value;
    T acc = T(0);
    sfor<0, 6>([&](auto C) {
      constexpr int c = decltype(C)::value;
      constexpr double k = M::at(J, r, c);
      if constexpr (k != 0.0) {
        if constexpr (nz_before<M, J>(r, c) == 0) {
          if constexpr (k == 1.0) acc = x[c];
          else if constexpr (k == -1.0) acc = -x[c];
          else acc = T(k) * x[c];
        } else {
          if constexpr (k == 1.0) acc = acc + x[c];
          else if constexpr (k == -1.0) acc = acc - x[c];
          else acc = fma_(T(k), x[c], acc);
        }
      }
    });
    y[r] = acc;
  });
}

// Per-joint trig: revolute (s, c) = (sin q, cos q); prismatic (s, c) = (q, unused).
template <class T>
struct JTrig {
  T s, c;
};

// y = X_J(q) t   (motion vectors, parent -> child); may alias.
template <int J, class T>
RBD_DEV void joint_fwd(const JTrig<T>& g, const T (&t)[6], T (&y)[6]) {
  constexpr int k = AXIS[J], a = (k + 1) % 3, b = (k + 2) % 3;
  if constexpr (JTYPE[J] == 0) {
    T ya = fma_(g.c, t[a], g.s * t[b]), yb = fma_(g.c, t[b], -(g.s * t[a]));
    T la = fma_(g.c, t[3 + a], g.s * t[3 + b]), lb = fma_(g.c, t[3 + b], -(g.s * t[3 + a]));
    y[a] = ya; y[b] = yb; y[k] = t[k];
    y[3 + a] = la; y[3 + b] = lb; y[3 + k] = t[3 + k];
  } else {
    T la = fma_(g.s, t[b], t[3 + a]), lb = fma_(-g.s, t[a], t[3 + b]);
    y[0] = t[0]; y[1] = t[1]; y[2] = t[2];
    y[3 + a] = la; y[3 + b] = lb; y[3 + k] = t[3 + k];
  }
}
// y = X_J(q)^T t  (forces, child -> parent); may alias.
template <int J, class T>
RBD_DEV void joint_bwd(const JTrig<T>& g, const T (&t)[6], T (&y)[6]) {
  constexpr int k = AXIS[J], a = (k + 1) % 3, b = (k + 2) % 3;
  if constexpr (JTYPE[J] == 0) {
    T ya = fma_(g.c, t[a], -(g.s * t[b])), yb = fma_(g.c, t[b], g.s * t[a]);
    T la = fma_(g.c, t[3 + a], -(g.s * t[3 + b])), lb = fma_(g.c, t[3 + b], g.s * t[3 + a]);
    y[a] = ya; y[b] = yb; y[k] = t[k];
    y[3 + a] = la; y[3 + b] = lb; y[3 + k] = t[3 + k];
  } else {
    T ya = fma_(-g.s, t[3 + b], t[a]), yb = fma_(g.s, t[3 + a], t[b]);
    y[a] = ya; y[b] = yb; y[k] = t[k];
    y[3] = t[3]; y[4] = t[4]; y[5] = t[5];
  }
}
// y = X_J(q) x = X_joint(q) (X_tree x)
template <int J, class T>
RBD_DEV void xform(const JTrig<T>& g, const T (&x)[6], T (&y)[6]) {
  T t[6];
  cmatvec<MatXT, J>(x, t);
  joint_fwd<J>(g, t, y);
}
// y = X_J(q)^T x = X_tree^T (X_joint(q)^T x)
template <int J, class T>
RBD_DEV void xform_T(const JTrig<T>& g, const T (&x)[6], T (&y)[6]) {
  T t[6];
  joint_bwd<J>(g, x, t);
  cmatvec<MatXTt, J>(t, y);
}

// out = alpha * crm(v) S_J  (mxS / _mxS, RBDReference.py:56-75); structural zeros are written as 0.
template <int J, class T>
RBD_DEV void mxS(const T (&v)[6], T alpha, T (&out)[6]) {
  constexpr int k = AXIS[J], a = (k + 1) % 3, b = (k + 2) % 3;
  sfor<0, 6>([&](auto R) { out[decltype(R)::value] = T(0); });
  if constexpr (JTYPE[J] == 0) {
    out[a] = alpha * v[b];
    out[b] = -(alpha * v[a]);
    out[3 + a] = alpha * v[3 + b];
    out[3 + b] = -(alpha * v[3 + a]);
  } else {
    out[3 + a] = alpha * v[b];
    out[3 + b] = -(alpha * v[a]);
  }
}
// acc += alpha * crm(v) S_J   (touches only the structurally non-zero components)
template <int J, class T>
RBD_DEV void add_mxS(const T (&v)[6], T alpha, T (&acc)[6]) {
  constexpr int k = AXIS[J], a = (k + 1) % 3, b = (k + 2) % 3;
  if constexpr (JTYPE[J] == 0) {
    acc[a] = fma_(alpha, v[b], acc[a]);
    acc[b] = fma_(-alpha, v[a], acc[b]);
    acc[3 + a] = fma_(alpha, v[3 + b], acc[3 + a]);
    acc[3 + b] = fma_(-alpha, v[3 + a], acc[3 + b]);
  } else {
    acc[3 + a] = fma_(alpha, v[b], acc[3 + a]);
    acc[3 + b] = fma_(-alpha, v[a], acc[3 + b]);
  }
}
// S_J^T f  (component pick) and  x += alpha * S_J
template <int J, class T>
RBD_DEV T S_dot(const T (&f)[6]) {
  return f[(JTYPE[J] == 0 ? 0 : 3) + AXIS[J]];
}
template <int J, class T>
RBD_DEV void add_S(T alpha, T (&x)[6]) {
  x[(JTYPE[J] == 0 ? 0 : 3) + AXIS[J]] += alpha;
}

// r (+)= crf(v) b   (fxv, RBDReference.py:149-164)
template <bool ACC, class T>
RBD_DEV void fxv(const T (&v)[6], const T (&b)[6], T (&r)[6]) {
  T r0 = fma_(v[1], b[2], fma_(-v[2], b[1], fma_(v[4], b[5], -(v[5] * b[4]))));
  T r1 = fma_(v[2], b[0], fma_(-v[0], b[2], fma_(v[5], b[3], -(v[3] * b[5]))));
  T r2 = fma_(v[0], b[1], fma_(-v[1], b[0], fma_(v[3], b[4], -(v[4] * b[3]))));
  T r3 = fma_(v[1], b[5], -(v[2] * b[4]));
  T r4 = fma_(v[2], b[3], -(v[0] * b[5]));
  T r5 = fma_(v[0], b[4], -(v[1] * b[3]));
  if constexpr (ACC) {
    r[0] += r0; r[1] += r1; r[2] += r2; r[3] += r3; r[4] += r4; r[5] += r5;
  } else {
    r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3; r[4] = r4; r[5] = r5;
  }
}

template <class T>
RBD_DEV T dot6(const T (&x)[6], const T (&y)[6]) {
  return fma_(x[5], y[5], fma_(x[4], y[4], fma_(x[3], y[3], fma_(x[2], y[2], fma_(x[1], y[1], x[0] * y[0])))));
}

// sin / cos of a joint angle: rbd_sincos.h -- own fast paths (Cody-Waite + minimax polynomials, bit-arithmetic quadrant
// fix-up) and an own BRANCH-FREE wide path (Payne-Hanek for huge angles, NaN for non-finite ones) behind a wave-uniform
// branch.  No libm / ocml routine is called anywhere in the device code: their lane-masked if / else bodies are where
// the round-3 aperture fault came from (the header tells the story; tools/isa_exec_audit.py checks the ISA).
using rbdsc::sincos_;
using rbdsc::sincos_core_;
using rbdsc::sincos_wide_;

template <int J, class T>
RBD_DEV JTrig<T> make_trig(T q) {
  JTrig<T> g;
  if constexpr (JTYPE[J] == 0) {
    sincos_(q, &g.s, &g.c);
  } else {
    g.s = q;
    g.c = T(0);
  }
  return g;
}

// One RNEA forward step for body J (RBDReference.py:569-596): given the parent's v, a produces
// xv = X v_p, xa = X a_p (X a0 at a root), v_J, a_J and the local force f_J.
template <int J, bool HAS_QDD, class T>
RBD_DEV void rnea_fwd_body(const JTrig<T>& g, T qd, T qdd, T grav, const T (&vp)[6], const T (&ap)[6],
                           T (&xv)[6], T (&xa)[6], T (&v)[6], T (&a)[6], T (&f)[6]) {
  if constexpr (PARENT[J] < 0) {
    // v_base = 0; a_base = [0,0,0,0,0,-GRAVITY]  (:565-566, :578)
    sfor<0, 6>([&](auto R) { xv[decltype(R)::value] = T(0); });
    T a0[6] = {T(0), T(0), T(0), T(0), T(0), -grav};
    xform<J>(g, a0, xa);
  } else {
    xform<J>(g, vp, xv);
    xform<J>(g, ap, xa);
  }
  sfor<0, 6>([&](auto R) {
    constexpr int r = decltype(R)::value;
    v[r] = xv[r];
    a[r] = xa[r];
  });
  add_S<J>(qd, v);             // v += S qd              (:586-587)
  add_mxS<J>(v, qd, a);        // a += crm(v) (S qd)     (:588)
  if constexpr (HAS_QDD) add_S<J>(qdd, a);  // (:589-593)
  T Iv[6], Ia[6];
  cmatvec<MatI, J>(v, Iv);
  cmatvec<MatI, J>(a, Ia);
  sfor<0, 6>([&](auto R) { f[decltype(R)::value] = Ia[decltype(R)::value]; });
  fxv<true>(v, Iv, f);         // f = I a + crf(v) I v   (:595-596)
}

// Rows R0 .. R0 + RN - 1 of Minv for NCFG configurations, from an LDS tile that holds only the group's OWN columns
// ([cfg][r * RN + c], configuration stride TS) to HBM as 16-byte pieces (4 floats or 2 doubles; the other groups' columns are structural
// zeros).  FULL blocks: thread t owns piece r4 = t of a configuration's RN * N / 4 pieces -- its four tile offsets and
// "own column" flags are computed ONCE, and every pass is four LDS reads, four selects and one store.  (The generic
// loop  g = tid, tid + NT, ...  divides by RN * N / 4 and by N four times in every iteration: 50 instructions per piece,
// a fifth of the one-launch kernel's instruction stream and 6 of its 29 us on the 30-body robot.)
// TRI: the tile holds only the UPPER TRIANGLE of the group's symmetric block, packed row-major (tri_off), TS = its stride;
// the mirror image (:799-804) is generated here -- or zeros below the diagonal when `dense` == 0.
RBD_DEV constexpr int tri_off(int a, int b, int rn) { return a * rn - a * (a - 1) / 2 + (b - a); }     // a <= b
template <class T, int R0, int RN, int NCFG, int TS, int NT, bool TRI = false>
RBD_DEV void minv_own_rows_flush(const T* tile, T* gdst, int tid, int nvalid, int dense = 1) {
  constexpr int VE = 16 / (int)sizeof(T);
  static_assert((RN * N) % VE == 0, "16-byte pieces");
  typedef T V __attribute__((ext_vector_type(VE)));
  constexpr int RV = RN * N / VE;
  if constexpr (RV <= NT) {
    if (nvalid == NCFG) {
      constexpr int K = NT / RV;                          // configurations per pass
      constexpr int PASSES = (NCFG + K - 1) / K;
      const int sub = tid / RV, r4 = tid - sub * RV;
      if (sub < K) {
        int off[VE];
        bool own[VE];
        sfor<0, VE>([&](auto I_) {
          constexpr int i = decltype(I_)::value;
          const int e = VE * r4 + i;
          const int r = e / N;
          const int cidx = e - r * N - R0;
          own[i] = cidx >= 0 && cidx < RN;
          if constexpr (TRI) {
            const int cx = own[i] ? cidx : r;
            off[i] = cx >= r ? tri_off(r, cx, RN) : tri_off(cx, r, RN);
            own[i] = own[i] && (dense != 0 || cx >= r);
          } else {
            off[i] = r * RN + (own[i] ? cidx : 0);
          }
        });
        V buf[PASSES];
        sfor<0, PASSES>([&](auto P_) {
          constexpr int p = decltype(P_)::value;
          const int cfg = p * K + sub;
          const int cc = (p + 1) * K <= NCFG ? cfg : (cfg < NCFG ? cfg : NCFG - 1);
          sfor<0, VE>([&](auto I_) { constexpr int i = decltype(I_)::value; buf[p][i] = tile[cc * TS + off[i]]; });
        });
        sfor<0, PASSES>([&](auto P_) {
          constexpr int p = decltype(P_)::value;
          const int cfg = p * K + sub;
          V x;
          sfor<0, VE>([&](auto I_) { constexpr int i = decltype(I_)::value; x[i] = own[i] ? buf[p][i] : T(0); });
          if ((p + 1) * K <= NCFG || cfg < NCFG) reinterpret_cast<V*>(gdst + (long long)cfg * (N * N))[r4] = x;
        });
      }
      return;
    }
  }
  auto elem = [&](int cfg, int e) -> T {                 // (row e / N, column e % N): own columns from the tile, the rest zero
    const int r = e / N;
    const int cidx = e - r * N - R0;
    bool own = cidx >= 0 && cidx < RN;
    int o;
    if constexpr (TRI) {
      const int cx = own ? cidx : r;
      o = cx >= r ? tri_off(r, cx, RN) : tri_off(cx, r, RN);
      own = own && (dense != 0 || cx >= r);
    } else {
      o = r * RN + (own ? cidx : 0);
    }
    const T x = tile[cfg * TS + o];
    return own ? x : T(0);
  };
  const int total = nvalid * RV;
#pragma unroll 2
  for (int g = tid; g < total; g += NT) {
    const int cfg = g / RV;
    const int r4 = g - cfg * RV;
    V x;
    sfor<0, VE>([&](auto I_) { constexpr int i = decltype(I_)::value; x[i] = elem(cfg, VE * r4 + i); });
    reinterpret_cast<V*>(gdst + (long long)cfg * (N * N))[r4] = x;
  }
}

// raw-buffer descriptor of an output region (base must be wave-uniform; it is pinned in SGPRs here)
template <class T>
RBD_DEV __amdgpu_buffer_rsrc_t out_tile_rsrc(T* base, int bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}

// FULL tile of NCFG configurations: RW contiguous scalars per configuration (LDS stride TS) -> HBM (stride DST) as 16-byte
// pieces by NT threads.  Piece g = tid + NT k of the NCFG x RV pieces is (configuration g / RV, piece g % RV); with
// tid = RV a + b the quotient of every step follows from compile-time constants and ONE comparison -- the plain loop
// "for (g = tid; g < nvalid * RV; g += NT) { cfg = g / RV; ... }" divides in every iteration and builds a 64-bit address
// per store: 12-20 instructions per piece, a sixth (two-lane gradient kernel) to a half (one-lane minv kernel) of the
// quadruped's instruction stream.  The reads of a batch are issued before its stores; the stores go through a
// raw-buffer descriptor of the block's output region.  Caller guarantees: full tile, RW % VE == 0, DST % VE == 0, the
// group's rows start on a 16-byte boundary, (NCFG * RV) % NT == 0.
template <class T, int NCFG, int RW, int TS, int DST, int NT>
RBD_DEV void flush_cfg_rows_full(const T* tile, T* gdst, int tid) {
  constexpr int VE = 16 / (int)sizeof(T);
  constexpr int RV = RW / VE;
  static_assert(RW % VE == 0 && DST % VE == 0 && (NCFG * RV) % NT == 0, "flush_cfg_rows_full: shape");
  typedef T V __attribute__((ext_vector_type(VE)));
  typedef unsigned U4 __attribute__((ext_vector_type(4)));
  constexpr int IT = NCFG * RV / NT;
  constexpr int BATCH = IT < 9 ? IT : 9;
  const __amdgpu_buffer_rsrc_t rs = out_tile_rsrc(gdst, (int)(((NCFG - 1) * DST + RW) * sizeof(T)));
  const int a = tid / RV, bq = tid - a * RV;
  sfor<0, (IT + BATCH - 1) / BATCH>([&](auto G_) {
    constexpr int g0 = decltype(G_)::value * BATCH;
    V buf[BATCH];
    int goff[BATCH];
    sfor<0, BATCH>([&](auto I_) {
      constexpr int i = decltype(I_)::value, k = g0 + i;
      if constexpr (k < IT) {
        constexpr int qk = (NT * k) / RV, rk = (NT * k) % RV;
        const int t = bq + rk;
        const bool wrap = t >= RV;
        const int cfg = a + qk + (wrap ? 1 : 0);
        const int r = wrap ? t - RV : t;
        if constexpr (TS % VE == 0) {
          buf[i] = *reinterpret_cast<const V*>(tile + cfg * TS + r * VE);
        } else {
          sfor<0, VE>([&](auto E_) { constexpr int e = decltype(E_)::value; buf[i][e] = tile[cfg * TS + r * VE + e]; });
        }
        goff[i] = (cfg * DST + r * VE) * (int)sizeof(T);
      }
    });
    sfor<0, BATCH>([&](auto I_) {
      constexpr int i = decltype(I_)::value, k = g0 + i;
      if constexpr (k < IT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(U4, buf[i]), rs, goff[i], 0, 0);
    });
  });
}

}  // namespace rbdk
