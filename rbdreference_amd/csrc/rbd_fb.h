// rbd_fb.h -- FLOATING-BASE rnea / minv / forward_dynamics, one configuration per lane.
//
// What the reference does when robot.floating_base is set (/root/reference/RBDReference.py:585-593,
// :652-691, :761-779): body 0 is attached to the world by a 6-DoF joint with S = eye(6) and owns
// q[0:6], qd[0:6] (qd[0:6] is the base twist in base coordinates); body i >= 1 owns index i + 5; the
// matrices have n = NB + 5 rows.  X_0(q[0:6]) = plux(Rz Ry Rx, p) (rbdreference_amd/robot.py,
// floating_base_X; the packer checks the robot's own Xmat against it).
//
//   rnea   (:559-628)  v_0 = qd[0:6], a_0 = X_0 a_grav + qdd[0:6] (crm(v_0) v_0 = 0), bodies >= 1 as
//                      usual; c[0:6] = f_0 (S = eye(6), :612), c[i + 5] = S_i^T f_i.
//   minv   (:630-806)  articulated inertias leaf -> base; the base is ONE 6 x 6 block:
//                      Minv[0:6, 0:6] = inv(IA_0) (:681-685), Minv[0:6, j] = -inv(IA_0) F_0[:, j] (:686-691);
//                      the forward pass starts from F_0 = Minv[0:6, :] (:779).  Per column j >= 6 (joint of
//                      body j - 5): its F vector climbs the root path, the base block gives rows 0..5,
//                      the forward sweep gives rows 6.. -- the reference's (n, 6, n) F tensor never exists.
//   forward_dynamics (:1371-1374)  Minv (u - c) by composition (rnea with qdd = None, minv, one product).
//   rnea_grad (:1345-1368)  the four passes with their floating-base branches; the base's six position
//                      columns are derivatives along a base-frame twist (crm(.) S with S = eye(6)).  The
//                      reference runs only when NB >= 6 (:1168 indexes bodies 0..5 and raises IndexError
//                      otherwise): the entry point reports smaller robots as unsupported.
// The reference's crba (:1063) and aba (:900) raise for floating bases: not part of this file.
//
// These are correctness-first kernels (a "next" row): one configuration per lane, scattered stores.
#pragma once
#include "rbd_spatial.h"

namespace rbdk {

constexpr int fb_s_index(int i) { return (JTYPE[i] == 0 ? 0 : 3) + AXIS[i]; }
constexpr unsigned long long fb_subtree_mask(int i) {
  unsigned long long m = 0;
  for (int j = 0; j < N; ++j) m |= is_anc_or_self(i, j) ? (1ull << j) : 0ull;
  return m;
}

// rotation part of the world -> base transform: E = Rz(rz) Ry(ry) Rx(rx), coordinate-transform rotations
template <class T>
RBD_DEV void fb_base_E(T rx, T ry, T rz, T (&E)[3][3]) {
  T sx, cx, sy, cy, sz, cz;
  sincos_(rx, &sx, &cx); sincos_(ry, &sy, &cy); sincos_(rz, &sz, &cz);
  // Ry Rx = [[cy, sy sx, -sy cx], [0, cx, sx], [sy, -cy sx, cy cx]]
  const T m00 = cy, m01 = sy * sx, m02 = -(sy * cx);
  const T m10 = T(0), m11 = cx, m12 = sx;
  const T m20 = sy, m21 = -(cy * sx), m22 = cy * cx;
  // Rz = [[cz, sz, 0], [-sz, cz, 0], [0, 0, 1]]
  E[0][0] = cz * m00 + sz * m10; E[0][1] = cz * m01 + sz * m11; E[0][2] = cz * m02 + sz * m12;
  E[1][0] = cz * m10 - sz * m00; E[1][1] = cz * m11 - sz * m01; E[1][2] = cz * m12 - sz * m02;
  E[2][0] = m20; E[2][1] = m21; E[2][2] = m22;
}

// ---------------------------------------------------------------------------------------------
// rnea: (q, qd, qdd) [B, NV] -> c [B, NV], v, a, f [B, 6, N] (f accumulated); v/a/f optional
// ---------------------------------------------------------------------------------------------
template <class T, bool HAS_QDD>
__global__ __launch_bounds__(64, 1) void rnea_fb_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                        const T* __restrict__ qdd, T grav, long long B,
                                                        T* __restrict__ c_out, T* __restrict__ v_out,
                                                        T* __restrict__ a_out, T* __restrict__ f_out) {
  const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const T* qb = q + b * NV; const T* qdb = qd + b * NV; const T* qddb = HAS_QDD ? qdd + b * NV : nullptr;
  JTrig<T> tr[N];
  T qdv[N], qddv[N];
  sfor<1, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    tr[j] = make_trig<j>(qb[j + 5]);
    qdv[j] = qdb[j + 5];
    if constexpr (HAS_QDD) qddv[j] = qddb[j + 5]; else qddv[j] = T(0);
  });
  T v[N][6], a[N][6], f[N][6];
  {
    // the base (:576-596 with the floating-base lines :585, :591): v_0 = qd[0:6]; a_0 = X_0 a_grav + qdd[0:6]
    T E[3][3];
    fb_base_E(qb[3], qb[4], qb[5], E);
    sfor<0, 6>([&](auto R) {
      constexpr int r = decltype(R)::value;
      v[0][r] = qdb[r];
      a[0][r] = HAS_QDD ? qddb[r] : T(0);
    });
    // a_grav = (0,0,0,0,0,-GRAVITY): X_0 a_grav = (0; -GRAVITY * E[:, 2])   (the translation drops out: no angular part)
    a[0][3] -= grav * E[0][2]; a[0][4] -= grav * E[1][2]; a[0][5] -= grav * E[2][2];
    T Iv[6], Ia[6];
    cmatvec<MatI, 0>(v[0], Iv);
    cmatvec<MatI, 0>(a[0], Ia);
    sfor<0, 6>([&](auto R) { f[0][decltype(R)::value] = Ia[decltype(R)::value]; });
    fxv<true>(v[0], Iv, f[0]);
  }
  sfor<1, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int p = PARENT[j];
    T xv[6], xa[6];
    rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddv[j], grav, v[p], a[p], xv, xa, v[j], a[j], f[j]);
  });
  if (v_out != nullptr) {
    sfor<0, N>([&](auto J) {
      sfor<0, 6>([&](auto R) {
        constexpr int j = decltype(J)::value, r = decltype(R)::value;
        v_out[b * (6 * N) + r * N + j] = v[j][r];
        a_out[b * (6 * N) + r * N + j] = a[j][r];
      });
    });
  }
  // backward pass (:607-619)
  sfor_down<1, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int p = PARENT[j];
    c_out[b * NV + j + 5] = S_dot<j>(f[j]);
    T t[6];
    xform_T<j>(tr[j], f[j], t);
    sfor<0, 6>([&](auto R) { f[p][decltype(R)::value] += t[decltype(R)::value]; });
  });
  sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; c_out[b * NV + r] = f[0][r]; });   // S = eye(6)
  if (f_out != nullptr) {
    sfor<0, N>([&](auto J) {
      sfor<0, 6>([&](auto R) {
        constexpr int j = decltype(J)::value, r = decltype(R)::value;
        f_out[b * (6 * N) + r * N + j] = f[j][r];
      });
    });
  }
}

// 6 x 6 SPD inverse (Gauss-Jordan without pivoting: the articulated inertia of the whole robot is SPD)
template <class T>
RBD_DEV void fb_inv6(const T (&A)[6][6], T (&Ai)[6][6]) {
  T M[6][6], R[6][6];
  sfor<0, 6>([&](auto I) { sfor<0, 6>([&](auto J) { constexpr int i = decltype(I)::value, j = decltype(J)::value; M[i][j] = A[i][j]; R[i][j] = i == j ? T(1) : T(0); }); });
  sfor<0, 6>([&](auto K) {
    constexpr int k = decltype(K)::value;
    const T piv = T(1) / M[k][k];
    sfor<0, 6>([&](auto J) { constexpr int j = decltype(J)::value; M[k][j] *= piv; R[k][j] *= piv; });
    sfor<0, 6>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (i != k) {
        const T fct = M[i][k];
        sfor<0, 6>([&](auto J) { constexpr int j = decltype(J)::value; M[i][j] = fma_(-fct, M[k][j], M[i][j]); R[i][j] = fma_(-fct, R[k][j], R[i][j]); });
      }
    });
  });
  sfor<0, 6>([&](auto I) { sfor<0, 6>([&](auto J) { constexpr int i = decltype(I)::value, j = decltype(J)::value; Ai[i][j] = R[i][j]; }); });
}

// ---------------------------------------------------------------------------------------------
// minv: q [B, NV] -> Minv [B, NV, NV]
// ---------------------------------------------------------------------------------------------
// FB_MINV_L lanes per configuration: every lane runs the articulated-inertia recursion of its configuration
// (redundantly: about as much work as three joint columns) and takes the joint columns jb = 1 + sub, 1 + sub + L, ...
// -- 4x the waves and 2x less work per lane than one lane per configuration (fp32 B = 65 536: 214 -> 80 us; 88 us with 8).
// The block's matrices ([16][nv * nv]) are assembled in LDS and leave as flat 16-byte copies of whole lines (the first
// version stored every entry on its own: 4-byte stores 72 bytes apart).
constexpr int FB_MINV_L = 4;
constexpr int FB_MINV_C = 64 / FB_MINV_L;    // configurations per block
template <class T>
constexpr size_t minv_fb_lds_bytes() { return sizeof(T) * (size_t)FB_MINV_C * NV * NV; }
template <class T>
__global__ __launch_bounds__(64, 1) void minv_fb_kernel(const T* __restrict__ q, long long B, int dense, T* __restrict__ Minv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* img = reinterpret_cast<T*>(smem_raw);
  const int sub = threadIdx.x % FB_MINV_L, slot = threadIdx.x / FB_MINV_L;
  const long long cfg0 = (long long)blockIdx.x * FB_MINV_C;
  const long long rem = B - cfg0;
  const int nvalid = rem < FB_MINV_C ? (int)rem : FB_MINV_C;
  const long long b = cfg0 + (slot < nvalid ? slot : nvalid - 1);   // lanes beyond the batch repeat its last configuration (their image slot is never flushed)
  const T* qb = q + b * NV;
  T* Mb = img + slot * (NV * NV);
  JTrig<T> tr[N];
  sfor<1, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j] = make_trig<j>(qb[j + 5]); });
  // ---- articulated inertias, U = IA S, D = S^T U (:662, :697-700, :728-733) -------------------------
  T U[N][6], Dinv[N];
  T fb6[6][6];
  {
    T IA[N][6][6];
    sfor<0, N>([&](auto J) {
      sfor<0, 6>([&](auto R) {
        sfor<0, 6>([&](auto C) {
          constexpr int j = decltype(J)::value, r = decltype(R)::value, c = decltype(C)::value;
          IA[j][r][c] = T(IM[j][r * 6 + c]);
        });
      });
    });
    sfor_down<1, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      constexpr int si = fb_s_index(i);
      sfor<0, 6>([&](auto R) { U[i][decltype(R)::value] = IA[i][decltype(R)::value][si]; });
      Dinv[i] = T(1) / U[i][si];
      T A[6][6];   // A = X^T Ia, Ia = IA - U U^T / D
      sfor<0, 6>([&](auto C) {
        constexpr int c = decltype(C)::value;
        T col[6], y[6];
        const T uc = U[i][c] * Dinv[i];
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; col[r] = fma_(-U[i][r], uc, IA[i][r][c]); });
        xform_T<i>(tr[i], col, y);
        sfor<0, 6>([&](auto R) { A[decltype(R)::value][c] = y[decltype(R)::value]; });
      });
      sfor<0, 6>([&](auto R) {   // (A X)[r][:] = X^T A[r][:]^T
        constexpr int r = decltype(R)::value;
        T y[6];
        xform_T<i>(tr[i], A[r], y);
        sfor<0, 6>([&](auto C) { IA[p][r][decltype(C)::value] += y[decltype(C)::value]; });
      });
    });
    fb_inv6(IA[0], fb6);                                            // fb_Dinv = inv(S^T IA_0 S), S = eye(6)  (:681-683)
  }
  // ---- the base block (:685): symmetric by mirroring its upper part ----------------------------------
  if (sub == 0) {
    sfor<0, 6>([&](auto R) {
      sfor<0, 6>([&](auto C) {
        constexpr int r = decltype(R)::value, c = decltype(C)::value;
        if constexpr (c >= r) {
          Mb[r * NV + c] = fb6[r][c];
          if constexpr (c > r) Mb[c * NV + r] = dense ? fb6[r][c] : T(0);
        }
      });
    });
  }
  // ---- joint columns j = 6 .. NV-1 (body jb = j - 5) -----------------------------------------------------
#pragma clang loop unroll(disable)
  for (int jb = 1 + sub; jb < N; jb += FB_MINV_L) {
    const int j = jb + 5;
    T mcol[N];
    T Fj[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    // backward sweep (:665-726): the column's F vector climbs the root path of body jb
    sfor_down<1, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr unsigned long long mask = fb_subtree_mask(i);
      const bool insub = ((mask >> jb) & 1ull) != 0;
      T m = sel(jb == i, Dinv[i], -(Dinv[i] * S_dot<i>(Fj)));           // :700, :702-708
      m = sel(insub, m, T(0));
      mcol[i] = m;
      T t[6], y[6];
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; t[r] = fma_(U[i][r], m, Fj[r]); });   // :721-723
      xform_T<i>(tr[i], t, y);                                                                              // :724-726
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Fj[r] = sel(insub, y[r], Fj[r]); });
    });
    // the base rows of the column: Minv[0:6, j] = -inv(IA_0) F_0[:, j]   (:686-691)
    T Ff[N][6];
    sfor<0, 6>([&](auto R) {
      constexpr int r = decltype(R)::value;
      T o = T(0);
      sfor<0, 6>([&](auto K) { constexpr int k = decltype(K)::value; o = fma_(-fb6[r][k], Fj[k], o); });
      Ff[0][r] = o;                                                     // F_0[:, j] = S Minv[0:6, j], S = eye(6)  (:779)
    });
    // forward sweep (:760-776)
    sfor<1, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      constexpr int si = fb_s_index(i);
      xform<i>(tr[i], Ff[p], Ff[i]);
      const T m = fma_(-Dinv[i], dot6(U[i], Ff[i]), mcol[i]);           // :771-773
      mcol[i] = m;
      Ff[i][si] += m;                                                   // :774-776
    });
    // rows r <= j of column j, mirrored below the diagonal
    sfor<0, 6>([&](auto R) {
      constexpr int r = decltype(R)::value;
      Mb[r * NV + j] = Ff[0][r];
      Mb[j * NV + r] = dense ? Ff[0][r] : T(0);
    });
    sfor<1, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if (i <= jb) {
        Mb[(i + 5) * NV + j] = mcol[i];
        if (i < jb) Mb[j * NV + i + 5] = dense ? mcol[i] : T(0);
      }
    });
  }
  // ---- the image leaves: [nvalid][nv * nv] is contiguous in Minv ------------------------------------------
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  {
    constexpr int VE = 16 / sizeof(T);
    T* gdst = Minv + cfg0 * (NV * NV);
    const int lane = threadIdx.x;
    if (nvalid == FB_MINV_C && (FB_MINV_C * NV * NV) % VE == 0) {
      typedef T V __attribute__((ext_vector_type(VE)));
      constexpr int NVEC = FB_MINV_C * NV * NV / VE;
#pragma unroll 4
      for (int g = lane; g < NVEC; g += 64) reinterpret_cast<V*>(gdst)[g] = reinterpret_cast<const V*>(img)[g];
    } else {
      for (int g = lane; g < nvalid * NV * NV; g += 64) gdst[g] = img[g];
    }
  }
}

// qdd = Minv (u - c)   (:1373-1374): one thread per (configuration, row)
template <class T>
__global__ __launch_bounds__(256) void fb_apply_kernel(const T* __restrict__ Minv, const T* __restrict__ u,
                                                       const T* __restrict__ c, long long B, T* __restrict__ qdd) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= B * NV) return;
  const long long b = g / NV;
  const int r = (int)(g - b * NV);
  const T* M = Minv + b * (NV * NV) + r * NV;
  T o = T(0);
  for (int k = 0; k < NV; ++k) o = fma_(M[k], u[b * NV + k] - c[b * NV + k], o);
  qdd[g] = o;
}

// ---------------------------------------------------------------------------------------------
// rnea_grad: (q, qd, qdd) [B, NV] -> dc_du [B, NV, 2 NV] (and c [B, NV]), one configuration per lane.
// Column by column, as the reference's dense (6, n, NB) updates do (:1139-1185, :1210-1252, :1264-1294,
// :1306-1341): for every derivative column the lane sweeps the bodies forward (dv, da, df) and backward
// (dc[:, col]).  FB_GRAD_L lanes per configuration share the 2 nv columns (fp32 B = 65 536, 13 bodies: 526 us with one
// lane, 294 with four, 257 with eight; lane `sub` takes out = sub, sub + L, ...;
// all of them run the rnea prologue).  v stays in registers; a (read once per column, by the own-column term) and the
// accumulated f live in LDS per configuration ([6 N][8]), the column's df of every body per lane ([6 N][64]).
// ---------------------------------------------------------------------------------------------
constexpr int FB_GRAD_L = 8;                 // lanes per configuration: lane `sub` takes the columns out = sub, sub + L, ...
constexpr int FB_GRAD_C = 64 / FB_GRAD_L;    // configurations per block
template <class T>
constexpr bool grad_fb_ok() { return N >= 6 && (size_t)6 * N * (2 * FB_GRAD_C + 64) * sizeof(T) <= 160 * 1024; }

template <class T>
RBD_DEV void crm_mul(const T (&x)[6], const T (&y)[6], T (&o)[6]) {   // o = crm(x) y   (:131-140)
  o[0] = x[1] * y[2] - x[2] * y[1];
  o[1] = x[2] * y[0] - x[0] * y[2];
  o[2] = x[0] * y[1] - x[1] * y[0];
  o[3] = x[1] * y[5] - x[2] * y[4] + x[4] * y[2] - x[5] * y[1];
  o[4] = x[2] * y[3] - x[0] * y[5] + x[5] * y[0] - x[3] * y[2];
  o[5] = x[0] * y[4] - x[1] * y[3] + x[3] * y[1] - x[4] * y[0];
}

template <class T, bool HAS_QDD>
__global__ __launch_bounds__(64, 1) void rnea_grad_fb_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                             const T* __restrict__ qdd, T grav, int use_damping,
                                                             long long B, T* __restrict__ c_out, T* __restrict__ dcdu) {
  static_assert(grad_fb_ok<T>(), "instantiated by the launch code only when grad_fb_ok");
  // a and the accumulated f are per CONFIGURATION (its lanes compute and store identical values, in lock step),
  // the column's df per lane
  __shared__ T facc[6 * N][FB_GRAD_C];
  __shared__ T dfl[6 * N][64];
  __shared__ T acl[6 * N][FB_GRAD_C];
  const int lane = threadIdx.x;
  const int sub = lane % FB_GRAD_L, slot = lane / FB_GRAD_L;
  const long long b = (long long)blockIdx.x * FB_GRAD_C + slot;
  if (b >= B) return;                                   // no barriers below (a ragged block's missing configurations are whole lane groups)
  const T* qb = q + b * NV; const T* qdb = qd + b * NV; const T* qddb = HAS_QDD ? qdd + b * NV : nullptr;
  JTrig<T> tr[N];
  T qdv[N];
  T v[N][6];
  T ag0[6];                                             // X_0 a_grav
  {
    // rnea (:559-621): a and the local f go to LDS as they are produced, the backward pass accumulates f there
    auto put = [&](T (&dst)[6 * N][FB_GRAD_C], int j, const T (&x)[6]) {
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; dst[6 * j + r][slot] = x[r]; });
    };
    auto get = [&](T (&src)[6 * N][FB_GRAD_C], int j, T (&x)[6]) {
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; x[r] = src[6 * j + r][slot]; });
    };
    T E[3][3];
    fb_base_E(qb[3], qb[4], qb[5], E);
    ag0[0] = ag0[1] = ag0[2] = T(0);
    ag0[3] = -(grav * E[0][2]); ag0[4] = -(grav * E[1][2]); ag0[5] = -(grav * E[2][2]);
    {
      T a0[6], f0[6], Iv[6], Ia[6];
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        v[0][r] = qdb[r];
        a0[r] = (HAS_QDD ? qddb[r] : T(0)) + ag0[r];
      });
      cmatvec<MatI, 0>(v[0], Iv);
      cmatvec<MatI, 0>(a0, Ia);
      sfor<0, 6>([&](auto R) { f0[decltype(R)::value] = Ia[decltype(R)::value]; });
      fxv<true>(v[0], Iv, f0);
      put(acl, 0, a0);
      put(facc, 0, f0);
    }
    sfor<1, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int p = PARENT[j];
      tr[j] = make_trig<j>(qb[j + 5]);
      qdv[j] = qdb[j + 5];
      T qddj = T(0);
      if constexpr (HAS_QDD) qddj = qddb[j + 5];
      T xv[6], xa[6], ap[6], aj[6], fj[6];
      get(acl, p, ap);
      rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddj, grav, v[p], ap, xv, xa, v[j], aj, fj);
      put(acl, j, aj);
      put(facc, j, fj);
    });
    sfor_down<1, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int p = PARENT[j];
      T fj[6], t[6];
      get(facc, j, fj);
      if (c_out != nullptr && sub == 0) c_out[b * NV + j + 5] = S_dot<j>(fj);
      xform_T<j>(tr[j], fj, t);
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; facc[6 * p + r][slot] += t[r]; });
    });
    if (c_out != nullptr && sub == 0) sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; c_out[b * NV + r] = facc[r][slot]; });
  }
  T* dcb = dcdu + b * (2 * NV * NV);
#pragma clang loop unroll(disable)
  for (int out = sub; out < 2 * NV; out += FB_GRAD_L) {
    const bool isqd = out >= NV;
    const int col = isqd ? out - NV : out;
    // Column-invariant products (X v_p, X a_p, I v, crm(.) S of every body) would be hoisted out of this
    // loop and kept live across it: ~30 N values, i.e. scratch.  Opaque copies keep them inside the iteration.
    sfor<1, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j].s = launder(tr[j].s); tr[j].c = launder(tr[j].c); });
    sfor<0, N>([&](auto J) {
      sfor<0, 6>([&](auto R) { constexpr int j = decltype(J)::value, r = decltype(R)::value; v[j][r] = launder(v[j][r]); });
    });
    // ---- forward sweep of the column ---------------------------------------------------------------
    T dv[N][6], da[N][6];
    {
      // the base: dq: dv = 0, da = crm(X_0 a_grav) e_col (:1175); dqd: dv = e_col (:1231),
      // da = crm(dv) v_0 (:1236-1238, sum_ii qd_ii crm(dv) e_ii) + crm(v_0) e_col (:1243)
      T e[6];
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; e[r] = col == r ? T(1) : T(0); });
      T t0[6], t1[6], t2[6];
      crm_mul(ag0, e, t0);
      crm_mul(e, v[0], t1);
      crm_mul(v[0], e, t2);
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        dv[0][r] = sel(isqd, e[r], T(0));
        da[0][r] = sel(isqd, t1[r] + t2[r], t0[r]);
      });
      T Iv[6], Idv[6], d[6];
      cmatvec<MatI, 0>(v[0], Iv);
      cmatvec<MatI, 0>(dv[0], Idv);
      cmatvec<MatI, 0>(da[0], d);
      fxv<true>(dv[0], Iv, d);
      fxv<true>(v[0], Idv, d);
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; dfl[r][lane] = d[r]; });
    }
    sfor<1, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      const bool own = col == i + 5;
      xform<i>(tr[i], dv[p], dv[i]);          // (:1158 / :1230)
      xform<i>(tr[i], da[p], da[i]);          // (:1163 / :1234)
      T xv[6], xa[6], ap[6], sdq[6], sS[6], e1[6], e2[6];
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; ap[r] = acl[6 * p + r][slot]; });
      xform<i>(tr[i], v[p], xv);
      xform<i>(tr[i], ap, xa);
      mxS<i>(xv, T(1), sdq);                  // crm(X v_p) S   (:1159)
      sfor<0, 6>([&](auto R) { sS[decltype(R)::value] = T(0); });
      add_S<i>(T(1), sS);                     // S              (:1231)
      mxS<i>(xa, T(1), e1);                   // crm(X a_p) S   (:1173)
      mxS<i>(v[i], T(1), e2);                 // crm(v_i) S     (:1243)
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; dv[i][r] += sel(own, sel(isqd, sS[r], sdq[r]), T(0)); });
      add_mxS<i>(dv[i], qdv[i], da[i]);       // (:1170 / :1240)
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; da[i][r] += sel(own, sel(isqd, e2[r], e1[r]), T(0)); });
      T Iv[6], Idv[6], d[6];
      cmatvec<MatI, i>(v[i], Iv);
      cmatvec<MatI, i>(dv[i], Idv);
      cmatvec<MatI, i>(da[i], d);
      fxv<true>(dv[i], Iv, d);                // (:1179-1185 / :1247-1252)
      fxv<true>(v[i], Idv, d);
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; dfl[(6 * i + r)][lane] = d[r]; });
    });
    // ---- backward sweep (:1264-1294 / :1306-1341) -----------------------------------------------------
    sfor_down<1, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      T d[6], fi[6], w[6], x[6], y[6];
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        d[r] = dfl[(6 * i + r)][lane];
        fi[r] = facc[(6 * i + r)][slot];
      });
      T o = S_dot<i>(d);                                                  // (:1284 / :1325)
      dcb[(i + 5) * (2 * NV) + out] = o;
      mxS<i>(fi, T(-1), w);                                               // fxS(S, f) = -crm(f) S  (:1292-1294)
      const bool ex = (col == i + 5) && !isqd;
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; x[r] = d[r] + sel(ex, w[r], T(0)); });
      xform_T<i>(tr[i], x, y);                                            // (:1291 / :1331)
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; dfl[(6 * p + r)][lane] += y[r]; });
    });
    sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; dcb[r * (2 * NV) + out] = dfl[r][lane]; });   // S = eye(6) (:1282)
  }
  if (use_damping) {
    // (:1336-1341) literally: the base adds its damping to a 5 x 5 block, body ind >= 1 to entry (ind, n + ind)
    sfor<0, 5>([&](auto R) {
      sfor<0, 5>([&](auto C) {
        constexpr int r = decltype(R)::value, c = decltype(C)::value;
        if constexpr (DAMPING[0] != 0.0) { if ((NV + c) % FB_GRAD_L == sub) dcb[r * (2 * NV) + NV + c] += T(DAMPING[0]); }
      });
    });
    sfor<1, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (DAMPING[i] != 0.0) { if ((NV + i) % FB_GRAD_L == sub) dcb[i * (2 * NV) + NV + i] += T(DAMPING[i]); }
    });
  }
}

}  // namespace rbdk
