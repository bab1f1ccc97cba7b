// rbd_passes.h -- the reference's PER-PASS methods as device kernels (gfx950).
//
// The reference's README designates the individual passes as the surface an accelerator back-end
// is tested through (/root/reference/README.md:19).  The fused product kernels (rnea_grad_*,
// minv_*) never materialise the intermediates of those passes; the kernels below do, in exactly
// the layouts the reference returns them:
//   rnea_grad_fpass_dq / _dqd  (RBDReference.py:1127-1187 / :1189-1255)  -> dv, da, df  [B,6,n,NB]
//   rnea_grad_bpass_dq / _dqd  (:1257-1297 / :1299-1343)   df mutated in place      -> dc  [B,n,n]
//   minv_bpass                 (:630-735)      -> Minv (upper part) [B,n,n], F [B,n,6,n], U [B,n,6],
//                                                 "Dinv" [B,n] holding D as the reference does (:698)
//   minv_fpass                 (:737-783)      Minv updated in place over WHOLE rows (:771), F rebuilt
// One configuration per lane; the derivative column (or Minv column) is the outer loop (run-time
// where the pass is dense over columns, so the code size stays O(n)), the bodies are unrolled.  These are verification entry points: they move the
// reference's O(n^2) six-vectors through HBM by definition, so they are bounded by their own output
// size, not by the product kernels' rooflines.
#pragma once
#include "rbd_spatial.h"

namespace rbdk {

template <class T>
RBD_DEV void load6(const T* base, int stride, T (&x)[6]) {
  sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; x[r] = base[r * stride]; });
}

// ---- rnea_grad_fpass_dq (DQ = true) / rnea_grad_fpass_dqd (DQ = false) ---------------------------
// The derivative column c is a run-time loop (wave-uniform), the body loop is unrolled: a column is
// seeded at body c (:1159,:1172-1175 / :1231,:1243) and propagated with X to every later body, so the
// bodies outside subtree(c) receive X * 0 = 0 exactly as the reference's dense column updates do.
template <class T, bool DQ>
__global__ __launch_bounds__(64) void grad_fpass_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                        const T* __restrict__ v_in, const T* __restrict__ a_in,
                                                        T grav, long long B, T* __restrict__ dv_out,
                                                        T* __restrict__ da_out, T* __restrict__ df_out) {
  const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  JTrig<T> tr[N];
  T qdv[N];
  sfor<0, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    tr[j] = make_trig<j>(q[b * N + j]);
    qdv[j] = qd[b * N + j];
  });
  const T* vb = v_in + b * (6 * N);
  const T* ab = DQ ? a_in + b * (6 * N) : nullptr;
  T* dvo = dv_out + b * (6 * N * N);
  T* dao = da_out + b * (6 * N * N);
  T* dfo = df_out + b * (6 * N * N);

#pragma clang loop unroll(disable)
  for (int c = 0; c < N; ++c) {
    T dv[N][6], da[N][6];
    sfor<0, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      T df[6];
      if (i < c) {      // bodies are numbered parents-first: column c cannot reach a body before c
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; dv[i][r] = T(0); da[i][r] = T(0); df[r] = T(0); });
      } else {
        T vi[6];
        load6(vb + i, N, vi);
        if constexpr (p >= 0) {
          xform<i>(tr[i], dv[p], dv[i]);     // (:1158 / :1230)
          xform<i>(tr[i], da[p], da[i]);     // (:1163 / :1234)
        } else {
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; dv[i][r] = T(0); da[i][r] = T(0); });
        }
        T sda[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
        if (i == c) {
          if constexpr (DQ) {
            // dv[:,i,i] += crm(X v_p) S (:1159);  da[:,i,i] += crm(X a_p) S, X a0 at a root (:1172-1175)
            T xa[6];
            if constexpr (p >= 0) {
              T vp[6], ap[6], xv[6];
              load6(vb + p, N, vp);
              load6(ab + p, N, ap);
              xform<i>(tr[i], vp, xv);
              xform<i>(tr[i], ap, xa);
              add_mxS<i>(xv, T(1), dv[i]);
            } else {
              const T a0[6] = {T(0), T(0), T(0), T(0), T(0), -grav};
              xform<i>(tr[i], a0, xa);
            }
            mxS<i>(xa, T(1), sda);
          } else {
            add_S<i>(T(1), dv[i]);           // dv[:,i,i] += S (:1231)
            mxS<i>(vi, T(1), sda);           // da[:,i,i] += crm(v_i) S (:1243)
          }
        }
        add_mxS<i>(dv[i], qdv[i], da[i]);    // da[:,c,i] += qd_i crm(dv[:,c,i]) S  (:1164-1170 / :1235-1240)
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; da[i][r] += sda[r]; });
        // df = I da + crf(dv)(I v) + crf(v)(I dv)  (:1179-1185 / :1247-1252)
        T Iv[6], Idv[6];
        cmatvec<MatI, i>(vi, Iv);
        cmatvec<MatI, i>(dv[i], Idv);
        cmatvec<MatI, i>(da[i], df);
        fxv<true>(dv[i], Iv, df);
        fxv<true>(vi, Idv, df);
      }
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        dvo[(r * N + c) * N + i] = dv[i][r]; dao[(r * N + c) * N + i] = da[i][r]; dfo[(r * N + c) * N + i] = df[r];
      });
    });
  }
}

// ---- rnea_grad_bpass_dq (DQ = true) / rnea_grad_bpass_dqd (DQ = false) ---------------------------
// df is an arbitrary dense [B,6,n,NB] input (the reference accepts any), accumulated in place
// child -> parent exactly like (:1291 / :1331); f is the ACCUMULATED RNEA force (:1353,:1362).
template <class T, bool DQ>
__global__ __launch_bounds__(64) void grad_bpass_kernel(const T* __restrict__ q, const T* __restrict__ f_in,
                                                        T* __restrict__ df_io, int use_damping, long long B,
                                                        T* __restrict__ dc_out) {
  const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  JTrig<T> tr[N];
  sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j] = make_trig<j>(q[b * N + j]); });
  const T* fb = DQ ? f_in + b * (6 * N) : nullptr;
  T* dfb = df_io + b * (6 * N * N);
  T* dcb = dc_out + b * (N * N);

#pragma clang loop unroll(disable)
  for (int c = 0; c < N; ++c) {
    T acc[N][6];     // child contributions waiting for their parent
    sfor<0, N>([&](auto I) {
      sfor<0, 6>([&](auto R) { acc[decltype(I)::value][decltype(R)::value] = T(0); });
    });
    sfor_down<0, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      T x[6];
      load6(dfb + c * N + i, N * N, x);
      if constexpr (has_child(i)) {
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; x[r] += acc[i][r]; });
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; dfb[(r * N + c) * N + i] = x[r]; });
      }
      T d = S_dot<i>(x);                                        // dc[i, c] = S^T df[:, c, i]  (:1284 / :1325)
      if constexpr (!DQ) d += sel(use_damping != 0 && i == c, T(DAMPING[i]), T(0));   // (:1336-1341)
      dcb[i * N + c] = d;
      if constexpr (p >= 0) {
        if constexpr (DQ) {
          if (i == c) {
            // df[:, i, p] += X^T fxS(S, f_i),  fxS(S, f) = -crm(f) S  (:166-168, :1292-1294)
            T fi[6];
            load6(fb + i, N, fi);
            add_mxS<i>(fi, T(-1), x);
          }
        }
        T y[6];
        xform_T<i>(tr[i], x, y);                                // df[:, c, p] += X^T df[:, c, i]  (:1291 / :1331)
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; acc[p][r] += y[r]; });
      }
    });
  }
}

// ---- minv_bpass (:630-735, fixed-base branch) -----------------------------------------------------
template <class T>
__global__ __launch_bounds__(64) void minv_bpass_kernel(const T* __restrict__ q, long long B, T* __restrict__ Minv,
                                                        T* __restrict__ F, T* U_out, T* D_out) {
  const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  JTrig<T> tr[N];
  sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j] = make_trig<j>(q[b * N + j]); });
  T* Mb = Minv + b * (N * N);
  T* Fb = F + b * (N * 6 * N);            // F[i][r][j]
  T* Ub = U_out + b * (N * 6);
  T* Db = D_out + b * N;
  // articulated inertias, U = IA S, D = S^T U  (:662, :697-698, :728-733)
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
    sfor_down<0, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      constexpr int si = s_index(i);
      T Ui[6];
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Ui[r] = IA[i][r][si]; Ub[i * 6 + r] = Ui[r]; });
      Db[i] = Ui[si];
      if constexpr (p >= 0) {
        const T dinv = T(1) / Ui[si];
        T A[6][6];
        sfor<0, 6>([&](auto C) {
          constexpr int c = decltype(C)::value;
          T col[6], y[6];
          const T uc = Ui[c] * dinv;
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; col[r] = fma_(-Ui[r], uc, IA[i][r][c]); });
          xform_T<i>(tr[i], col, y);
          sfor<0, 6>([&](auto R) { A[decltype(R)::value][c] = y[decltype(R)::value]; });
        });
        sfor<0, 6>([&](auto R) {
          constexpr int r = decltype(R)::value;
          T y[6];
          xform_T<i>(tr[i], A[r], y);
          sfor<0, 6>([&](auto C) { IA[p][r][decltype(C)::value] += y[decltype(C)::value]; });
        });
      }
    });
  }
  // U, D come back from memory (same thread, program order) so that they need not stay in registers
  const volatile T* Uv = Ub;
  const volatile T* Dv = Db;
  // column j of Minv / F: only the root path of j is non-zero (:700-726)
  sfor<0, N>([&](auto JC) {
    constexpr int jc = decltype(JC)::value;
    T Fj[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};   // F[i][:, jc] as accumulated from the child on the path
    sfor_down<0, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (!is_anc_or_self(i, jc)) {
        Mb[i * N + jc] = T(0);
        sfor<0, 6>([&](auto R) { Fb[(i * 6 + decltype(R)::value) * N + jc] = T(0); });
      } else {
        constexpr int p = PARENT[i];
        const T dinv = T(1) / Dv[i];
        T m;
        if constexpr (i == jc) m = dinv;                       // (:700)
        else m = -(dinv * S_dot<i>(Fj));                       // (:702-708)
        Mb[i * N + jc] = m;
        if constexpr (p >= 0) {
          T t[6], y[6];
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; t[r] = fma_(T(Uv[i * 6 + r]), m, Fj[r]); });   // (:721-723)
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Fb[(i * 6 + r) * N + jc] = t[r]; });
          xform_T<i>(tr[i], t, y);                             // F[p][:, jc] += X^T F[i][:, jc]  (:724-726)
          sfor<0, 6>([&](auto R) { Fj[decltype(R)::value] = y[decltype(R)::value]; });
        } else {
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Fb[(i * 6 + r) * N + jc] = Fj[r]; });
        }
      }
    });
  });
}

// ---- minv_fpass (:737-783) -------------------------------------------------------------------------
// Whole rows are updated (:771), so the strict lower triangle receives the same by-products as in the
// reference.  F is rebuilt from scratch (:774-781); its incoming contents are never read.
template <class T>
__global__ __launch_bounds__(64) void minv_fpass_kernel(const T* __restrict__ q, long long B, T* __restrict__ Minv,
                                                        T* __restrict__ F, const T* __restrict__ U_in,
                                                        const T* __restrict__ D_in) {
  const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
  // big robots in fp64 (30 bodies: the column's F of every body is 360 registers): sin / cos wait in lane-private LDS
  constexpr bool TRIG_LDS = N * sizeof(T) > 128;
  __shared__ T trig_lds[TRIG_LDS ? 2 * N : 1][TRIG_LDS ? 64 : 1];
  if (b >= B) return;                                        // (no barrier below)
  JTrig<T> tr[N];
  sfor<0, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    tr[j] = make_trig<j>(q[b * N + j]);
    if constexpr (TRIG_LDS) { trig_lds[2 * j][threadIdx.x] = tr[j].s; trig_lds[2 * j + 1][threadIdx.x] = tr[j].c; }
  });
  T* Mb = Minv + b * (N * N);
  T* Fb = F + b * (N * 6 * N);
  const T* Ub = U_in + b * (N * 6);
  const T* Db = D_in + b * N;
#pragma clang loop unroll(disable)
  for (int c = 0; c < N; ++c) {
    T Ff[N][6];
    sfor<0, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      constexpr int si = s_index(i);
      T m = Mb[i * N + c];
      if constexpr (p < 0) {
        sfor<0, 6>([&](auto R) { Ff[i][decltype(R)::value] = T(0); });
        Ff[i][si] = m;                                           // F[i] = outer(S, Minv[i, :])  (:781)
      } else {
        T Ui[6];
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Ui[r] = Ub[i * 6 + r]; });
        if constexpr (TRIG_LDS) { tr[i].s = trig_lds[2 * i][threadIdx.x]; tr[i].c = trig_lds[2 * i + 1][threadIdx.x]; }
        xform<i>(tr[i], Ff[p], Ff[i]);
        m = fma_(-(T(1) / Db[i]), dot6(Ui, Ff[i]), m);           // Minv[i, c] -= (1/D)(U^T X) F[p][:, c]  (:771-773)
        Mb[i * N + c] = m;
        Ff[i][si] += m;                                          // F[i] = X F[p] + outer(S, Minv[i, :])  (:774-776)
      }
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Fb[(i * 6 + r) * N + c] = Ff[i][r]; });
    });
  }
}

}  // namespace rbdk
