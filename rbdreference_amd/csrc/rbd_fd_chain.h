// rbd_fd_chain.h -- forward_dynamics_grad for robots that are ONE chain (the 7-DoF arm), as TWO launches whose
// intermediate never takes the shape the reference gives it.
//
// Reference (/root/reference/RBDReference.py:1376-1384):
//     qdd = forward_dynamics(q, qd, u)          = minv(q) (u - rnea(q, qd)[0])                  (:1371-1374)
//     dc_du = rnea_grad(q, qd, qdd)                                                             (:1345-1368)
//     return -minv(q) dc_dq, -minv(q) dc_dqd                                                    (:1381-1383)
// Round 3 ran that as three launches -- rnea (bias force c), minv with qdd = Minv (u - c) fused in, and the round-2
// gradient kernel with a -Minv epilogue that re-read the dense [B, n, n] Minv through an LDS tile -- 258-274 us for the
// arm at B = 2^20 (0.23 of HBM peak), of which 208 us the gradient leg, because the epilogue had never been ported to
// the software-pipelined kernel that serves plain rnea_grad in 128 us.
//
// Here:
//   launch 1, fd_pre_kernel: one configuration per lane, everything of a configuration in that lane --
//       RNEA with qdd = None (the bias force, :559-621), the articulated-inertia recursion and the per-column sweeps of
//       minv (:630-783, as rbd_minv_lane.h), qdd = Minv (u - c) accumulated column by column.  It writes qdd [B, n] and
//       the UPPER TRIANGLE of Minv in a lane-major workspace  [tile][n (n + 1) / 2][64 lanes]  -- every store of a wave
//       is one contiguous 256-byte piece, no LDS, no dense matrix, no mirror;
//   launch 2, rnea_grad_idsva_pipe_kernel<T, true, FDG = true> (rbd_idsva_pipe.h): when a tile's sweep is over and its
//       2 n^2 finished entries sit in registers, the lane reads ITS n (n + 1) / 2 Minv values back with coalesced dword
//       loads (same layout, same lane: 112 B per configuration at n = 7 instead of 196 B through LDS) and overwrites
//       the entries column by column with -Minv E; the flush that follows is the pipelined one, untouched.
// HBM traffic per configuration at n = 7, fp32: launch 1 reads 84 B, writes 28 + 112; launch 2 reads 84 + 112, writes
// 392 -- 812 B against 1 176 B before (c 28 + 28, dense Minv 196 + 196), and one launch and its sin / cos fewer.
#pragma once
#include "rbd_idsva_pipe.h"

namespace rbdk {

// one chain, revolute joints, rigid-body inertias (what the world-frame chain kernels serve): fp32 -- the software-pipelined
// kernel where it applies, else rnea_grad_idsva_kernel<T, true, FDG>; fp64 -- rnea_grad_idsva_kernel<double, true, FDG> for
// chains of up to 7 bodies (rbd_kernels.hip: grad_chain_kernel)
template <class T>
constexpr bool fd_chain_ok() { return GRAD_IDSVA_OK && n_groups() == 1 && grad_max_rows() == N && (sizeof(T) == 4 || N <= 7); }

#ifndef RBD_FDP_MINW
#define RBD_FDP_MINW 2
#endif
template <class T>
__global__ __launch_bounds__(64, sizeof(T) == 4 ? RBD_FDP_MINW : 1) void fd_pre_kernel(const T* __restrict__ q, const T* __restrict__ qd, const T* __restrict__ u,
                                                       T grav, long long B, T* __restrict__ qdd_out, T* __restrict__ minv_pk) {
  const int lane = threadIdx.x;
  const long long cfg0 = (long long)blockIdx.x * 64;
  const long long rem = B - cfg0;
  const int nvalid = rem < 64 ? (int)rem : 64;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);
  T qv[N], qdv[N], tau[N];
  sfor<0, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    qv[j] = q[b * N + j]; qdv[j] = qd[b * N + j]; tau[j] = u[b * N + j];
  });
  JTrig<T> tr[N];
  sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j] = make_trig<j>(qv[j]); });

  // ---- bias force: rnea(q, qd) with qdd = None (:559-621) -> tau = u - c -----------------------------------------
#ifndef RBD_FDP_EXP_NOBIAS      // (timing experiments: results are wrong with any RBD_FDP_EXP_* switch)
  {
    T f[N][6], vb[N][6], ab[N][6];
    sfor<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int p = PARENT[j];
      T xv[6], xa[6];
      const T zero6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
      if constexpr (p < 0) rnea_fwd_body<j, false>(tr[j], qdv[j], T(0), grav, zero6, zero6, xv, xa, vb[j], ab[j], f[j]);
      else rnea_fwd_body<j, false>(tr[j], qdv[j], T(0), grav, vb[p], ab[p], xv, xa, vb[j], ab[j], f[j]);
    });
    sfor_down<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int p = PARENT[j];
      tau[j] -= S_dot<j>(f[j]);                                                   // c_j = S^T f_j (:612)
      if constexpr (p >= 0) {
        T y[6];
        xform_T<j>(tr[j], f[j], y);                                               // f_p += X^T f_j (:618-619)
        sfor<0, 6>([&](auto R) { f[p][decltype(R)::value] += y[decltype(R)::value]; });
      }
    });
  }
#endif

  // ---- articulated inertias (:662, :697-700, :728-733).  IA is symmetric: 21 scalars per body (upper triangle) instead of
  //      rbd_minv_lane.h's 36 -- the lower half of X^T Ia X is never formed (its instructions are dead code once nothing
  //      reads them), and two live inertias are 42 registers instead of 72 ------------------------------------------------
  T U[N][6], Dinv[N];
  {
    constexpr auto sy = [](int r, int c) constexpr { return r <= c ? r * 6 - r * (r - 1) / 2 + (c - r) : c * 6 - c * (c - 1) / 2 + (r - c); };
    T IA[N][21];
    sfor<0, N>([&](auto J) {
      sfor<0, 6>([&](auto R) {
        sfor<0, 6>([&](auto C) {
          constexpr int j = decltype(J)::value, r = decltype(R)::value, c = decltype(C)::value;
          if constexpr (r <= c) IA[j][sy(r, c)] = T(IM[j][r * 6 + c]);
        });
      });
    });
    sfor_down<0, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      constexpr int si = s_index(i);
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; U[i][r] = IA[i][sy(r, si)]; });   // U = IA S
      Dinv[i] = rcp_inertia(U[i][si]);                                                            // 1 / (S^T U): v_rcp_f32 + one Newton step in fp32 (rbd_spatial.h)
#ifdef RBD_FDP_EXP_NOIA
      if constexpr (false) {
#else
      if constexpr (p >= 0) {
#endif
        T A[6][6];   // A = X^T Ia, Ia = IA - U U^T / D
        sfor<0, 6>([&](auto C) {
          constexpr int c = decltype(C)::value;
          T col[6], y[6];
          const T uc = U[i][c] * Dinv[i];
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; col[r] = fma_(-U[i][r], uc, IA[i][sy(r, c)]); });
          xform_T<i>(tr[i], col, y);
          sfor<0, 6>([&](auto R) { A[decltype(R)::value][c] = y[decltype(R)::value]; });
        });
        sfor<0, 6>([&](auto R) {   // IA_p += (A X), upper triangle: row r of A X = X^T A[r][:]^T
          constexpr int r = decltype(R)::value;
          T y[6];
          xform_T<i>(tr[i], A[r], y);
          sfor<r, 6>([&](auto C) { constexpr int c = decltype(C)::value; IA[p][sy(r, c)] += y[c]; });
        });
      }
    });
  }

  // ---- Minv.  Backward sweeps only (:700-726): column jc climbs its root path and leaves m[k][jc] = Minv_bpass[k, jc] for every
  //      ancestor-or-self k.  The reference's forward pass (:760-781, n (n + 1) / 2 more six-vector transforms) is replaced by
  //      the factorisation it evaluates (the articulated-body "innovations" form, exact -- checked at 2e-16 against the
  //      reference's Minv on every golden robot, tools/check_minv_factorisation.py):
  //          Minv[i, j] = sum over k in anc(i) & anc(j) of  D_k m[k][i] m[k][j],      m[k][k] = 1 / D_k
  //      i.e. ~n^3 / 6 scalar FMAs.  qdd = Minv tau from the symmetric result. ------------------------------------------------
  T Mb[N][N];
  sfor<0, N>([&](auto JC) {
    constexpr int jc = decltype(JC)::value;
#ifdef RBD_FDP_EXP_NOCOLS
    Mb[jc][jc] = Dinv[jc] + U[jc][0] + tau[jc];
#else
    T Fj[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    sfor_down<0, jc + 1>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (is_anc_or_self(i, jc)) {
        constexpr int p = PARENT[i];
        T m;
        if constexpr (i == jc) m = Dinv[i];
        else m = -(Dinv[i] * S_dot<i>(Fj));
        Mb[i][jc] = m;
        if constexpr (p >= 0 && i > 0) {
          T t[6], y[6];
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; t[r] = (i == jc) ? U[i][r] * m : fma_(U[i][r], m, Fj[r]); });
          xform_T<i>(tr[i], t, y);
          sfor<0, 6>([&](auto R) { Fj[decltype(R)::value] = y[decltype(R)::value]; });
        }
      }
    });
#endif
  });
  T qacc[N];
  sfor<0, N>([&](auto J) { qacc[decltype(J)::value] = T(0); });
  T* mp = minv_pk + (size_t)blockIdx.x * (FDC_NP * 64) + lane;
#ifndef RBD_FDP_EXP_NOCOLS
  {
    T W[N][N];                                                    // W[k][i] = D_k m[k][i]  (k a proper ancestor of i)
    sfor<0, N>([&](auto I) {
      sfor<0, N>([&](auto K) {
        constexpr int i = decltype(I)::value, k = decltype(K)::value;
        if constexpr (k != i && is_anc_or_self(k, i)) W[k][i] = U[k][s_index(k)] * Mb[k][i];
      });
    });
    sfor<0, N>([&](auto J) {
      sfor<0, N>([&](auto I) {
        constexpr int i = decltype(I)::value, j = decltype(J)::value;
        if constexpr (i <= j && is_anc_or_self(i, j)) {           // (a chain: every i <= j is an ancestor-or-self of j)
          T acc = Mb[i][j];                                       // k = i: D_i m[i][i] m[i][j] = m[i][j]
          sfor<0, N>([&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (k != i && is_anc_or_self(k, i)) acc = fma_(W[k][i], Mb[k][j], acc);
          });
          mp[fdc_slot(i, j) * 64] = acc;
          qacc[i] = fma_(acc, tau[j], qacc[i]);
          if constexpr (i < j) qacc[j] = fma_(acc, tau[i], qacc[j]);
        }
      });
    });
  }
#else
  sfor<0, N>([&](auto JC) { constexpr int jc = decltype(JC)::value; mp[fdc_slot(jc, jc) * 64] = Mb[jc][jc]; });
#endif
  if (lane < nvalid) {
    sfor<0, N>([&](auto I) { constexpr int i = decltype(I)::value; qdd_out[b * N + i] = qacc[i]; });
  }
}

}  // namespace rbdk
