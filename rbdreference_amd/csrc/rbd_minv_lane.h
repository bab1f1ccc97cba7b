// rbd_minv_lane.h -- minv (and the forward-dynamics product) as ONE fused kernel, one configuration
// per lane, for robots whose root subtrees ("groups") have at most 8 bodies.
//
// The two-phase kernels (minv_ia_kernel + minv_cols_kernel) hand {U, 1/D, sin, cos} per body through
// an HBM workspace of 48 B x n per configuration -- for a 7-DoF arm that round trip (336 B written,
// 336 B read) is more than the 224 B of q + Minv the algorithm has to move, and two thirds of the
// measured time.  Here the articulated-inertia recursion (RBDReference.py:697-700, :728-733) and
// the per-column backward / forward sweeps (:702-726, :771-776) of a configuration run in the same
// lane, U / D stay in registers, columns are processed one after the other (each column only visits
// the bodies on its own root path in the backward sweep and the bodies with index <= its own in the
// forward sweep -- all resolved at compile time), and a finished column goes to the LDS image of
// the output tile.  Independent roots are processed as groups exactly as in rnea_grad_kernel:
// Minv is block-diagonal over groups, so other groups' columns are structural zeros.
#pragma once
#include "rbd_spatial.h"

namespace rbdk {

// (round 4: also robots whose root subtrees INTERLEAVE in the numbering -- GRAD_PER_ROOT false, the whole robot is one group --
// when that group has at most 8 bodies: columns of the other root's bodies are structural zeros the sweeps pass through, and
// the two-phase path such a robot used to take ran at 7.5 % of HBM peak, 596 GB/s for the 8-body forest at B = 65 536)
constexpr bool MINV_LANE_OK = grad_max_rows() <= 8;
// (round 4, "topology lottery": for mid-size trees -- a 14-body torso with three limbs, two 9-body chains, dense frames -- the
// two-phase / eight-lane paths run at 0.12-0.19 of HBM peak.  With the articulated inertias kept as 21-scalar symmetric
// matrices a lone wave's registers do hold such a group in fp32 (223 / 167 VGPRs, no scratch: -DMINV_LANE_MAX_F32=14), but
// the unrolled kernel is 15-17 k instructions = 120-140 KB of code that lone waves stream through a 64 KB instruction
// cache: 52 / 77 us at B = 65 536 against 46-49 / 68 us for the one-lane phase A + column kernel.  The limit stays at 8.)
#ifndef MINV_LANE_MAX_F32
#define MINV_LANE_MAX_F32 8
#endif
template <class T>
constexpr bool minv_lane_ok() { return grad_max_rows() <= (sizeof(T) == 4 ? MINV_LANE_MAX_F32 : 8); }
// LDS stride between configurations (odd => conflict-free per-lane rows)
constexpr int MINV_LANE_TS = (grad_max_rows() * N) | 1;

template <class T>
__global__ __launch_bounds__(64, (sizeof(T) == 4 && grad_max_rows() <= 8) ? 2 : 1) void minv_lane_kernel(const T* __restrict__ q, long long B, int dense,
                                                                              T* __restrict__ Minv, const T* __restrict__ u_in,
                                                                              const T* __restrict__ c_in, T* __restrict__ qdd_out,
                                                                              const T* __restrict__ qd_in = nullptr, T grav = T(0)) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x;
  const long long cfg0 = (long long)blockIdx.x * 64;
  const long long rem = B - cfg0;
  const int nvalid = rem < 64 ? (int)rem : 64;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);
  const bool fd = qdd_out != nullptr;

  JTrig<T> tr[N];
  T qv[N];
  sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; qv[j] = q[b * N + j]; });
  T tau[N], qacc[N];
  // forward dynamics (:1371-1374): tau = u - c.  With qd_in the bias force c = rnea(q, qd, qdd = None) (:559-621) is computed
  // HERE, group by group, from the sin / cos this lane holds anyway -- the separate c-only rnea launch and its [B, n] round
  // trip are gone (the quadruped's forward_dynamics_grad in fp64, B = 65 536: 11 of 88 us)
  const bool own_bias = fd && qd_in != nullptr;                  // (uniform over the launch)
  if (fd) {
    sfor<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      tau[j] = own_bias ? u_in[b * N + j] : u_in[b * N + j] - c_in[b * N + j];
      qacc[j] = T(0);
    });
  } else {
    sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; tau[j] = T(0); qacc[j] = T(0); });
  }
  sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j] = make_trig<j>(qv[j]); });

  T U[N][6], Dinv[N];
  sfor<0, N>([&](auto Rt) {
   constexpr int rt = decltype(Rt)::value;
   if constexpr (grp_head(rt)) {
    constexpr int row0 = grp_row0(rt);
    constexpr int rows = grp_rows(rt);
    T* my = tile + lane * MINV_LANE_TS - row0 * N;     // my[i * N + c], rows of this group
    if (own_bias) {
      T f[N][6], vb[N][6], ab[N][6];
      sfor<row0, row0 + rows>([&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr int p = PARENT[j];
        T xv[6], xa[6];
        const T zero6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
        const T qdj = qd_in[b * N + j];
        if constexpr (p < 0) rnea_fwd_body<j, false>(tr[j], qdj, T(0), grav, zero6, zero6, xv, xa, vb[j], ab[j], f[j]);
        else rnea_fwd_body<j, false>(tr[j], qdj, T(0), grav, vb[p], ab[p], xv, xa, vb[j], ab[j], f[j]);
      });
      sfor_down<row0, row0 + rows>([&](auto J) {
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
    // ---- articulated inertias of the group (:662, :697-700, :728-733); IA is symmetric: 21 scalars per body, the lower half
    //      of X^T Ia X is never formed ------------------------------------------------------------------------------------
    {
      constexpr auto sy = [](int r, int c) constexpr { return r <= c ? r * 6 - r * (r - 1) / 2 + (c - r) : c * 6 - c * (c - 1) / 2 + (r - c); };
      T IA[N][21];
      sfor<row0, row0 + rows>([&](auto J) {
        sfor<0, 6>([&](auto R) {
          sfor<0, 6>([&](auto C) {
            constexpr int j = decltype(J)::value, r = decltype(R)::value, c = decltype(C)::value;
            if constexpr (r <= c) IA[j][sy(r, c)] = T(IM[j][r * 6 + c]);
          });
        });
      });
      sfor_down<row0, row0 + rows>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int p = PARENT[i];
        constexpr int si = s_index(i);
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; U[i][r] = IA[i][sy(r, si)]; });   // U = IA S
        Dinv[i] = rcp_inertia(U[i][si]);                                                                  // 1 / (S^T U)
        if constexpr (p >= 0) {
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
    // ---- Minv of the group.  Backward sweeps only (:700-726): column jc climbs its root path and leaves
    //      m[k][jc] = minv_bpass's Minv[k, jc] for every ancestor-or-self k.  The reference's forward pass (:760-781: for every
    //      column a six-vector transform per body of the group) is replaced by the factorisation it evaluates,
    //          Minv[i, j] = sum over k in anc(i) & anc(j) of  D_k m[k][i] m[k][j],      m[k][k] = 1 / D_k
    //      (exact: 2e-16 against the reference's Minv on every golden robot, tools/check_minv_factorisation.py) -- scalar FMAs
    //      over COMMON ANCESTORS, all resolved at compile time; unrelated pairs of one root share ancestors too. ----------
    {
      T Mb[N][N];
      sfor<row0, row0 + rows>([&](auto JC) {
        constexpr int jc = decltype(JC)::value;
        T Fj[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
        sfor_down<row0, jc + 1>([&](auto I) {
          constexpr int i = decltype(I)::value;
          if constexpr (is_anc_or_self(i, jc)) {
            constexpr int p = PARENT[i];
            T m;
            if constexpr (i == jc) m = Dinv[i];
            else m = -(Dinv[i] * S_dot<i>(Fj));
            Mb[i][jc] = m;
            if constexpr (p >= 0) {
              T t[6], y[6];
              sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; t[r] = (i == jc) ? U[i][r] * m : fma_(U[i][r], m, Fj[r]); });
              xform_T<i>(tr[i], t, y);
              sfor<0, 6>([&](auto R) { Fj[decltype(R)::value] = y[decltype(R)::value]; });
            }
          }
        });
      });
      T W[N][N];                                                  // W[k][i] = D_k m[k][i], k a proper ancestor of i (D_k = S^T U_k)
      sfor<row0, row0 + rows>([&](auto I) {
        sfor<row0, row0 + rows>([&](auto K) {
          constexpr int i = decltype(I)::value, k = decltype(K)::value;
          if constexpr (k != i && is_anc_or_self(k, i)) W[k][i] = U[k][s_index(k)] * Mb[k][i];
        });
      });
      sfor<row0, row0 + rows>([&](auto J) {
        sfor<row0, row0 + rows>([&](auto I) {
          constexpr int i = decltype(I)::value, j = decltype(J)::value;
          if constexpr (i <= j) {
            T acc = T(0);
            bool any = false;
            if constexpr (is_anc_or_self(i, j)) { acc = Mb[i][j]; any = true; }          // k = i: D_i m[i][i] m[i][j] = m[i][j]
            sfor<row0, row0 + rows>([&](auto K) {
              constexpr int k = decltype(K)::value;
              if constexpr (k != i && is_anc_or_self(k, i) && is_anc_or_self(k, j)) { acc = any ? fma_(W[k][i], Mb[k][j], acc) : W[k][i] * Mb[k][j]; any = true; }
            });
            my[i * N + j] = acc;
            if constexpr (i < j) my[j * N + i] = sel(dense != 0, acc, T(0));
            qacc[i] = fma_(acc, tau[j], qacc[i]);
            if constexpr (i < j) qacc[j] = fma_(acc, tau[i], qacc[j]);
          }
        });
      });
    }
    // columns of other groups are structural zeros
    sfor<row0, row0 + rows>([&](auto I) {
      sfor<0, N>([&](auto C) {
        constexpr int i = decltype(I)::value, c = decltype(C)::value;
        if constexpr (!grp_has(rt, c)) my[i * N + c] = T(0);
      });
    });
    if (fd && lane < nvalid) {
      sfor<row0, row0 + rows>([&](auto I) { constexpr int i = decltype(I)::value; qdd_out[b * N + i] = qacc[i]; });
    }
    if (Minv != nullptr) {
      __syncthreads();
      constexpr int RW = rows * N;
      T* gdst = Minv + cfg0 * (N * N) + row0 * N;
      constexpr int VE = 16 / sizeof(T);
      bool done = false;
      if constexpr (RW == N * N && MINV_LANE_TS == N * N && (64 * N * N) % VE == 0) {
        if (nvalid == 64) {     // LDS image == HBM image: flat 16-byte copies
          typedef T V __attribute__((ext_vector_type(VE)));
          const V* src = reinterpret_cast<const V*>(tile);
          V* dst = reinterpret_cast<V*>(gdst);
#pragma unroll 4
          for (int g = lane; g < 64 * N * N / VE; g += 64) dst[g] = src[g];
          done = true;
        }
      }
      if constexpr (RW % VE == 0 && (N * N) % VE == 0 && (row0 * N) % VE == 0 && (64 * (RW / VE)) % 64 == 0) {
        if (!done && nvalid == 64) {   // rows of one group among several: 16-byte pieces without a division per element
          flush_cfg_rows_full<T, 64, RW, MINV_LANE_TS, N * N, 64>(tile, gdst, lane);
          done = true;
        }
      }
      if (!done) {
#pragma unroll 4
        for (int g = lane; g < nvalid * RW; g += 64) {
          const int cfg = g / RW;
          const int rem2 = g - cfg * RW;
          gdst[cfg * (N * N) + rem2] = tile[cfg * MINV_LANE_TS + rem2];
        }
      }
      if constexpr (rows != N) __syncthreads();
    }
   }
  });
}

}  // namespace rbdk
