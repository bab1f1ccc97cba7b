// rbd_aba.h -- articulated-body algorithm, one configuration per lane (gfx950).
//
// RBDReference.aba(q, qd, tau, f_ext=[], GRAVITY) fixed-base branch (/root/reference/RBDReference.py:
// 940-1024) -> qdd.  Three sweeps per configuration:
//   1 (i up)   v_i = X v_p + S qd_i;  c_i = crm(v_i) S qd_i;  pA_i = crf(v_i) I_i v_i      (:963-984)
//   2 (i down) U_i = IA_i S; d_i = S^T U_i; u_i = tau_i - S^T pA_i;                          (:990-992)
//              Ia = IA_i - U U^T / d;  pa = pA_i + Ia c_i + U u / d;
//              IA_p += X^T Ia X;  pA_p += X^T pa                                             (:996-1007)
//   3 (i up)   a_i = X a_p + c_i (X a0 + c_i at a root);  qdd_i = (u_i - U_i^T a_i) / d_i;
//              a_i += S qdd_i                                                                 (:1015-1022)
// It yields the same qdd as forward_dynamics = Minv (tau - c) (:1371-1374) without ever forming Minv
// or c: algorithmic traffic 4n scalars per configuration.
//
// State that crosses sweeps (c_i, pA_i, then U_i, 1/d_i, u_i: ABA_SLOTS scalars per body) lives in
// registers for robots made of small root subtrees (each subtree is an independent problem and is
// taken through all three sweeps before the next one starts).  For larger trees it is parked,
// together with sin/cos of every joint, in lane-private LDS columns (slot * LANES + lane:
// conflict-free, no barrier needed); qd and tau are then read from memory where they are used, and
// every root subtree gets its own blocks (blockIdx.y).
#pragma once
#include "rbd_spatial.h"

namespace rbdk {

constexpr bool ABA_PARK = !MINV_LANE_OK;             // MINV_LANE_OK: root subtrees of at most 8 bodies
// per-body slots: c: 0..3 (qdd reuses 0), pA then U: 4..9, 1/d: 10, u: 11; parked robots also keep
// sin/cos there (12, 13) so that no per-body value has to stay in a register across the sweeps
constexpr int ABA_SLOTS = ABA_PARK ? 14 : 12;
// parked robots run one block per (64 configurations, root subtree): a block parks only its own
// group's bodies (Atlas: 18 / 6 / 6 instead of 30 => 64 KB instead of 107 KB, two blocks per CU)
constexpr int ABA_PARK_ROWS = GRAD_PER_ROOT ? grad_max_rows() : N;
template <class T>
constexpr int aba_lanes() {                          // configurations per block when parked
  int l = 64;
  while (l > 8 && (size_t)ABA_PARK_ROWS * ABA_SLOTS * l * sizeof(T) > 160u * 1024u) l /= 2;
  return l;
}
template <class T>
constexpr size_t aba_lds_bytes() {
  const size_t park = ABA_PARK ? (size_t)ABA_PARK_ROWS * ABA_SLOTS * aba_lanes<T>() * sizeof(T) : 0;
  const size_t stage = (size_t)64 * odd_pad<N>() * sizeof(T);
  return park > stage ? park : stage;
}

// per-lane state: registers (small root subtrees) or lane-private LDS columns (parked)
template <class T>
struct AbaRegs {
  JTrig<T> tr[N];
  T qdv[N], tauv[N], qddv[N];
  T r[N * ABA_SLOTS];
  template <int K> RBD_DEV void put(T x) { r[K] = x; }
  template <int K> RBD_DEV T get() const { return r[K]; }
  template <int I> RBD_DEV JTrig<T> trig() const { return tr[I]; }
  template <int I> RBD_DEV T qd() const { return qdv[I]; }
  template <int I> RBD_DEV T tau() const { return tauv[I]; }
  template <int I> RBD_DEV void set_qdd(T x) { qddv[I] = x; }
};
template <class T, int LANES, int ROW0>
struct AbaParked {
  T* base;              // lds + lane; slot K of body I sits at ((I - ROW0) * ABA_SLOTS + K) * LANES
  const T* qd_row;      // qd + b * N
  const T* tau_row;     // tau + b * N
  template <int K> RBD_DEV void put(T x) { base[(K - ROW0 * ABA_SLOTS) * LANES] = x; }
  template <int K> RBD_DEV T get() const { return base[(K - ROW0 * ABA_SLOTS) * LANES]; }
  template <int I> RBD_DEV JTrig<T> trig() const { return JTrig<T>{get<I * ABA_SLOTS + 12>(), get<I * ABA_SLOTS + 13>()}; }
  template <int I> RBD_DEV T qd() const { return qd_row[I]; }
  template <int I> RBD_DEV T tau() const { return tau_row[I]; }
  template <int I> RBD_DEV void set_qdd(T x) { put<I * ABA_SLOTS>(x); }
};

// structural non-zeros of c_i = crm(v) S qd  (revolute about k: a, b, 3+a, 3+b; prismatic: 3+a, 3+b)
constexpr bool c_nonzero(int j, int r) {
  const int k = AXIS[j], a = (k + 1) % 3, b = (k + 2) % 3;
  if (JTYPE[j] == 0) return r == a || r == b || r == 3 + a || r == 3 + b;
  return r == 3 + a || r == 3 + b;
}
constexpr int c_slot(int j, int r) {   // slot of component r among the non-zeros (0..3)
  int s = 0;
  for (int t = 0; t < r; ++t) s += c_nonzero(j, t) ? 1 : 0;
  return s;
}

template <class T, int ROW0, int ROWS, class Ctx>
RBD_DEV void aba_group(T grav, Ctx& st) {
  // ---- sweep 1 -----------------------------------------------------------------------------------
  {
    T v[N][6];
    sfor<ROW0, ROW0 + ROWS>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      const JTrig<T> g = st.template trig<i>();
      const T qdi = st.template qd<i>();
      if constexpr (p < 0) {
        sfor<0, 6>([&](auto R) { v[i][decltype(R)::value] = T(0); });
      } else {
        xform<i>(g, v[p], v[i]);
      }
      add_S<i>(qdi, v[i]);
      T ci[6], Iv[6], pa[6];
      mxS<i>(v[i], qdi, ci);                        // (:968); zero at a root since v = S qd there
      cmatvec<MatI, i>(v[i], Iv);
      fxv<false>(v[i], Iv, pa);                        // pA = crf(v) I v  (:974-984)
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        if constexpr (c_nonzero(i, r)) st.template put<i * ABA_SLOTS + c_slot(i, r)>(ci[r]);
        st.template put<i * ABA_SLOTS + 4 + r>(pa[r]);
      });
      if constexpr (ABA_PARK) pin6(v[i]);              // bodies stay in program order (bounds live state)
    });
  }
  // ---- sweep 2 -----------------------------------------------------------------------------------
  {
    T IA[N][6][6];
    sfor<ROW0, ROW0 + ROWS>([&](auto J) {
      sfor<0, 6>([&](auto R) {
        sfor<0, 6>([&](auto C) {
          constexpr int j = decltype(J)::value, r = decltype(R)::value, c = decltype(C)::value;
          IA[j][r][c] = T(IM[j][r * 6 + c]);
        });
      });
    });
    sfor_down<ROW0, ROW0 + ROWS>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      constexpr int si = s_index(i);
      T U[6], pA[6], ci[6];
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        U[r] = IA[i][r <= si ? r : si][r <= si ? si : r];          // (IA is symmetric: only its upper triangle is kept up to date)
        pA[r] = st.template get<i * ABA_SLOTS + 4 + r>();
        if constexpr (c_nonzero(i, r)) ci[r] = st.template get<i * ABA_SLOTS + c_slot(i, r)>();
        else ci[r] = T(0);
      });
      const T dinv = rcp_inertia(U[si]);
      const T u = st.template tau<i>() - pA[si];                    // (:992)
      if constexpr (p >= 0) {
        const JTrig<T> g = st.template trig<i>();
        // pa = pA + Ia c + U u / d with Ia c = IA c - U (U.c) / d                (:999)
        const T k = (u - dot6(U, ci)) * dinv;
        T pa[6];
        sfor<0, 6>([&](auto R) {
          constexpr int r = decltype(R)::value;
          T acc = fma_(U[r], k, pA[r]);
          sfor<0, 6>([&](auto C) {
            constexpr int c = decltype(C)::value;
            if constexpr (c_nonzero(i, c)) acc = fma_(IA[i][r <= c ? r : c][r <= c ? c : r], ci[c], acc);
          });
          pa[r] = acc;
        });
        T y[6];
        xform_T<i>(g, pa, y);                      // pA_p += X^T pa  (:1006-1007)
        sfor<0, 6>([&](auto R) {
          constexpr int r = decltype(R)::value;
          st.template put<p * ABA_SLOTS + 4 + r>(st.template get<p * ABA_SLOTS + 4 + r>() + y[r]);
        });
        // IA_p += X^T (IA - U U^T / d) X  (:996-1004)
        T A[6][6];
        sfor<0, 6>([&](auto C) {
          constexpr int c = decltype(C)::value;
          T col[6], yc[6];
          const T uc = U[c] * dinv;
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; col[r] = fma_(-U[r], uc, IA[i][r <= c ? r : c][r <= c ? c : r]); });
          xform_T<i>(g, col, yc);
          sfor<0, 6>([&](auto R) { A[decltype(R)::value][c] = yc[decltype(R)::value]; });
        });
        sfor<0, 6>([&](auto R) {
          constexpr int r = decltype(R)::value;
          T yr[6];
          xform_T<i>(g, A[r], yr);
          sfor<r, 6>([&](auto C) { IA[p][r][decltype(C)::value] += yr[decltype(C)::value]; });   // upper triangle of the symmetric X^T Ia X
        });
      }
      // hand U, 1/d, u to sweep 3 (U takes over pA's slots)
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; st.template put<i * ABA_SLOTS + 4 + r>(U[r]); });
      st.template put<i * ABA_SLOTS + 10>(dinv);
      st.template put<i * ABA_SLOTS + 11>(u);
      if constexpr (ABA_PARK) pin6(U);
    });
  }
  // ---- sweep 3 -----------------------------------------------------------------------------------
  {
    T a[N][6];
    sfor<ROW0, ROW0 + ROWS>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      const JTrig<T> g = st.template trig<i>();
      if constexpr (p < 0) {
        const T a0[6] = {T(0), T(0), T(0), T(0), T(0), -grav};   // (:954-955, :1015)
        xform<i>(g, a0, a[i]);
      } else {
        xform<i>(g, a[p], a[i]);
      }
      T U[6];
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        if constexpr (c_nonzero(i, r)) a[i][r] += st.template get<i * ABA_SLOTS + c_slot(i, r)>();
        U[r] = st.template get<i * ABA_SLOTS + 4 + r>();
      });
      const T qi = (st.template get<i * ABA_SLOTS + 11>() - dot6(U, a[i])) * st.template get<i * ABA_SLOTS + 10>();   // (:1020-1021)
      st.template set_qdd<i>(qi);                                 // (c's first slot is free by now)
      add_S<i>(qi, a[i]);                                         // (:1022)
      if constexpr (ABA_PARK) pin6(a[i]);
    });
  }
}

template <class T>
__global__ __launch_bounds__(64) void aba_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                 const T* __restrict__ tau, T grav, long long B,
                                                 T* __restrict__ qdd_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  constexpr int LANES = ABA_PARK ? aba_lanes<T>() : 64;
  const int lane = threadIdx.x;
  const long long cfg0 = (long long)blockIdx.x * LANES;
  const long long rem = B - cfg0;
  const int nvalid = rem < LANES ? (int)rem : LANES;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);

  if constexpr (ABA_PARK) {
    const int gsel = blockIdx.y;
    sfor<0, N>([&](auto Rt) {
      constexpr int rt = decltype(Rt)::value;
      if constexpr (grp_head(rt)) {
        constexpr int gi = grp_index(rt);
        constexpr int row0 = grp_row0(rt), rows = grp_rows(rt);
        if (gi == gsel) {
          AbaParked<T, LANES, row0> st{lds + (lane < LANES ? lane : 0), qd + b * N, tau + b * N};
          if (lane < LANES) {
            sfor<row0, row0 + rows>([&](auto J) {
              constexpr int j = decltype(J)::value;
              const JTrig<T> g = make_trig<j>(q[b * N + j]);
              st.template put<j * ABA_SLOTS + 12>(g.s);
              st.template put<j * ABA_SLOTS + 13>(g.c);
            });
            aba_group<T, row0, rows>(grav, st);
          }
          __syncthreads();
          // qdd sits in slot 0 of every body of the group -> qdd_out[cfg0 + cfg][row0 + i]
          for (int g = lane; g < nvalid * rows; g += 64) {
            const int cfg = g / rows;
            const int i = g - cfg * rows;
            qdd_out[(cfg0 + cfg) * N + row0 + i] = lds[(i * ABA_SLOTS) * LANES + cfg];
          }
        }
      }
    });
  } else {
    AbaRegs<T> st;
    sfor<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      st.tr[j] = make_trig<j>(q[b * N + j]);
      st.qdv[j] = qd[b * N + j];
      st.tauv[j] = tau[b * N + j];
    });
    sfor<0, N>([&](auto Rt) {
      constexpr int rt = decltype(Rt)::value;
      if constexpr (grp_head(rt)) aba_group<T, grp_row0(rt), grp_rows(rt)>(grav, st);
    });
    staged_store<N>(lds, st.qddv, qdd_out + cfg0 * N, lane, nvalid);
  }
}

}  // namespace rbdk
