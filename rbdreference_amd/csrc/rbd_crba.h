// rbd_crba.h -- crba(q): joint-space inertia matrix H [B, n, n]  (RBDReference.py:1091-1124, fixed base).
// SURVEY.md §8f-4 "next" row; the golden files already hold the reference's H (it is the Minv H = I
// witness).  One configuration per lane, one sweep from the leaves up: a body's composite inertia is final when the
// sweep reaches it (IC_p += X^T IC X, :1096-1103), fh = IC_i S_i is carried up its root path right then,
// H[i, j] = S_j^T fh (:1107-1122), and the composite moves on to the parent.  Rows are staged per group (root subtree) in LDS and streamed out coalesced; robots
// whose largest group tile exceeds LDS write H with per-lane strided stores instead.
#pragma once
#include "rbd_spatial.h"

namespace rbdk {

constexpr int CRBA_TS = (grad_max_rows() * N) | 1;
template <class T>
constexpr bool crba_tile_fits() { return (size_t)64 * CRBA_TS * sizeof(T) <= 150 * 1024; }

template <class T>
__global__ __launch_bounds__(64) void crba_kernel(const T* __restrict__ q, long long B, T* __restrict__ H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x;
  const long long cfg0 = (long long)blockIdx.x * 64;
  const long long rem = B - cfg0;
  const int nvalid = rem < 64 ? (int)rem : 64;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);
  constexpr bool TILE = crba_tile_fits<T>();

  JTrig<T> tr[N];
  T qv[N];
  sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; qv[j] = q[b * N + j]; });
  sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j] = make_trig<j>(qv[j]); });

  sfor<0, N>([&](auto Rt) {
   constexpr int rt = decltype(Rt)::value;
   if constexpr (grp_head(rt)) {
    constexpr int row0 = grp_row0(rt);
    constexpr int rows = grp_rows(rt);
    T* my = TILE ? tile + lane * CRBA_TS - row0 * N : H + b * (N * N);   // my[i * N + c]
    // composite inertias (:1096-1103)
    T IC[N][6][6];
    sfor<row0, row0 + rows>([&](auto J) {
      sfor<0, 6>([&](auto R) {
        sfor<0, 6>([&](auto C) {
          constexpr int j = decltype(J)::value, r = decltype(R)::value, c = decltype(C)::value;
          IC[j][r][c] = T(IM[j][r * 6 + c]);
        });
      });
    });
    // One descending sweep: when body i is reached every child has added its composite, so IC_i is final --
    // fh = IC_i S_i climbs the root path NOW (:1107-1122) and IC_i is handed to the parent and dies (:1096-1103).
    // (Two separate sweeps keep the composite of every body of the group alive between them: 36 n values, scratch
    // for 18 bodies in fp64.)
    const bool ok = TILE || lane < nvalid;
    sfor_down<row0, row0 + rows>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      constexpr int si = s_index(i);
      {
        T fh[6];
        sfor<0, 6>([&](auto R) { fh[decltype(R)::value] = IC[i][decltype(R)::value][si]; });   // IC_i S_i
        if (ok) my[i * N + i] = fh[si];
        // structural zeros: bodies of this group that are unrelated to i, and every other group
        sfor<0, N>([&](auto C) {
          constexpr int c = decltype(C)::value;
          if constexpr (!related(i, c)) { if (ok) my[i * N + c] = T(0); }
        });
        // climb the root path
        sfor_down<row0, i + 1>([&](auto JJ) {
          constexpr int jj = decltype(JJ)::value;           // ancestors-or-self of i, in descending index order
          if constexpr (is_anc_or_self(jj, i) && PARENT[jj] >= 0) {
            constexpr int pj = PARENT[jj];
            T y[6];
            xform_T<jj>(tr[jj], fh, y);
            sfor<0, 6>([&](auto R) { fh[decltype(R)::value] = y[decltype(R)::value]; });
            const T h = S_dot<pj>(fh);
            if (ok) { my[i * N + pj] = h; my[pj * N + i] = h; }
          }
        });
      }
      if constexpr (p >= 0) {
        T A[6][6];   // A = X^T IC
        sfor<0, 6>([&](auto C) {
          constexpr int c = decltype(C)::value;
          T col[6], y[6];
          sfor<0, 6>([&](auto R) { col[decltype(R)::value] = IC[i][decltype(R)::value][c]; });
          xform_T<i>(tr[i], col, y);
          sfor<0, 6>([&](auto R) { A[decltype(R)::value][c] = y[decltype(R)::value]; });
        });
        sfor<0, 6>([&](auto R) {
          constexpr int r = decltype(R)::value;
          T y[6];
          xform_T<i>(tr[i], A[r], y);
          sfor<0, 6>([&](auto C) { IC[p][r][decltype(C)::value] += y[decltype(C)::value]; });
        });
      }
    });
    if constexpr (TILE) {
      __syncthreads();
      constexpr int RW = rows * N;
      T* gdst = H + cfg0 * (N * N) + row0 * N;
#pragma unroll 4
      for (int g = lane; g < nvalid * RW; g += 64) {
        const int cfg = g / RW;
        const int rem2 = g - cfg * RW;
        gdst[cfg * (N * N) + rem2] = tile[cfg * CRBA_TS + rem2];
      }
      if constexpr (rows != N) __syncthreads();
    }
   }
  });
}

}  // namespace rbdk
