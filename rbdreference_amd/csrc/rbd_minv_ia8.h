// rbd_minv_ia8.h -- phase A of minv (articulated inertias, RBDReference.py:697-700, :728-733) with EIGHT
// lanes per configuration, for launches that are too small to fill the chip with one lane per
// configuration (Atlas, B = 16 384: 256 waves on 1 024 SIMDs).
//
// Lane c (< 6) of a configuration's 8-lane group owns COLUMN c of every articulated inertia IA_i:
//   U = IA S          = column s of IA          -> broadcast from lane s with ds_swizzle (no LDS memory)
//   Ia[:, c]          = IA[:, c] - U (U[c] / D) -> U[c] is the lane's own element s (IA is symmetric)
//   A[:, c]           = X^T Ia[:, c]            -> one xform_T per lane
//   (X^T Ia X)[:, c]  = X^T (A[c, :])^T         -> row c of A is gathered through a 6 x 6 LDS transpose,
//                                                  then one more xform_T per lane
// i.e. 2 transforms per lane per body instead of 12 in one lane, and 8x the waves.  sin/cos of the
// joints are computed once per group (lane c takes joints c, c + 8, ...) and broadcast the same way.
// Writes the same [body][config][12] records as minv_ia_kernel.
#pragma once
#include "rbd_spatial.h"

namespace rbdk {

// broadcast the value of lane L (0..7) of every aligned 8-lane group: ds_swizzle bit-mode,
// lane' = (lane & 0x18) | L  within each 32-lane half
template <int L>
RBD_DEV float grp8_bcast(float x) {
  constexpr int pattern = (0x18) | (L << 5) | (0 << 10);   // and_mask[4:0] | or_mask[9:5] | xor_mask[14:10]
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), pattern));
}
template <int L>
RBD_DEV double grp8_bcast(double x) {
  constexpr int pattern = (0x18) | (L << 5);
  unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  int lo = __builtin_amdgcn_ds_swizzle((int)(u & 0xffffffffu), pattern);
  int hi = __builtin_amdgcn_ds_swizzle((int)(u >> 32), pattern);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

constexpr int last_child_of(int p) {   // the child of p with the largest index (visited first going down)
  int m = -1;
  for (int j = 0; j < N; ++j)
    if (PARENT[j] == p) m = j;
  return m;
}

template <class T>
__global__ __launch_bounds__(64, 2) void minv_ia8_kernel(const T* __restrict__ q, long long B, T* __restrict__ ws) {
  __shared__ T tr_lds[64 * 6];                 // per-lane 6-vector exchange (one group = 8 x 6 values)
  __shared__ T im_lds[N * 36];                 // the robot's spatial inertias
  const int lane = threadIdx.x;
  const int c = lane & 7;                      // column owned by this lane (6, 7: idle columns)
  const int cc = c < 6 ? c : 0;                // clamp for table reads
  const int grp = lane >> 3;
  const long long b0 = (long long)blockIdx.x * 8 + grp;
  const bool valid = b0 < B;
  const long long b = valid ? b0 : B - 1;

  // sin / cos (or q for prismatic joints): lane c handles joints c, c + 8, c + 16, ...
  constexpr int NR = (N + 7) / 8;
  T s_l[NR], c_l[NR];
  sfor<0, NR>([&](auto K) {
    constexpr int k = decltype(K)::value;
    const int j = k * 8 + c;
    const int jj = j < N ? j : N - 1;
    const T qv = q[b * N + jj];
    T sv, cv;
    sincos_(qv, &sv, &cv);
    const bool pris = JTYPE[jj] != 0;           // runtime-indexed constexpr table
    s_l[k] = sel(pris, qv, sv);
    c_l[k] = sel(pris, T(0), cv);
  });
  // Column cc of a body's spatial inertia (runtime column index) is read from an LDS copy of the
  // constant table when the body is first needed.  (Read lazily from the constant segment itself, each
  // body paid one L2 round trip on its critical path and the kernel was no faster than one lane per
  // configuration; read up front, the 6 N values spilled.)
  for (int k = lane; k < N * 36; k += 64) im_lds[k] = T(IM[k / 36][k % 36]);
  __syncthreads();
  // independent root subtrees (groups) run in separate blocks: blockIdx.y picks the group, which
  // shortens the serial body chain of a wave from n to the group's size
  const int gsel = blockIdx.y;
  sfor<0, N>([&](auto Rt_) {
   constexpr int rt = decltype(Rt_)::value;
   if constexpr (grp_head(rt)) {
   constexpr int gi = grp_index(rt);   // constexpr on purpose (a plain call would walk the tree at run time)
   if (gi == gsel) {
  T IAc[N][6];
  sfor_down<grp_row0(rt), grp_row0(rt) + grp_rows(rt)>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int p = PARENT[i];
    constexpr int si = s_index(i);
    JTrig<T> tri;
    tri.s = grp8_bcast<i % 8>(s_l[i / 8]);
    tri.c = grp8_bcast<i % 8>(c_l[i / 8]);
    if constexpr (!has_child(i)) {
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; IAc[i][r] = im_lds[i * 36 + r * 6 + cc]; });
    }
    T U[6];
    sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; U[r] = grp8_bcast<si>(IAc[i][r]); });   // U = IA S (:697)
    const T Dinv = T(1) / U[si];                                                                          // :698,:700
    // record {U[6], 1/D, s, c, 0, 0, 0}: every lane of the group holds all of it, so lanes 0..VPB-1 each
    // store one 16-byte piece -> one store instruction per body covers the block's 8 x 48 contiguous bytes
    {
      constexpr int VE = 16 / sizeof(T);
      constexpr int VPB = MINV_WS / VE;
      typedef T V __attribute__((ext_vector_type(VE)));
      const T flat[MINV_WS] = {U[0], U[1], U[2], U[3], U[4], U[5], Dinv, tri.s, tri.c, T(0), T(0), T(0)};
      V piece;
      sfor<0, VE>([&](auto E) {
        constexpr int e = decltype(E)::value;
        T x = flat[e];
        sfor<1, VPB>([&](auto P) { constexpr int pp = decltype(P)::value; x = sel(c == pp, flat[pp * VE + e], x); });
        piece[e] = x;
      });
      if (valid && c < VPB) reinterpret_cast<V*>(ws + ((long long)i * B + b) * MINV_WS)[c] = piece;
    }
    if constexpr (p >= 0) {
      // Ia[:, c] = IA[:, c] - U * (U[c] / D), with U[c] = IA[s][c] = this lane's element s
      const T uc = IAc[i][si] * Dinv;
      T col[6], y[6];
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; col[r] = fma_(-U[r], uc, IAc[i][r]); });
      xform_T<i>(tri, col, y);                         // column c of A = X^T Ia
      // row c of A through LDS: lane k wrote A[:, k]; lane c reads A[c][k] for k = 0..5.  The block
      // is ONE wave and a wave's LDS operations execute in order, so a wave-level fence (compiler
      // ordering only) replaces __syncthreads() -- whose s_waitcnt vmcnt(0) would also wait for the
      // record stores above: ~1.5 k cycles per body, the whole run time of the first version.
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; tr_lds[lane * 6 + r] = y[r]; });
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      T row[6], z[6];
      sfor<0, 6>([&](auto K) { constexpr int k = decltype(K)::value; row[k] = tr_lds[(grp * 8 + k) * 6 + cc]; });
      xform_T<i>(tri, row, z);                         // column c of X^T Ia X (symmetric)
      // IA_p += X^T Ia X (:732-733); the child processed first also brings in I_p itself
      constexpr bool first = last_child_of(p) == i;
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        if constexpr (first) IAc[p][r] = im_lds[p * 36 + r * 6 + cc] + z[r]; else IAc[p][r] += z[r];
      });
      pin6(IAc[p]);   // ordering point: keeps the bodies in program order (bounds live registers)
    }
  });
   }
   }
  });
}

}  // namespace rbdk
