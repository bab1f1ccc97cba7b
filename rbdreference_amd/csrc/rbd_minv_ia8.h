// rbd_minv_ia8.h -- phase A of minv (articulated inertias, RBDReference.py:697-700, :728-733) with EIGHT
// lanes per configuration, for launches that are too small to fill the chip with one lane per
// configuration (Atlas, B = 16 384: 256 waves on 1 024 SIMDs).
//
// Lane c (< 6) of a configuration's 8-lane group owns COLUMN c of every articulated inertia IA_i:
//   U = IA S          = column s of IA          -> broadcast from lane s with two DPP moves (no LDS)
//   Ia[:, c]          = IA[:, c] - U (U[c] / D) -> U[c] is the lane's own element s (IA is symmetric)
//   A[:, c]           = X^T Ia[:, c]            -> one xform_T per lane
//   (X^T Ia X)[:, c]  = X^T (A[c, :])^T         -> row c of A is gathered through a 6 x 6 LDS transpose,
//                                                  then one more xform_T per lane
// i.e. 2 transforms per lane per body instead of 12 in one lane, and 8x the waves.  sin/cos of the block's
// joints are computed once per 8-lane group (lane c takes the c-th, (c + 8)-th ... joint) and broadcast the same way.
// Writes the same [body][config][12] records as minv_ia_kernel.
#pragma once
#include "rbd_spatial.h"

namespace rbdk {

// broadcast the value of lane L (0..7) of every aligned 8-lane group with two DPP moves (VALU, no LDS
// traffic): quad_perm spreads lane L % 4 of every quad over its quad, then row_half_mirror (lane i <- lane 7 - i
// of its 8-lane half row) carries the value of L's quad into the other quad of the group -- bank_mask
// enables only the quads that do not hold L, the others keep the first move's result.  (The first version
// used ds_swizzle: 8 LDS-pipe operations per body, and at B = 16 384 the kernel's body steps were bound by
// the CU's LDS issue rate, 0.5 us per body against 0.2 us for a lone block.)
template <int L>
RBD_DEV int grp8_bcast_i(int x) {
  constexpr int q = L % 4;
  constexpr int quad = q | (q << 2) | (q << 4) | (q << 6);          // quad_perm [q, q, q, q]
#ifdef RBD_EXP_DPP_TIED
  const int x1 = __builtin_amdgcn_update_dpp(x, x, quad, 0xF, 0xF, false);
#else
  const int x1 = __builtin_amdgcn_mov_dpp(x, quad, 0xF, 0xF, true);   // every lane written: no tied `old` operand, so no copy of x when x stays live
#endif
  return __builtin_amdgcn_update_dpp(x1, x1, 0x141 /* row_half_mirror */, 0xF, L < 4 ? 0xA : 0x5, false);
}
template <int L>
RBD_DEV float grp8_bcast(float x) { return __builtin_bit_cast(float, grp8_bcast_i<L>(__builtin_bit_cast(int, x))); }
template <int L>
RBD_DEV double grp8_bcast(double x) {
  unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  int lo = grp8_bcast_i<L>((int)(u & 0xffffffffu));
  int hi = grp8_bcast_i<L>((int)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

constexpr int last_child_of(int p) {   // the child of p with the largest index (visited first going down)
  int m = -1;
  for (int j = 0; j < N; ++j)
    if (PARENT[j] == p) m = j;
  return m;
}

// Root subtrees of at most 8 bodies (Atlas' legs) are FINISHED here when `fuse_small` is set: the 8 lanes of a
// configuration hold U, 1/D, sin, cos of every body anyway, and the articulated-inertia recursion visits the bodies
// in the order of the columns' backward sweep, so lane c (< rows) carries column row0 + c along: its backward step
// right after U_i, D_i exist, its forward sweep from the records kept in registers, its column into an LDS tile,
// the group's rows out from there (and qdd = Minv (u - c) for forward dynamics).  No workspace round trip, and no
// blocks of the column kernel: at B = 16 384 the legs' blocks there (12 body steps each) had cost as much as the
// torso's (LDS capacity serialised the two: 12.8 of the column kernel's 25.5 us).
constexpr bool minv_small_group(int rt) { return GRAD_PER_ROOT && grp_rows(rt) <= 8; }
constexpr int IA8_TS = 8 * 8 + 1;              // LDS tile stride of a small group's [rows][rows] block

// One group (root subtree) RT for the 8 configurations cfg0 .. cfg0 + 7, by ONE wave: `lane` = lane of that wave,
// tr_lds [64 * 6], im_lds [N * 36], tile_s [8 * IA8_TS], tau_s [64] = LDS of that wave (im_lds may be shared by
// waves that work on the same group).  Contains block barriers: every wave of the block must call it.
template <class T, int rt>
RBD_DEV void ia8_group(const T* __restrict__ q, long long B, T* __restrict__ ws, int fuse_small, int dense, T* __restrict__ Minv,
                       const T* __restrict__ u_in, const T* __restrict__ c_in, T* __restrict__ qdd_out, long long cfg0, int lane,
                       T* tr_lds, T* im_lds, T* tile_s, T* tau_s) {
  const int c = lane & 7;                      // column owned by this lane (6, 7: idle columns)
  const int cc = c < 6 ? c : 0;                // clamp for table reads
  const int grp = lane >> 3;
  const long long b0 = cfg0 + grp;
  const bool valid = b0 < B;
  const long long b = valid ? b0 : B - 1;
  // sin / cos (or q for prismatic joints) of THIS group's joints only (the group's bodies are contiguous):
  // lane c handles joints row0 + c, row0 + c + 8, ...  (all 30 joints in every wave had been a third of the
  // kernel's VALU work)
  constexpr int R0 = grp_row0(rt), RN = grp_rows(rt), NR = (RN + 7) / 8;
  T s_l[NR], c_l[NR];
  sfor<0, NR>([&](auto K) {
    constexpr int k = decltype(K)::value;
    const int j = R0 + k * 8 + c;
    const int jj = j < R0 + RN ? j : R0 + RN - 1;
    const T qv = q[b * N + jj];
    T sv, cv;
    sincos_(qv, &sv, &cv);
    const bool pris = JTYPE[jj] != 0;           // runtime-indexed constexpr table
    s_l[k] = sel(pris, qv, sv);
    c_l[k] = sel(pris, T(0), cv);
  });
  for (int k = lane; k < RN * 36; k += 64) im_lds[R0 * 36 + k] = T(IM[R0 + k / 36][k % 36]);   // this group's inertias
  __syncthreads();                            // (every wave of the block is in a call of this function)
  T IAc[N][6];
  constexpr bool SMALL = minv_small_group(rt);
  const bool fused = SMALL && fuse_small != 0;
  // this lane's column of a small group (lanes c >= rows repeat the last one and store nothing)
  const int jc = R0 + (c < RN ? c : RN - 1);
  T mcol[N], wj[N];                              // column jc: m[i][jc] = minv_bpass's Minv[i, jc], and D_i m[i][jc]
  T Fj[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  sfor_down<grp_row0(rt), grp_row0(rt) + grp_rows(rt)>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int p = PARENT[i];
    constexpr int si = s_index(i);
    JTrig<T> tri;
    tri.s = grp8_bcast<(i - R0) % 8>(s_l[(i - R0) / 8]);
    tri.c = grp8_bcast<(i - R0) % 8>(c_l[(i - R0) / 8]);
    if constexpr (!has_child(i)) {
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; IAc[i][r] = im_lds[i * 36 + r * 6 + cc]; });
    }
    T U[6];
    sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; U[r] = grp8_bcast<si>(IAc[i][r]); });   // U = IA S (:697)
    const T Dinv = rcp_inertia(U[si]);                                                                          // :698,:700
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
      if (valid && c < VPB && !fused) reinterpret_cast<V*>(ws + ((long long)i * B + b) * MINV_WS)[c] = piece;
    }
    if constexpr (SMALL) {
      if (fused) {
        // backward step of column jc at body i (:700-726), as in minv_cols_group
        constexpr unsigned long long mask = subtree_mask(i);
        const bool insub = ((mask >> jc) & 1ull) != 0;
        T m = sel(jc == i, Dinv, -(Dinv * S_dot<i>(Fj)));
        m = sel(insub, m, T(0));
        mcol[i] = m;
        wj[i] = U[si] * m;                               // D_i m[i][jc]   (D_i = S^T U_i, :698)
        if constexpr (p >= 0) {
          T t[6], y[6];
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; t[r] = fma_(U[r], m, Fj[r]); });
          xform_T<i>(tri, t, y);
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Fj[r] = sel(insub, y[r], Fj[r]); });
        }
      }
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
  if constexpr (SMALL) {
    if (fused) {
      // ---- no forward sweep (round 4): Minv = Psi^T D^-1 Psi (rbd_kernels.hip, minv_cols_class_bwd; exact, tools/
      //      check_minv_factorisation.py):  Minv[i, jc] = sum over k in anc(i) & anc(jc) of (D_k m[k][jc]) m[k][i].  The table of
      //      all columns' m is the tile's upper triangle; U, 1/D, sin / cos of the group's bodies need not stay in registers.
      T* myt = tile_s + grp * IA8_TS;                    // myt[(i - R0) * RN + (col - R0)]
      const int jl = jc - R0;
      if (c < RN) {
        sfor<R0, R0 + RN>([&](auto I) { constexpr int i = decltype(I)::value; if (i <= jc) myt[(i - R0) * RN + jl] = mcol[i]; });
      }
      __syncthreads();
      T accv[N];
      sfor<R0, R0 + RN>([&](auto I) {
        constexpr int i = decltype(I)::value;
        T a = T(0);
        sfor<R0, i + 1>([&](auto K) {
          constexpr int k = decltype(K)::value;
          if constexpr (is_anc_or_self(k, i)) a = fma_(wj[k], myt[(k - R0) * RN + (i - R0)], a);
        });
        accv[i] = a;
      });
      __syncthreads();
      // ---- the column into the tile, mirrored (:799-804) -----------------------------------------------------------
      if (c < RN) {
        sfor<R0, R0 + RN>([&](auto I) {
          constexpr int i = decltype(I)::value;
          if (i <= jc) myt[(i - R0) * RN + jl] = accv[i];
          if (i < jc) myt[jl * RN + (i - R0)] = sel(dense != 0, accv[i], T(0));
        });
      }
      __syncthreads();
      const long long rem = B - cfg0;
      const int nvalid = rem < 8 ? (rem > 0 ? (int)rem : 0) : 8;
      if (qdd_out != nullptr) {                          // forward dynamics (:1371-1374): lane jc owns row jc
        if (valid && c < RN) tau_s[grp * 8 + jl] = u_in[b * N + jc] - c_in[b * N + jc];
        __syncthreads();
        if (valid && c < RN) {
          T o = T(0);
          sfor<0, RN>([&](auto K) { constexpr int k = decltype(K)::value; o = fma_(myt[jl * RN + k], tau_s[grp * 8 + k], o); });
          qdd_out[b * N + jc] = o;
        }
      }
      if (Minv != nullptr) {
        constexpr int RW = RN * N;
        T* gdst = Minv + cfg0 * (N * N) + R0 * N;
        auto elem = [&](int cfg, int e) -> T {           // (row e / N, column e % N): own columns from the tile, the rest zero
          const int r = e / N;
          const int cidx = e - r * N - R0;
          const bool own = cidx >= 0 && cidx < RN;
          const T x = tile_s[cfg * IA8_TS + r * RN + (own ? cidx : 0)];
          return own ? x : T(0);
        };
        if constexpr (minv_piece_flush<T>(rt)) {
          minv_own_rows_flush<T, R0, RN, 8, IA8_TS, 64>(tile_s, gdst, lane, nvalid);
        } else {
          const int total = nvalid * RW;
#pragma unroll 4
          for (int g = lane; g < total; g += 64) {
            const int cfg = g / RW;
            const int r2 = g - cfg * RW;
            gdst[(long long)cfg * (N * N) + r2] = elem(cfg, r2);
          }
        }
      }
    }
  }
}

template <class T>
__global__ __launch_bounds__(64, 2) void minv_ia8_kernel(const T* __restrict__ q, long long B, T* __restrict__ ws, int fuse_small,
                                                         int dense, T* __restrict__ Minv, const T* __restrict__ u_in,
                                                         const T* __restrict__ c_in, T* __restrict__ qdd_out) {
  __shared__ T tr_lds[64 * 6];                 // per-lane 6-vector exchange (one group = 8 x 6 values)
  __shared__ T im_lds[N * 36];                 // the robot's spatial inertias
  __shared__ T tile_s[8 * IA8_TS];             // small groups: [configuration][rows][rows] of Minv
  __shared__ T tau_s[8 * 8];                   //               u - c of the group's joints
  // independent root subtrees (groups) run in separate blocks: blockIdx.y picks the group, which
  // shortens the serial body chain of a wave from n to the group's size
  const int gsel = blockIdx.y;
  sfor<0, N>([&](auto Rt_) {
    constexpr int rt = decltype(Rt_)::value;
    if constexpr (grp_head(rt)) {
      constexpr int gi = grp_index(rt);   // constexpr on purpose (a plain call would walk the tree at run time)
      if (gi == gsel)
        ia8_group<T, rt>(q, B, ws, fuse_small, dense, Minv, u_in, c_in, qdd_out, (long long)blockIdx.x * 8, (int)threadIdx.x,
                         tr_lds, im_lds, tile_s, tau_s);
    }
  });
}

}  // namespace rbdk
