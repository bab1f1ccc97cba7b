// rbd_idsva_tree.h -- rnea_grad for TREES, one configuration per lane, chain by chain.
//
// Same world-frame identities as rbd_idsva.h (first-order part of the reference's IDSVA scheme,
// /root/reference/RBDReference.py:1413-1484; result = RBDReference.rnea_grad, :1345-1368), extended
// from single chains to arbitrary fixed-base trees of revolute joints with rigid-body inertias:
//
//   * the tree is cut into chains at compile time (heavy-path decomposition: every body continues
//     the chain of its parent iff it is the child with the largest subtree);
//   * chains are processed from the highest head index down, so every chain hanging off a body has
//     been finished -- and has parked its composite {IC, Sym, momentum, force} (31 scalars) in LDS --
//     before the chain that carries that body starts;
//   * per chain: world kinematics root -> leaf (S, psid, psidd of the whole root path stay in
//     registers), then leaf -> head: local inertia terms, composites (+ parked child chains),
//     t-vectors, and for every ancestor jj of the body j
//        row entries   (j, jj)  = t4.psid_jj + t1.psidd_jj | t4.S_jj + 2 t1.psid_jj  -> row j, now
//        column entries (jj, j) = S_jj.t3 | S_jj.t2          -> parked in LDS until row jj is built;
//   * a row is complete the moment its body is processed (descendants have already delivered their
//     column entries), so it is assembled in registers (structural zeros included), written to a
//     64 x 2n LDS image and streamed out as 8n-byte contiguous segments -- every entry of dc_du is
//     produced exactly once, no accumulation, no tile of the whole matrix.
//
// Why: the column-recursion kernel needs the whole [rows x 2n] accumulator tile of a root subtree in
// LDS (Atlas: 139 KB => one 32-configuration wave per CU, 50 k instructions per lane); here a block
// needs 66 KB for 64 configurations of the same robot and ~20 k instructions per lane.
#pragma once
#include "rbd_idsva.h"

namespace rbdk {

// ---- compile-time chain decomposition (tables, so that constexpr evaluation stays O(n^2)) ---------
struct TreePlan {
  int sub[N > 0 ? N : 1] = {};       // subtree size
  int heavy[N > 0 ? N : 1] = {};     // heavy child or -1
  int head[N > 0 ? N : 1] = {};      // head of the body's chain
  int pos[N > 0 ? N : 1] = {};       // position in the chain (0 at the head)
  int leaf[N > 0 ? N : 1] = {};      // leaf of the chain that starts at a head (valid at heads)
  int len[N > 0 ? N : 1] = {};       // chain length (valid at heads)
  int park[N > 0 ? N : 1] = {};      // slot of the parked composite (heads with a parent)
  int cross0[N > 0 ? N : 1] = {};    // number of cross pairs (x, y) with y < j
  int rootidx[N > 0 ? N : 1] = {};   // index of the body's root among the roots (0, 1, ...)
  int side_head[N > 0 ? N : 1] = {}; // per ROOT INDEX: head of the side subtree that runs on the block's second wave, -1: none
  bool on_side[N > 0 ? N : 1] = {};  // body belongs to its root's side subtree
  bool any_side = false;
  int wave_of[N > 0 ? N : 1] = {};   // wave (of a one-block-per-64-configurations layout) that runs the body's chain
  int wave_len[16] = {};             // longest chain of a wave
  int inch_off[17] = {};             // first in-chain pending slot of a wave (prefix sums of L (L - 1))
  int n_waves = 0;
  int n_cross = 0, n_park = 0, max_len = 0, n_roots = 0;
  constexpr TreePlan() {
    for (int i = 0; i < N; ++i) rootidx[i] = PARENT[i] < 0 ? n_roots++ : rootidx[PARENT[i]];
    for (int i = 0; i < N; ++i) sub[i] = 1;
    for (int i = N - 1; i >= 0; --i)
      if (PARENT[i] >= 0) sub[PARENT[i]] += sub[i];
    for (int i = 0; i < N; ++i) heavy[i] = -1;
    for (int i = 0; i < N; ++i) {          // lowest index wins ties
      const int p = PARENT[i];
      if (p >= 0 && (heavy[p] < 0 || sub[i] > sub[heavy[p]])) heavy[p] = i;
    }
    for (int i = 0; i < N; ++i) {
      const int p = PARENT[i];
      if (p >= 0 && heavy[p] == i) { head[i] = head[p]; pos[i] = pos[p] + 1; }
      else { head[i] = i; pos[i] = 0; }
    }
    for (int i = 0; i < N; ++i) { leaf[i] = i; len[i] = 0; }
    for (int i = 0; i < N; ++i) {
      const int h = head[i];
      if (pos[i] + 1 > len[h]) { len[h] = pos[i] + 1; leaf[h] = i; }
    }
    for (int i = 0; i < N; ++i) {
      if (head[i] == i) {
        if (len[i] > max_len) max_len = len[i];
        if (PARENT[i] >= 0) park[i] = n_park++;
      }
    }
    for (int j = 0; j < N; ++j) {
      cross0[j] = n_cross;
      for (int x = PARENT[j]; x >= 0; x = PARENT[x])
        if (head[x] != head[j]) ++n_cross;
    }
    // the side subtree of a root: the biggest chain-head subtree (>= TREE_SIDE_MIN bodies) hanging directly off
    // the root's heavy chain; it depends on nothing outside itself but the root path's kinematics (recomputed) and
    // is needed only when the heavy chain's upward sweep reaches its parent
    for (int r = 0; r < N; ++r) side_head[r] = -1;
    for (int h = 0; h < N; ++h) {
      const int p = PARENT[h];
      if (head[h] != h || p < 0) continue;
      int rt = h;
      while (PARENT[rt] >= 0) rt = PARENT[rt];
      if (head[p] != rt) continue;                       // parent is not on the root's heavy chain
      if (sub[h] < 4) continue;
      const int ri = rootidx[h];
      if (side_head[ri] < 0 || sub[h] > sub[side_head[ri]]) side_head[ri] = h;
    }
    for (int j = 0; j < N; ++j) {
      const int sh = side_head[rootidx[j]];
      bool in = false;
      if (sh >= 0)
        for (int x = j; x >= 0; x = PARENT[x])
          if (x == sh) in = true;
      on_side[j] = in;
      any_side = any_side || in;
    }
    // waves: per root (in index order) its main wave, then its side wave if it has one
    {
      int base[N > 0 ? N : 1] = {};
      int w = 0;
      for (int r = 0; r < N; ++r)
        if (PARENT[r] < 0) { base[rootidx[r]] = w; w += 1 + (side_head[rootidx[r]] >= 0 ? 1 : 0); }
      n_waves = w;
      for (int j = 0; j < N; ++j) wave_of[j] = base[rootidx[j]] + (on_side[j] ? 1 : 0);
      for (int j = 0; j < N; ++j)
        if (head[j] == j && wave_of[j] < 16 && len[j] > wave_len[wave_of[j]]) wave_len[wave_of[j]] = len[j];
      for (int k = 0; k < 16; ++k) inch_off[k + 1] = inch_off[k] + wave_len[k] * (wave_len[k] - 1);
    }
  }
};
constexpr TreePlan TP{};
constexpr bool is_chain_head(int i) { return TP.head[i] == i; }
constexpr int chain_head_of(int i) { return TP.head[i]; }
constexpr int chain_leaf(int h) { return TP.leaf[h]; }
constexpr bool in_chain(int j, int h) { return TP.head[j] == h; }
constexpr int pos_in_chain(int j) { return TP.pos[j]; }
constexpr int max_chain_len() { return TP.max_len; }
// (jj, j): jj a proper ancestor of j in ANOTHER chain; rank = position among all such pairs
constexpr int cross_rank(int jj, int j) {
  int k = TP.cross0[j];
  for (int x = PARENT[j]; x >= 0 && x != jj; x = PARENT[x])
    if (TP.head[x] != TP.head[j]) ++k;
  return k;
}
constexpr int n_cross_pairs() { return TP.n_cross; }
constexpr int park_rank(int h) { return TP.park[h]; }
constexpr int n_parked_chains() { return TP.n_park; }
constexpr int tree_n_roots() { return TP.n_roots; }

// ---- LDS plan (scalars of T) -----------------------------------------------------------------------
// [0, 64 * TREE_KP)                      row image, 64 configurations x 2n (+ pad), float2-granular
// then lane-private columns (slot * 64 + lane):
//   TREE_INCH  in-chain pending column entries, reused by every chain      2 * L (L - 1) / 2
//   TREE_CROSS cross-chain pending column entries, live until consumed     2 * n_cross_pairs
//   TREE_PARK  composites of finished chains                               31 * n_parked_chains
// row stride: odd in units of the flush vector (float4 when n is even, else float2) => conflict-free
constexpr int TREE_KP = (N % 2 == 0) ? 4 * ((N / 2) | 1) : 2 * N;
constexpr int TREE_KP2 = TREE_KP / 2;
constexpr int TREE_COMP = 31;
constexpr int TREE_INCH = 0;
// Robots with a side subtree (Atlas' right arm next to the heavy path back -> left arm): ONE block per 64
// configurations, one wave per root subtree plus one per side subtree, each with its own row image and its own
// in-chain pending region (sized by its longest chain).  Other robots: one single-wave block per (64 configurations,
// root), blockIdx.y = root, as before.
constexpr int TREE_MULTI_SCALARS = TP.n_waves * TREE_KP + TP.inch_off[TP.n_waves < 16 ? TP.n_waves : 16] + 2 * TP.n_cross + 31 * TP.n_park;
constexpr bool TREE_MULTI = TP.any_side && TP.n_waves <= 8 && (size_t)64 * TREE_MULTI_SCALARS * (N <= 12 ? 8 : 4) <= 156 * 1024;
constexpr int TREE_W = TREE_MULTI ? TP.n_waves : 1;
constexpr int TREE_INCH_TOTAL = TREE_MULTI ? TP.inch_off[TP.n_waves] : max_chain_len() * (max_chain_len() - 1);
constexpr int TREE_CROSS = TREE_INCH + TREE_INCH_TOTAL;
constexpr int TREE_PARK = TREE_CROSS + 2 * n_cross_pairs();
constexpr int TREE_PRIV = TREE_PARK + TREE_COMP * n_parked_chains();
template <class T>
constexpr size_t tree_lds_bytes() { return sizeof(T) * (size_t)64 * (TREE_W * TREE_KP + TREE_PRIV); }

constexpr int tree_pend_slot(int jj, int j) {   // slot of the (dq, dqd) pair for row jj, column j
  if (chain_head_of(jj) == chain_head_of(j)) {
    const int a = pos_in_chain(jj), b = pos_in_chain(j);
    return TREE_INCH + (TREE_MULTI ? TP.inch_off[TP.wave_of[j]] : 0) + 2 * (b * (b - 1) / 2 + a);
  }
  return TREE_CROSS + 2 * cross_rank(jj, j);
}

constexpr bool grad_tree_ok_() {
  for (int j = 0; j < N; ++j) {
    if (JTYPE[j] != 0) return false;
    if (!rigid_inertia_(j)) return false;
  }
  return N >= 2 && N <= 64;
}
constexpr bool GRAD_TREE_OK = grad_tree_ok_();

// ---- world-frame state of one body and the sweeps' building blocks -------------------------------
template <class T>
struct WState {
  T R[3][3], p[3], v[6], a[6];   // body -> world rotation, origin, spatial velocity / acceleration (world frame)
};

// parent(J) -> J  (:1413-1434); for a root the incoming state is ignored
template <int J, class T>
RBD_DEV void ws_down(WState<T>& s, const JTrig<T>& g, T qd, T qdd, T grav, T (&Sv)[6], T (&Pd)[6], T (&Pdd)[6]) {
  constexpr int k = AXIS[J], ka = (k + 1) % 3, kb = (k + 2) % 3;
  constexpr bool root = PARENT[J] < 0;
  T Tm[3][3];
  sfor<0, 3>([&](auto R_) {
    sfor<0, 3>([&](auto C_) {
      constexpr int r = decltype(R_)::value, c = decltype(C_)::value;
      if constexpr (root) {
        Tm[r][c] = T(Et_(J, c, r));
      } else {
        T acc = T(0);
        sfor<0, 3>([&](auto M_) {
          constexpr int m = decltype(M_)::value;
          constexpr double e = Et_(J, c, m);
          if constexpr (e == 1.0) acc = acc + s.R[r][m];
          else if constexpr (e == -1.0) acc = acc - s.R[r][m];
          else if constexpr (e != 0.0) acc = fma_(T(e), s.R[r][m], acc);
        });
        Tm[r][c] = acc;
      }
    });
  });
  T pn[3];
  sfor<0, 3>([&](auto R_) {
    constexpr int r = decltype(R_)::value;
    if constexpr (root) {
      pn[r] = T(rt_(J, r));
    } else {
      T acc = s.p[r];
      sfor<0, 3>([&](auto M_) {
        constexpr int m = decltype(M_)::value;
        constexpr double e = rt_(J, m);
        if constexpr (e != 0.0) acc = fma_(T(e), s.R[r][m], acc);
      });
      pn[r] = acc;
    }
  });
  sfor<0, 3>([&](auto R_) {
    constexpr int r = decltype(R_)::value;
    s.R[r][ka] = fma_(g.c, Tm[r][ka], g.s * Tm[r][kb]);
    s.R[r][kb] = fma_(g.c, Tm[r][kb], -(g.s * Tm[r][ka]));
    s.R[r][k] = Tm[r][k];
    s.p[r] = pn[r];
  });
  const T ang[3] = {s.R[0][k], s.R[1][k], s.R[2][k]};
  T sl[3];
  cross3(s.p, ang, sl);
  sfor<0, 3>([&](auto R_) { constexpr int r = decltype(R_)::value; Sv[r] = ang[r]; Sv[3 + r] = sl[r]; });
  if constexpr (root) {
    sfor<0, 6>([&](auto R_) { Pd[decltype(R_)::value] = T(0); });
    Pdd[0] = T(0); Pdd[1] = T(0); Pdd[2] = T(0);
    Pdd[3] = grav * ang[1];
    Pdd[4] = -(grav * ang[0]);
    Pdd[5] = T(0);
    sfor<0, 6>([&](auto R_) {
      constexpr int r = decltype(R_)::value;
      s.v[r] = Sv[r] * qd;
      s.a[r] = Sv[r] * qdd;
    });
    s.a[5] -= grav;
  } else {
    T t1[6], t2[6];
    crm6(s.v, Sv, Pd);
    crm6(s.a, Sv, t1);
    crm6(s.v, Pd, t2);
    sfor<0, 6>([&](auto R_) {
      constexpr int r = decltype(R_)::value;
      Pdd[r] = t1[r] + t2[r];
      s.v[r] = fma_(Sv[r], qd, s.v[r]);
      s.a[r] = fma_(Sv[r], qdd, fma_(Pd[r], qd, s.a[r]));
    });
  }
}

// J -> parent(J): exact inverse of ws_down for a non-root body
template <int J, class T>
RBD_DEV void ws_up(WState<T>& s, const JTrig<T>& g, T qd, T qdd, const T (&Sv)[6], const T (&Pd)[6]) {
  constexpr int k = AXIS[J], ka = (k + 1) % 3, kb = (k + 2) % 3;
  sfor<0, 6>([&](auto R_) {
    constexpr int r = decltype(R_)::value;
    s.v[r] = fma_(-Sv[r], qd, s.v[r]);
    s.a[r] = fma_(-Sv[r], qdd, fma_(-Pd[r], qd, s.a[r]));
  });
  T Tm[3][3];
  sfor<0, 3>([&](auto R_) {
    constexpr int r = decltype(R_)::value;
    Tm[r][ka] = fma_(g.c, s.R[r][ka], -(g.s * s.R[r][kb]));
    Tm[r][kb] = fma_(g.s, s.R[r][ka], g.c * s.R[r][kb]);
    Tm[r][k] = s.R[r][k];
  });
  sfor<0, 3>([&](auto R_) {
    sfor<0, 3>([&](auto C_) {
      constexpr int r = decltype(R_)::value, c = decltype(C_)::value;
      T acc = T(0);
      sfor<0, 3>([&](auto M_) {
        constexpr int m = decltype(M_)::value;
        constexpr double e = Et_(J, m, c);
        if constexpr (e == 1.0) acc = acc + Tm[r][m];
        else if constexpr (e == -1.0) acc = acc - Tm[r][m];
        else if constexpr (e != 0.0) acc = fma_(T(e), Tm[r][m], acc);
      });
      s.R[r][c] = acc;
    });
  });
  sfor<0, 3>([&](auto R_) {
    constexpr int r = decltype(R_)::value;
    T acc = s.p[r];
    sfor<0, 3>([&](auto M_) {
      constexpr int m = decltype(M_)::value;
      constexpr double e = rt_(J, m);
      if constexpr (e != 0.0) acc = fma_(T(-e), s.R[r][m], acc);
    });
    s.p[r] = acc;
  });
}

// composite of a subtree in the world frame: rigid inertia, Sym part of BC, momentum, force (:1436-1448)
template <class T>
struct Comp {
  RInertia<T> IC;
  SymB<T> SC;
  T pm[6], f[6];
};
template <class T>
RBD_DEV void comp_add(Comp<T>& a, const Comp<T>& b) {
  a.IC.m += b.IC.m;
  sfor<0, 3>([&](auto R_) { constexpr int r = decltype(R_)::value; a.IC.h[r] += b.IC.h[r]; a.SC.G[r] += b.SC.G[r]; });
  sfor<0, 6>([&](auto R_) {
    constexpr int r = decltype(R_)::value;
    a.IC.I[r] += b.IC.I[r]; a.SC.TL[r] += b.SC.TL[r]; a.pm[r] += b.pm[r]; a.f[r] += b.f[r];
  });
}
// body J's own terms from its world state
template <int J, class T>
RBD_DEV void comp_local(const WState<T>& s, Comp<T>& L) {
  T cw[3];
  sfor<0, 3>([&](auto R_) {
    constexpr int r = decltype(R_)::value;
    T acc = s.p[r];
    sfor<0, 3>([&](auto M_) {
      constexpr int m = decltype(M_)::value;
      constexpr double e = com_(J, m);
      if constexpr (e != 0.0) acc = fma_(T(e), s.R[r][m], acc);
    });
    cw[r] = acc;
  });
  L.IC.m = T(mass_(J));
  sfor<0, 3>([&](auto R_) { L.IC.h[decltype(R_)::value] = T(mass_(J)) * cw[decltype(R_)::value]; });
  {
    T A[3][3];
    sfor<0, 3>([&](auto R_) {
      sfor<0, 3>([&](auto C_) {
        constexpr int r = decltype(R_)::value, c = decltype(C_)::value;
        T acc = T(0);
        sfor<0, 3>([&](auto M_) {
          constexpr int m = decltype(M_)::value;
          constexpr double e = Ic_(J, m, c);
          if constexpr (e != 0.0) acc = fma_(T(e), s.R[r][m], acc);
        });
        A[r][c] = acc;
      });
    });
    const T cc = fma_(cw[0], cw[0], fma_(cw[1], cw[1], cw[2] * cw[2]));
    constexpr int IR[6] = {0, 0, 0, 1, 1, 2}, IC_[6] = {0, 1, 2, 1, 2, 2};
    sfor<0, 6>([&](auto E_) {
      constexpr int e = decltype(E_)::value, r = IR[e], c = IC_[e];
      T x = fma_(A[r][0], s.R[c][0], fma_(A[r][1], s.R[c][1], A[r][2] * s.R[c][2]));
      const T mcc = L.IC.h[r] * cw[c];
      if constexpr (r == c) x += fma_(L.IC.m, cc, -mcc); else x -= mcc;
      L.IC.I[e] = x;
    });
  }
  T Ia[6];
  rin_apply(L.IC, s.v, L.pm);
  rin_apply(L.IC, s.a, Ia);
  fxv<false>(s.v, L.pm, L.f);
  sfor<0, 6>([&](auto R_) { L.f[decltype(R_)::value] += Ia[decltype(R_)::value]; });
  {
    const T w[3] = {s.v[0], s.v[1], s.v[2]}, u[3] = {s.v[3], s.v[4], s.v[5]};
    const T Ifull[3][3] = {{L.IC.I[0], L.IC.I[1], L.IC.I[2]}, {L.IC.I[1], L.IC.I[3], L.IC.I[4]}, {L.IC.I[2], L.IC.I[4], L.IC.I[5]}};
    T K[3][3];
    sfor<0, 3>([&](auto C_) {
      constexpr int c = decltype(C_)::value;
      const T col[3] = {Ifull[0][c], Ifull[1][c], Ifull[2][c]};
      T o[3];
      cross3(w, col, o);
      K[0][c] = o[0]; K[1][c] = o[1]; K[2][c] = o[2];
    });
    const T uh2 = T(2) * fma_(u[0], L.IC.h[0], fma_(u[1], L.IC.h[1], u[2] * L.IC.h[2]));
    constexpr int IR[6] = {0, 0, 0, 1, 1, 2}, IC_[6] = {0, 1, 2, 1, 2, 2};
    sfor<0, 6>([&](auto E_) {
      constexpr int e = decltype(E_)::value, r = IR[e], c = IC_[e];
      T x = K[r][c] + K[c][r];
      x = fma_(-L.IC.h[r], u[c], fma_(-u[r], L.IC.h[c], x));
      if constexpr (r == c) x += uh2;
      L.SC.TL[e] = x;
    });
    T g[3];
    cross3(w, L.IC.h, g);
    sfor<0, 3>([&](auto R_) { constexpr int r = decltype(R_)::value; L.SC.G[r] = fma_(L.IC.m, u[r], g[r]); });
  }
}
// flat view of a composite (31 scalars) for parking
template <class T, class F>
RBD_DEV void comp_each(Comp<T>& c, F&& f) {
  f(std::integral_constant<int, 0>{}, c.IC.m);
  sfor<0, 3>([&](auto R_) { constexpr int r = decltype(R_)::value; f(std::integral_constant<int, 1 + r>{}, c.IC.h[r]); });
  sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; f(std::integral_constant<int, 4 + r>{}, c.IC.I[r]); });
  sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; f(std::integral_constant<int, 10 + r>{}, c.SC.TL[r]); });
  sfor<0, 3>([&](auto R_) { constexpr int r = decltype(R_)::value; f(std::integral_constant<int, 16 + r>{}, c.SC.G[r]); });
  sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; f(std::integral_constant<int, 19 + r>{}, c.pm[r]); });
  sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; f(std::integral_constant<int, 25 + r>{}, c.f[r]); });
}

template <class T, bool HAS_QDD>
__global__ __launch_bounds__(64 * TREE_W, 1) void rnea_grad_tree_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                               const T* __restrict__ qdd, T grav, int use_damping,
                                                               long long B, T* __restrict__ c_out, T* __restrict__ dcdu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  T* rowimg = reinterpret_cast<T*>(smem_raw) + wave * (64 * TREE_KP);   // [64][TREE_KP], one image per wave
  T* priv = reinterpret_cast<T*>(smem_raw) + TREE_W * 64 * TREE_KP + lane;   // priv[slot * 64], shared by the waves (same lane = same configuration)
  const long long cfg0 = (long long)blockIdx.x * 64;
  const long long rem = B - cfg0;
  const int nvalid = rem < 64 ? (int)rem : 64;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);
  const T* qrow = q + b * N;
  const T* qdrow = qd + b * N;
  const T* qddrow = HAS_QDD ? qdd + b * N : nullptr;

  // flush geometry: a row is N float2 (or N/2 float4 when n is even: 16-byte aligned segments);
  // lanes (sub, e) cover CPI configurations x FW vectors per step
  typedef T V2 __attribute__((ext_vector_type(2)));
  typedef T V4 __attribute__((ext_vector_type(4)));
  constexpr bool WIDE = (N % 2 == 0) && (TREE_KP % 4 == 0);
  constexpr int FW = WIDE ? N / 2 : N;                    // vectors per row
  constexpr int CPI = 64 / FW > 0 ? 64 / FW : 1;          // configurations per flush step
  const int fsub = lane / FW, fe = lane - fsub * FW;
  const bool factive = lane < CPI * FW;
  const int myroot = blockIdx.y;                          // (single-wave layout) independent root subtrees run in separate blocks

  sfor_down<0, N>([&](auto H_) {
    constexpr int h = decltype(H_)::value;
    if constexpr (is_chain_head(h)) {
     constexpr int hroot = TP.rootidx[h];
     constexpr int hwave = TREE_MULTI ? TP.wave_of[h] : 0;  // multi-wave layout: the wave that runs this chain
     constexpr int side = TREE_MULTI ? TP.side_head[hroot] : -1;
     if (TREE_MULTI ? (wave == hwave) : (hroot == myroot)) {
      constexpr int leaf = chain_leaf(h);
      // ---- inputs and trig of the root path of this chain ------------------------------------------
      JTrig<T> tr[N];
      T qdv[N], qddv[N];
      sfor<0, N>([&](auto J_) {
        constexpr int j = decltype(J_)::value;
        if constexpr (is_anc_or_self(j, leaf)) {
          tr[j] = make_trig<j>(qrow[j]);
          qdv[j] = qdrow[j];
          if constexpr (HAS_QDD) qddv[j] = qddrow[j]; else qddv[j] = T(0);
        }
      });
      // ---- world kinematics root -> leaf (:1413-1434) -------------------------------------------------
      WState<T> s;
      T Sv[N][6], Pd[N][6], Pdd[N][6];
      sfor<0, N>([&](auto J_) {
        constexpr int j = decltype(J_)::value;
        if constexpr (is_anc_or_self(j, leaf)) ws_down<j>(s, tr[j], qdv[j], qddv[j], grav, Sv[j], Pd[j], Pdd[j]);
      });
      // ---- leaf -> head ---------------------------------------------------------------------------------
      Comp<T> C;
      sfor_down<0, N>([&](auto J_) {
        constexpr int j = decltype(J_)::value;
        if constexpr (in_chain(j, h)) {
          if constexpr (j == leaf) {
            comp_local<j>(s, C);
          } else {
            Comp<T> L;
            comp_local<j>(s, L);
            comp_add(C, L);
          }
          // the side subtree (other wave) must have parked its composite and column entries before its parent is built
          if constexpr (side >= 0 && !TP.on_side[h]) {
            if constexpr (j == PARENT[side >= 0 ? side : 0]) __syncthreads();
          }
          // finished chains hanging off this body (:1446-1448)
          sfor<0, N>([&](auto K_) {
            constexpr int kk = decltype(K_)::value;
            if constexpr (PARENT[kk] == j && is_chain_head(kk) && kk != j) {
              Comp<T> P;
              comp_each(P, [&](auto I_, T& x) { x = priv[(TREE_PARK + TREE_COMP * park_rank(kk) + decltype(I_)::value) * 64]; });
              comp_add(C, P);
            }
          });
          const T cj = dot6(Sv[j], C.f);
          if (c_out != nullptr && lane < nvalid) c_out[b * N + j] = cj;
          // t-vectors (:1481-1484)
          T t1[6], t2[6], t3[6], t4[6];
          {
            T y3[6], s1[6], z1[6], zf[6];
            rin_apply(C.IC, Sv[j], t1);
            rin_apply(C.IC, Pdd[j], y3);
            sym_apply(C.SC, Sv[j], s1);
            fxv<false>(Sv[j], C.pm, z1);
            fxv<false>(Sv[j], C.f, zf);
            if constexpr (PARENT[j] < 0) {   // psid of a root is identically zero
              sfor<0, 6>([&](auto R_) {
                constexpr int r = decltype(R_)::value;
                t4[r] = s1[r] - z1[r];
                t3[r] = y3[r] + zf[r];
                t2[r] = s1[r] + z1[r];
              });
            } else {
              T y2[6], s2[6], z2[6];
              rin_apply(C.IC, Pd[j], y2);
              sym_apply(C.SC, Pd[j], s2);
              fxv<false>(Pd[j], C.pm, z2);
              sfor<0, 6>([&](auto R_) {
                constexpr int r = decltype(R_)::value;
                t4[r] = s1[r] - z1[r];
                t3[r] = (s2[r] + z2[r]) + (y3[r] + zf[r]);
                t2[r] = fma_(T(2), y2[r], s1[r] + z1[r]);
              });
            }
          }
          // ---- row j ------------------------------------------------------------------------------------
          T row[2 * N];
          sfor<0, N>([&](auto C_) {
            constexpr int c = decltype(C_)::value;
            if constexpr (c == j || (is_anc_or_self(c, j))) {
              T dq, dqd;
              if constexpr (PARENT[c] < 0) {   // root column: psid = 0, psidd = (0, 0, 0, g S_y, -g S_x, 0)
                dq = fma_(t1[3], Pdd[c][3], t1[4] * Pdd[c][4]);
                dqd = dot6(t4, Sv[c]);
              } else {
                dq = dot6(t4, Pd[c]) + dot6(t1, Pdd[c]);
                dqd = fma_(T(2), dot6(t1, Pd[c]), dot6(t4, Sv[c]));
              }
              if constexpr (c == j) dqd += sel(use_damping != 0, T(DAMPING[j]), T(0));   // (:1336-1341)
              row[c] = dq;
              row[N + c] = dqd;
              if constexpr (c != j) {   // column entries of the ancestor's row, parked until that row is built
                priv[(tree_pend_slot(c, j)) * 64] = dot6(Sv[c], t3);
                priv[(tree_pend_slot(c, j) + 1) * 64] = dot6(Sv[c], t2);
              }
            } else if constexpr (is_anc_or_self(j, c)) {   // descendant: delivered earlier
              row[c] = priv[(tree_pend_slot(j, c)) * 64];
              row[N + c] = priv[(tree_pend_slot(j, c) + 1) * 64];
            } else {
              row[c] = T(0);
              row[N + c] = T(0);
            }
          });
          // image of row j for the 64 configurations, then 8n-byte segments to dc_du[b][j][:].  The
          // block is ONE wave and a wave's LDS operations execute in order, so a wave-level fence
          // (compiler ordering) is all the image needs -- no s_barrier, no wait on the previous
          // row's global stores.
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          {
            V2* mine = reinterpret_cast<V2*>(rowimg) + lane * TREE_KP2;
            sfor<0, N>([&](auto E_) {
              constexpr int e = decltype(E_)::value;
              V2 x; x[0] = row[2 * e]; x[1] = row[2 * e + 1];
              mine[e] = x;
            });
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (factive) {
            if constexpr (WIDE) {
              const V4* src = reinterpret_cast<const V4*>(rowimg);
              V4* dst = reinterpret_cast<V4*>(dcdu + (cfg0 * N + j) * (2 * N));
#pragma unroll 4
              for (int c0 = 0; c0 < 64; c0 += CPI) {
                const int cfg = c0 + fsub;
                if (cfg < nvalid) dst[(long long)cfg * (N * N / 2) + fe] = src[cfg * (TREE_KP / 4) + fe];
              }
            } else {
              const V2* src = reinterpret_cast<const V2*>(rowimg);
              V2* dst = reinterpret_cast<V2*>(dcdu + (cfg0 * N + j) * (2 * N));
#pragma unroll 4
              for (int c0 = 0; c0 < 64; c0 += CPI) {
                const int cfg = c0 + fsub;
                if (cfg < nvalid) dst[(long long)cfg * (N * N) + fe] = src[cfg * TREE_KP2 + fe];
              }
            }
          }
          // step back to the parent inside the chain, or park the finished chain's composite
          if constexpr (j != h) {
            ws_up<j>(s, tr[j], qdv[j], qddv[j], Sv[j], Pd[j]);
          } else if constexpr (PARENT[h] >= 0) {
            comp_each(C, [&](auto I_, T& x) { priv[(TREE_PARK + TREE_COMP * park_rank(h) + decltype(I_)::value) * 64] = x; });
          }
        }
      });
     }
    }
  });
  if constexpr (TREE_MULTI) {
    // every wave passes exactly ONE block barrier: a root's main wave before it builds the side subtree's parent
    // (above), every other wave here, when its chains are done
    bool main_with_side = false;
    sfor<0, N>([&](auto R_) {
      constexpr int r = decltype(R_)::value;
      if constexpr (PARENT[r] < 0) {
        constexpr int w0 = TP.wave_of[r];
        constexpr bool hs = TP.side_head[TP.rootidx[r]] >= 0;
        if (wave == w0 && hs) main_with_side = true;
      }
    });
    if (!main_with_side) __syncthreads();
  }
}

}  // namespace rbdk
