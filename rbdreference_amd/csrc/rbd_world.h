// rbd_world.h -- building blocks of the WORLD-FRAME gradient kernels (rbd_idsva.h: one chain; rbd_idsva_tree.h:
// fixed-base trees; rbd_fb_world.h: trees under a floating base), shared so that the three kernels run the same
// arithmetic: small spatial-vector helpers, the rigid-inertia / Sym operators, the compile-time chain
// decomposition of the tree, the world-frame state of a body with its root -> leaf and leaf -> root steps, and the
// subtree composites.  Identities: the first-order part of the reference's IDSVA scheme,
// /root/reference/RBDReference.py:1413-1484; what they reproduce: RBDReference.rnea_grad, :1345-1368.
#pragma once
#include "rbd_spatial.h"

namespace rbdk {

// ---- small world-frame helpers -------------------------------------------------------------------
template <class T>
RBD_DEV void cross3(const T (&a)[3], const T (&b)[3], T (&o)[3]) {
  o[0] = fma_(a[1], b[2], -(a[2] * b[1]));
  o[1] = fma_(a[2], b[0], -(a[0] * b[2]));
  o[2] = fma_(a[0], b[1], -(a[1] * b[0]));
}
// o += a x b
template <class T>
RBD_DEV void cross3_acc(const T (&a)[3], const T (&b)[3], T (&o)[3]) {
  o[0] = fma_(a[1], b[2], fma_(-a[2], b[1], o[0]));
  o[1] = fma_(a[2], b[0], fma_(-a[0], b[2], o[1]));
  o[2] = fma_(a[0], b[1], fma_(-a[1], b[0], o[2]));
}
// motion cross: o = crm(v) x
template <class T>
RBD_DEV void crm6(const T (&v)[6], const T (&x)[6], T (&o)[6]) {
  const T w[3] = {v[0], v[1], v[2]}, u[3] = {v[3], v[4], v[5]};
  const T xa[3] = {x[0], x[1], x[2]}, xb[3] = {x[3], x[4], x[5]};
  T oa[3], ob[3];
  cross3(w, xa, oa);
  cross3(u, xa, ob);
  cross3_acc(w, xb, ob);
  o[0] = oa[0]; o[1] = oa[1]; o[2] = oa[2]; o[3] = ob[0]; o[4] = ob[1]; o[5] = ob[2];
}
// rigid-body inertia (m, h, Ibar sym: xx xy xz yy yz zz) times a motion vector
template <class T>
struct RInertia {
  T m, h[3], I[6];
};
template <class T>
RBD_DEV void rin_apply(const RInertia<T>& R, const T (&x)[6], T (&y)[6]) {
  const T w[3] = {x[0], x[1], x[2]}, u[3] = {x[3], x[4], x[5]};
  T top[3] = {fma_(R.I[0], w[0], fma_(R.I[1], w[1], R.I[2] * w[2])),
              fma_(R.I[1], w[0], fma_(R.I[3], w[1], R.I[4] * w[2])),
              fma_(R.I[2], w[0], fma_(R.I[4], w[1], R.I[5] * w[2]))};
  cross3_acc(R.h, u, top);
  T hw[3];
  cross3(R.h, w, hw);
  y[0] = top[0]; y[1] = top[1]; y[2] = top[2];
  y[3] = fma_(R.m, u[0], -hw[0]); y[4] = fma_(R.m, u[1], -hw[1]); y[5] = fma_(R.m, u[2], -hw[2]);
}
// Sym = [[TL, G^x], [G^x^T, 0]] times a motion vector: [TL a + G x b ; -G x a]
template <class T>
struct SymB {
  T TL[6], G[3];
};
template <class T>
RBD_DEV void sym_apply(const SymB<T>& S, const T (&x)[6], T (&y)[6]) {
  const T a[3] = {x[0], x[1], x[2]}, b[3] = {x[3], x[4], x[5]};
  T top[3] = {fma_(S.TL[0], a[0], fma_(S.TL[1], a[1], S.TL[2] * a[2])),
              fma_(S.TL[1], a[0], fma_(S.TL[3], a[1], S.TL[4] * a[2])),
              fma_(S.TL[2], a[0], fma_(S.TL[4], a[1], S.TL[5] * a[2]))};
  cross3_acc(S.G, b, top);
  T ga[3];
  cross3(S.G, a, ga);
  y[0] = top[0]; y[1] = top[1]; y[2] = top[2];
  y[3] = -ga[0]; y[4] = -ga[1]; y[5] = -ga[2];
}

// ---- accumulate forms (rbd_idsva_pipe.h) -----------------------------------------------------------------
// "X = A + (a product chain)" costs a multiply to start the chain and an add to join it; started FROM A the chain is
// FMAs only: one instruction less per component.  The one-lane chain kernel is bound by VALU issue (DESIGN.md
// §3.1 a'), so the ~55 joins per body of the backward sweep are worth folding.  Same values up to rounding order.
template <class T>
RBD_DEV T dot6_acc(const T (&x)[6], const T (&y)[6], T acc) {
  return fma_(x[5], y[5], fma_(x[4], y[4], fma_(x[3], y[3], fma_(x[2], y[2], fma_(x[1], y[1], fma_(x[0], y[0], acc))))));
}
// r = acc + crf(v) b   (fxv, rbd_spatial.h)
template <class T>
RBD_DEV void fxv_add(const T (&v)[6], const T (&b)[6], const T (&acc)[6], T (&r)[6]) {
  const T r0 = fma_(v[1], b[2], fma_(-v[2], b[1], fma_(v[4], b[5], fma_(-v[5], b[4], acc[0]))));
  const T r1 = fma_(v[2], b[0], fma_(-v[0], b[2], fma_(v[5], b[3], fma_(-v[3], b[5], acc[1]))));
  const T r2 = fma_(v[0], b[1], fma_(-v[1], b[0], fma_(v[3], b[4], fma_(-v[4], b[3], acc[2]))));
  const T r3 = fma_(v[1], b[5], fma_(-v[2], b[4], acc[3]));
  const T r4 = fma_(v[2], b[3], fma_(-v[0], b[5], acc[4]));
  const T r5 = fma_(v[0], b[4], fma_(-v[1], b[3], acc[5]));
  r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3; r[4] = r4; r[5] = r5;
}
// r = acc - crf(v) b
template <class T>
RBD_DEV void fxv_sub(const T (&v)[6], const T (&b)[6], const T (&acc)[6], T (&r)[6]) {
  const T r0 = fma_(-v[1], b[2], fma_(v[2], b[1], fma_(-v[4], b[5], fma_(v[5], b[4], acc[0]))));
  const T r1 = fma_(-v[2], b[0], fma_(v[0], b[2], fma_(-v[5], b[3], fma_(v[3], b[5], acc[1]))));
  const T r2 = fma_(-v[0], b[1], fma_(v[1], b[0], fma_(-v[3], b[4], fma_(v[4], b[3], acc[2]))));
  const T r3 = fma_(-v[1], b[5], fma_(v[2], b[4], acc[3]));
  const T r4 = fma_(-v[2], b[3], fma_(v[0], b[5], acc[4]));
  const T r5 = fma_(-v[0], b[4], fma_(v[1], b[3], acc[5]));
  r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3; r[4] = r4; r[5] = r5;
}
// o = acc + crm(v) x
template <class T>
RBD_DEV void crm6_add(const T (&v)[6], const T (&x)[6], const T (&acc)[6], T (&o)[6]) {
  const T w[3] = {v[0], v[1], v[2]}, u[3] = {v[3], v[4], v[5]};
  const T xa[3] = {x[0], x[1], x[2]}, xb[3] = {x[3], x[4], x[5]};
  T oa[3] = {acc[0], acc[1], acc[2]}, ob[3] = {acc[3], acc[4], acc[5]};
  cross3_acc(w, xa, oa);
  cross3_acc(u, xa, ob);
  cross3_acc(w, xb, ob);
  o[0] = oa[0]; o[1] = oa[1]; o[2] = oa[2]; o[3] = ob[0]; o[4] = ob[1]; o[5] = ob[2];
}
// y = acc + R x   (rin_apply)
template <class T>
RBD_DEV void rin_apply_acc(const RInertia<T>& R, const T (&x)[6], const T (&acc)[6], T (&y)[6]) {
  const T w[3] = {x[0], x[1], x[2]}, u[3] = {x[3], x[4], x[5]};
  T top[3] = {fma_(R.I[0], w[0], fma_(R.I[1], w[1], fma_(R.I[2], w[2], acc[0]))),
              fma_(R.I[1], w[0], fma_(R.I[3], w[1], fma_(R.I[4], w[2], acc[1]))),
              fma_(R.I[2], w[0], fma_(R.I[4], w[1], fma_(R.I[5], w[2], acc[2])))};
  cross3_acc(R.h, u, top);
  y[0] = top[0]; y[1] = top[1]; y[2] = top[2];
  // bottom: m u - h x w  =  m u + w x h
  T bot[3] = {fma_(R.m, u[0], acc[3]), fma_(R.m, u[1], acc[4]), fma_(R.m, u[2], acc[5])};
  cross3_acc(w, R.h, bot);
  y[3] = bot[0]; y[4] = bot[1]; y[5] = bot[2];
}
// y = acc + Sym x   (sym_apply): [TL a + G x b ; -G x a] = [.. ; a x G]
template <class T>
RBD_DEV void sym_apply_acc(const SymB<T>& S, const T (&x)[6], const T (&acc)[6], T (&y)[6]) {
  const T a[3] = {x[0], x[1], x[2]}, b[3] = {x[3], x[4], x[5]};
  T top[3] = {fma_(S.TL[0], a[0], fma_(S.TL[1], a[1], fma_(S.TL[2], a[2], acc[0]))),
              fma_(S.TL[1], a[0], fma_(S.TL[3], a[1], fma_(S.TL[4], a[2], acc[1]))),
              fma_(S.TL[2], a[0], fma_(S.TL[4], a[1], fma_(S.TL[5], a[2], acc[2])))};
  cross3_acc(S.G, b, top);
  T bot[3] = {acc[3], acc[4], acc[5]};
  cross3_acc(a, S.G, bot);
  y[0] = top[0]; y[1] = top[1]; y[2] = top[2]; y[3] = bot[0]; y[4] = bot[1]; y[5] = bot[2];
}

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
      // Load balance: a root's main wave is the block's critical path (Atlas: neck + left arm + back against one leg on
      // each leg wave).  A chain-head subtree that hangs off the root's heavy chain AT OR ABOVE the side subtree's parent
      // is consumed by the main wave only after its block barrier, so any other wave may run it ahead of ITS barrier
      // (the end of its chains): it goes to the least loaded wave if that shortens the main wave.
      // cost of a chain: its downward kinematic sweep + twice its upward steps
      if (w <= 16) {
        int load[16] = {};
        for (int hd = 0; hd < N; ++hd)
          if (head[hd] == hd) load[wave_of[hd]] += DEPTH[leaf[hd]] + 1 + 2 * len[hd];
        for (int x = 0; x < N; ++x) {
          const int p = PARENT[x];
          if (head[x] != x || p < 0) continue;
          const int sh = side_head[rootidx[x]];
          if (sh < 0 || on_side[x]) continue;
          int rt = x;
          while (PARENT[rt] >= 0) rt = PARENT[rt];
          if (head[p] != rt) continue;
          bool above = false;
          for (int y = PARENT[sh]; y >= 0; y = PARENT[y]) above = above || y == p;
          if (!above) continue;
          int cost = 0;
          for (int hd = 0; hd < N; ++hd) {
            bool in = false;
            for (int y = hd; y >= 0; y = PARENT[y]) in = in || y == x;
            if (in && head[hd] == hd) cost += DEPTH[leaf[hd]] + 1 + 2 * len[hd];
          }
          const int wm = wave_of[rt];
          int best = -1;
          for (int k = 0; k < w; ++k)
            if (k != wm && (best < 0 || load[k] < load[best])) best = k;
          if (best < 0 || load[best] + cost >= load[wm]) continue;
          for (int j = 0; j < N; ++j) {
            bool in = false;
            for (int y = j; y >= 0; y = PARENT[y]) in = in || y == x;
            if (in) wave_of[j] = best;
          }
          load[best] += cost;
          load[wm] -= cost;
        }
      }
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
    T t1[6];
    crm6(s.v, Sv, Pd);
    crm6(s.a, Sv, t1);
    crm6_add(s.v, Pd, t1, Pdd);                  // (accumulate forms: the join is part of the FMA chain)
    sfor<0, 6>([&](auto R_) {
      constexpr int r = decltype(R_)::value;
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
constexpr int TREE_COMP_SCALARS = 31;   // 10 + 9 + 6 + 6
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
  fxv_add(s.v, L.pm, Ia, L.f);
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
// t-vectors of body J from its composite and its S / psid / psidd (:1481-1484), joins folded into the FMA chains:
//   t1 = IC S,  t4 = SC S - S x* pm,  t3 = IC psidd + S x* f + SC psid + psid x* pm,  t2 = 2 IC psid + SC S + S x* pm
template <bool ROOT, class T>
RBD_DEV void tvectors(const Comp<T>& C, const T (&Sj)[6], const T (&Pdj)[6], const T (&Pddj)[6], T (&t1)[6], T (&t2)[6], T (&t3)[6], T (&t4)[6]) {
  T y3[6], s1[6], yf[6];
  rin_apply(C.IC, Sj, t1);
  sym_apply(C.SC, Sj, s1);
  fxv_sub(Sj, C.pm, s1, t4);
  rin_apply(C.IC, Pddj, y3);
  fxv_add(Sj, C.f, y3, yf);
  if constexpr (ROOT) {   // psid of a root is identically zero
    sfor<0, 6>([&](auto R_) {
      constexpr int r = decltype(R_)::value;
      t3[r] = yf[r];
      t2[r] = fma_(T(2), s1[r], -t4[r]);
    });
  } else {
    T y2[6], ys[6];
    rin_apply(C.IC, Pdj, y2);
    sym_apply_acc(C.SC, Pdj, yf, ys);
    fxv_add(Pdj, C.pm, ys, t3);
    sfor<0, 6>([&](auto R_) {
      constexpr int r = decltype(R_)::value;
      t2[r] = fma_(T(2), y2[r] + s1[r], -t4[r]);
    });
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

// Row image -> HBM: configuration cfg's segment of the image (vector fe of it: src[cfg * SST + fe]) goes to
// dst[cfg * dstride + fe]; lanes (fsub, fe) cover CPI configurations per step.  A FULL tile issues the LDS reads of a
// whole batch of steps before their stores: with the tail predicate `cfg < nvalid` inside the loop every step is an
// exec-masked region of its own, i.e. read -> wait -> store, 16 times per row (measured on the 30-body robot's tree
// kernel: 3 300 cycles per row on the wave that is the block's critical path, 15 of 36 us at B = 16 384).
// The same for FULL tiles of 16-byte vectors through a raw-buffer descriptor of the block's output region (base = the
// block's first configuration, made by the caller once per kernel): a store is  s_mov soffset; buffer_store_dwordx4
// v, v_lane_offset, s[desc], soffset  instead of a 64-bit per-lane address built with four instructions per store.
// voff = the lane's byte offset (fsub * dstride + fe) * 16, soff = the row's byte offset in a configuration's matrix.
template <int CPI, int SST, class V>
RBD_DEV void flush_image_rows_buf(const V* src, __amdgpu_buffer_rsrc_t rs, int voff, int soff, int dstride_bytes, int fsub, int fe) {
  static_assert(sizeof(V) == 16, "16-byte vectors");
  typedef unsigned U4 __attribute__((ext_vector_type(4)));
  constexpr int IT = (64 + CPI - 1) / CPI;
  constexpr int BATCH = 16 < IT ? 16 : IT;
  sfor<0, (IT + BATCH - 1) / BATCH>([&](auto G_) {
    constexpr int g = decltype(G_)::value;
    V buf[BATCH];
    sfor<0, BATCH>([&](auto I_) {
      constexpr int i = decltype(I_)::value, it = g * BATCH + i;
      if constexpr (it < IT) {
        const int cfg = it * CPI + fsub;
        if constexpr ((it + 1) * CPI <= 64) buf[i] = src[cfg * SST + fe];
        else buf[i] = src[(cfg < 64 ? cfg : 63) * SST + fe];
      }
    });
    sfor<0, BATCH>([&](auto I_) {
      constexpr int i = decltype(I_)::value, it = g * BATCH + i;
      if constexpr (it < IT) {
        if constexpr ((it + 1) * CPI <= 64) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(U4, buf[i]), rs, voff, soff + it * CPI * dstride_bytes, 0);
        } else {
          if (it * CPI + fsub < 64) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(U4, buf[i]), rs, voff, soff + it * CPI * dstride_bytes, 0);
        }
      }
    });
  });
}

// BATCHED = false keeps the plain loop (the fp64 workspace kernel, at 460 of 512 registers, lost 3 % with the staging).
template <int CPI, int SST, bool BATCHED = true, class V>
RBD_DEV void flush_image_rows(const V* src, V* dst, long long dstride, int fsub, int fe, int nvalid) {
  constexpr int IT = (64 + CPI - 1) / CPI;
  constexpr int BATCH = (int)(256 / sizeof(V)) < IT ? (int)(256 / sizeof(V)) : IT;   // 64 registers of staging
  if (BATCHED && nvalid == 64) {
    sfor<0, (IT + BATCH - 1) / BATCH>([&](auto G_) {
      constexpr int g = decltype(G_)::value;
      V buf[BATCH];
      sfor<0, BATCH>([&](auto I_) {
        constexpr int i = decltype(I_)::value, it = g * BATCH + i;
        if constexpr (it < IT) {
          const int cfg = it * CPI + fsub;
          if constexpr ((it + 1) * CPI <= 64) buf[i] = src[cfg * SST + fe];
          else buf[i] = src[(cfg < 64 ? cfg : 63) * SST + fe];
        }
      });
      sfor<0, BATCH>([&](auto I_) {
        constexpr int i = decltype(I_)::value, it = g * BATCH + i;
        if constexpr (it < IT) {
          const int cfg = it * CPI + fsub;
          if constexpr ((it + 1) * CPI <= 64) dst[cfg * dstride + fe] = buf[i];
          else if (cfg < 64) dst[cfg * dstride + fe] = buf[i];
        }
      });
    });
  } else {
#pragma unroll 4
    for (int c0 = 0; c0 < 64; c0 += CPI) {
      const int cfg = c0 + fsub;
      if (cfg < nvalid) dst[cfg * dstride + fe] = src[cfg * SST + fe];
    }
  }
}

}  // namespace rbdk
