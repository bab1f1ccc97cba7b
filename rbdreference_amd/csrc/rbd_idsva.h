// rbd_idsva.h -- rnea_grad as ONE lane per configuration via world-frame spatial-vector identities.
//
// The reference computes dc_du with O(n * depth) 6-vector recursions per derivative column
// (/root/reference/RBDReference.py:1127-1343).  For all-revolute robots with rigid-body inertias the
// same matrix follows from per-body quantities expressed in the WORLD frame (the first-order part
// of the IDSVA scheme the reference itself uses in second_order_idsva_parallel, :1413-1484):
//
//   forward   S_i, psid_i = v_p x S_i, psidd_i = a_p x S_i + v_p x psid_i, v_i, a_i       (:1427-1434)
//   backward  composites over the subtree (plain sums in the world frame):
//               IC (rigid inertia: m, h = m c, Ibar),  BC = crf(v) I + icrf(I v) - I crm(v)  (:1439),
//               f (:1440);   BC = Sym + icrf(pm) with Sym = [[TL, G^x], [G^x^T, 0]] (9 numbers)
//             t1 = IC S, t4 = BC^T S, t3 = BC psid + IC psidd + S x* f, t2 = BC S + 2 IC psid  (:1481-1484)
//             for every ancestor-or-self j of i:
//               dc_dq [i,j] = t4.psid_j + t1.psidd_j      dc_dq [j,i] = S_j.t3   (j != i)
//               dc_dqd[i,j] = t4.S_j   + 2 t1.psid_j      dc_dqd[j,i] = S_j.t2   (j != i)
//
// Every entry of dc_du is produced exactly once (no accumulation), about half the arithmetic of
// the column recursions, and one lane per configuration instead of two.  The numpy prototype of
// exactly this arithmetic matches the oracle to 2e-15 (fp64) / 7e-7 (fp32) on all test robots.
//
// Eligibility (compile time, GRAD_IDSVA_OK): all joints revolute (for prismatic joints the
// reference's dc_dq is its literal fxS form, which only the column recursion reproduces), rigid-body
// structured inertias, every group (root subtree) a chain of at most 8 bodies (the S / psid / psidd
// vectors of a whole chain stay in registers), and chains long enough for the scheme to pay off.
// Other robots keep rnea_grad_kernel.
#pragma once
#include "rbd_spatial.h"

namespace rbdk {

// ---- compile-time body constants ---------------------------------------------------------------
constexpr double cabs_(double x) { return x < 0 ? -x : x; }
constexpr double Et_(int j, int r, int c) { return XT[j][r * 6 + c]; }              // E_tree
constexpr double rx_(int j, int r, int c) {                                         // (r_tree)^x = -E^T B
  double s = 0;
  for (int m = 0; m < 3; ++m) s -= XT[j][m * 6 + r] * XT[j][(3 + m) * 6 + c];
  return s;
}
constexpr double rt_(int j, int k) { return k == 0 ? rx_(j, 2, 1) : k == 1 ? rx_(j, 0, 2) : rx_(j, 1, 0); }
constexpr double mass_(int j) { return IM[j][3 * 6 + 3]; }
constexpr double hb_(int j, int k) {   // h = m c from the top-right block H = h^x
  return k == 0 ? IM[j][2 * 6 + 3 + 1] : k == 1 ? IM[j][0 * 6 + 3 + 2] : IM[j][1 * 6 + 3 + 0];
}
constexpr double com_(int j, int k) { return hb_(j, k) / mass_(j); }
constexpr double Ic_(int j, int r, int c) {   // inertia about the centre of mass: Ibar + m (c c^T - |c|^2 1)
  const double cc = com_(j, 0) * com_(j, 0) + com_(j, 1) * com_(j, 1) + com_(j, 2) * com_(j, 2);
  return IM[j][r * 6 + c] + mass_(j) * (com_(j, r) * com_(j, c) - (r == c ? cc : 0.0));
}
constexpr bool rigid_inertia_(int j) {
  const double m = mass_(j);
  if (!(m > 0)) return false;
  const double tol = 1e-12 * (m > 1 ? m : 1);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      if (cabs_(IM[j][(3 + r) * 6 + 3 + c] - (r == c ? m : 0.0)) > tol) return false;          // m 1
      if (cabs_(IM[j][r * 6 + 3 + c] + IM[j][c * 6 + 3 + r]) > tol) return false;              // H skew
      if (cabs_(IM[j][(3 + r) * 6 + c] - IM[j][c * 6 + 3 + r]) > tol) return false;            // lower-left = H^T
    }
  return true;
}
constexpr int n_children_(int i) {
  int k = 0;
  for (int j = 0; j < N; ++j) k += (PARENT[j] == i) ? 1 : 0;
  return k;
}
constexpr bool grad_idsva_ok_() {
  if (!GRAD_PER_ROOT) return false;
  for (int j = 0; j < N; ++j) {
    if (JTYPE[j] != 0) return false;
    if (!rigid_inertia_(j)) return false;
    if (n_children_(j) > 1) return false;            // chains only
    if (PARENT[j] != -1 && PARENT[j] != j - 1) return false;
  }
  // The per-body cost of this scheme is fixed (~500 instructions) while the column recursion costs
  // ~150 per (body, ancestor) pair: measured break-even is an average chain position of about 3
  // (quadruped legs, depth 3: the column kernel is 3-18 % faster; 7-DoF arm: this one is 1.6x faster).
  int pairs = 0;
  for (int j = 0; j < N; ++j) pairs += DEPTH[j] + 1;
  if (pairs < 3 * N) return false;
  return grad_max_rows() <= 8;
}
constexpr bool GRAD_IDSVA_OK = grad_idsva_ok_();

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

// The block is ONE wave and a wave's LDS operations execute in order: ordering LDS traffic between
// lanes needs a compiler-level fence only.  __syncthreads() would add s_waitcnt vmcnt(0), i.e. wait
// for every global store / load still in flight.
#ifdef RBD_EXP_BLOCKSYNC
#define IDS_WAVE_SYNC() __syncthreads()
#else
#define IDS_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// rnea_grad, one configuration per lane (64 per block).  Same signature and output layout as
// rnea_grad_kernel<T, HAS_QDD, false>.
// ---------------------------------------------------------------------------------------------
// FDG (single-group robots only) = forward_dynamics_grad epilogue (RBDReference.py:1376-1384): when
// the sweep is over and the registers are free, every lane pulls its finished dc_du row out of the
// LDS tile, the block's Minv rows ([64][n*n], contiguous in `minv_in`) are staged through the now
// idle tile with coalesced loads, and [qdd_dq | qdd_dqd] = -Minv dc_du goes back into the tile for
// the usual flush.  No extra LDS (an earlier version with a separate Minv tile dropped to 4 blocks per
// CU and ran 2x slower), no dc_du round trip through HBM.
template <class T, bool HAS_QDD, bool FDG = false>
__global__ __launch_bounds__(64, sizeof(T) == 4 ? 2 : 1) void rnea_grad_idsva_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                                const T* __restrict__ qdd, T grav, int use_damping,
                                                                long long B, T* __restrict__ c_out,
                                                                T* __restrict__ dcdu, const T* __restrict__ minv_in = nullptr) {
  static_assert(!FDG || grad_max_rows() == N, "fused -Minv epilogue: single-group robots only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x;
  constexpr int CFGS = 64;
  const long long cfg0 = (long long)blockIdx.x * CFGS;
  const long long rem = B - cfg0;
  const int nvalid = rem < CFGS ? (int)rem : CFGS;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);

#ifdef RBD_EXP_STAMPS
#define IDS_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[k] = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
  unsigned long long stamps[8];
#else
#define IDS_STAMP(k) do {} while (0)
#endif
  IDS_STAMP(0);
  JTrig<T> tr[N];
  T qv[N], qdv[N], qddv[N];
  auto load_group = [&](auto G) {
    constexpr int g = decltype(G)::value;
    sfor<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (grp_has(g, j)) {
        qv[j] = q[b * N + j];
        qdv[j] = qd[b * N + j];
        if constexpr (HAS_QDD) qddv[j] = qdd[b * N + j]; else qddv[j] = T(0);
      }
    });
  };

  T Sv[N][6], Pd[N][6], Pdd[N][6];
  sfor<0, N>([&](auto Rt) {
   constexpr int rt = decltype(Rt)::value;
   if constexpr (grp_head(rt)) {
    constexpr int row0 = grp_row0(rt);
    constexpr int rows = grp_rows(rt);
    constexpr int last = row0 + rows - 1;
    T* my = tile + lane * GRAD_TS - row0 * GRAD_ROW;   // my[i * 2N + c] (dq), + N (dqd)
    if constexpr (rt == grp_first()) load_group(Rt);
#ifdef RBD_EXP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    IDS_STAMP(1);
    sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; if constexpr (grp_has(rt, j)) tr[j] = make_trig<j>(qv[j]); });
    IDS_STAMP(2);
    if constexpr (grp_next(rt) >= 0) load_group(std::integral_constant<int, grp_next(rt) >= 0 ? grp_next(rt) : 0>{});

    // ---- forward: world kinematics of the chain (:1413-1434) ------------------------------------
    T Rm[3][3], pw[3], v[6], a[6];          // state of the current body: R (body -> world), origin, v, a
    sfor<row0, row0 + rows>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int k = AXIS[j], ka = (k + 1) % 3, kb = (k + 2) % 3;
      constexpr bool root = PARENT[j] < 0;
      T Tm[3][3];
      // T = R_p E_tree^T ; p = p_p + R_p r_tree
      sfor<0, 3>([&](auto R_) {
        sfor<0, 3>([&](auto C_) {
          constexpr int r = decltype(R_)::value, c = decltype(C_)::value;
          if constexpr (root) {
            Tm[r][c] = T(Et_(j, c, r));
          } else {
            T acc = T(0);
            sfor<0, 3>([&](auto M_) {
              constexpr int m = decltype(M_)::value;
              constexpr double e = Et_(j, c, m);
              if constexpr (e == 1.0) acc = acc + Rm[r][m];
              else if constexpr (e == -1.0) acc = acc - Rm[r][m];
              else if constexpr (e != 0.0) acc = fma_(T(e), Rm[r][m], acc);
            });
            Tm[r][c] = acc;
          }
        });
      });
      T pn[3];
      sfor<0, 3>([&](auto R_) {
        constexpr int r = decltype(R_)::value;
        if constexpr (root) {
          pn[r] = T(rt_(j, r));
        } else {
          T acc = pw[r];
          sfor<0, 3>([&](auto M_) {
            constexpr int m = decltype(M_)::value;
            constexpr double e = rt_(j, m);
            if constexpr (e != 0.0) acc = fma_(T(e), Rm[r][m], acc);
          });
          pn[r] = acc;
        }
      });
      // R = T Rj^T  (columns ka, kb rotate)
      sfor<0, 3>([&](auto R_) {
        constexpr int r = decltype(R_)::value;
        Rm[r][ka] = fma_(tr[j].c, Tm[r][ka], tr[j].s * Tm[r][kb]);
        Rm[r][kb] = fma_(tr[j].c, Tm[r][kb], -(tr[j].s * Tm[r][ka]));
        Rm[r][k] = Tm[r][k];
        pw[r] = pn[r];
      });
      const T ang[3] = {Rm[0][k], Rm[1][k], Rm[2][k]};
      T sl[3];
      cross3(pw, ang, sl);
      sfor<0, 3>([&](auto R_) { constexpr int r = decltype(R_)::value; Sv[j][r] = ang[r]; Sv[j][3 + r] = sl[r]; });
      if constexpr (root) {
        // v_p = 0, a_p = [0,0,0,0,0,-GRAVITY]  (:1417-1420): psid = 0, psidd = a_p x S
        sfor<0, 6>([&](auto R_) { Pd[j][decltype(R_)::value] = T(0); });
        Pdd[j][0] = T(0); Pdd[j][1] = T(0); Pdd[j][2] = T(0);
        Pdd[j][3] = grav * ang[1];           // (0,0,g) x ang with g = -GRAVITY
        Pdd[j][4] = -(grav * ang[0]);
        Pdd[j][5] = T(0);
        sfor<0, 6>([&](auto R_) {
          constexpr int r = decltype(R_)::value;
          v[r] = Sv[j][r] * qdv[j];
          a[r] = Sv[j][r] * qddv[j];
        });
        a[5] -= grav;
      } else {
        T t1[6], t2[6];
        crm6(v, Sv[j], Pd[j]);               // psid  = v_p x S                 (:1431)
        crm6(a, Sv[j], t1);                  // psidd = a_p x S + v_p x psid    (:1432)
        crm6(v, Pd[j], t2);
        sfor<0, 6>([&](auto R_) {
          constexpr int r = decltype(R_)::value;
          Pdd[j][r] = t1[r] + t2[r];
          v[r] = fma_(Sv[j][r], qdv[j], v[r]);                                   // (:1433)
          a[r] = fma_(Sv[j][r], qddv[j], fma_(Pd[j][r], qdv[j], a[r]));         // (:1430,:1434)
        });
      }
    });

    IDS_STAMP(3);
    // ---- backward: local inertia terms, composites, t-vectors, all pairs of the body -------------
    RInertia<T> IC;
    SymB<T> SC;
    T pmC[6], fC[6];
    sfor_down<row0, row0 + rows>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int k = AXIS[j], ka = (k + 1) % 3, kb = (k + 2) % 3;
      // world rigid inertia of body j about the world origin
      RInertia<T> L;
      T cw[3];
      sfor<0, 3>([&](auto R_) {
        constexpr int r = decltype(R_)::value;
        T acc = pw[r];
        sfor<0, 3>([&](auto M_) {
          constexpr int m = decltype(M_)::value;
          constexpr double e = com_(j, m);
          if constexpr (e != 0.0) acc = fma_(T(e), Rm[r][m], acc);
        });
        cw[r] = acc;
      });
      L.m = T(mass_(j));
      sfor<0, 3>([&](auto R_) { L.h[decltype(R_)::value] = T(mass_(j)) * cw[decltype(R_)::value]; });
      {
        // A = R Ic ; Ibar = A R^T + m (|cw|^2 1 - cw cw^T)
        T A[3][3];
        sfor<0, 3>([&](auto R_) {
          sfor<0, 3>([&](auto C_) {
            constexpr int r = decltype(R_)::value, c = decltype(C_)::value;
            T acc = T(0);
            sfor<0, 3>([&](auto M_) {
              constexpr int m = decltype(M_)::value;
              constexpr double e = Ic_(j, m, c);
              if constexpr (e != 0.0) acc = fma_(T(e), Rm[r][m], acc);
            });
            A[r][c] = acc;
          });
        });
        const T cc = fma_(cw[0], cw[0], fma_(cw[1], cw[1], cw[2] * cw[2]));
        constexpr int IR[6] = {0, 0, 0, 1, 1, 2}, IC_[6] = {0, 1, 2, 1, 2, 2};
        sfor<0, 6>([&](auto E_) {
          constexpr int e = decltype(E_)::value, r = IR[e], c = IC_[e];
          T s = fma_(A[r][0], Rm[c][0], fma_(A[r][1], Rm[c][1], A[r][2] * Rm[c][2]));
          const T mcc = L.h[r] * cw[c];                      // m cw_r cw_c
          if constexpr (r == c) s += fma_(L.m, cc, -mcc); else s -= mcc;
          L.I[e] = s;
        });
      }
      T pm[6], fl[6], Ia[6];
      rin_apply(L, v, pm);                                    // momentum I v
      rin_apply(L, a, Ia);
      fxv<false>(v, pm, fl);                                  // f = I a + v x* (I v)   (:1440)
      sfor<0, 6>([&](auto R_) { fl[decltype(R_)::value] += Ia[decltype(R_)::value]; });
      // Sym part of B = crf(v) I + icrf(I v) - I crm(v):  TL = K + K^T - (h u^T + u h^T) + 2 (u.h) 1,
      // K = w^x Ibar ;  G = w x h + m u
      SymB<T> Sl;
      {
        const T w[3] = {v[0], v[1], v[2]}, u[3] = {v[3], v[4], v[5]};
        const T Ifull[3][3] = {{L.I[0], L.I[1], L.I[2]}, {L.I[1], L.I[3], L.I[4]}, {L.I[2], L.I[4], L.I[5]}};
        T K[3][3];
        sfor<0, 3>([&](auto C_) {
          constexpr int c = decltype(C_)::value;
          const T col[3] = {Ifull[0][c], Ifull[1][c], Ifull[2][c]};
          T o[3];
          cross3(w, col, o);
          K[0][c] = o[0]; K[1][c] = o[1]; K[2][c] = o[2];
        });
        const T uh2 = T(2) * fma_(u[0], L.h[0], fma_(u[1], L.h[1], u[2] * L.h[2]));
        constexpr int IR[6] = {0, 0, 0, 1, 1, 2}, IC_[6] = {0, 1, 2, 1, 2, 2};
        sfor<0, 6>([&](auto E_) {
          constexpr int e = decltype(E_)::value, r = IR[e], c = IC_[e];
          T s = K[r][c] + K[c][r];
          s = fma_(-L.h[r], u[c], fma_(-u[r], L.h[c], s));
          if constexpr (r == c) s += uh2;
          Sl.TL[e] = s;
        });
        T g[3];
        cross3(w, L.h, g);
        sfor<0, 3>([&](auto R_) { constexpr int r = decltype(R_)::value; Sl.G[r] = fma_(L.m, u[r], g[r]); });
      }
      // composites (plain sums in the world frame, :1446-1448)
      if constexpr (j == last) {
        IC = L; SC = Sl;
        sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; pmC[r] = pm[r]; fC[r] = fl[r]; });
      } else {
        IC.m += L.m;
        sfor<0, 3>([&](auto R_) { constexpr int r = decltype(R_)::value; IC.h[r] += L.h[r]; SC.G[r] += Sl.G[r]; });
        sfor<0, 6>([&](auto R_) {
          constexpr int r = decltype(R_)::value;
          IC.I[r] += L.I[r]; SC.TL[r] += Sl.TL[r]; pmC[r] += pm[r]; fC[r] += fl[r];
        });
      }
      // c_j and the t-vectors (:1481-1484)
      const T cj = dot6(Sv[j], fC);
      if (c_out != nullptr && lane < nvalid) c_out[b * N + j] = cj;
      T y1[6], y3[6], s1[6], z1[6], zf[6];
      rin_apply(IC, Sv[j], y1);
      rin_apply(IC, Pdd[j], y3);
      sym_apply(SC, Sv[j], s1);
      fxv<false>(Sv[j], pmC, z1);
      fxv<false>(Sv[j], fC, zf);
      T t1[6], t2[6], t3[6], t4[6];
      if constexpr (PARENT[j] < 0) {
        // psid of a root is identically zero (v_parent = 0): its three products drop out
        sfor<0, 6>([&](auto R_) {
          constexpr int r = decltype(R_)::value;
          t1[r] = y1[r];
          t4[r] = s1[r] - z1[r];
          t3[r] = y3[r] + zf[r];
          t2[r] = s1[r] + z1[r];
        });
      } else {
        T y2[6], s2[6], z2[6];
        rin_apply(IC, Pd[j], y2);
        sym_apply(SC, Pd[j], s2);
        fxv<false>(Pd[j], pmC, z2);
        sfor<0, 6>([&](auto R_) {
          constexpr int r = decltype(R_)::value;
          t1[r] = y1[r];
          t4[r] = s1[r] - z1[r];
          t3[r] = (s2[r] + z2[r]) + (y3[r] + zf[r]);
          t2[r] = fma_(T(2), y2[r], s1[r] + z1[r]);
        });
      }
      // all pairs (j, jj) with jj an ancestor-or-self of j (chain: row0 .. j)
      sfor<row0, j + 1>([&](auto JJ) {
        constexpr int jj = decltype(JJ)::value;
        T dq_ij, dqd_ij;
        if constexpr (PARENT[jj] < 0) {
          // root ancestor: psid = 0 and psidd = a_base x S = (0, 0, 0, g S_y, -g S_x, 0)
          dq_ij = fma_(t1[3], Pdd[jj][3], t1[4] * Pdd[jj][4]);
          dqd_ij = dot6(t4, Sv[jj]);
        } else {
          dq_ij = dot6(t4, Pd[jj]) + dot6(t1, Pdd[jj]);
          dqd_ij = fma_(T(2), dot6(t1, Pd[jj]), dot6(t4, Sv[jj]));
        }
        if constexpr (jj == j) dqd_ij += sel(use_damping != 0, T(DAMPING[j]), T(0));   // :1336-1341
        my[j * GRAD_ROW + jj] = dq_ij;
        my[j * GRAD_ROW + N + jj] = dqd_ij;
        if constexpr (jj != j) {
          my[jj * GRAD_ROW + j] = dot6(Sv[jj], t3);
          my[jj * GRAD_ROW + N + j] = dot6(Sv[jj], t2);
        }

      });
      // structural zeros of row j / column j: bodies of other groups
      sfor<0, N>([&](auto C_) {
        constexpr int c = decltype(C_)::value;
        if constexpr (!grp_has(rt, c)) { my[j * GRAD_ROW + c] = T(0); my[j * GRAD_ROW + N + c] = T(0); }
      });
      // step the kinematic state back to the parent: v_p, a_p, R_p, p_p
      if constexpr (j > row0) {
        sfor<0, 6>([&](auto R_) {
          constexpr int r = decltype(R_)::value;
          v[r] = fma_(-Sv[j][r], qdv[j], v[r]);
          a[r] = fma_(-Sv[j][r], qddv[j], fma_(-Pd[j][r], qdv[j], a[r]));
        });
        T Tm[3][3];
        sfor<0, 3>([&](auto R_) {
          constexpr int r = decltype(R_)::value;
          Tm[r][ka] = fma_(tr[j].c, Rm[r][ka], -(tr[j].s * Rm[r][kb]));
          Tm[r][kb] = fma_(tr[j].s, Rm[r][ka], tr[j].c * Rm[r][kb]);
          Tm[r][k] = Rm[r][k];
        });
        sfor<0, 3>([&](auto R_) {
          sfor<0, 3>([&](auto C_) {
            constexpr int r = decltype(R_)::value, c = decltype(C_)::value;
            T acc = T(0);
            sfor<0, 3>([&](auto M_) {
              constexpr int m = decltype(M_)::value;
              constexpr double e = Et_(j, m, c);
              if constexpr (e == 1.0) acc = acc + Tm[r][m];
              else if constexpr (e == -1.0) acc = acc - Tm[r][m];
              else if constexpr (e != 0.0) acc = fma_(T(e), Tm[r][m], acc);
            });
            Rm[r][c] = acc;
          });
        });
        sfor<0, 3>([&](auto R_) {
          constexpr int r = decltype(R_)::value;
          T acc = pw[r];
          sfor<0, 3>([&](auto M_) {
            constexpr int m = decltype(M_)::value;
            constexpr double e = rt_(j, m);
            if constexpr (e != 0.0) acc = fma_(T(-e), Rm[r][m], acc);
          });
          pw[r] = acc;
        });
      }
    });

    if constexpr (FDG) {
      // the block's Minv rows: n^2 coalesced loads per lane, all issued before anything waits on them
      // (as a run-time loop of load -> ds_write pairs this staging paid one memory round trip per
      // iteration)
      T ms[N * N];
      {
        const T* msrc = minv_in + cfg0 * (N * N);
        const int lim = nvalid * N * N;
        sfor<0, N * N>([&](auto K_) {
          constexpr int k = decltype(K_)::value;
          const int g = lane + CFGS * k;
          ms[k] = msrc[g < lim ? g : 0];
        });
      }
      T D[GRAD_TILE];
      sfor<0, GRAD_TILE>([&](auto K_) { constexpr int k = decltype(K_)::value; D[k] = my[k]; });
      IDS_WAVE_SYNC();                                   // every lane has its row in registers
      sfor<0, N * N>([&](auto K_) { constexpr int k = decltype(K_)::value; tile[lane + CFGS * k] = ms[k]; });
      IDS_WAVE_SYNC();
      T Mm[N * N];
      {
        const T* mrow = tile + (lane < nvalid ? lane : 0) * (N * N);
        sfor<0, N * N>([&](auto K_) { constexpr int k = decltype(K_)::value; Mm[k] = mrow[k]; });
      }
      IDS_WAVE_SYNC();                                   // Minv is in registers; the tile can take the outputs
      sfor<0, N>([&](auto I_) {
        sfor<0, 2 * N>([&](auto C_) {
          constexpr int i = decltype(I_)::value, c = decltype(C_)::value;
          T o = T(0);
          sfor<0, N>([&](auto K_) { constexpr int k = decltype(K_)::value; o = fma_(-Mm[i * N + k], D[k * GRAD_ROW + c], o); });
          my[i * GRAD_ROW + c] = o;
        });
      });
    }
    // ---- stream this group's rows out ---------------------------------------------------------------
    IDS_STAMP(4);
    IDS_WAVE_SYNC();
    {
      constexpr int RW = rows * GRAD_ROW;
      T* gdst = dcdu + cfg0 * GRAD_TILE + row0 * GRAD_ROW;
      constexpr int VE = 16 / sizeof(T);
      bool done = false;
      if constexpr (RW == GRAD_TILE && GRAD_TS == GRAD_TILE && (CFGS * GRAD_TILE) % VE == 0) {
        typedef T V __attribute__((ext_vector_type(VE)));
        if (nvalid == CFGS) {
          const V* src = reinterpret_cast<const V*>(tile);
          V* dst = reinterpret_cast<V*>(gdst);
#pragma unroll 4
          for (int g = lane; g < CFGS * GRAD_TILE / VE; g += CFGS) dst[g] = src[g];
          done = true;
        }
      }
      if (!done) {
#pragma unroll 4
        for (int g = lane; g < nvalid * RW; g += CFGS) {
          const int cfg = g / RW;
          const int rem2 = g - cfg * RW;
          gdst[cfg * GRAD_TILE + rem2] = tile[cfg * GRAD_TS + rem2];
        }
      }
    }
    if constexpr (rows != N) IDS_WAVE_SYNC();
#ifdef RBD_EXP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IDS_STAMP(5);
    if (lane == 0 && c_out != nullptr) {   // DIAGNOSTIC BUILD ONLY
      for (int k = 0; k < 5; ++k) c_out[cfg0 * N + k] = (T)(float)(stamps[k + 1] - stamps[k]);
    }
#endif
   }
  });
}

}  // namespace rbdk
