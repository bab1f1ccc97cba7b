// rbd_idsva_pipe.h -- the one-lane world-frame rnea_grad kernel (rbd_idsva.h) for robots that are ONE chain,
// with the tile loop software-pipelined: the flush of tile t is interleaved with the start of tile t + 1.
//
// Why (measured, tools/ubench/pk_issue.hip, issue_mix.hip): one wave issues at most one VALU instruction per
// ~5 cycles while the SIMD's pipe takes one per < 2, so with the two waves per SIMD that 250 VGPRs allow the
// pipe is idle whenever either wave waits -- and the flush is the part of a tile that mostly waits: LDS
// read-back of the parked entries, write of the [32][2n^2] row image, 16-byte reads of it, stores, twice, then
// c.  None of that depends on the next tile's arithmetic, whose inputs are already in registers (prefetched in
// the backward sweep), and the start of a tile is where registers are plentiful.  So the flush is cut into
// stages (each issues its LDS / global operations and returns) and the next tile's sin/cos and first forward
// bodies run between them; the stage that needs a previous stage's data finds it arrived.  The forward bodies
// of the next tile park their cold values in the tile only after the last image read has been issued (one
// wave = in-order LDS: no barrier needed, a compiler-level fence only).
//
// Mathematics, tile layout, parking and the row image are those of rnea_grad_idsva_kernel (rbd_idsva.h);
// reference: /root/reference/RBDReference.py:1345-1368 (rnea_grad), :1413-1484 (the world-frame identities).
#pragma once
#include "rbd_idsva.h"

namespace rbdk {

constexpr bool ids_pipe_ok_() {
  if (!GRAD_IDSVA_OK) return false;
  if (n_groups() != 1 || grad_max_rows() != N) return false;      // one chain = one group
  if (N < 4) return false;
  // the finished rows of 32 configurations leave as flat 16-byte copies
  if ((32 * GRAD_TILE) % 4 != 0 || GRAD_TILE % 2 != 0) return false;
  return true;
}
constexpr bool IDS_PIPE_OK = ids_pipe_ok_();

#define IDS_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef IDS_FDG_LOAD_AT
#define IDS_FDG_LOAD_AT 0                           // FDG: body (from the root) at the start of whose backward step the tile's Minv is requested
#endif
// Parking of cold per-body values in the tile (rbd_idsva.h) is OFF here: with the flush out of the way this
// kernel's register peak is lower (236 VGPRs with every park, 248 with none, no scratch either way), and every
// park is two LDS writes, two reads and a wait in a wave that has only one partner to hide it -- interleaved A/B
// on the 7-DoF arm, B = 2^20: all parks 134.4 us, none 132.4, none + 50 register entries 130.7 (medians of 30 rounds).
#ifndef IDS_PIPE_LEAF_PARK
#define IDS_PIPE_LEAF_PARK 0                        // the leaf's {sin, cos, qd, qdd} during its own backward step
#endif
#ifndef IDS_PIPE_C_PARK
#define IDS_PIPE_C_PARK 0                           // c of the top bodies
#endif
#ifndef IDS_PIPE_PARK_UPTO
#define IDS_PIPE_PARK_UPTO 0                        // inner bodies 1 .. this index park their {sin, cos, qd, qdd}
#endif
#ifndef IDS_PIPE_REG_ENTRIES_MAX
#define IDS_PIPE_REG_ENTRIES_MAX 50                 // entries a lane keeps in registers during the sweep (rbd_idsva.h: 36)
#endif
constexpr int ids_pipe_rows_a(int rows) {           // bodies whose entries wait in LDS: as few as that budget allows
  for (int a = 0; a < rows; ++a)
    if (ids_regs_for(rows, a) <= IDS_PIPE_REG_ENTRIES_MAX && ids_tile_for(rows, a) <= IDS_TS) return a;
  return ids_rows_a(rows);
}
// "x is defined here" without an instruction (ends the live range of whatever x held)
template <class X>
RBD_DEV void ids_undef(X& x) { asm volatile("" : "=v"(x)); }

template <class T, bool HAS_QDD, bool FDG = false>
__global__ __launch_bounds__(64, 2) void rnea_grad_idsva_pipe_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                                    const T* __restrict__ qdd, T grav, int use_damping,
                                                                    long long B, T* __restrict__ c_out,
                                                                    T* __restrict__ dcdu, const T* __restrict__ minv_pk) {
  static_assert(sizeof(T) == 4, "fp32 kernel");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x;
  constexpr int CFGS = 64;
  constexpr int TS = IDS_TS;
  constexpr int row0 = 0, rows = N, last = N - 1;
  constexpr int ra = ids_pipe_rows_a(rows);                // top bodies: their entries wait in LDS
  static_assert(ids_tile_for(rows, ra) <= IDS_TS, "tile overflow");
  constexpr int rs = rows - ra;                            // first of them
  constexpr int PEND = ra * GRAD_ROW;                      // tile offset of the pending column entries
  constexpr int RW = GRAD_TILE;                            // finished scalars per configuration
  constexpr int VE = 4;
  constexpr int NV = 32 * RW / VE;                         // 16-byte pieces of a half-tile image
  constexpr int NRD = (NV + 63) / 64;                      // reads / stores per lane and half
  constexpr int NVC = CFGS * N / VE;                       // 16-byte pieces of the c image (when 64 n % 4 == 0)
  constexpr int NRC = (NVC + 63) / 64;
  typedef T V __attribute__((ext_vector_type(VE)));
  const long long ntiles = (B + CFGS - 1) / CFGS;
  long long t = blockIdx.x;
  if (t >= ntiles) return;
  T* my = tile + lane * TS;
  // the grid size stays in an SGPR: re-read from the dispatch packet inside the loop (what the compiler does
  // otherwise) it is a scalar load in flight, and with one in flight every LDS wait of the staged flush
  // becomes lgkmcnt(0) -- scalar loads return out of order
  int gdim = (int)gridDim.x;
  asm volatile("" : "+s"(gdim));

  JTrig<T> tr[N];
  T qv[N], qdv[N], qddv[N];
  T qn[N], qdn[N], qddn[N];                                // inputs in flight: the next tile
  T Sv[N][6], Pd[N][6], Pdd[N][6];
  T cv[N];
  T Rm[3][3], pw[3], v[6], a[6];                           // state of the current body: R (body -> world), origin, v, a
  T E[RW];                                                 // finished rows, final layout [r * 2n + c]
  T mk[FDG ? FDC_NP : 1];                                  // forward_dynamics_grad: this lane's Minv (upper triangle) of the current tile
  long long tcur = 0;                                      // ... whose index the backward sweep needs for that request

  auto load_inputs = [&](long long tt, bool dummy) {
    const long long c0 = tt * CFGS;
    const long long rm = dummy ? 1 : B - c0;               // dummy: every lane reads row 0 of the tile
    unsigned lo = (unsigned)((lane < rm ? lane : (int)rm - 1) * N);   // clamped in the last tile
    asm volatile("" : "+v"(lo));
    const T* qt = q + c0 * N + lo;
    const T* qdt = qd + c0 * N + lo;
    const T* qddt = HAS_QDD ? qdd + c0 * N + lo : nullptr;
    sfor<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      qn[j] = qt[j];
      qdn[j] = qdt[j];
      if constexpr (HAS_QDD) qddn[j] = qddt[j]; else qddn[j] = T(0);
    });
  };
  // where the wait for the loads lands (rbd_idsva.h, settle_group: before a flush's stores are issued)
  auto settle_inputs = [&]() {
    sfor<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      qn[j] = launder(qn[j]); qdn[j] = launder(qdn[j]); qddn[j] = launder(qddn[j]);
    });
  };
  auto take_inputs = [&]() {
    sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; qv[j] = qn[j]; qdv[j] = qdn[j]; qddv[j] = qddn[j]; });
  };
  // sin / cos of bodies [J0, J1): ONE wave-uniform branch for the chunk (any lane, any joint beyond the range of
  // the fast reduction or non-finite -> the branch-free wide routine of rbd_sincos.h for the chunk)
  auto trig_chunk = [&](auto J0_, auto J1_) {
    constexpr int J0 = decltype(J0_)::value, J1 = decltype(J1_)::value;
    T mx = T(0);
    sfor<J0, J1>([&](auto J) { constexpr int j = decltype(J)::value; mx = __builtin_fmaxf(mx, __builtin_fabsf(qv[j])); });
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(mx <= 8192.0f)) != 0, 0)) {
      sfor<J0, J1>([&](auto J) { constexpr int j = decltype(J)::value; sincos_wide_(qv[j], &tr[j].s, &tr[j].c); });
    } else {
      sfor<J0, J1>([&](auto J) { constexpr int j = decltype(J)::value; sincos_core_(qv[j], &tr[j].s, &tr[j].c); });
    }
  };

  // ---- forward step of body j: world kinematics (:1413-1434) -------------------------------------------
  auto fwd_body = [&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int k = AXIS[j], ka = (k + 1) % 3, kb = (k + 2) % 3;
    constexpr bool root = PARENT[j] < 0;
    T Tm[3][3];
    sfor<0, 3>([&](auto R_) {                              // T = R_p E_tree^T
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
    sfor<0, 3>([&](auto R_) {                              // p = p_p + R_p r_tree
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
    sfor<0, 3>([&](auto R_) {                              // R = T Rj^T  (columns ka, kb rotate)
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
      Pdd[j][3] = grav * ang[1];
      Pdd[j][4] = -(grav * ang[0]);
      Pdd[j][5] = T(0);
      sfor<0, 6>([&](auto R_) {
        constexpr int r = decltype(R_)::value;
        v[r] = Sv[j][r] * qdv[j];
        a[r] = Sv[j][r] * qddv[j];
      });
      a[5] -= grav;
    } else {
      T t1[6];
      crm6(v, Sv[j], Pd[j]);                               // psid  = v_p x S                 (:1431)
      crm6(a, Sv[j], t1);                                  // psidd = a_p x S + v_p x psid    (:1432)
      crm6_add(v, Pd[j], t1, Pdd[j]);
      sfor<0, 6>([&](auto R_) {
        constexpr int r = decltype(R_)::value;
        v[r] = fma_(Sv[j][r], qdv[j], v[r]);                                   // (:1433)
        a[r] = fma_(Sv[j][r], qddv[j], fma_(Pd[j][r], qdv[j], a[r]));         // (:1430,:1434)
      });
    }
  };
  // cold until the backward sweep steps back over body j: {sin, cos, qd, qdd} wait in the tile
  auto park = [&](auto J) {
    constexpr int j = decltype(J)::value;
    if constexpr (j > row0 && j < last && j <= IDS_PIPE_PARK_UPTO) {
      my[ids_park_slot(row0, rows, ra, j, 0)] = tr[j].s; my[ids_park_slot(row0, rows, ra, j, 1)] = tr[j].c;
      my[ids_park_slot(row0, rows, ra, j, 2)] = qdv[j]; my[ids_park_slot(row0, rows, ra, j, 3)] = qddv[j];
    }
    if constexpr (j == last && (IDS_PIPE_LEAF_PARK && ids_leaf_parks(row0, rows, ra))) {
      my[ids_leaf_slot(row0, rows, ra, 0)] = tr[j].s; my[ids_leaf_slot(row0, rows, ra, 1)] = tr[j].c;
      my[ids_leaf_slot(row0, rows, ra, 2)] = qdv[j]; my[ids_leaf_slot(row0, rows, ra, 3)] = qddv[j];
    }
  };

  // ---- backward sweep of a tile: local inertia terms, composites, t-vectors, all pairs (:1439-1484) ----
  auto backward = [&](long long tnext_or_t, bool dummy) {
    RInertia<T> IC;
    SymB<T> SC;
    T pmC[6], fC[6];
    sfor_down<row0, row0 + rows>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int k = AXIS[j], ka = (k + 1) % 3, kb = (k + 2) % 3;
      constexpr bool top = j >= rs;                          // this body's entries go to LDS (else: registers)
      if constexpr (j == rs && ra > 1 && (IDS_PIPE_C_PARK && ids_c_parks(row0, rows, ra))) {
        asm volatile("" ::: "memory");
        sfor<rs + 1, row0 + rows>([&](auto JC) { constexpr int jc = decltype(JC)::value; cv[jc] = my[ids_c_slot(row0, rows, ra, jc - rs - 1)]; });
      }
      if constexpr (j > row0 && j < last && j <= IDS_PIPE_PARK_UPTO) {
        // read-back of what the forward sweep parked: needed at the END of this step (the step back to the
        // parent) -- issued here, a whole body ahead, and kept here (the scheduler would sink the reads to
        // their use and wait for them there)
        asm volatile("" ::: "memory");
        tr[j].s = my[ids_park_slot(row0, rows, ra, j, 0)]; tr[j].c = my[ids_park_slot(row0, rows, ra, j, 1)];
        qdv[j] = my[ids_park_slot(row0, rows, ra, j, 2)]; qddv[j] = my[ids_park_slot(row0, rows, ra, j, 3)];
#ifndef RBD_EXP_NO_PIN_READBACK
        IDS_SB();
#endif
      }
      if constexpr (FDG && j == row0 + (rows > IDS_FDG_LOAD_AT ? IDS_FDG_LOAD_AT : rows - 1)) {
        // forward_dynamics_grad: this lane's Minv (upper triangle) comes back from the workspace fd_pre_kernel filled --
        // one coalesced dword load per entry, requested late in the sweep (registers are free by now) so that the rest
        // of the sweep and the read-back of the parked entries cover the latency
        IDS_SB();
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const T* mp = minv_pk + (size_t)tcur * (FDC_NP * 64) + lo;
#ifdef RBD_FDG_EXP_NOLOAD        // timing experiment (results are wrong): no Minv loads
        sfor<0, FDC_NP>([&](auto S_) { constexpr int s_ = decltype(S_)::value; mk[s_] = launder(T(1) + T(s_)); });
#else
        sfor<0, FDC_NP>([&](auto S_) { constexpr int s_ = decltype(S_)::value; mk[s_] = mp[s_ * 64]; });
#endif
        IDS_SB();
      }
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
          s = fma_(-L.h[r], cw[c], s);                       // - m cw_r cw_c
          if constexpr (r == c) s = fma_(L.m, cc, s);
          L.I[e] = s;
        });
      }
      // momentum I v, force f = I a + v x* (I v) (:1440), Sym part of B = crf(v) I + icrf(I v) - I crm(v):
      //   TL = K + K^T - (h u^T + u h^T) + 2 (u.h) 1,  K = w^x Ibar ;  G = w x h + m u
      // -- every one of them joins its composite (plain sums in the world frame, :1446-1448) as the START of its own
      // FMA chain instead of through a separate add (rbd_world.h, "accumulate forms")
      T pm[6];
      rin_apply(L, v, pm);
      if constexpr (j == last) {
        T Ia[6];
        rin_apply(L, a, Ia);
        fxv_add(v, pm, Ia, fC);
        IC = L;
        sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; pmC[r] = pm[r]; });
      } else {
        T t[6];
        rin_apply_acc(L, a, fC, t);
        fxv_add(v, pm, t, fC);
        IC.m += L.m;
        sfor<0, 3>([&](auto R_) { constexpr int r = decltype(R_)::value; IC.h[r] += L.h[r]; });
        sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; IC.I[r] += L.I[r]; pmC[r] += pm[r]; });
      }
      {
        const T w[3] = {v[0], v[1], v[2]}, u[3] = {v[3], v[4], v[5]};
        const T Ifull[3][3] = {{L.I[0], L.I[1], L.I[2]}, {L.I[1], L.I[3], L.I[4]}, {L.I[2], L.I[4], L.I[5]}};
        const T uh = fma_(u[0], L.h[0], fma_(u[1], L.h[1], u[2] * L.h[2]));
        constexpr int IR[6] = {0, 0, 0, 1, 1, 2}, IC_[6] = {0, 1, 2, 1, 2, 2};
        sfor<0, 6>([&](auto E_) {
          constexpr int e = decltype(E_)::value, r = IR[e], c = IC_[e];
          constexpr int r1 = (r + 1) % 3, r2 = (r + 2) % 3, c1 = (c + 1) % 3, c2 = (c + 2) % 3;
          const T base = (j == last) ? T(0) : SC.TL[e];
          if constexpr (r == c) {       // 2 K_rr - 2 h_r u_r + 2 (u.h)
            const T d = fma_(-L.h[r], u[r], fma_(w[r1], Ifull[r2][r], fma_(-w[r2], Ifull[r1][r], uh)));
            SC.TL[e] = fma_(T(2), d, base);
          } else {                      // K_rc + K_cr - h_r u_c - u_r h_c
            T x = fma_(w[r1], Ifull[r2][c], fma_(-w[r2], Ifull[r1][c], base));
            x = fma_(w[c1], Ifull[c2][r], fma_(-w[c2], Ifull[c1][r], x));
            SC.TL[e] = fma_(-L.h[r], u[c], fma_(-u[r], L.h[c], x));
          }
        });
        sfor<0, 3>([&](auto R_) {
          constexpr int r = decltype(R_)::value, r1 = (r + 1) % 3, r2 = (r + 2) % 3;
          const T base = (j == last) ? T(0) : SC.G[r];
          SC.G[r] = fma_(L.m, u[r], fma_(w[r1], L.h[r2], fma_(-w[r2], L.h[r1], base)));
        });
      }
      // c_j and the t-vectors (:1481-1484)
      cv[j] = dot6(Sv[j], fC);
      if constexpr (j > rs && (IDS_PIPE_C_PARK && ids_c_parks(row0, rows, ra))) my[ids_c_slot(row0, rows, ra, j - rs - 1)] = cv[j];
      T t1[6], t2[6], t3[6], t4[6];
      {
        T y3[6], s1[6], yf[6];
        rin_apply(IC, Sv[j], t1);                             // t1 = IC S
        sym_apply(SC, Sv[j], s1);
        fxv_sub(Sv[j], pmC, s1, t4);                          // t4 = SC S - S x* pmC
        rin_apply(IC, Pdd[j], y3);
        fxv_add(Sv[j], fC, y3, yf);                           // IC psidd + S x* fC
        if constexpr (PARENT[j] < 0) {                        // psid of a root is identically zero
          sfor<0, 6>([&](auto R_) {
            constexpr int r = decltype(R_)::value;
            t3[r] = yf[r];
            t2[r] = fma_(T(2), s1[r], -t4[r]);                // SC S + S x* pmC = 2 SC S - t4
          });
        } else {
          T y2[6], ys[6];
          rin_apply(IC, Pd[j], y2);
          sym_apply_acc(SC, Pd[j], yf, ys);                   // + SC psid
          fxv_add(Pd[j], pmC, ys, t3);                        // + psid x* pmC
          sfor<0, 6>([&](auto R_) {
            constexpr int r = decltype(R_)::value;
            t2[r] = fma_(T(2), y2[r] + s1[r], -t4[r]);        // 2 IC psid + SC S + S x* pmC
          });
        }
      }
      // all pairs (j, jj) with jj an ancestor-or-self of j
      sfor<row0, j + 1>([&](auto JJ) {
        constexpr int jj = decltype(JJ)::value;
        T dq_ij, dqd_ij;
        if constexpr (PARENT[jj] < 0) {
          dq_ij = fma_(t1[3], Pdd[jj][3], t1[4] * Pdd[jj][4]);
          dqd_ij = dot6(t4, Sv[jj]);
        } else {
          dq_ij = dot6_acc(t1, Pdd[jj], dot6(t4, Pd[jj]));
          dqd_ij = fma_(T(2), dot6(t1, Pd[jj]), dot6(t4, Sv[jj]));
        }
        if constexpr (jj == j) dqd_ij += sel(use_damping != 0, T(DAMPING[j]), T(0));   // :1336-1341
        if constexpr (top) {
          my[(j - rs) * GRAD_ROW + jj] = dq_ij;
          my[(j - rs) * GRAD_ROW + N + jj] = dqd_ij;
        } else {
          E[(j - row0) * GRAD_ROW + jj] = dq_ij;
          E[(j - row0) * GRAD_ROW + N + jj] = dqd_ij;
        }
        if constexpr (jj != j) {
          const T e3 = dot6(Sv[jj], t3), e2 = dot6(Sv[jj], t2);
          if constexpr (!top) {
            E[(jj - row0) * GRAD_ROW + j] = e3;
            E[(jj - row0) * GRAD_ROW + N + j] = e2;
          } else if constexpr (jj < rs) {          // a lower row: park (jj, j), (jj, N + j)
            constexpr int slot = PEND + ((jj - row0) * ra + (j - rs)) * 2;
            my[slot] = e3;
            my[slot + 1] = e2;
          } else {
            my[(jj - rs) * GRAD_ROW + j] = e3;
            my[(jj - rs) * GRAD_ROW + N + j] = e2;
          }
        }
      });
      // inputs of the next tile: requested late in the sweep, when the registers of the bodies already
      // processed are free.  Unconditional (a block's last tile re-reads one row): a branch would end in
      // register copies of the loaded values, i.e. in a wait right behind the loads' issue.
      if constexpr (j == row0 + (rows > IDS_PREFETCH_AT ? IDS_PREFETCH_AT : rows - 1)) {
        IDS_SB();
        load_inputs(tnext_or_t, dummy);
        IDS_SB();
      }
      if constexpr (j == last && (IDS_PIPE_LEAF_PARK && ids_leaf_parks(row0, rows, ra))) {
        asm volatile("" ::: "memory");
        tr[j].s = my[ids_leaf_slot(row0, rows, ra, 0)]; tr[j].c = my[ids_leaf_slot(row0, rows, ra, 1)];
        qdv[j] = my[ids_leaf_slot(row0, rows, ra, 2)]; qddv[j] = my[ids_leaf_slot(row0, rows, ra, 3)];
      }
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
  };

  // ---- the flush, in pieces ---------------------------------------------------------------------------
  // parked entries -> registers (after this every lane holds its 2 n^2 finished scalars and the tile is free)
  auto readback = [&]() {
    sfor<rs, row0 + rows>([&](auto R_) {
      sfor<0, GRAD_ROW>([&](auto C_) {
        constexpr int r = decltype(R_)::value, c = decltype(C_)::value;
        E[(r - row0) * GRAD_ROW + c] = my[(r - rs) * GRAD_ROW + c];
      });
    });
    sfor<row0, rs>([&](auto JJ) {
      sfor<rs, row0 + rows>([&](auto JA) {
        constexpr int jj = decltype(JJ)::value, ja = decltype(JA)::value;
        constexpr int slot = PEND + ((jj - row0) * ra + (ja - rs)) * 2;
        E[(jj - row0) * GRAD_ROW + ja] = my[slot];
        E[(jj - row0) * GRAD_ROW + N + ja] = my[slot + 1];
      });
    });
  };
  // lanes of half h write their rows into the compact [32][RW] image at the front of the tile
  auto image_write = [&](int ln, int h) {
    if ((ln >> 5) == h) {
      T* img = tile + (ln & 31) * RW;
      typedef T V2 __attribute__((ext_vector_type(2)));
      sfor<0, RW / 2>([&](auto K_) {
        constexpr int k = decltype(K_)::value;
        V2 x; x[0] = E[2 * k]; x[1] = E[2 * k + 1];
        reinterpret_cast<V2*>(img)[k] = x;
      });
    }
  };
  // (the staging registers are arguments, declared INSIDE the tile loop by the caller: declared outside they
  // would be loop-carried for the compiler -- live, i.e. spilled, across the whole backward sweep)
  auto image_read = [&](int ln, V (&buf)[NRD]) {
    const V* src = reinterpret_cast<const V*>(tile);
    sfor<0, NRD>([&](auto I_) {
      constexpr int i = decltype(I_)::value;
      if constexpr ((i + 1) * 64 <= NV) buf[i] = src[ln + 64 * i];
      else buf[i] = src[(ln + 64 * i < NV) ? ln + 64 * i : 0];
    });
  };
  auto image_store = [&](int ln, long long cfg0, int h, const V (&buf)[NRD]) {
    V* dst = reinterpret_cast<V*>(dcdu + (cfg0 + 32 * h) * GRAD_TILE);
    sfor<0, NRD>([&](auto I_) {
      constexpr int i = decltype(I_)::value;
      if constexpr ((i + 1) * 64 <= NV) dst[ln + 64 * i] = buf[i];
      else { if (ln + 64 * i < NV) dst[ln + 64 * i] = buf[i]; }
    });
  };
  // the unpipelined flush of one half (ragged tiles, and a block's last tile): as in rnea_grad_idsva_kernel
  auto flush_half_plain = [&](int ln, long long cfg0, int nvalid, int h) {
    image_write(ln, h);
    IDS_WAVE_SYNC();
    const int nv = nvalid - 32 * h;
    if (nv >= 32) {
      V buf[NRD];
      image_read(ln, buf);
      image_store(ln, cfg0, h, buf);
    } else if (nv > 0) {
      T* gdst = dcdu + (cfg0 + 32 * h) * GRAD_TILE;
#pragma unroll 2
      for (int g = ln; g < nv * RW; g += 64) gdst[g] = tile[g];
    }
    IDS_WAVE_SYNC();
  };
  auto flush_c_plain = [&](int ln, long long cfg0, int nvalid) {
    if (c_out != nullptr) {
      sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; tile[ln * N + j] = cv[j]; });
      IDS_WAVE_SYNC();
      T* cdst = c_out + cfg0 * N;
      if (nvalid == CFGS && (CFGS * N) % VE == 0) {
        sfor<0, NRC>([&](auto I_) {
          constexpr int i = decltype(I_)::value;
          if (ln + 64 * i < NVC) reinterpret_cast<V*>(cdst)[ln + 64 * i] = reinterpret_cast<const V*>(tile)[ln + 64 * i];
        });
      } else {
        for (int g = ln; g < nvalid * N; g += 64) cdst[g] = tile[g];
      }
      IDS_WAVE_SYNC();
    }
  };

  // ---- tile loop.  Iteration = [flush of the PREVIOUS tile, in stages] interleaved with [sin/cos and the first
  // forward bodies of this tile], the rest of the forward sweep, the backward sweep.  Order in the (in-order)
  // LDS queue: image 0 write, image 0 reads, image 1 write, image 1 reads, c image write, c reads, and only
  // then the first park of this tile.  The previous tile is never the ragged one (it had a successor).
  load_inputs(t, false);
  settle_inputs();
  bool first = true;
  long long pcfg0 = 0;                                     // first configuration of the previous tile
  constexpr bool c_flat = (CFGS * N) % VE == 0;
  for (;;) {
    V buf[NRD], cbuf[NRC];
    // "defined here" for the compiler (no instruction): without it the staging registers of the previous
    // iteration stay live around the back edge -- the uses below sit under `!first`, the definitions too, and
    // the two branches are not correlated for it -- and are spilled across the whole backward sweep
    sfor<0, NRD>([&](auto I_) { ids_undef(buf[decltype(I_)::value]); });
    sfor<0, NRC>([&](auto I_) { ids_undef(cbuf[decltype(I_)::value]); });
    const long long cfg0 = t * CFGS;
    const long long tnext = t + gdim;
    const bool has_next = tnext < ntiles;
    int ln = lane;
    asm volatile("" : "+v"(ln));                           // (what derives from the lane id is computed here, not hoisted out of the loop)

    take_inputs();
    IDS_SB();
    constexpr int TA = N >= 6 ? 2 : 1, TB = N >= 6 ? 4 : 2;   // sin / cos in three chunks: [0, TA) [TA, TB) [TB, N)
    trig_chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, TA>{});                 // ... the first one covers the read-back of the parked entries
    IDS_SB();
    // every LDS read of the read-back has landed.  Said explicitly (a real s_waitcnt the compiler's counter
    // bookkeeping sees): the image writes below sit in lane-masked branches, and on the branch-not-taken path
    // the read-back's destination registers would count as still pending -- every later reuse of one of them
    // (temporaries of the sin / cos code) then waits for most of the image reads in flight
    __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0) only
    if (!first) {
      image_write(ln, 0);
      IDS_WAVE_SYNC();
      image_read(ln, buf);
    }
    IDS_SB();
    trig_chunk(std::integral_constant<int, TA>{}, std::integral_constant<int, TB>{});                // sin / cos of the next bodies while the image is read
    IDS_SB();
    if (!first) {
      image_store(ln, pcfg0, 0, buf);
      IDS_WAVE_SYNC();                                     // image 0 has been read (in-order LDS): image 1 may overwrite it
      image_write(ln, 1);
      IDS_WAVE_SYNC();
      image_read(ln, buf);
    }
    IDS_SB();
    trig_chunk(std::integral_constant<int, TB>{}, std::integral_constant<int, N>{});
    fwd_body(std::integral_constant<int, 0>{});
    IDS_SB();
    if (!first) {
      image_store(ln, pcfg0, 1, buf);
      IDS_WAVE_SYNC();
      if (c_out != nullptr) {
        sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; tile[ln * N + j] = cv[j]; });
        IDS_WAVE_SYNC();
        if constexpr (c_flat) {
          sfor<0, NRC>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            cbuf[i] = reinterpret_cast<const V*>(tile)[(ln + 64 * i < NVC) ? ln + 64 * i : 0];
          });
        }
      }
    }
    IDS_SB();
    fwd_body(std::integral_constant<int, 1>{});
    IDS_SB();
    if (!first && c_out != nullptr) {
      T* cdst = c_out + pcfg0 * N;
      if constexpr (c_flat) {
        sfor<0, NRC>([&](auto I_) {
          constexpr int i = decltype(I_)::value;
          if (ln + 64 * i < NVC) reinterpret_cast<V*>(cdst)[ln + 64 * i] = cbuf[i];
        });
      } else {
        for (int g = ln; g < CFGS * N; g += 64) cdst[g] = tile[g];
      }
    }
    IDS_WAVE_SYNC();                                       // the last image read has been issued: the tile is this tile's
    park(std::integral_constant<int, 1>{});
    sfor<2, N>([&](auto J) { fwd_body(J); park(J); });

    tcur = t;
    backward(has_next ? tnext : t, !has_next);

    readback();
#ifdef RBD_FDG_EXP_NOPROD          // timing experiment (results are wrong): no product
    if constexpr (FDG) { sfor<0, FDC_NP>([&](auto S_) { constexpr int s_ = decltype(S_)::value; E[s_] += mk[s_]; }); }
    if constexpr (false) {
#else
    if constexpr (FDG) {
#endif
      // E <- -Minv E (:1381-1383), column by column in place: 2 n columns x n^2 FMAs; Minv is symmetric (:799-804)
      sfor<0, GRAD_ROW>([&](auto C_) {
        constexpr int c = decltype(C_)::value;
        T x[N];
        sfor<0, N>([&](auto K_) { constexpr int k = decltype(K_)::value; x[k] = E[k * GRAD_ROW + c]; });
        sfor<0, N>([&](auto I_) {
          constexpr int i = decltype(I_)::value;
          T o = -(mk[fdc_sym(i, 0)] * x[0]);
          sfor<1, N>([&](auto K_) { constexpr int k = decltype(K_)::value; o = fma_(-mk[fdc_sym(i, k)], x[k], o); });
          E[i * GRAD_ROW + c] = o;
        });
      });
    }
    settle_inputs();                                       // the next inputs have landed before any store is issued
    IDS_WAVE_SYNC();                                       // every lane has left the tile
    if (!has_next) break;                                  // uniform
    first = false;
    pcfg0 = cfg0;
    t = tnext;
  }
  // ---- a block's last tile (possibly the ragged one): nothing to overlap its flush with ---------------------
  {
    const long long cfg0 = t * CFGS;
    const long long rem = B - cfg0;
    const int nvalid = rem < CFGS ? (int)rem : CFGS;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    flush_half_plain(ln, cfg0, nvalid, 0);
    flush_half_plain(ln, cfg0, nvalid, 1);
    flush_c_plain(ln, cfg0, nvalid);
  }
}

}  // namespace rbdk
