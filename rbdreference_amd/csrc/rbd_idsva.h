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
#include "rbd_world.h"

namespace rbdk {

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

// The block is ONE wave and a wave's LDS operations execute in order: ordering LDS traffic between
// lanes needs a compiler-level fence only.  __syncthreads() would add s_waitcnt vmcnt(0), i.e. wait
// for every global store / load still in flight.
#define IDS_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// ---- output tile: late entries stay in registers --------------------------------------------------
// The backward sweep visits a chain leaf -> root.  Registers are scarce at its start (the leaf: every
// body's S, psid, psidd is still live) and plentiful at its end.  So only the entries produced by the TOP
// `a` bodies go to the LDS tile -- their own rows plus the 2 a (rows - a) column entries (jj, j),
// (jj, N + j) they contribute to the rows below ("pending") -- while everything the lower bodies
// produce stays in REGISTERS (2 (rows - a)^2 values).  For the 7-DoF arm: 3 + 4 bodies, 66 scalars of
// LDS per configuration instead of 98 -> 16.9 KB (+ parking) per 64-configuration wave instead of
// 25 KB -> 8 waves per CU (= the 2 waves per SIMD the VGPR count allows) instead of 6.  When the sweep is
// over every lane pulls its LDS entries into registers as well, and the finished rows leave through a
// compact [32][rows * 2n] LDS image, 32 configurations at a time, as flat 16-byte full-line stores
// (an earlier version flushed the top rows in the middle of the sweep: the partial cache lines cost
// +23 % HBM write traffic and 50 % run time).
constexpr int ids_tile_for(int rows, int a) {           // LDS scalars per configuration during the sweep
  if (a == 0) return 0;
  return a * GRAD_ROW + 2 * a * (rows - a);
}
constexpr int ids_regs_for(int rows, int a) { return 2 * (rows - a) * (rows - a); }
#ifndef IDS_PREFETCH_AT
#define IDS_PREFETCH_AT 3                               // body (counted from the chain's root) behind which the next inputs are requested
#endif
#ifndef IDS_PREFETCH_AT_F64
#define IDS_PREFETCH_AT_F64 0                           // fp64: at the root, when every other body of the sweep is done
#endif
#ifndef IDS_REG_ENTRIES_MAX
#define IDS_REG_ENTRIES_MAX 36                          // entries a lane may keep in registers during the sweep
#endif
constexpr int ids_rows_a(int rows) {                     // bodies whose entries go to LDS: as few as the register budget allows
  for (int a = 0; a < rows; ++a)
    if (ids_regs_for(rows, a) <= IDS_REG_ENTRIES_MAX) return a;
  return rows - 1;
}
// Cold per-body values (sin q, cos q, qd, qdd of the chain's inner bodies: written by the forward
// sweep, not needed again until the backward sweep steps the kinematic state back over that body)
// are parked in LDS too -- the register peak is at the LEAF body of the backward sweep.  A top body
// parks in its own row (columns 0, 1, N, N + 1: entries (j, c <= 1 <= j) that only body j itself
// writes, and it reloads first); the lower bodies park behind the tile.
constexpr int IDS_PARK = 4;
constexpr int ids_park_slot(int row0, int rows, int ra, int j, int k) {
  const int rs = row0 + rows - ra;
  if (j >= rs) return (j - rs) * GRAD_ROW + (k & 1) + (k >> 1) * N;
  return ids_tile_for(rows, ra) + IDS_PARK * (j - row0 - 1) + k;
}
// the leaf parks during its own backward step when the row above it is a top row with >= 4 own columns
constexpr bool ids_leaf_parks(int row0, int rows, int ra) { return ra >= 2 && rows >= 5; }
constexpr int ids_leaf_slot(int row0, int rows, int ra, int k) {
  const int rs = row0 + rows - ra, r = row0 + rows - 2;
  return (r - rs) * GRAD_ROW + 2 + (k & 1) + (k >> 1) * N;
}
// c of the top bodies above body rs waits in row rs, columns 2.. of both halves (written by body rs only)
constexpr bool ids_c_parks(int row0, int rows, int ra) { return ra > 1 && ra - 1 <= 2 * (rows - ra - 1); }
constexpr int ids_c_slot(int row0, int rows, int ra, int k) { return 2 + (k >> 1) + (k & 1) * N; }   // row rs is the tile's first row
constexpr int ids_spare_for(int rows, int a) { return rows - a > 1 ? IDS_PARK * (rows - a - 1) : 0; }
// LDS stride between configurations: == 2 (mod 4), i.e. even with an odd half => the 4-byte per-lane
// writes of a 32-lane group fall on 16 distinct banks (2-way, free); and 64 strides must hold the
// [32][rows * 2n] image the finished rows leave through.
constexpr int ids_tile_stride() {
  int m = 0;
  for (int rt = 0; rt < N; ++rt)
    if (grp_head(rt)) {
      const int rows = grp_rows(rt), a = ids_rows_a(rows);
      int t = ids_tile_for(rows, a) + ids_spare_for(rows, a);
      const int img = (rows * GRAD_ROW + 1) / 2;       // 32 x rows x 2n scalars over 64 strides
      t = t > img ? t : img;
      m = t > m ? t : m;
    }
  while (m % 4 != 2) ++m;
  return m;
}
constexpr int IDS_TS = ids_tile_stride();

#ifdef RBD_EXP_STAMPS
// DIAGNOSTIC BUILDS ONLY (-DRBD_EXP_STAMPS): s_memtime phase stamps of the first blocks go to a buffer
// of their own that nothing else reads (rbd_debug_read_stamps copies it out); no output element is
// ever touched by a stamp.
constexpr int IDS_STAMP_SLOTS = 8, IDS_STAMP_BLOCKS = 4096;
static __device__ unsigned long long ids_stamp_buf[IDS_STAMP_BLOCKS * IDS_STAMP_SLOTS];
#define IDS_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[k] = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define IDS_STAMP(k) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------
// rnea_grad, one configuration per lane, 64 per tile; a block (ONE wave) walks the tiles
// blockIdx.x, blockIdx.x + gridDim.x, ... (the launch sizes the grid to what is resident at once), and
// loads the next tile's q, qd, qdd late in the current tile's backward sweep -- when the registers of
// the bodies already processed are free -- so that no wave waits for its inputs (with two waves per
// SIMD a waiting wave halves the SIMD's issue rate: a lone wave issues one VALU instruction per 4
// cycles).  Same signature and output layout as rnea_grad_kernel<T, HAS_QDD, false>.
// ---------------------------------------------------------------------------------------------
// FDG (one-chain robots only) = forward_dynamics_grad epilogue (RBDReference.py:1376-1384): when the sweep is over and
// every lane holds its finished dc_du rows in registers, the lane reads ITS Minv -- the upper triangle, from the lane-major
// workspace [tile][n (n + 1) / 2][64 lanes] that fd_pre_kernel (rbd_fd_chain.h) filled: one coalesced load per entry -- and
// [qdd_dq | qdd_dqd] = -Minv dc_du replaces the rows column by column, in place, before they leave.  No LDS, no dense
// Minv, no dc_du round trip through HBM.  (Round 3 staged the dense [64][n n] Minv rows through the idle tile and built
// the product in a second set of 2 n^2 registers: fp32 only -- in fp64 that copy alone is 196 VGPRs.)
constexpr int FDC_NP = N * (N + 1) / 2;                       // packed upper triangle of Minv, row-major: (i, j), i <= j
constexpr int fdc_slot(int i, int j) { return i * N - i * (i - 1) / 2 + (j - i); }
constexpr int fdc_sym(int i, int j) { return i <= j ? fdc_slot(i, j) : fdc_slot(j, i); }
// workspace scalars for a batch of B rows: whole tiles (the lanes past the batch hold copies of its last row, which the
// ragged tile of the gradient kernels relies on)
constexpr size_t fdc_ws_scalars(long long B) { return (size_t)((B + 63) / 64) * FDC_NP * 64; }
template <class T, bool HAS_QDD, bool FDG = false>
__global__ __launch_bounds__(64, sizeof(T) == 4 ? 2 : 1) void rnea_grad_idsva_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                                const T* __restrict__ qdd, T grav, int use_damping,
                                                                long long B, T* __restrict__ c_out,
                                                                T* __restrict__ dcdu, const T* __restrict__ minv_in = nullptr) {
  static_assert(!FDG || (grad_max_rows() == N && n_groups() == 1), "fused -Minv epilogue: one-chain robots only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x;
  constexpr int CFGS = 64;
  constexpr int TS = IDS_TS;
  const long long ntiles = (B + CFGS - 1) / CFGS;
  long long t = blockIdx.x;
  if (t >= ntiles) return;
  T* my = tile + lane * TS;

#ifdef RBD_EXP_STAMPS
  unsigned long long stamps[IDS_STAMP_SLOTS] = {};
#endif
  IDS_STAMP(0);
  JTrig<T> tr[N];
  T qv[N], qdv[N], qddv[N];
  T qn[N], qdn[N], qddn[N];                      // inputs in flight: the next group / the next tile
  // global addresses are (uniform tile base: SGPR pair) + (32-bit lane offset): no 64-bit per-lane
  // address is kept in VGPRs across the sweep
  auto load_group = [&](auto G, long long tt, bool dummy = false) {
    constexpr int g = decltype(G)::value;
    const long long c0 = tt * CFGS;
    const long long rm = dummy ? 1 : B - c0;     // dummy: every lane reads row 0 of the tile (84 B instead of 5 KB)
    unsigned lo = (unsigned)((lane < rm ? lane : (int)rm - 1) * N);   // clamped in the last tile
    asm volatile("" : "+v"(lo));
    const T* qt = q + c0 * N + lo;
    const T* qdt = qd + c0 * N + lo;
    const T* qddt = HAS_QDD ? qdd + c0 * N + lo : nullptr;
    sfor<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (grp_has(g, j)) {
        qn[j] = qt[j];
        qdn[j] = qdt[j];
        if constexpr (HAS_QDD) qddn[j] = qddt[j]; else qddn[j] = T(0);
      }
    });
  };
  // The compiler's s_waitcnt bookkeeping treats loads and stores pending on the same counter as
  // unordered: once a flush's stores are in flight, ANY wait for a load becomes vmcnt(0), i.e. a wait for
  // those stores.  So prefetched inputs are "settled" (pulled through an empty asm, which is where the
  // wait lands) BEFORE the next flush is issued, never after it -- and the first tile's before the loop.
  auto settle_group = [&](auto G) {
    constexpr int g = decltype(G)::value;
    sfor<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (grp_has(g, j)) { qn[j] = launder(qn[j]); qdn[j] = launder(qdn[j]); qddn[j] = launder(qddn[j]); }
    });
  };
  load_group(std::integral_constant<int, grp_first()>{}, t);
  settle_group(std::integral_constant<int, grp_first()>{});

  for (; t < ntiles; t += gridDim.x) {
  const long long cfg0 = t * CFGS;
  const long long rem = B - cfg0;
  const int nvalid = rem < CFGS ? (int)rem : CFGS;
  const long long tnext = t + gridDim.x;
  const bool has_next = tnext < ntiles;

  T Sv[N][6], Pd[N][6], Pdd[N][6];
  T cv[N];
  sfor<0, N>([&](auto Rt) {
   constexpr int rt = decltype(Rt)::value;
   if constexpr (grp_head(rt)) {
    constexpr int row0 = grp_row0(rt);
    constexpr int rows = grp_rows(rt);
    constexpr int last = row0 + rows - 1;
    constexpr int ra = ids_rows_a(rows);                    // top bodies: their entries are parked in LDS
    constexpr int rs = row0 + rows - ra;                     // first of them (== row0 + rows when ra == 0)
    constexpr int PEND = ra * GRAD_ROW;                      // tile offset of the pending column entries
    constexpr int RW = rows * GRAD_ROW;                      // finished scalars of this group per configuration
    T E[RW];                                                 // the group's rows, final layout [(r - row0) * 2n + c]
    // the group's inputs have arrived in qn, qdn, qddn (prefetched)
    sfor<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (grp_has(rt, j)) { qv[j] = qn[j]; qdv[j] = qdn[j]; qddv[j] = qddn[j]; }
    });
    IDS_STAMP(1);
    sfor<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (grp_has(rt, j)) tr[j] = make_trig<j>(qv[j]);
      // fp64: one joint's sin / cos at a time (interleaved, the seven polynomial chains cost 30-110 spilled registers)
      if constexpr (sizeof(T) == 8) __builtin_amdgcn_sched_barrier(0);
    });
    IDS_STAMP(2);
    // issues the loads of what comes next: the next group of this tile, else the first group of the next tile
    auto prefetch = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (grp_next(rt) >= 0) {
        load_group(std::integral_constant<int, grp_next(rt) >= 0 ? grp_next(rt) : 0>{}, t);
      } else {
        // unconditional (a block's last tile re-reads one row of its own): a branch here would end in
        // register copies of the loaded values, i.e. in a wait for the loads right behind their issue
        load_group(std::integral_constant<int, grp_first()>{}, has_next ? tnext : t, !has_next);
      }
      __builtin_amdgcn_sched_barrier(0);
    };

    // ---- forward: world kinematics of the chain (:1413-1434) ------------------------------------
    T Rm[3][3], pw[3], v[6], a[6];          // state of the current body: R (body -> world), origin, v, a
    sfor<row0, row0 + rows>([&](auto J) {
      if constexpr (sizeof(T) == 8) __builtin_amdgcn_sched_barrier(0);   // fp64: body steps are not interleaved (register peak)
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
      if constexpr (j > row0 && j < last) {
        // cold until the backward sweep steps back over body j: park {sin, cos, qd, qdd}
        my[ids_park_slot(row0, rows, ra, j, 0)] = tr[j].s; my[ids_park_slot(row0, rows, ra, j, 1)] = tr[j].c;
        my[ids_park_slot(row0, rows, ra, j, 2)] = qdv[j]; my[ids_park_slot(row0, rows, ra, j, 3)] = qddv[j];
      }
      if constexpr (j == last && ids_leaf_parks(row0, rows, ra)) {
        // the leaf's own four are cold while ITS backward step runs (the register peak): they wait in
        // the row above (columns 2, 3, N + 2, N + 3, written by that row's body only, i.e. later)
        my[ids_leaf_slot(row0, rows, ra, 0)] = tr[j].s; my[ids_leaf_slot(row0, rows, ra, 1)] = tr[j].c;
        my[ids_leaf_slot(row0, rows, ra, 2)] = qdv[j]; my[ids_leaf_slot(row0, rows, ra, 3)] = qddv[j];
      }
    });

    IDS_STAMP(3);
    // ---- backward: local inertia terms, composites, t-vectors, all pairs of the body -------------
    RInertia<T> IC;
    SymB<T> SC;
    T pmC[6], fC[6];
    sfor_down<row0, row0 + rows>([&](auto J) {
      if constexpr (sizeof(T) == 8) __builtin_amdgcn_sched_barrier(0);   // fp64: body steps are not interleaved (register peak)
      constexpr int j = decltype(J)::value;
      constexpr int k = AXIS[j], ka = (k + 1) % 3, kb = (k + 2) % 3;
      // tile slot of entry (row r, column c) while body j is being processed: bodies >= rs write the
      // top rows at the front of the tile and park what they contribute to lower rows behind them
      constexpr bool top = j >= rs;         // this body's entries go to LDS (else: registers)
      if constexpr (!FDG && j == rs && ra > 1 && ids_c_parks(row0, rows, ra)) {
        asm volatile("" ::: "memory");
        sfor<rs + 1, row0 + rows>([&](auto JC) { constexpr int jc = decltype(JC)::value; cv[jc] = my[ids_c_slot(row0, rows, ra, jc - rs - 1)]; });
      }
      if constexpr (j > row0 && j < last) {
        asm volatile("" ::: "memory");                       // a real LDS read-back, not a forwarded register
        tr[j].s = my[ids_park_slot(row0, rows, ra, j, 0)]; tr[j].c = my[ids_park_slot(row0, rows, ra, j, 1)];
        qdv[j] = my[ids_park_slot(row0, rows, ra, j, 2)]; qddv[j] = my[ids_park_slot(row0, rows, ra, j, 3)];
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
      if constexpr (!FDG) cv[j] = dot6(Sv[j], fC);   // c_j (one value per body: kept, leaves with the tile; the -Minv variant has no c output)
      if constexpr (!FDG && j > rs && ids_c_parks(row0, rows, ra)) {
        // ... except during the register peak: the bodies above the lowest top body leave theirs in
        // that body's (still empty) row and take them back when it starts
        my[ids_c_slot(row0, rows, ra, j - rs - 1)] = cv[j];
      }
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
      // structural zeros of row j: bodies of other groups
      sfor<0, N>([&](auto C_) {
        constexpr int c = decltype(C_)::value;
        if constexpr (!grp_has(rt, c)) {
          if constexpr (top) { my[(j - rs) * GRAD_ROW + c] = T(0); my[(j - rs) * GRAD_ROW + N + c] = T(0); }
          else { E[(j - row0) * GRAD_ROW + c] = T(0); E[(j - row0) * GRAD_ROW + N + c] = T(0); }
        }
      });
      // inputs of what comes next: issued late in the sweep, when the registers of the bodies already
      // processed are free (bodies and the epilogue still lie between these loads and their first use)
      // (fp64, one wave per SIMD at 512 VGPRs: the 3 n values in flight are 6 n registers -- requested two bodies later)
      constexpr int PF_AT = sizeof(T) == 8 ? IDS_PREFETCH_AT_F64 : IDS_PREFETCH_AT;
      if constexpr (!FDG && j == row0 + (rows > PF_AT ? PF_AT : rows - 1)) prefetch();
      if constexpr (j == last && ids_leaf_parks(row0, rows, ra)) {
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

    IDS_STAMP(4);
    // ---- the sweep is over: everything else is dead, pull the parked entries into registers ---------
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
    // the prefetched inputs must have landed before this tile's stores are issued (see settle_group)
    if constexpr (!FDG) settle_group(std::integral_constant<int, grp_next(rt) >= 0 ? grp_next(rt) : grp_first()>{});
    IDS_WAVE_SYNC();                                       // every lane has left the tile

    if constexpr (FDG) {
      T mk[FDC_NP];
      {
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const T* mp = minv_in + (size_t)t * (FDC_NP * 64) + lo;
        sfor<0, FDC_NP>([&](auto S_) { constexpr int s_ = decltype(S_)::value; mk[s_] = mp[s_ * 64]; });
      }
      // E <- -Minv E (:1381-1383), column by column in place; Minv is symmetric (:799-804)
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
      // this variant requests the next inputs behind its epilogue; they are waited for at the top of the next tile
      // (together with this tile's stores)
      prefetch();
    }

    // ---- the rows leave, 32 configurations at a time, through a compact [32][RW] image ----------------
    int ln = lane;
    asm volatile("" : "+v"(ln));       // (everything derived from the lane id below is computed here, not hoisted out of the tile loop and kept)
    {
      T* gbase = dcdu + cfg0 * GRAD_TILE + row0 * GRAD_ROW;
      constexpr int VE = 16 / sizeof(T);
      constexpr bool FLAT = RW == GRAD_TILE && (32 * RW) % VE == 0 && RW % (8 / (int)sizeof(T)) == 0;
      sfor<0, 2>([&](auto H_) {
        constexpr int h = decltype(H_)::value;
        if ((ln >> 5) == h) {
          T* img = tile + (ln & 31) * RW;
          if constexpr (RW % 2 == 0 && sizeof(T) == 4) {
            typedef T V2 __attribute__((ext_vector_type(2)));
            sfor<0, RW / 2>([&](auto K_) {
              constexpr int k = decltype(K_)::value;
              V2 x; x[0] = E[2 * k]; x[1] = E[2 * k + 1];
              reinterpret_cast<V2*>(img)[k] = x;
            });
          } else {
            sfor<0, RW>([&](auto K_) { constexpr int k = decltype(K_)::value; img[k] = E[k]; });
          }
        }
        IDS_WAVE_SYNC();
        const int nv = nvalid - 32 * h;                    // configurations of this half that exist
        T* gdst = gbase + (long long)(32 * h) * GRAD_TILE;
        bool done = false;
        if constexpr (FLAT) {
          if (nv >= 32) {                                  // LDS image == HBM image: flat 16-byte copies
            typedef T V __attribute__((ext_vector_type(VE)));
            const V* src = reinterpret_cast<const V*>(tile);
            V* dst = reinterpret_cast<V*>(gdst);
            const int g0 = ln;
            constexpr int NV = 32 * RW / VE;
            sfor<0, (NV + 63) / 64>([&](auto I_) {
              constexpr int i = decltype(I_)::value;
              if constexpr ((i + 1) * 64 <= NV) dst[g0 + 64 * i] = src[g0 + 64 * i];
              else { if (g0 + 64 * i < NV) dst[g0 + 64 * i] = src[g0 + 64 * i]; }
            });
            done = true;
          }
        }
        if (!done && nv > 0) {                             // ragged last tile, or rows of one group among several
          const int nvc = nv < 32 ? nv : 32;
#pragma unroll 2
          for (int g = ln; g < nvc * RW; g += 64) {
            const int cfg = g / RW;
            const int r2 = g - cfg * RW;
            gdst[(long long)cfg * GRAD_TILE + r2] = tile[cfg * RW + r2];
          }
        }
        IDS_WAVE_SYNC();                                   // the image has been read: next half / next tile may write
      });
    }
    // ---- c [64][n] leaves the same way once the last group is done (a 64 x n image, flat stores) ------
    if constexpr (!FDG && grp_next(rt) < 0) {
      if (c_out != nullptr) {
        sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; tile[ln * N + j] = cv[j]; });
        IDS_WAVE_SYNC();
        T* cdst = c_out + cfg0 * N;
        constexpr int VE = 16 / sizeof(T);
        if (nvalid == CFGS && (CFGS * N) % VE == 0) {
          typedef T V __attribute__((ext_vector_type(VE)));
          const int g0 = ln;
          constexpr int NV = CFGS * N / VE;
          sfor<0, (NV + 63) / 64>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            if (g0 + 64 * i < NV) reinterpret_cast<V*>(cdst)[g0 + 64 * i] = reinterpret_cast<const V*>(tile)[g0 + 64 * i];
          });
        } else {
          for (int g = ln; g < nvalid * N; g += 64) cdst[g] = tile[g];
        }
        IDS_WAVE_SYNC();
      }
    }
#ifdef RBD_EXP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IDS_STAMP(5);
    if (lane == 0 && t < IDS_STAMP_BLOCKS && t == blockIdx.x) {
      for (int kk = 0; kk < IDS_STAMP_SLOTS; ++kk) ids_stamp_buf[t * IDS_STAMP_SLOTS + kk] = stamps[kk];
    }
#endif
   }
  });
  }  // tiles
}

}  // namespace rbdk
