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

#ifndef RBD_TREE_NO_BUF_FLUSH
#define RBD_TREE_NO_BUF_FLUSH 0
#endif
namespace rbdk {

// ---- LDS plan (scalars of T) -----------------------------------------------------------------------
// [0, 64 * TREE_KP)                      row image, 64 configurations x 2n (+ pad), float2-granular
// then lane-private columns (slot * 64 + lane):
//   TREE_INCH  in-chain pending column entries, reused by every chain      2 * L (L - 1) / 2
//   TREE_CROSS cross-chain pending column entries, live until consumed     2 * n_cross_pairs
//   TREE_PARK  composites of finished chains                               31 * n_parked_chains
// row stride: odd in units of the flush vector (float4 when n is even, else float2) => conflict-free
constexpr int TREE_KP = (N % 2 == 0) ? 4 * ((N / 2) | 1) : 2 * N;
constexpr int TREE_KP2 = TREE_KP / 2;
constexpr int TREE_COMP = TREE_COMP_SCALARS;
constexpr int TREE_INCH = 0;
// Robots with a side subtree (Atlas' right arm next to the heavy path back -> left arm): ONE block per 64
// configurations, one wave per root subtree plus one per side subtree, each with its own row image and its own
// in-chain pending region (sized by its longest chain).  Other robots: one single-wave block per (64 configurations,
// root), blockIdx.y = root, as before.
constexpr int TREE_MULTI_SCALARS = TP.n_waves * TREE_KP + TP.inch_off[TP.n_waves < 16 ? TP.n_waves : 16] + 2 * TP.n_cross + 31 * TP.n_park;
constexpr bool TREE_MULTI = TP.any_side && TP.n_waves <= 8 && (size_t)64 * TREE_MULTI_SCALARS * (N <= 12 ? 8 : 4) <= 156 * 1024;
constexpr int TREE_W = TREE_MULTI ? TP.n_waves : 1;
// The multi-wave kernel has ONE block barrier per wave, reached at different program points: a root's main wave
// reaches it when it gets to the side subtree's parent, every other wave at the end of the kernel.  That is a
// barrier in wave-divergent control flow; it is sound on gfx9 because s_barrier only COUNTS arrivals, and only if
// every wave arrives exactly once -- i.e. the side subtree's parent is a body of the main wave's own heavy chain
// (each body of a chain is processed exactly once).  Checked here for the robot at hand rather than assumed.
constexpr bool tree_barrier_plan_ok() {
  for (int r = 0; r < N; ++r) {
    if (PARENT[r] >= 0) continue;
    const int sh = TP.side_head[TP.rootidx[r]];
    if (sh < 0) continue;
    const int p = PARENT[sh];
    if (p < 0 || TP.on_side[p]) return false;          // the parent is a body of the MAIN wave ...
    if (TP.rootidx[p] != TP.rootidx[r]) return false;  // ... of the same root ...
    if (TP.head[p] != r) return false;                 // ... on the root's heavy chain
    if (TP.wave_of[p] == TP.wave_of[sh]) return false; // and the side subtree runs on another wave
  }
  // subtrees handed to another wave for load balance (TreePlan): consumed by the main wave after its barrier only
  for (int x = 0; x < N; ++x) {
    int rt = x;
    while (PARENT[rt] >= 0) rt = PARENT[rt];
    if (TP.on_side[x] || TP.wave_of[x] == TP.wave_of[rt]) continue;
    int top = x;                                          // head of the handed-over subtree
    while (PARENT[top] >= 0 && TP.wave_of[PARENT[top]] == TP.wave_of[x] && !(TP.head[PARENT[top]] == rt)) top = PARENT[top];
    const int p = PARENT[top], sh = TP.side_head[TP.rootidx[rt]];
    if (p < 0 || sh < 0 || TP.head[p] != rt || TP.wave_of[p] != TP.wave_of[rt]) return false;
    bool above = false;
    for (int y = PARENT[sh]; y >= 0; y = PARENT[y]) above = above || y == p;
    if (!above) return false;
  }
  return true;
}
static_assert(!TREE_MULTI || tree_barrier_plan_ok(), "multi-wave tree kernel: a wave would pass its block barrier zero or two times");
constexpr int TREE_INCH_TOTAL = TREE_MULTI ? TP.inch_off[TP.n_waves] : max_chain_len() * (max_chain_len() - 1);
constexpr int TREE_CROSS = TREE_INCH + TREE_INCH_TOTAL;
constexpr int TREE_PARK = TREE_CROSS + 2 * n_cross_pairs();
constexpr int TREE_PRIV = TREE_PARK + TREE_COMP * n_parked_chains();
// ---- two rows at a time ------------------------------------------------------------------------------------------------------
// A row of dc_du is 8n bytes (240 B at n = 30) at 16-byte alignment: written alone it touches 4-5 granules of 64 bytes (1.20 x
// its bytes at the L2's memory side), and beyond the last-level cache the write path, not the arithmetic, sets the kernel's
// time (every block writing ONE block's region: 97.6 instead of 185 us at B = 65 536).  Consecutive bodies of a chain
// (j + 1, j with PARENT[j + 1] == j) own ADJACENT rows: the first waits in 2n registers (one wave per SIMD: 512 registers) and
// the pair leaves as one 16n-byte piece per configuration, two configurations per store instruction, through a
// [32][2 x 2n (+ pad)] image, 32 configurations at a time.
#ifndef RBD_TREE_PAIR
#define RBD_TREE_PAIR 0
#endif
template <class T>
constexpr bool tree_pair_ok() { return RBD_TREE_PAIR && sizeof(T) == 4 && (N % 2 == 0) && (TREE_KP % 4 == 0) && !RBD_TREE_NO_BUF_FLUSH && N <= 32; }
constexpr int TREE_PKP = 4 * (N | 1);                         // scalars per configuration of the pair image: N 16-byte pieces + pad (odd count)
template <class T>
constexpr int tree_img_scalars() {                            // one wave's image
  return tree_pair_ok<T>() && 32 * TREE_PKP > 64 * TREE_KP ? 32 * TREE_PKP : 64 * TREE_KP;
}
// role of body j's row: 1 = first of a pair (kept in registers), 2 = second (the pair leaves), 0 = alone
constexpr int tree_pair_role(int j) {
  const int h = chain_head_of(j);
  int k = chain_leaf(h);
  for (int guard = 0; guard < N + 1; ++guard) {
    if (k == h) return 0;
    const int p = PARENT[k];
    if (p == k - 1 && in_chain(p, h)) {
      if (j == k) return 1;
      if (j == p) return 2;
      if (p == h) return 0;
      k = PARENT[p];
    } else {
      if (j == k) return 0;
      k = p;
    }
  }
  return 0;
}
template <class T>
constexpr size_t tree_lds_bytes() { return sizeof(T) * ((size_t)TREE_W * tree_img_scalars<T>() + (size_t)64 * TREE_PRIV); }

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

template <class T, bool HAS_QDD>
__global__ __launch_bounds__(64 * TREE_W, 1) void rnea_grad_tree_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                               const T* __restrict__ qdd, T grav, int use_damping,
                                                               long long B, T* __restrict__ c_out, T* __restrict__ dcdu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  T* rowimg = reinterpret_cast<T*>(smem_raw) + wave * tree_img_scalars<T>();   // [64][TREE_KP] (or the pair image [32][TREE_PKP]), one per wave
  T* priv = reinterpret_cast<T*>(smem_raw) + TREE_W * tree_img_scalars<T>() + lane;   // priv[slot * 64], shared by the waves (same lane = same configuration)
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
  // full tiles store through a descriptor of the block's 64 matrices (rbd_world.h: flush_image_rows_buf)
  const __amdgpu_buffer_rsrc_t out_rs = out_tile_rsrc(dcdu + cfg0 * N * (2 * N), 64 * N * 2 * N * (int)sizeof(T));

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
      constexpr bool PAIR = tree_pair_ok<T>();
      T rowA[PAIR ? 2 * N : 1];                               // the first row of a pair waits here for its neighbour
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
          tvectors<(PARENT[j] < 0)>(C, Sv[j], Pd[j], Pdd[j], t1, t2, t3, t4);
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
                dq = dot6_acc(t1, Pdd[c], dot6(t4, Pd[c]));
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
          // image of a row for the 64 configurations, then 8n-byte segments to dc_du[b][row][:].  The
          // block is ONE wave and a wave's LDS operations execute in order, so a wave-level fence
          // (compiler ordering) is all the image needs -- no s_barrier, no wait on the previous
          // row's global stores.
          auto emit_single = [&](auto JR_, const T (&rw)[2 * N]) {
            constexpr int jr = decltype(JR_)::value;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
              V2* mine = reinterpret_cast<V2*>(rowimg) + lane * TREE_KP2;
              sfor<0, N>([&](auto E_) {
                constexpr int e = decltype(E_)::value;
                V2 x; x[0] = rw[2 * e]; x[1] = rw[2 * e + 1];
                mine[e] = x;
              });
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#ifdef RBD_TREE_EXP_NOFLUSH      // timing experiment: rows stay in the image (results are wrong)
            if (factive && use_damping == 12345) {
#else
            if (factive) {
#endif
              if constexpr (WIDE && sizeof(T) == 4 && !RBD_TREE_NO_BUF_FLUSH) {
                if (nvalid == 64) {
                  flush_image_rows_buf<CPI, TREE_KP / 4>(reinterpret_cast<const V4*>(rowimg), out_rs, (fsub * (N * N / 2) + fe) * 16,
                                                        jr * 2 * N * (int)sizeof(T), N * 2 * N * (int)sizeof(T), fsub, fe);
                } else {
                  flush_image_rows<CPI, TREE_KP / 4, false>(reinterpret_cast<const V4*>(rowimg), reinterpret_cast<V4*>(dcdu + (cfg0 * N + jr) * (2 * N)),
                                                            (long long)(N * N / 2), fsub, fe, nvalid);
                }
              } else if constexpr (WIDE) {
#ifdef RBD_TREE_EXP_L2ONLY       // timing experiment: every block writes block 0's region (stays in L2; results are wrong)
                flush_image_rows<CPI, TREE_KP / 4>(reinterpret_cast<const V4*>(rowimg), reinterpret_cast<V4*>(dcdu + (0 * N + jr) * (2 * N)),
#else
                flush_image_rows<CPI, TREE_KP / 4>(reinterpret_cast<const V4*>(rowimg), reinterpret_cast<V4*>(dcdu + (cfg0 * N + jr) * (2 * N)),
#endif
                                                   (long long)(N * N / 2), fsub, fe, nvalid);
              } else {
                flush_image_rows<CPI, TREE_KP2>(reinterpret_cast<const V2*>(rowimg), reinterpret_cast<V2*>(dcdu + (cfg0 * N + jr) * (2 * N)),
                                                (long long)(N * N), fsub, fe, nvalid);
              }
            }
          };
          constexpr int role = PAIR ? tree_pair_role(j) : 0;
          if constexpr (role == 1) {
            sfor<0, 2 * N>([&](auto E_) { constexpr int e = decltype(E_)::value; rowA[e] = row[e]; });
          } else if constexpr (role == 2) {
            if (nvalid == 64) {
              // rows j (this one) and j + 1 (rowA) of 32 configurations at a time: [32][N 16-byte pieces (+ 1 pad)], row j first;
              // a store instruction = TWO whole pairs (lanes 0 .. 2N-1: configuration 2 it + lane / N, piece lane % N)
              constexpr int PV = TREE_PKP / 4;                          // image stride in 16-byte pieces (odd: conflict-free)
              const int psub = lane / N, pe = lane - psub * N;          // (N <= 32: two configurations per instruction)
              sfor<0, 2>([&](auto H_) {
                constexpr int hh = decltype(H_)::value;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if ((lane >> 5) == hh) {
                  V4* mine = reinterpret_cast<V4*>(rowimg) + (lane & 31) * PV;
                  sfor<0, N / 2>([&](auto E_) {
                    constexpr int e = decltype(E_)::value;
                    V4 x; x[0] = row[4 * e]; x[1] = row[4 * e + 1]; x[2] = row[4 * e + 2]; x[3] = row[4 * e + 3];
                    mine[e] = x;
                    V4 y; y[0] = rowA[4 * e]; y[1] = rowA[4 * e + 1]; y[2] = rowA[4 * e + 2]; y[3] = rowA[4 * e + 3];
                    mine[N / 2 + e] = y;
                  });
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane < 2 * N) {
                  typedef unsigned U4 __attribute__((ext_vector_type(4)));
                  typedef float F4 __attribute__((ext_vector_type(4)));      // (the pair path is fp32 only; F4 keeps this branch well-formed for T = double)
                  F4 buf[16];
                  sfor<0, 16>([&](auto I_) { constexpr int it = decltype(I_)::value; buf[it] = reinterpret_cast<const F4*>(rowimg)[(2 * it + psub) * PV + pe]; });
                  const int voff = (psub * (N * N / 2) + pe) * 16;
                  sfor<0, 16>([&](auto I_) {
                    constexpr int it = decltype(I_)::value;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(U4, buf[it]), out_rs, voff,
                                                           ((32 * hh + 2 * it) * N + j) * 2 * N * (int)sizeof(T), 0);
                  });
                }
              });
            } else {       // a ragged tile: the two rows one after the other
              emit_single(std::integral_constant<int, j + 1>{}, rowA);
              emit_single(std::integral_constant<int, j>{}, row);
            }
          } else {
            emit_single(std::integral_constant<int, j>{}, row);
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
