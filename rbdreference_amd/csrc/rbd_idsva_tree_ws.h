// rbd_idsva_tree_ws.h -- rnea_grad for BIG trees in fp64: the chain-by-chain world-frame kernel of
// rbd_idsva_tree.h with its lane-private state in a global workspace instead of registers and LDS.
//
// Result = RBDReference.rnea_grad (/root/reference/RBDReference.py:1345-1368); identities and chain order as in
// rbd_idsva_tree.h (first-order part of :1413-1484).  What differs, and why: in fp64 a 30-body tree does not
// fit the fp32 kernel's plan -- the root path's S / psid / psidd are 36 registers per body (ten bodies deep: 360
// of 512 before anything else is live), a 2n-entry row 120, and the lane-private LDS columns (pending column
// entries, parked composites) would be 155 KB next to 123 KB of row images.  Here
//   * S, psid, psidd, sin q, cos q, qd, qdd of a body deeper than TWS_DREG are written to the workspace by the
//     downward kinematic sweep and read back where they are used (the body's own step; one read per
//     (body, ancestor) pair), bodies near the root keep them in registers;
//   * pending column entries and parked composites live in the workspace too (same slot plan as the LDS
//     columns of the fp32 kernel);
//   * a row is written to the wave's LDS image entry by entry as it is formed, never held in registers;
//   * LDS holds the row images only.
// The workspace is [block of the launch][slot][64 lanes] (consecutive lanes = consecutive addresses: every access is
// one contiguous 512-byte piece per wave; a block's region is 64 * TWS_SLOTS scalars, addressed through a buffer
// descriptor) and is private to a lane, except across the ONE block barrier of the multi-wave layout (side subtree
// -> main wave), as in the fp32 kernel.  It belongs to the library (one buffer per stream, rbd_stream_workspace); a
// launch covers at most the blocks that are resident at once and the launcher walks larger batches chunk by chunk on
// the same stream, so its size does not grow with the batch (Atlas: 256 blocks x 492 KB = 126 MB).
#pragma once
#include "rbd_idsva_tree.h"

namespace rbdk {

#ifndef RBD_TWS_DREG
#define RBD_TWS_DREG 4
#endif
constexpr int TWS_DREG = RBD_TWS_DREG;                       // bodies with DEPTH < TWS_DREG keep their vectors in registers
constexpr bool tws_in_regs(int j) { return DEPTH[j] < TWS_DREG; }
constexpr int TWS_PATH = 22;                                  // S, psid, psidd (18), sin, cos, qd, qdd
constexpr int TWS_PATH_SLOTS = TWS_PATH * N;
constexpr int TWS_E_SLOTS = TREE_PRIV;                        // pending entries + parked composites (tree_pend_slot, TREE_PARK)
constexpr int TWS_SLOTS = TWS_PATH_SLOTS + TWS_E_SLOTS;
constexpr bool TWS_MULTI = TREE_MULTI;                        // the slot plan of the in-chain region follows TREE_MULTI
constexpr int TWS_W = TWS_MULTI ? TP.n_waves : 1;
template <class T>
constexpr size_t tws_lds_bytes() { return sizeof(T) * (size_t)64 * TWS_W * TREE_KP; }
template <class T>
constexpr bool tws_ok() { return GRAD_TREE_OK && sizeof(T) == 8 && tws_lds_bytes<T>() <= 156 * 1024; }

// one block's workspace region as a raw buffer: [slot][64 lanes] of T
template <class T>
struct TwsBuf {
  __amdgpu_buffer_rsrc_t rs;
  int voff;
  RBD_DEV TwsBuf(T* base, int lane) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                           (int)(TWS_SLOTS * 64 * sizeof(T)), 0x00020000);
    voff = lane * (int)sizeof(T);
  }
};
template <class T>
struct TwsRef {   // pw(slot) = x;  x = pw(slot);
  const TwsBuf<T>& b;
  int slot;
  RBD_DEV operator T() const {
    static_assert(sizeof(T) == 8, "workspace kernel: fp64 only");
    typedef unsigned U2 __attribute__((ext_vector_type(2)));
#ifdef RBD_TWS_EXP_NOLOAD        // timing experiment: no workspace reads (results are wrong)
    T r;
    asm volatile("; no load" : "=v"(r) : "v"(b.voff));   // stays where the load was: volatile, ordered with the clobbers
    return r;
#else
    const U2 v = __builtin_amdgcn_raw_buffer_load_b64(b.rs, b.voff, slot * 64 * (int)sizeof(T), 0);
    return __builtin_bit_cast(T, v);
#endif
  }
  RBD_DEV void operator=(T x) const {
    typedef unsigned U2 __attribute__((ext_vector_type(2)));
#ifdef RBD_TWS_EXP_NOSTORE       // timing experiment: no workspace writes (results are wrong)
    asm volatile("" :: "v"(x));
#else
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(U2, x), b.rs, b.voff, slot * 64 * (int)sizeof(T), 0);
#endif
  }
};

#ifndef RBD_TWS_MINBLOCKS                                      // experiments: blocks per CU the register budget must allow
#define RBD_TWS_MINBLOCKS 1
#endif
template <class T, bool HAS_QDD>
__global__ __launch_bounds__(64 * TWS_W, RBD_TWS_MINBLOCKS) void rnea_grad_tree_ws_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                                  const T* __restrict__ qdd, T grav, int use_damping,
                                                                  long long B, T* __restrict__ c_out, T* __restrict__ dcdu,
                                                                  T* __restrict__ pws, T* __restrict__ ews) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  T* rowimg = reinterpret_cast<T*>(smem_raw) + wave * (64 * TREE_KP);   // [64][TREE_KP], one image per wave
  const long long cfg0 = (long long)blockIdx.x * 64;
  const long long rem = B - cfg0;
  const int nvalid = rem < 64 ? (int)rem : 64;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);
  const T* qrow = q + b * N;
  const T* qdrow = qd + b * N;
  const T* qddrow = HAS_QDD ? qdd + b * N : nullptr;
  // workspace of this block: [slot][64 lanes], compile-time slot offsets (lanes beyond the batch compute the last row
  // again, in their own column)
  // Buffer addressing: the block's base sits in a descriptor (four SGPRs), the lane's byte offset in ONE VGPR, the
  // slot's offset is a constant of the instruction stream.  (Plain pointers became 64-bit per-lane addresses, one
  // live register pair per eight slots: 600 spilled registers.)
  // one region per BLOCK: in the single-wave layout the grid is (x, root) and the blocks (x, 0), (x, 1), ... of a
  // multi-root robot run concurrently on the same lanes' slots (tree_pend_slot's in-chain region starts at 0 for every
  // chain) -- they must not share a region (they did, indexed by blockIdx.x alone)
  const size_t wblk = TWS_MULTI ? (size_t)blockIdx.x : (size_t)blockIdx.x * gridDim.y + blockIdx.y;
  const TwsBuf<T> PWB(pws + wblk * (TWS_SLOTS * 64), lane);
  const TwsBuf<T> EWB(ews + wblk * (TWS_SLOTS * 64), lane);
  auto PW = [&](int slot) { return TwsRef<T>{PWB, slot}; };
  auto EW = [&](int slot) { return TwsRef<T>{EWB, slot}; };

  typedef T V2 __attribute__((ext_vector_type(2)));
  typedef T V4 __attribute__((ext_vector_type(4)));
  constexpr bool WIDE = (N % 2 == 0) && (TREE_KP % 4 == 0);
  constexpr int FW = WIDE ? N / 2 : N;                    // vectors per row
  constexpr int CPI = 64 / FW > 0 ? 64 / FW : 1;          // configurations per flush step
  const int fsub = lane / FW, fe = lane - fsub * FW;
  const bool factive = lane < CPI * FW;
  const int myroot = blockIdx.y;                          // (single-wave layout) independent root subtrees run in separate blocks
  T* myrow = rowimg + lane * TREE_KP;

  sfor_down<0, N>([&](auto H_) {
    constexpr int h = decltype(H_)::value;
    if constexpr (is_chain_head(h)) {
     constexpr int hroot = TP.rootidx[h];
     constexpr int hwave = TWS_MULTI ? TP.wave_of[h] : 0;
     constexpr int side = TWS_MULTI ? TP.side_head[hroot] : -1;
     if (TWS_MULTI ? (wave == hwave) : (hroot == myroot)) {
      constexpr int leaf = chain_leaf(h);
      // ---- inputs of the root path, all loads first -------------------------------------------------------
      T qv[N], qdv[N], qddv[N];
      sfor<0, N>([&](auto J_) {
        constexpr int j = decltype(J_)::value;
        if constexpr (is_anc_or_self(j, leaf)) {
          qv[j] = qrow[j];
          qdv[j] = qdrow[j];
          if constexpr (HAS_QDD) qddv[j] = qddrow[j]; else qddv[j] = T(0);
        }
      });
      // ---- world kinematics root -> leaf (:1413-1434); deep bodies leave their vectors in the workspace ----
      WState<T> s;
      JTrig<T> trr[N];
      T Sv[N][6], Pd[N][6], Pdd[N][6];   // used at register-resident bodies only
      sfor<0, N>([&](auto J_) {
        constexpr int j = decltype(J_)::value;
        if constexpr (is_anc_or_self(j, leaf)) {
          const JTrig<T> g = make_trig<j>(qv[j]);
          if constexpr (tws_in_regs(j)) {
            trr[j] = g;
            ws_down<j>(s, g, qdv[j], qddv[j], grav, Sv[j], Pd[j], Pdd[j]);
          } else {
            T a[6], bb[6], cc[6];
            ws_down<j>(s, g, qdv[j], qddv[j], grav, a, bb, cc);
            sfor<0, 6>([&](auto R_) {
              constexpr int r = decltype(R_)::value;
              PW(TWS_PATH * j + r) = a[r];
              PW(TWS_PATH * j + 6 + r) = bb[r];
              PW(TWS_PATH * j + 12 + r) = cc[r];
            });
            PW(TWS_PATH * j + 18) = g.s;
            PW(TWS_PATH * j + 19) = g.c;
            PW(TWS_PATH * j + 20) = qdv[j];
            PW(TWS_PATH * j + 21) = qddv[j];
          }
        }
      });
      // ---- leaf -> head ---------------------------------------------------------------------------------
      Comp<T> C;
      sfor_down<0, N>([&](auto J_) {
        constexpr int j = decltype(J_)::value;
        if constexpr (in_chain(j, h)) {
          // the body's own vectors
          T Sj[6], Pdj[6], Pddj[6], qdj, qddj;
          JTrig<T> gj;
          if constexpr (tws_in_regs(j)) {
            sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; Sj[r] = Sv[j][r]; Pdj[r] = Pd[j][r]; Pddj[r] = Pdd[j][r]; });
            gj = trr[j]; qdj = qdv[j]; qddj = qddv[j];
          } else {
            sfor<0, 6>([&](auto R_) {
              constexpr int r = decltype(R_)::value;
              Sj[r] = PW(TWS_PATH * j + r);
              Pdj[r] = PW(TWS_PATH * j + 6 + r);
              Pddj[r] = PW(TWS_PATH * j + 12 + r);
            });
            gj.s = PW(TWS_PATH * j + 18);
            gj.c = PW(TWS_PATH * j + 19);
            qdj = PW(TWS_PATH * j + 20);
            qddj = PW(TWS_PATH * j + 21);
          }
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
              comp_each(P, [&](auto I_, T& x) { x = EW(TREE_PARK + TREE_COMP * park_rank(kk) + decltype(I_)::value); });
              comp_add(C, P);
            }
          });
          const T cj = dot6(Sj, C.f);
          if (c_out != nullptr && lane < nvalid) c_out[b * N + j] = cj;
          // t-vectors (:1481-1484)
          T t1[6], t2[6], t3[6], t4[6];
          tvectors<(PARENT[j] < 0)>(C, Sj, Pdj, Pddj, t1, t2, t3, t4);
          // ---- row j, straight into the image (the wave's LDS operations execute in order: the previous row's
          //      flush reads are ahead of these writes) ---------------------------------------------------------
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          // descendants' column entries (delivered earlier) and structural zeros
          sfor<0, N>([&](auto C_) {
            constexpr int c = decltype(C_)::value;
            if constexpr (c != j && is_anc_or_self(j, c)) {
              myrow[c] = EW(tree_pend_slot(j, c));
              myrow[N + c] = EW(tree_pend_slot(j, c) + 1);
            } else if constexpr (c != j && !is_anc_or_self(c, j)) {
              myrow[c] = T(0);
              myrow[N + c] = T(0);
            }
          });
          // ancestors root -> parent, then the body itself; a workspace-resident ancestor's vectors are requested one
          // pair ahead of use (the memory clobbers keep the compiler from requesting all of them at once)
          constexpr int D = DEPTH[j];
          T abuf[2][18];
          auto request = [&](auto K_) {
            constexpr int k = decltype(K_)::value;
            constexpr int c = anc_at(j, k);
            if constexpr (k < D && !tws_in_regs(c)) {
              sfor<0, 18>([&](auto R_) { constexpr int r = decltype(R_)::value; abuf[k & 1][r] = PW(TWS_PATH * c + r); });
            }
          };
          request(std::integral_constant<int, 0>{});
          sfor<0, D + 1>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            constexpr int c = anc_at(j, k);
            if constexpr (k + 1 < D) request(std::integral_constant<int, (k + 1 < D ? k + 1 : 0)>{});
            T Sc[6], Pdc[6], Pddc[6];
            if constexpr (k == D) {
              sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; Sc[r] = Sj[r]; Pdc[r] = Pdj[r]; Pddc[r] = Pddj[r]; });
            } else if constexpr (tws_in_regs(c)) {
              sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; Sc[r] = Sv[c][r]; Pdc[r] = Pd[c][r]; Pddc[r] = Pdd[c][r]; });
            } else {
              sfor<0, 6>([&](auto R_) {
                constexpr int r = decltype(R_)::value;
                Sc[r] = abuf[k & 1][r]; Pdc[r] = abuf[k & 1][6 + r]; Pddc[r] = abuf[k & 1][12 + r];
              });
            }
            T dq, dqd;
            if constexpr (PARENT[c] < 0) {   // root column: psid = 0, psidd = (0, 0, 0, g S_y, -g S_x, 0)
              dq = fma_(t1[3], Pddc[3], t1[4] * Pddc[4]);
              dqd = dot6(t4, Sc);
            } else {
              dq = dot6_acc(t1, Pddc, dot6(t4, Pdc));
              dqd = fma_(T(2), dot6(t1, Pdc), dot6(t4, Sc));
            }
            if constexpr (c == j) dqd += sel(use_damping != 0, T(DAMPING[j]), T(0));   // (:1336-1341)
            myrow[c] = dq;
            myrow[N + c] = dqd;
            if constexpr (c != j) {   // column entries of the ancestor's row, parked until that row is built
              EW(tree_pend_slot(c, j)) = dot6(Sc, t3);
              EW(tree_pend_slot(c, j) + 1) = dot6(Sc, t2);
            }
            asm volatile("" ::: "memory");
          });
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
#ifdef RBD_TWS_EXP_NOFLUSH       // timing experiment: rows stay in the image (results are wrong)
          if (factive && use_damping == 12345) {
#else
          if (factive) {
#endif
            if constexpr (WIDE) {
              flush_image_rows<CPI, TREE_KP / 4, false>(reinterpret_cast<const V4*>(rowimg), reinterpret_cast<V4*>(dcdu + (cfg0 * N + j) * (2 * N)),
                                                 (long long)(N * N / 2), fsub, fe, nvalid);
            } else {
              flush_image_rows<CPI, TREE_KP / 2, false>(reinterpret_cast<const V2*>(rowimg), reinterpret_cast<V2*>(dcdu + (cfg0 * N + j) * (2 * N)),
                                                 (long long)(N * N), fsub, fe, nvalid);
            }
          }
          // step back to the parent inside the chain, or park the finished chain's composite
          if constexpr (j != h) {
            ws_up<j>(s, gj, qdj, qddj, Sj, Pdj);
          } else if constexpr (PARENT[h] >= 0) {
            comp_each(C, [&](auto I_, T& x) { EW(TREE_PARK + TREE_COMP * park_rank(h) + decltype(I_)::value) = x; });
          }
        }
      });
     }
    }
  });
  if constexpr (TWS_MULTI) {
    // every wave passes exactly ONE block barrier (rbd_idsva_tree.h: tree_barrier_plan_ok)
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
