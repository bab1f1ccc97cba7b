// rbd_minv_fused.h -- minv of an Atlas-size robot in ONE launch.
//
// The two-launch path (minv_ia8_kernel, then minv_cols_kernel through a [body][config][12] HBM workspace) is, at
// B = 16 384, a sum of latencies: phase A 17-19 us (the torso's 18-body articulated-inertia chain; its 2 048
// single-wave blocks all start and end together), a launch boundary, and the torso's column blocks 15 us (stage the
// records, sweep, flush -- again in lockstep, one round).  Here a block owns 8 configurations of a group WITH LIMBS
// from q to Minv:
//   phase A   one wave per SEGMENT (stem, limb 1, limb 2, ...: the limbs' recursions run side by side, 7 + 4 serial
//             body steps instead of 18 for Atlas' torso): 8 lanes per configuration as in rbd_minv_ia8.h, the
//             {U, 1/D, sin, cos} records go to LDS, a limb's root parks column c of X^T Ia X for the stem's wave
//   phase B   one wave per column CLASS (stem columns, limb 1 columns, ...): minv_cols_class on the LDS records
//   epilogue  qdd = Minv (u - c) if asked, the group's rows out as 16-byte pieces
// and the waves of a block of a SMALL group (<= 8 bodies) each run ia8_group's fused path on their own 8
// configurations.  No workspace, no second launch, and blocks of different groups and different progress share the
// chip, so the recursions of some overlap with the sweeps and stores of others.  Atlas fp32 against the two
// launches: B = 4 096 11.5 vs 17.3 us, 16 384 27.0 vs 33.8, 131 072 174 vs 233, 524 288 785 vs 928.  The phase-A
// exchange areas and parks live in the tile's LDS space (the tile is first written in phase B): 29 -> 20 KB per
// block, 5 -> 8 blocks per CU, worth 31.5 -> 27.0 us at B = 16 384; forcing 64 VGPRs for 8 waves per SIMD spills
// and was slower.
#pragma once
#include "rbd_spatial.h"

namespace rbdk {

constexpr int MF_W = MINV_COLS_W;          // waves per block
constexpr int MF_CPB = 8;                  // configurations per block of a group with limbs
constexpr int MF_XA = 8 * 6 * 6;           // scalars of one phase-A exchange area of such a block (8 configurations x 6 columns x 6)
constexpr bool mf_group_ok(int rt) {
  if (minv_small_group(rt)) return true;
  if (mcl_limbs(rt) == 0) return false;
  for (int c = 0; c <= mcl_limbs(rt); ++c)
    if (mcl_count(rt, c) * MF_CPB > 64) return false;
  return true;
}
constexpr bool mf_ok() {
  if (!GRAD_PER_ROOT || MF_W < 2) return false;
  for (int rt = 0; rt < N; ++rt)
    if (grp_head(rt) && !mf_group_ok(rt)) return false;
  return true;
}
constexpr bool MINV_FUSED_OK = mf_ok();
// LDS scalars of a block.  Limbs group: records + tile + the group's inertias; the per-wave exchange areas and
// the parks of phase A live in the tile's space (the tile is first written in phase B).  Small group: the group's
// inertias + per wave max(exchange area, tile) + tau.
constexpr size_t mf_max(size_t a, size_t b) { return a > b ? a : b; }
// (round 4: the tile is the packed UPPER TRIANGLE of the group's block, 171 instead of 325 scalars per configuration for an
// 18-body group, the inertia table is symmetric-packed and lives in the space the tile takes over in phase B, and the exchange
// areas hold 6 x 6 values per configuration without the idle lanes' slots: 19.9 -> 14.2 KB per block, i.e. TEN three-wave blocks
// per CU instead of eight, with 60 VGPRs since the column phase lost its forward sweep.  Measured: no change at B = 16 384
// (all 2 048 + 1 366 blocks were co-resident before, too), 777 -> 689-733 us at B = 524 288.)
constexpr size_t mf_limbs_scalars(int rt) {
  return (size_t)MF_CPB * grp_rows(rt) * MINV_WS +
         mf_max((size_t)MF_CPB * minv_tst(rt), (size_t)(2 * MF_W - 1) * MF_XA + (size_t)grp_rows(rt) * 21);
}
constexpr int MF_SMALL_WAVE = (64 * 6 > 8 * IA8_TS ? 64 * 6 : 8 * IA8_TS) + 64;   // scalars per wave of a small-group block
constexpr size_t mf_small_scalars() { return (size_t)8 * 36 + (size_t)MF_W * MF_SMALL_WAVE; }
template <class T>
constexpr size_t mf_lds_bytes() {
  size_t m = mf_small_scalars();
  for (int rt = 0; rt < N; ++rt)
    if (grp_head(rt) && mcl_limbs(rt) > 0 && mf_limbs_scalars(rt) > m) m = mf_limbs_scalars(rt);
  return sizeof(T) * ((m + 3) / 4 * 4);
}
constexpr int mf_cfgs_per_block(int rt) { return mcl_limbs(rt) > 0 ? MF_CPB : 8 * MF_W; }
inline long long mf_blocks(long long B) {
  long long nb = 0;
  for (int rt = 0; rt < N; ++rt)
    if (grp_head(rt)) nb += (B + mf_cfgs_per_block(rt) - 1) / mf_cfgs_per_block(rt);
  return nb;
}
// k-th body of segment SEG (0: stem, k: k-th limb) of group rt, and the segment's size
constexpr int mf_seg_count(int rt, int seg) { return mcl_count(rt, seg); }
constexpr int mf_seg_body(int rt, int seg, int k) { return mcl_col(rt, seg, k); }
constexpr bool mf_has_limb_child(int p) {
  for (int x = 0; x < N; ++x)
    if (PARENT[x] == p && limb_head(x)) return true;
  return false;
}

// phase A of one segment by one wave (see rbd_minv_ia8.h for the 8-lanes-per-configuration scheme)
template <class T, int RT, int SEG>
RBD_DEV void mf_segment_chain(const T* __restrict__ q, long long B, long long cfg0, int lane, T* recs, T* tr_lds, const T* im_lds, T* park) {
  constexpr int row0 = grp_row0(RT), rows = grp_rows(RT), NL = mcl_limbs(RT);
  constexpr int CNT = mf_seg_count(RT, SEG), NR = (CNT + 7) / 8;
  const int c = lane & 7;
  const int cc = c < 6 ? c : 0;
  const int grp = lane >> 3;
  const long long b0 = cfg0 + grp;
  const long long b = b0 < B ? b0 : B - 1;
  // sin / cos of the segment's joints: lane c takes the segment's c-th, (c + 8)-th ... body
  T s_l[NR], c_l[NR];
  sfor<0, NR>([&](auto K) {
    constexpr int k = decltype(K)::value;
    constexpr int last_body = mf_seg_body(RT, SEG, CNT - 1);   // (bound to a constant first: a constexpr call in a run-time expression is not folded)
    int jj = last_body;                                     // lanes beyond the segment repeat its last body
    sfor<0, 8>([&](auto E) {
      constexpr int e = decltype(E)::value;
      if constexpr (k * 8 + e < CNT) { constexpr int body = mf_seg_body(RT, SEG, k * 8 + e); jj = sel(c == e, body, jj); }
    });
    const T qv = q[b * N + jj];
    T sv, cv;
    sincos_(qv, &sv, &cv);
    const bool pris = JTYPE[jj] != 0;
    s_l[k] = sel(pris, qv, sv);
    c_l[k] = sel(pris, T(0), cv);
  });
  // exchange areas hold the 6 x 6 values of the wave's 8 configurations compactly (the two idle lanes of a configuration's eight
  // write nothing): MF_XA = 288 scalars per area instead of 64 x 6
  const int x6 = (grp * 6 + cc) * 6;
  // im_lds holds the group's inertias symmetric-packed (21 scalars per body): this lane's column cc of I_i is im_lds[i * 21 + so[r]]
  int so[6];
  sfor<0, 6>([&](auto R) {
    constexpr int r = decltype(R)::value;
    so[r] = r <= cc ? tri_off(r, cc, 6) : tri_off(cc, r, 6);
  });
  T IAc[N][6];
  if constexpr (SEG == 0) {
    __syncthreads();                       // the limbs have parked what their roots hand to this stem
    sfor<0, N>([&](auto P) {
      constexpr int pp = decltype(P)::value;
      if constexpr (mcl_has(RT, 0, pp) && mf_has_limb_child(pp)) {
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; IAc[pp][r] = im_lds[pp * 21 + so[r]]; });
        sfor_down<0, N>([&](auto L) {      // descending: the order of the reference's loop (:732)
          constexpr int l = decltype(L)::value;
          if constexpr (limb_head(l) && PARENT[l] == pp) {
            constexpr int lk = limb_rank_in_group(l);
            sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; IAc[pp][r] += park[lk * MF_XA + x6 + r]; });
          }
        });
      }
    });
  }
  sfor_down<row0, row0 + rows>([&](auto I) {
    constexpr int i = decltype(I)::value;
    if constexpr (mcl_has(RT, SEG, i)) {
      constexpr int p = PARENT[i];
      constexpr int si = s_index(i);
      constexpr int idx = mcl_index(RT, SEG, i);            // position of body i in the segment
      JTrig<T> tri;
      tri.s = grp8_bcast<idx % 8>(s_l[idx / 8]);
      tri.c = grp8_bcast<idx % 8>(c_l[idx / 8]);
      if constexpr (!has_child(i)) {
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; IAc[i][r] = im_lds[i * 21 + so[r]]; });
      }
      T U[6];
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; U[r] = grp8_bcast<si>(IAc[i][r]); });   // U = IA S (:697)
      const T Dinv = rcp_inertia(U[si]);                                                                          // :698,:700
      {   // record {U[6], 1/D, s, c, 0, 0, 0} of (configuration grp, body i) into LDS: lanes 0..VPB-1 one 16-byte piece each
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
        if (c < VPB) reinterpret_cast<V*>(recs + (grp * rows + (i - row0)) * MINV_WS)[c] = piece;
      }
      if constexpr (p >= 0) {
        const T uc = IAc[i][si] * Dinv;
        T col[6], y[6];
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; col[r] = fma_(-U[r], uc, IAc[i][r]); });
        xform_T<i>(tri, col, y);                         // column c of A = X^T Ia
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (c < 6) { sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; tr_lds[x6 + r] = y[r]; }); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        T row[6], z[6];
        sfor<0, 6>([&](auto K) { constexpr int k = decltype(K)::value; row[k] = tr_lds[(grp * 6 + k) * 6 + cc]; });
        xform_T<i>(tri, row, z);                         // column c of X^T Ia X (symmetric)
        if constexpr (SEG > 0 && !mcl_has(RT, SEG, p)) {  // the limb's root: parked for the stem's wave
          if (c < 6) { sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; park[(SEG - 1) * MF_XA + x6 + r] = z[r]; }); }
        } else {
          constexpr bool pre = SEG == 0 && mf_has_limb_child(p);
          constexpr bool first = !pre && last_child_of(p) == i;
          sfor<0, 6>([&](auto R) {
            constexpr int r = decltype(R)::value;
            if constexpr (first) IAc[p][r] = im_lds[p * 21 + so[r]] + z[r]; else IAc[p][r] += z[r];
          });
          pin6(IAc[p]);
        }
      }
    }
  });
  if constexpr (SEG > 0) __syncthreads();  // parked: the stem may start
}

// a group with limbs: 8 configurations per block from q to Minv
template <class T, int RT>
RBD_DEV void mf_limbs_group(const T* __restrict__ q, long long B, int dense, T* __restrict__ Minv, const T* __restrict__ u_in,
                            const T* __restrict__ c_in, T* __restrict__ qdd_out, long long blk, T* smem) {
  constexpr int row0 = grp_row0(RT), rows = grp_rows(RT), NL = mcl_limbs(RT), TS = minv_tst(RT), NT = 64 * MF_W;
  T* recs = smem;                                   // [8][rows][MINV_WS]
  T* tile = recs + MF_CPB * rows * MINV_WS;         // [8][TS]   (phase B onwards): packed upper triangles
  T* tr_all = tile;                                 // [W][MF_XA]       phase A only: in the tile's space
  T* park = tr_all + MF_W * MF_XA;                  // [W - 1][MF_XA]   phase A only
  T* im_lds = park + (MF_W - 1) * MF_XA - row0 * 21;    // phase A only: im_lds[i * 21 + packed (r, c)] for the group's bodies i
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const long long cfg0 = blk * MF_CPB;
  const long long rem = B - cfg0;
  const int nvalid = rem < MF_CPB ? (int)rem : MF_CPB;
  {
    constexpr int TR[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
    constexpr int TC[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};
    for (int k = tid; k < rows * 21; k += NT) {     // this group's inertias, upper triangles
      const int e = k % 21;
      im_lds[row0 * 21 + k] = T(IM[row0 + k / 21][TR[e] * 6 + TC[e]]);
    }
  }
  __syncthreads();
  // ---- phase A: one wave per segment (waves beyond the group's segments only keep the barrier count) -------------
  bool ran = false;
  sfor<0, MF_W>([&](auto W) {
    constexpr int w = decltype(W)::value;
    if constexpr (w <= NL) {
      if (wave == w) { mf_segment_chain<T, RT, w>(q, B, cfg0, lane, recs, tr_all + w * MF_XA, im_lds, park); ran = true; }
    }
  });
  if (!ran) __syncthreads();
  __syncthreads();                                  // every record of the block is in LDS
#if defined(RBD_MF_EXP_STOP) && RBD_MF_EXP_STOP == 1     // timing experiments (tools/pmc_atlas_variants.sh): the block ends after phase A /
  if (dense != 12345) return;                           // the sweeps / the entries; results are wrong
#endif
  // ---- phase B: one wave per column class ---------------------------------------------------------------------------
  int slot = 0, j = row0;
  bool spare = true;
  // backward sweeps -> table (the tile's upper triangle) | entries from the table (Minv = Psi^T D^-1 Psi) | final values
  T wj[N], accv[N];
  sfor<0, MF_W>([&](auto W) {
    constexpr int w = decltype(W)::value;
    if constexpr (w <= NL) { if (wave == w) minv_cols_class_bwd<T, RT, w, MF_CPB, true>(recs, tile, lane, slot, j, spare, wj); }
  });
  __syncthreads();
#if defined(RBD_MF_EXP_STOP) && RBD_MF_EXP_STOP == 2
  if (dense != 12345) return;
#endif
  sfor<0, MF_W>([&](auto W) {
    constexpr int w = decltype(W)::value;
    if constexpr (w <= NL) { if (wave == w) minv_cols_class_fin<T, RT, w, true>(tile, slot, wj, accv); }
  });
  __syncthreads();
#if defined(RBD_MF_EXP_STOP) && RBD_MF_EXP_STOP == 3
  if (dense != 12345) return;
#endif
  sfor<0, MF_W>([&](auto W) {
    constexpr int w = decltype(W)::value;
    if constexpr (w <= NL) { if (wave == w) minv_cols_class_put<T, RT, w, true>(tile, dense, slot, j, spare, accv); }
  });
  __syncthreads();
  if (qdd_out != nullptr) {
    T* tau = recs + slot * rows - row0;              // tau[k], k in the group (the record area is free now)
    if (!spare && slot < nvalid) tau[j] = u_in[(cfg0 + slot) * N + j] - c_in[(cfg0 + slot) * N + j];
    __syncthreads();
    if (!spare && slot < nvalid) {
      const T* myt = tile + slot * TS;              // row j of the symmetric block from the packed upper triangle
      const int jl = j - row0;
      T o = T(0);
      sfor<row0, row0 + rows>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const int off = (k - row0) <= jl ? tri_off(k - row0, jl, rows) : tri_off(jl, k - row0, rows);
        o = fma_(myt[off], tau[k], o);
      });
      qdd_out[(cfg0 + slot) * N + j] = o;
    }
  }
#ifdef RBD_MF_EXP_NOEPI            // timing experiment: the torso's rows are not written (results are wrong)
  if (Minv != nullptr && dense == 12345) {
#else
  if (Minv != nullptr) {
#endif
    constexpr int RW = rows * N;
    T* gdst = Minv + cfg0 * (N * N) + row0 * N;
    auto elem = [&](int cfg, int e) -> T {
      const int r = e / N;
      const int cx = e - r * N - row0;
      bool own = cx >= 0 && cx < rows;
      const int cc2 = own ? cx : r;
      const int off = cc2 >= r ? tri_off(r, cc2, rows) : tri_off(cc2, r, rows);
      own = own && (dense != 0 || cc2 >= r);
      const T x = tile[cfg * TS + off];
      return own ? x : T(0);
    };
    if constexpr (minv_piece_flush<T>(RT)) {
      minv_own_rows_flush<T, row0, rows, MF_CPB, TS, NT, true>(tile, gdst, tid, nvalid, dense);
    } else {
      const int total = nvalid * RW;
#pragma unroll 4
      for (int g = tid; g < total; g += NT) {
        const int cfg = g / RW;
        const int r2 = g - cfg * RW;
        gdst[(long long)cfg * (N * N) + r2] = elem(cfg, r2);
      }
    }
  }
}

template <class T>
__global__ __launch_bounds__(64 * MF_W, 2) void minv_fused_kernel(const T* __restrict__ q, long long B, int dense, T* __restrict__ Minv,
                                                                const T* __restrict__ u_in, const T* __restrict__ c_in,
                                                                T* __restrict__ qdd_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  long long blk = blockIdx.x;
  bool done = false;
  sfor<0, N>([&](auto Rt) {
    constexpr int rt = decltype(Rt)::value;
    if constexpr (grp_head(rt)) {
      constexpr int cpb = mf_cfgs_per_block(rt);     // constexpr on purpose (a plain call would walk the tree at run time)
      const long long nb = (B + cpb - 1) / cpb;
      if (!done) {
        if (blk < nb) {
          if constexpr (mcl_limbs(rt) > 0) {
#ifndef RBD_MF_EXP_NOTORSO        // timing experiment: the blocks of groups with limbs do nothing
            mf_limbs_group<T, rt>(q, B, dense, Minv, u_in, c_in, qdd_out, blk, smem);
#endif
          } else {
#ifdef RBD_MF_EXP_NOLEGS          // timing experiment: the blocks of small groups do nothing
            if (dense != 12345) return;
#endif
            // a small group: every wave of the block is an independent 8-configuration unit of ia8_group's fused path
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            constexpr int r0 = grp_row0(rt);
            T* im_lds = smem - r0 * 36;                         // the group's inertias (<= 8 x 36), shared by the block's waves
            T* mine = smem + 8 * 36 + wave * MF_SMALL_WAVE;     // exchange area and tile share a space (used one after the other)
            ia8_group<T, rt>(q, B, static_cast<T*>(nullptr), 1, dense, Minv, u_in, c_in, qdd_out, (blk * MF_W + wave) * 8, lane,
                             mine, im_lds, mine, mine + MF_SMALL_WAVE - 64);
          }
          done = true;
        } else {
          blk -= nb;
        }
      }
    }
  });
}

}  // namespace rbdk
