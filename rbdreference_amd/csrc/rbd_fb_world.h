// rbd_fb_world.h -- FLOATING-BASE rnea and rnea_grad as batch kernels with coalesced traffic.
//
// rnea_grad_fbw_kernel: RBDReference.rnea_grad with the reference's floating-base branches
// (/root/reference/RBDReference.py:1127-1368: body 0 is a 6-DoF joint with S = eye(6) in base coordinates, its
// six position columns are derivatives along a base-frame twist) from WORLD-FRAME identities, one
// configuration per lane -- the scheme of rbd_idsva.h / rbd_idsva_tree.h (first-order part of :1413-1484) with
// a multi-DoF root.  Numpy prototype of exactly this arithmetic against the pinned oracle:
// tools/proto_fb_idsva.py (3e-13).  What the 6-DoF joint changes:
//   world columns of the base  s_k = X_0^{-1} e_k:  k < 3: (R_0[:, k]; p_0 x R_0[:, k]),  k >= 3: (0; R_0[:, k - 3])
//   psid_k = 0 (the parent is the world),  psidd_k = a_grav x s_k (zero for k >= 3),
//   Sdot_k = v_0 x s_k != psid_k: the velocity columns of a multi-DoF joint carry  t1.(psid_k + Sdot_k)  where a
//   1-DoF joint has  2 t1.psid  -- for the base  t1.Sdot_k = -(v_0 x* t1).s_k, so a row needs ONE vector
//   u = t4 - v_0 x* t1 and six dot products u.s_k;
//   two columns of the SAME joint use the row identity in both orders (the base's 6 x 6 blocks), so the base
//   needs t1, t4 per column and no t2, t3.
// Per body j >= 1 on top of the fixed-base work: the six base columns of its own row and the twelve entries
// (s_k.t3_j, s_k.t2_j) of the base's rows, which wait in lane-private LDS until the base is built.
// The tree is cut into chains (rbd_world.h, TreePlan); chains are processed from the highest head down, every
// chain recomputes the world kinematics of its root path from q (base included), finished chains park their
// composite; a row is complete when its body is processed, goes through a [64][2 nv] LDS image and leaves as
// 8 nv-byte contiguous segments in 16-byte pieces.  The velocity-damping quirks of :1336-1341 (body id instead
// of matrix index, a 5 x 5 block for the base) are kept literally.
//
// rnea_fbw_kernel: RBDReference.rnea for a floating base (:559-628 with :585, :591), one configuration per
// lane in body coordinates as the reference, v / a / f / c through unpadded LDS images that equal the HBM
// layout: flat 16-byte full-line stores (the first floating-base kernel stored element by element).
#pragma once
#include "rbd_fb.h"
#include "rbd_world.h"

namespace rbdk {

#define FBW_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// ---------------------------------------------------------------------------------------------------------
// rnea (floating base), coalesced
// ---------------------------------------------------------------------------------------------------------
template <class T>
constexpr size_t rnea_fbw_lds_bytes() { return sizeof(T) * (size_t)64 * 6 * N; }

// flat copy of a [64][K] LDS image to HBM (K scalars per configuration, 64 K % (16 / sizeof T) == 0)
template <int K, class T>
RBD_DEV void fbw_flush_flat(const T* img, T* gdst, int lane, int nvalid) {
  constexpr int VE = 16 / sizeof(T);
  static_assert((64 * K) % VE == 0, "image size");
  if (nvalid == 64) {
    typedef T V __attribute__((ext_vector_type(VE)));
    const V* src = reinterpret_cast<const V*>(img);
    V* dst = reinterpret_cast<V*>(gdst);
    constexpr int NVEC = 64 * K / VE;
#pragma unroll
    for (int i = 0; i < (NVEC + 63) / 64; ++i) {
      const int g = lane + 64 * i;
      if (g < NVEC) dst[g] = src[g];
    }
  } else {
    for (int g = lane; g < nvalid * K; g += 64) gdst[g] = img[g];
  }
}

// MODE 0: rnea.  MODE 1: rnea_fpass (:559-598): v, a and the LOCAL f, no backward pass.  MODE 2: rnea_bpass
// (:600-621): f [B, 6, N] is read from f_out, accumulated and written back in place, c out.
template <class T, bool HAS_QDD, int MODE = 0>
__global__ __launch_bounds__(64, 1) void rnea_fbw_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                         const T* __restrict__ qdd, T grav, long long B,
                                                         T* __restrict__ c_out, T* __restrict__ v_out,
                                                         T* __restrict__ a_out, T* __restrict__ f_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* img = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x;
  const long long cfg0 = (long long)blockIdx.x * 64;
  const long long rem = B - cfg0;
  const int nvalid = rem < 64 ? (int)rem : 64;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);
  const T* qb = q + b * NV; const T* qdb = MODE == 2 ? nullptr : qd + b * NV; const T* qddb = HAS_QDD ? qdd + b * NV : nullptr;
  constexpr int K6 = 6 * N;
  JTrig<T> tr[N];
  T qdv[N], qddv[N];
  sfor<1, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    tr[j] = make_trig<j>(qb[j + 5]);
    if constexpr (MODE != 2) {
      qdv[j] = qdb[j + 5];
      if constexpr (HAS_QDD) qddv[j] = qddb[j + 5]; else qddv[j] = T(0);
    }
  });
  T v[N][6], a[N][6], f[N][6];
  if constexpr (MODE == 2) {
    sfor<0, N>([&](auto J) { sfor<0, 6>([&](auto R) { constexpr int j = decltype(J)::value, r = decltype(R)::value; f[j][r] = f_out[b * K6 + r * N + j]; }); });
  } else {
    // the base (:576-596 with the floating-base lines :585, :591): v_0 = qd[0:6]; a_0 = X_0 a_grav + qdd[0:6]
    T E[3][3];
    fb_base_E(qb[3], qb[4], qb[5], E);
    sfor<0, 6>([&](auto R) {
      constexpr int r = decltype(R)::value;
      v[0][r] = qdb[r];
      a[0][r] = HAS_QDD ? qddb[r] : T(0);
    });
    a[0][3] -= grav * E[0][2]; a[0][4] -= grav * E[1][2]; a[0][5] -= grav * E[2][2];
    T Iv[6], Ia[6];
    cmatvec<MatI, 0>(v[0], Iv);
    cmatvec<MatI, 0>(a[0], Ia);
    sfor<0, 6>([&](auto R) { f[0][decltype(R)::value] = Ia[decltype(R)::value]; });
    fxv<true>(v[0], Iv, f[0]);
  }
  if constexpr (MODE != 2) {
  sfor<1, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int p = PARENT[j];
    T xv[6], xa[6];
    rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddv[j], grav, v[p], a[p], xv, xa, v[j], a[j], f[j]);
  });
  }
  // a [B, 6, N] tensor of 64 configurations through the image: element (r, j) of configuration `lane` at lane * 6N + r * N + j
  auto put6 = [&](const T (&x)[N][6], T* gdst) {
    FBW_WAVE_SYNC();                                           // the previous image has been read
    sfor<0, N>([&](auto J) {
      sfor<0, 6>([&](auto R) {
        constexpr int j = decltype(J)::value, r = decltype(R)::value;
        img[lane * K6 + r * N + j] = x[j][r];
      });
    });
    FBW_WAVE_SYNC();
    fbw_flush_flat<K6>(img, gdst + cfg0 * K6, lane, nvalid);
  };
  if constexpr (MODE != 2) {
    if (v_out != nullptr) {
      put6(v, v_out);
      put6(a, a_out);
    }
  }
  if constexpr (MODE == 1) {
    put6(f, f_out);                                            // the local forces: no backward pass
    return;
  }
  // backward pass (:607-619)
  T c[NV];
  sfor_down<1, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int p = PARENT[j];
    c[j + 5] = S_dot<j>(f[j]);
    T t[6];
    xform_T<j>(tr[j], f[j], t);
    sfor<0, 6>([&](auto R) { f[p][decltype(R)::value] += t[decltype(R)::value]; });
  });
  sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; c[r] = f[0][r]; });   // S = eye(6)
  if (f_out != nullptr) put6(f, f_out);
  FBW_WAVE_SYNC();
  sfor<0, NV>([&](auto I) { constexpr int i = decltype(I)::value; img[lane * NV + i] = c[i]; });
  FBW_WAVE_SYNC();
  fbw_flush_flat<NV>(img, c_out + cfg0 * NV, lane, nvalid);
}

// ---------------------------------------------------------------------------------------------------------
// rnea_grad (floating base), world frame
// ---------------------------------------------------------------------------------------------------------
// LDS plan (scalars of T): [0, FBW_W * 64 * FBW_KP) one row image per wave; then lane-private columns (slot * 64 + lane):
//   FBW_PAIR  (jj, j) with jj >= 1 a proper ancestor of j: the column entries (dq, dqd) of row jj     2 per pair
//   FBW_BASE  (k, j), k < 6, j >= 1: the entries (dq, dqd) of base row k in the columns of body j       12 (N - 1)
//   FBW_PARK  composites of finished chains (heads with a parent) that do not wait in a wave's image     31 per chain
constexpr int fbw_pairs_before(int j) {          // pairs (jj, x) with x < j
  int k = 0;
  for (int x = 1; x < j; ++x)
    for (int y = PARENT[x]; y >= 1; y = PARENT[y]) ++k;
  return k;
}
constexpr int fbw_pair_rank(int jj, int j) {     // jj >= 1 a proper ancestor of j
  int k = fbw_pairs_before(j);
  for (int y = PARENT[j]; y >= 1 && y != jj; y = PARENT[y]) ++k;
  return k;
}
constexpr int FBW_NPAIRS = fbw_pairs_before(N);
constexpr int FBW_ROW = 2 * NV;
constexpr int FBW_KP = (NV % 2 == 0) ? 4 * ((NV / 2) | 1) : 2 * NV;   // row stride: odd in units of the flush vector
// One block = 64 configurations x FBW_W waves.  The subtrees hanging off the base are independent of each other
// until the base itself is built, so each gets a wave (round-robin when there are more than FBW_W_MAX of them);
// the one that continues the base's chain (its heavy child) runs on wave 0, which then builds the base.  The
// waves share the lane-private LDS columns (same lane = same configuration): pending entries and parked
// composites of one wave are read by another across ONE block barrier -- wave 0 passes it just before it builds
// the base, every other wave when its chains are done (s_barrier counts arrivals).
constexpr int FBW_W_MAX = 4;
constexpr int fbw_child_root(int j) {            // the child of the base that body j >= 1 hangs under
  while (PARENT[j] != 0) j = PARENT[j];
  return j;
}
constexpr int fbw_n_children() {
  int k = 0;
  for (int j = 1; j < N; ++j) k += PARENT[j] == 0 ? 1 : 0;
  return k;
}
constexpr int FBW_W = fbw_n_children() < FBW_W_MAX ? (fbw_n_children() < 1 ? 1 : fbw_n_children()) : FBW_W_MAX;
constexpr int fbw_wave_of_child(int r) {         // r: a child of the base
  if (TP.heavy[0] == r) return 0;
  int k = 0;                                     // ordinal among the other children
  for (int j = 1; j < r; ++j) k += (PARENT[j] == 0 && TP.heavy[0] != j) ? 1 : 0;
  return FBW_W == 1 ? 0 : 1 + k % (FBW_W - 1);
}
constexpr int fbw_wave_of(int h) { return h == 0 ? 0 : fbw_wave_of_child(fbw_child_root(h)); }
// A child subtree's composite waits for the base in the row image of ITS wave (free once that wave's last row has
// been flushed) when that wave runs no other child subtree; everything else parks in the FBW_PARK area.
constexpr int fbw_children_on_wave(int w) {
  int k = 0;
  for (int j = 1; j < N; ++j) k += (PARENT[j] == 0 && fbw_wave_of_child(j) == w) ? 1 : 0;
  return k;
}
constexpr bool fbw_park_in_image(int h) {        // h: a chain head with a parent
  return PARENT[h] == 0 && fbw_wave_of_child(h) != 0 && fbw_children_on_wave(fbw_wave_of_child(h)) == 1 && TREE_COMP_SCALARS <= 2 * NV;
}
constexpr int fbw_area_parks() {
  int k = 0;
  for (int h = 1; h < N; ++h) k += (is_chain_head(h) && !fbw_park_in_image(h)) ? 1 : 0;
  return k;
}
constexpr int fbw_area_rank(int h) {
  int k = 0;
  for (int x = 1; x < h; ++x) k += (is_chain_head(x) && !fbw_park_in_image(x)) ? 1 : 0;
  return k;
}
constexpr int FBW_PAIR = 0;
constexpr int FBW_BASE = FBW_PAIR + 2 * FBW_NPAIRS;
constexpr int FBW_PARK = FBW_BASE + 12 * (N - 1);
constexpr int FBW_PRIV = FBW_PARK + TREE_COMP_SCALARS * fbw_area_parks();
template <class T>
constexpr size_t fbw_lds_bytes() { return sizeof(T) * (size_t)64 * (FBW_W * FBW_KP + FBW_PRIV); }
constexpr bool fbw_model_ok_() {
  if (N < 6) return false;                                  // the reference raises below six bodies (:1168)
  for (int j = 1; j < N; ++j) {
    if (JTYPE[j] != 0) return false;                        // revolute joints (the reference's fxS form for prismatic ones is not an identity)
    if (!rigid_inertia_(j)) return false;
  }
  return rigid_inertia_(0);
}
template <class T>
constexpr bool grad_fbw_ok() { return fbw_model_ok_() && fbw_lds_bytes<T>() <= 160 * 1024; }

// base state from q[0:6], qd[0:6], qdd[0:6]: R_0 = E^T (base -> world), p_0 = q[0:3]; v_0, a_0 in world coordinates
template <class T>
RBD_DEV void fbw_base_state(WState<T>& s, const T* qb, const T* qdb, const T* qddb, T grav) {
  T E[3][3];
  fb_base_E(qb[3], qb[4], qb[5], E);
  sfor<0, 3>([&](auto R_) { sfor<0, 3>([&](auto C_) { constexpr int r = decltype(R_)::value, c = decltype(C_)::value; s.R[r][c] = E[c][r]; }); });
  s.p[0] = qb[0]; s.p[1] = qb[1]; s.p[2] = qb[2];
  // X_0^{-1} x = (R w; R u + p x (R w))
  auto to_world = [&](const T (&x)[6], T (&y)[6]) {
    T w[3], u[3];
    sfor<0, 3>([&](auto R_) {
      constexpr int r = decltype(R_)::value;
      w[r] = fma_(s.R[r][0], x[0], fma_(s.R[r][1], x[1], s.R[r][2] * x[2]));
      u[r] = fma_(s.R[r][0], x[3], fma_(s.R[r][1], x[4], s.R[r][2] * x[5]));
    });
    cross3_acc(s.p, w, u);
    y[0] = w[0]; y[1] = w[1]; y[2] = w[2]; y[3] = u[0]; y[4] = u[1]; y[5] = u[2];
  };
  T x[6];
  sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; x[r] = qdb[r]; });
  to_world(x, s.v);
  sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; x[r] = qddb != nullptr ? qddb[r] : T(0); });
  to_world(x, s.a);
  s.a[5] -= grav;                                           // a_grav = (0, 0, 0, 0, 0, -GRAVITY) in the world frame
}
// world column k of the base joint
template <int K, class T>
RBD_DEV void fbw_base_col(const WState<T>& s, T (&sk)[6]) {
  if constexpr (K < 3) {
    const T ang[3] = {s.R[0][K], s.R[1][K], s.R[2][K]};
    T lin[3];
    cross3(s.p, ang, lin);
    sk[0] = ang[0]; sk[1] = ang[1]; sk[2] = ang[2]; sk[3] = lin[0]; sk[4] = lin[1]; sk[5] = lin[2];
  } else {
    sk[0] = T(0); sk[1] = T(0); sk[2] = T(0);
    sk[3] = s.R[0][K - 3]; sk[4] = s.R[1][K - 3]; sk[5] = s.R[2][K - 3];
  }
}
// x . s_k using the structure of the base columns
template <int K, class T>
RBD_DEV T fbw_dot_col(const T (&x)[6], const T (&sk)[6]) {
  if constexpr (K < 3) return dot6(x, sk);
  else return fma_(x[5], sk[5], fma_(x[4], sk[4], x[3] * sk[3]));
}

template <class T, bool HAS_QDD>
__global__ __launch_bounds__(64 * FBW_W, 1) void rnea_grad_fbw_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                              const T* __restrict__ qdd, T grav, int use_damping,
                                                              long long B, T* __restrict__ c_out, T* __restrict__ dcdu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  T* rowimg = reinterpret_cast<T*>(smem_raw) + wave * (64 * FBW_KP);             // [64][FBW_KP], one per wave
  T* priv = reinterpret_cast<T*>(smem_raw) + FBW_W * 64 * FBW_KP + lane;         // priv[slot * 64], shared by the waves
  // where the composite of chain h waits: its wave's image (slot * 64 + lane, like the private columns) or the park area
  auto park_at = [&](auto H_, int slot) -> T* {
    constexpr int hh = decltype(H_)::value;
    if constexpr (fbw_park_in_image(hh)) return reinterpret_cast<T*>(smem_raw) + fbw_wave_of(hh) * (64 * FBW_KP) + slot * 64 + lane;
    else return priv + (FBW_PARK + TREE_COMP_SCALARS * fbw_area_rank(hh) + slot) * 64;
  };
  const long long cfg0 = (long long)blockIdx.x * 64;
  const long long rem = B - cfg0;
  const int nvalid = rem < 64 ? (int)rem : 64;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);
  const T* qrow = q + b * NV;
  const T* qdrow = qd + b * NV;
  const T* qddrow = HAS_QDD ? qdd + b * NV : nullptr;
  const T dsel = use_damping != 0 ? T(1) : T(0);

  // flush geometry: a row is NV float2 (NV / 2 float4 when nv is even); lanes (fsub, fe) cover CPI configurations
  typedef T V2 __attribute__((ext_vector_type(2)));
  typedef T V4 __attribute__((ext_vector_type(4)));
  constexpr bool WIDE = (NV % 2 == 0) && (FBW_KP % 4 == 0) && sizeof(T) == 4;
  constexpr int FW = WIDE ? NV / 2 : NV;                  // vectors per row
  constexpr int CPI = 64 / FW > 0 ? 64 / FW : 1;          // configurations per flush step
  const int fsub = lane / FW, fe = lane - fsub * FW;
  const bool factive = lane < CPI * FW;
  // full tiles store through a descriptor of the block's 64 matrices (rbd_world.h: flush_image_rows_buf)
  const __amdgpu_buffer_rsrc_t out_rs = out_tile_rsrc(dcdu + cfg0 * NV * (long long)FBW_ROW, 64 * NV * FBW_ROW * (int)sizeof(T));
  // row[2 nv] of matrix row `mrow` -> image -> dc_du[b][mrow][:]
  auto flush_row = [&](const T (&row)[FBW_ROW], int mrow) {
    FBW_WAVE_SYNC();
    {
      V2* mine = reinterpret_cast<V2*>(rowimg) + lane * (FBW_KP / 2);
      sfor<0, NV>([&](auto E_) {
        constexpr int e = decltype(E_)::value;
        V2 x; x[0] = row[2 * e]; x[1] = row[2 * e + 1];
        mine[e] = x;
      });
    }
    FBW_WAVE_SYNC();
    if (factive) {
      if constexpr (WIDE) {
        if (nvalid == 64) {
          flush_image_rows_buf<CPI, FBW_KP / 4>(reinterpret_cast<const V4*>(rowimg), out_rs, (fsub * (NV * FBW_ROW / 4) + fe) * 16,
                                                mrow * FBW_ROW * (int)sizeof(T), NV * FBW_ROW * (int)sizeof(T), fsub, fe);
        } else
        flush_image_rows<CPI, FBW_KP / 4, false>(reinterpret_cast<const V4*>(rowimg), reinterpret_cast<V4*>(dcdu + (cfg0 * NV + mrow) * (long long)FBW_ROW),
                                          (long long)(NV * FBW_ROW / 4), fsub, fe, nvalid);
      } else {
        flush_image_rows<CPI, FBW_KP / 2>(reinterpret_cast<const V2*>(rowimg), reinterpret_cast<V2*>(dcdu + (cfg0 * NV + mrow) * (long long)FBW_ROW),
                                          (long long)(NV * FBW_ROW / 2), fsub, fe, nvalid);
      }
    }
  };
  // (:1336-1341) literally: the base adds its damping to a 5 x 5 block, body ind >= 1 to entry (ind, nv + ind) --
  // the BODY id, not its matrix index
  auto add_damping = [&](T (&row)[FBW_ROW], auto MR_) {
    constexpr int mr = decltype(MR_)::value;
    if constexpr (mr < 5 && DAMPING[0] != 0.0) {
      sfor<0, 5>([&](auto C_) { constexpr int c = decltype(C_)::value; row[NV + c] = fma_(dsel, T(DAMPING[0]), row[NV + c]); });
    }
    if constexpr (mr >= 1 && mr < N) {
      if constexpr (DAMPING[mr] != 0.0) row[NV + mr] = fma_(dsel, T(DAMPING[mr]), row[NV + mr]);
    }
  };

  sfor_down<0, N>([&](auto H_) {
    constexpr int h = decltype(H_)::value;
    if constexpr (is_chain_head(h)) {
     if (wave == fbw_wave_of(h)) {
      constexpr int leaf = chain_leaf(h);
      // ---- inputs and trig of the root path of this chain; the base's state and columns ------------------
      JTrig<T> tr[N];
      T qdv[N], qddv[N];
      sfor<1, N>([&](auto J_) {
        constexpr int j = decltype(J_)::value;
        if constexpr (is_anc_or_self(j, leaf)) {
          tr[j] = make_trig<j>(qrow[j + 5]);
          qdv[j] = qdrow[j + 5];
          if constexpr (HAS_QDD) qddv[j] = qddrow[j + 5]; else qddv[j] = T(0);
        }
      });
      WState<T> s;
      fbw_base_state(s, qrow, qdrow, qddrow, grav);
      T sb[6][6];                                           // world columns of the base
      sfor<0, 6>([&](auto K_) { fbw_base_col<decltype(K_)::value>(s, sb[decltype(K_)::value]); });
      T v0[6];
      sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; v0[r] = s.v[r]; });
      // ---- world kinematics base -> leaf (:1413-1434) ---------------------------------------------------
      T Sv[N][6], Pd[N][6], Pdd[N][6];
      sfor<1, N>([&](auto J_) {
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
          // the other waves' subtrees (composites, base-row entries) are complete behind this barrier
          if constexpr (j == 0 && FBW_W > 1) __syncthreads();
          // finished chains hanging off this body (:1446-1448)
          sfor<0, N>([&](auto K_) {
            constexpr int kk = decltype(K_)::value;
            if constexpr (PARENT[kk] == j && is_chain_head(kk) && kk != j) {
              Comp<T> P;
              comp_each(P, [&](auto I_, T& x) { x = *park_at(K_, decltype(I_)::value); });
              comp_add(C, P);
            }
          });
          if constexpr (j >= 1) {
            const T cj = dot6(Sv[j], C.f);
            if (c_out != nullptr && lane < nvalid) c_out[b * NV + j + 5] = cj;
            // t-vectors (:1481-1484)
            T t1[6], t2[6], t3[6], t4[6];
            tvectors<false>(C, Sv[j], Pd[j], Pdd[j], t1, t2, t3, t4);
            // ---- row j + 5 --------------------------------------------------------------------------------
            T row[FBW_ROW];
            {
              // base columns: dq = t1.psidd_k (k < 3; psidd_k = (0, 0, 0, g s_y, -g s_x, 0), zero for k >= 3),
              // dqd = (t4 - v_0 x* t1).s_k; and the entries (k, j + 5) of the base's rows
              T w[6], u[6];
              fxv<false>(v0, t1, w);
              sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; u[r] = t4[r] - w[r]; });
              sfor<0, 6>([&](auto K_) {
                constexpr int k = decltype(K_)::value;
                if constexpr (k < 3) row[k] = grav * fma_(t1[3], sb[k][1], -(t1[4] * sb[k][0]));
                else row[k] = T(0);
                row[NV + k] = fbw_dot_col<k>(u, sb[k]);
                priv[(FBW_BASE + ((j - 1) * 6 + k) * 2) * 64] = fbw_dot_col<k>(t3, sb[k]);
                priv[(FBW_BASE + ((j - 1) * 6 + k) * 2 + 1) * 64] = fbw_dot_col<k>(t2, sb[k]);
              });
            }
            sfor<1, N>([&](auto C_) {
              constexpr int c = decltype(C_)::value;
              if constexpr (is_anc_or_self(c, j)) {
                T dq = dot6_acc(t1, Pdd[c], dot6(t4, Pd[c]));
                T dqd = fma_(T(2), dot6(t1, Pd[c]), dot6(t4, Sv[c]));
                row[c + 5] = dq;
                row[NV + c + 5] = dqd;
                if constexpr (c != j) {   // column entries of the ancestor's row, parked until that row is built
                  priv[(FBW_PAIR + 2 * fbw_pair_rank(c, j)) * 64] = dot6(Sv[c], t3);
                  priv[(FBW_PAIR + 2 * fbw_pair_rank(c, j) + 1) * 64] = dot6(Sv[c], t2);
                }
              } else if constexpr (is_anc_or_self(j, c)) {   // descendant: delivered earlier
                row[c + 5] = priv[(FBW_PAIR + 2 * fbw_pair_rank(j, c)) * 64];
                row[NV + c + 5] = priv[(FBW_PAIR + 2 * fbw_pair_rank(j, c) + 1) * 64];
              } else {
                row[c + 5] = T(0);
                row[NV + c + 5] = T(0);
              }
            });
            add_damping(row, std::integral_constant<int, j + 5>{});
            flush_row(row, j + 5);
            // step back to the parent inside the chain, or park the finished chain's composite
            if constexpr (j != h) {
              ws_up<j>(s, tr[j], qdv[j], qddv[j], Sv[j], Pd[j]);
            } else {
              FBW_WAVE_SYNC();                               // (an image park: the row just flushed has been read)
              comp_each(C, [&](auto I_, T& x) { *park_at(H_, decltype(I_)::value) = x; });
            }
          } else {
            // ---- the base: c[0:6] and its six rows (s now holds the base's state again) ----------------------
            sfor<0, 6>([&](auto K_) {
              constexpr int k = decltype(K_)::value;
              const T ck = fbw_dot_col<k>(C.f, sb[k]);
              if (c_out != nullptr && lane < nvalid) c_out[b * NV + k] = ck;
            });
            sfor<0, 6>([&](auto K_) {
              constexpr int k = decltype(K_)::value;
              T t1[6], t4[6], u[6];
              {
                T s1[6], z1[6], w[6];
                rin_apply(C.IC, sb[k], t1);
                sym_apply(C.SC, sb[k], s1);
                fxv<false>(sb[k], C.pm, z1);
                sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; t4[r] = s1[r] - z1[r]; });
                fxv<false>(v0, t1, w);
                sfor<0, 6>([&](auto R_) { constexpr int r = decltype(R_)::value; u[r] = t4[r] - w[r]; });
              }
              T row[FBW_ROW];
              sfor<0, 6>([&](auto L_) {
                constexpr int l = decltype(L_)::value;
                if constexpr (l < 3) row[l] = grav * fma_(t1[3], sb[l][1], -(t1[4] * sb[l][0]));
                else row[l] = T(0);
                row[NV + l] = fbw_dot_col<l>(u, sb[l]);
              });
              sfor<1, N>([&](auto C_) {
                constexpr int c = decltype(C_)::value;
                row[c + 5] = priv[(FBW_BASE + ((c - 1) * 6 + k) * 2) * 64];
                row[NV + c + 5] = priv[(FBW_BASE + ((c - 1) * 6 + k) * 2 + 1) * 64];
              });
              add_damping(row, std::integral_constant<int, k>{});
              flush_row(row, k);
            });
          }
        }
      });
     }
    }
  });
  if constexpr (FBW_W > 1) {
    if (wave != 0) __syncthreads();                          // wave 0 passed its barrier before it built the base
  }
}

}  // namespace rbdk
