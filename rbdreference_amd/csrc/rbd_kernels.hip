// rbd_kernels.hip -- per-robot HIP kernels (gfx950 / CDNA4) + the C-ABI of include/rbd_hip.h.
//
// Compiled once per robot:  hipcc --offload-arch=gfx950 -include <generated model header> ...
// (rbdreference_amd/build.py).  The path restated here is RBDReference.rnea / rnea_grad / minv,
// /root/reference/RBDReference.py:559-806, 1127-1368; line cites below are into that file.
//
// Work mapping; which kernel serves a robot is decided at compile time:
//   rnea        rnea_kernel: one configuration per lane; outputs leave through LDS tiles as coalesced
//               (flat 16-byte where the tile is unpadded) stores; big robots park v, a in LDS.
//   rnea_grad   (a) rnea_grad_idsva_kernel (rbd_idsva.h): ONE lane per configuration, world-frame
//               composite quantities, every dc_du entry produced once -- long all-revolute chains.
//               (b) rnea_grad_kernel (below): TWO lanes per configuration (even = d/dq columns, odd =
//               d/dqd columns, one instruction stream; the recursions :1139-1185 / :1210-1252 differ
//               only in their seeds).  The reference's backward passes (:1257-1343) are folded into
//               the forward sweep through
//                   dc[i, c] = sum_{j in subtree(i) & subtree(c)} Phi[i, j]^T df[c, j],
//                   Phi[i, j] = X_{j<-i} S_i = dv_dqd[:, i, j],
//               accumulated in registers and parked once in the LDS image of the output tile.
//   minv        minv_lane_kernel (rbd_minv_lane.h): one lane per configuration, fused, for robots with
//               small root groups; otherwise phase A (minv_ia_kernel / minv_ia8_kernel) + phase B
//               (minv_cols_kernel, one lane per COLUMN, one wave per column class of a group with limbs)
//               through a [body][config][12] workspace; minv_ia8_kernel also finishes the groups of <= 8 bodies;
//               robots whose big groups have limbs (Atlas): minv_fused_kernel (rbd_minv_fused.h), one launch.
//   crba, rnea_fpass / rnea_bpass, forward_dynamics(_grad): further rows, same building blocks.
// The library is built from several translation units of this one file (rbdreference_amd/build.py
// compiles them in parallel): -DRBD_TU_COMMON, _RNEA_F32, _RNEA_F64, _GRAD_F32, _GRAD_F64,
// _GRADN_F32, _GRADN_F64 (the qdd = None instantiations of the gradient kernels: half of a gradient unit's compile
// time), _MINV_F32, _MINV_F64, _FD_F32, _FD_F64, _PASS_F32, _PASS_F64 (each together with -DRBD_TU_SPLIT);
// without RBD_TU_SPLIT everything is compiled in one unit.
// -DRBD_FAST_STAGE=1 (first-use family libraries of the gradient, rbdreference_amd/build.py): only the kernel AUTO
// picks for large batches is compiled -- no small-batch column kernel, no kernels that exist to be forced by
// rbd_set_option -- which cuts the unit's compile time to a third; the full library replaces it when it is ready.
#ifndef RBD_FAST_STAGE
#define RBD_FAST_STAGE 0
#endif
#if !defined(RBD_TU_SPLIT)
#define RBD_TU_COMMON 1
#define RBD_TU_RNEA_F32 1
#define RBD_TU_RNEA_F64 1
#define RBD_TU_GRAD_F32 1
#define RBD_TU_GRAD_F64 1
#define RBD_TU_GRADN_F32 1
#define RBD_TU_GRADN_F64 1
#define RBD_TU_MINV_F32 1
#define RBD_TU_MINV_F64 1
#define RBD_TU_FD_F32 1
#define RBD_TU_FD_F64 1
#define RBD_TU_PASS_F32 1
#define RBD_TU_PASS_F64 1
#endif

// Which kernel families this unit needs (everything it does not need is dropped by the preprocessor,
// so that build.py's object cache -- keyed by the preprocessed text -- survives unrelated edits).
#if defined(RBD_TU_RNEA_F32) || defined(RBD_TU_RNEA_F64)
#define RBD_NEED_RNEA 1
#endif
#if defined(RBD_TU_GRAD_F32) || defined(RBD_TU_GRAD_F64) || defined(RBD_TU_GRADN_F32) || defined(RBD_TU_GRADN_F64) || \
    defined(RBD_TU_FD_F32) || defined(RBD_TU_FD_F64)
#define RBD_NEED_GRAD 1
#endif
#if defined(RBD_TU_MINV_F32) || defined(RBD_TU_MINV_F64)
#define RBD_NEED_MINV 1
#endif
#if defined(RBD_TU_FD_F32) || defined(RBD_TU_FD_F64)
#define RBD_NEED_FD 1
#endif
#if defined(RBD_TU_PASS_F32) || defined(RBD_TU_PASS_F64)
#define RBD_NEED_PASS 1
#endif
#include "rbd_spatial.h"

namespace rbdk {

// ---------------------------------------------------------------------------------------------
// LDS-staged coalesced store: each of the block's 64 lanes holds K values of its configuration
// (global layout [cfg][K], row-major).  Lanes park them in LDS with row stride odd_pad<K>(), then
// the wave streams the tile out.  When K itself gives at most 4-way bank conflicts on the per-lane
// 4-byte writes (2-way is free, 4-way costs 2x on ds_write_b32) the tile is left UNPADDED: the LDS
// image then equals the HBM image and leaves as flat 16-byte copies (one ds_read_b128 + one
// global_store_dwordx4 per 4 elements instead of ~10 instructions per element).  Otherwise the stride
// is padded to an odd number (conflict-free writes, element-wise read-out).
// ---------------------------------------------------------------------------------------------
constexpr int gcd32(int k) {
  int g = 32;
  while (k % g != 0) g /= 2;
  return g;
}
template <int K>
constexpr int odd_pad() { return gcd32(K) <= 4 && K % 4 == 0 ? K : gcd32(K) == 2 ? K : (K % 2 == 0) ? K + 1 : K; }

// stream a [64][odd_pad<K>]-strided LDS tile out as [nvalid][K] rows
template <int K, class T>
RBD_DEV void flush_tile(const T* lds, T* gdst, int lane, int nvalid) {
  constexpr int KP = odd_pad<K>();
  constexpr int VE = 16 / sizeof(T);
  if constexpr (KP == K && (64 * K) % VE == 0) {
    if (nvalid == 64) {
      typedef T V __attribute__((ext_vector_type(VE)));
      const V* src = reinterpret_cast<const V*>(lds);
      V* dst = reinterpret_cast<V*>(gdst);
#pragma unroll 4
      for (int g = lane; g < 64 * K / VE; g += 64) dst[g] = src[g];
      return;
    }
  }
  const int total = nvalid * K;
#pragma unroll 4
  for (int g = lane; g < total; g += 64) {
    int cfg = g / K;
    int r = g - cfg * K;
    gdst[g] = lds[cfg * KP + r];
  }
}

template <int K, class T>
RBD_DEV void staged_store(T* lds, const T (&vals)[K], T* gdst, int lane, int nvalid) {
  constexpr int KP = odd_pad<K>();
  __syncthreads();  // previous users of `lds` are done
  sfor<0, K>([&](auto I) { lds[lane * KP + decltype(I)::value] = vals[decltype(I)::value]; });
  __syncthreads();
  flush_tile<K>(lds, gdst, lane, nvalid);
}

// ---------------------------------------------------------------------------------------------
// rnea:  (q, qd, qdd) -> c [B,n], v, a, f [B,6,n]   (f = ACCUMULATED force, :619, :628)
// ---------------------------------------------------------------------------------------------

// Ordering point: the six values pass through an empty volatile asm with a memory clobber, so code
// that produces them stays above it and later memory reads stay below it.
RBD_DEV void pin6(float (&x)[6]) {
  asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]) : : "memory");
}
RBD_DEV void pin6(double (&x)[6]) {
  asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]) : : "memory");
}

// Robots with more than RNEA_REG_BODIES bodies cannot hold v, a, f of every body in registers
// (18 n values); for them v and a are parked in LDS tiles as they are produced and flushed after the
// forward pass, so only f (6 n) stays live across the backward pass.
constexpr int RNEA_REG_BODIES = 10;
constexpr bool RNEA_PARK_VA = N > RNEA_REG_BODIES;
template <class T>
constexpr bool rnea_two_tiles() { return 2ull * 64 * odd_pad<6 * N>() * sizeof(T) <= 150 * 1024; }
template <class T>
constexpr size_t rnea_lds_bytes(bool vaf) {
  if (!vaf && RNEA_PARK_VA && !rnea_two_tiles<T>()) return sizeof(T) * 64 * (size_t)odd_pad<6 * N>();   // f lives in LDS
  if (!vaf) return sizeof(T) * 64 * (size_t)odd_pad<N>();
  const size_t one = sizeof(T) * 64 * (size_t)odd_pad<6 * N>();
  return (RNEA_PARK_VA && rnea_two_tiles<T>()) ? 2 * one : one;
}

// flush_tile for a block of NT threads (tid = dense rank of the participating thread)
template <int K, int NT, class T>
RBD_DEV void flush_tile_nt(const T* lds, T* gdst, int tid, int nvalid) {
  constexpr int KP = odd_pad<K>();
  constexpr int VE = 16 / sizeof(T);
  if constexpr (KP == K && (64 * K) % VE == 0) {
    if (nvalid == 64) {                  // unpadded tile: the LDS image is the HBM image, flat 16-byte copies
      typedef T V __attribute__((ext_vector_type(VE)));
      const V* src = reinterpret_cast<const V*>(lds);
      V* dst = reinterpret_cast<V*>(gdst);
#pragma unroll 4
      for (int g = tid; g < 64 * K / VE; g += NT) dst[g] = src[g];
      return;
    }
  }
  const int total = nvalid * K;
#pragma unroll 4
  for (int g = tid; g < total; g += NT) {
    const int cfg = g / K;
    const int r = g - cfg * K;
    gdst[g] = lds[cfg * KP + r];
  }
}

#ifdef RBD_NEED_RNEA
template <class T, bool HAS_QDD, bool WITH_VAF>
__global__ __launch_bounds__(64) void rnea_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                  const T* __restrict__ qdd, T grav, long long B,
                                                  T* __restrict__ c_out, T* __restrict__ v_out,
                                                  T* __restrict__ a_out, T* __restrict__ f_out, int fpass_only) {
  // fpass_only != 0: the reference's rnea_fpass (:559-598) -- f stays LOCAL, no backward pass, no c
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x;
  const long long cfg0 = (long long)blockIdx.x * 64;
  const long long rem = B - cfg0;
  const int nvalid = rem < 64 ? (int)rem : 64;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);
  constexpr int K6 = 6 * N;
  constexpr int KP6 = odd_pad<K6>();

  JTrig<T> tr[N];
  T qv[N], qdv[N], qddv[N];
  sfor<0, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    qv[j] = q[b * N + j];
    qdv[j] = qd[b * N + j];
    if constexpr (HAS_QDD) qddv[j] = qdd[b * N + j]; else qddv[j] = T(0);
  });
  sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j] = make_trig<j>(qv[j]); });

  T v[N][6], a[N][6], f[N][6];
  const T zero6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  constexpr bool PARK = WITH_VAF && RNEA_PARK_VA;
  constexpr bool TWO = rnea_two_tiles<T>();
  // one tile only (fp64, big robots): f cannot stay in registers either (6 n doubles) -- it is
  // recomputed into the tile once v and a have left, and the backward pass runs on the LDS rows
  constexpr bool FLDS = RNEA_PARK_VA && !TWO;      // (c-only launches of such robots write f to the tile at once)
  constexpr bool FLDS_C = FLDS && !WITH_VAF;
  T* ldsV = lds + lane * KP6;
  T* ldsA = lds + (TWO ? 64 * KP6 : 0) + lane * KP6;
  sfor<0, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int p = PARENT[j];
    T xv[6], xa[6];
    if constexpr (p < 0)
      rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddv[j], grav, zero6, zero6, xv, xa, v[j], a[j], f[j]);
    else
      rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddv[j], grav, v[p], a[p], xv, xa, v[j], a[j], f[j]);
    if constexpr (PARK) {   // reference layout (6, NB): element [r][j]
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        ldsV[r * N + j] = v[j][r];
        if constexpr (TWO) ldsA[r * N + j] = a[j][r];
      });
      // keep the bodies in program order (bounds the live v/a set)
      if constexpr (FLDS) pin6(v[j]); else pin6(f[j]);
    }
    if constexpr (FLDS_C) {
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; ldsV[r * N + j] = f[j][r]; });
      pin6(v[j]);
    }
  });
  if constexpr (PARK) {
    __syncthreads();
    flush_tile<K6>(lds, v_out + cfg0 * K6, lane, nvalid);
    if constexpr (TWO) {
      flush_tile<K6>(lds + 64 * KP6, a_out + cfg0 * K6, lane, nvalid);
    } else {
      // a does not fit beside v: redo the cheap v/a recursion from laundered inputs and park a
      __syncthreads();
      T v2[N][6], a2[N][6];
      sfor<0, N>([&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr int p = PARENT[j];
        const JTrig<T> g{launder(tr[j].s), launder(tr[j].c)};
        T xv[6], xa[6], fdead[6];
        if constexpr (p < 0)
          rnea_fwd_body<j, HAS_QDD>(g, launder(qdv[j]), launder(qddv[j]), grav, zero6, zero6, xv, xa, v2[j], a2[j], fdead);
        else
          rnea_fwd_body<j, HAS_QDD>(g, launder(qdv[j]), launder(qddv[j]), grav, v2[p], a2[p], xv, xa, v2[j], a2[j], fdead);
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; ldsA[r * N + j] = a2[j][r]; });
      });
      __syncthreads();
      flush_tile<K6>(lds, a_out + cfg0 * K6, lane, nvalid);
      // third recursion: the local forces go into the (now idle) tile, lane-private rows
      __syncthreads();
      T v3[N][6], a3[N][6];
      sfor<0, N>([&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr int p = PARENT[j];
        const JTrig<T> g{launder(tr[j].s), launder(tr[j].c)};
        T xv[6], xa[6], f3[6];
        if constexpr (p < 0)
          rnea_fwd_body<j, HAS_QDD>(g, launder(qdv[j]), launder(qddv[j]), grav, zero6, zero6, xv, xa, v3[j], a3[j], f3);
        else
          rnea_fwd_body<j, HAS_QDD>(g, launder(qdv[j]), launder(qddv[j]), grav, v3[p], a3[p], xv, xa, v3[j], a3[j], f3);
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; ldsV[r * N + j] = f3[r]; });
        pin6(v3[j]);
      });
    }
  }
  // backward pass (:607-619)
  T c[N];
  sfor<0, N>([&](auto J) { c[decltype(J)::value] = T(0); });
  if (!fpass_only) {
    sfor_down<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int p = PARENT[j];
      if constexpr (FLDS) {
        T fj[6];
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; fj[r] = ldsV[r * N + j]; });
        c[j] = S_dot<j>(fj);
        if constexpr (p >= 0) {
          T t[6];
          xform_T<j>(tr[j], fj, t);
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; ldsV[r * N + p] += t[r]; });
        }
        pin6(fj);
      } else {
        c[j] = S_dot<j>(f[j]);
        if constexpr (p >= 0) {
          T t[6];
          xform_T<j>(tr[j], f[j], t);
          sfor<0, 6>([&](auto R) { f[p][decltype(R)::value] += t[decltype(R)::value]; });
        }
      }
    });
  }

  if constexpr (WITH_VAF) {
    // reference layout (6, NB) per configuration: element [r][i]
    T tmp[K6];
    if constexpr (!PARK) {
      sfor<0, N>([&](auto J) { sfor<0, 6>([&](auto R) { tmp[decltype(R)::value * N + decltype(J)::value] = v[decltype(J)::value][decltype(R)::value]; }); });
      staged_store<K6>(lds, tmp, v_out + cfg0 * K6, lane, nvalid);
      sfor<0, N>([&](auto J) { sfor<0, 6>([&](auto R) { tmp[decltype(R)::value * N + decltype(J)::value] = a[decltype(J)::value][decltype(R)::value]; }); });
      staged_store<K6>(lds, tmp, a_out + cfg0 * K6, lane, nvalid);
    }
    if constexpr (FLDS) {
      __syncthreads();
      flush_tile<K6>(lds, f_out + cfg0 * K6, lane, nvalid);
    } else {
      sfor<0, N>([&](auto J) { sfor<0, 6>([&](auto R) { tmp[decltype(R)::value * N + decltype(J)::value] = f[decltype(J)::value][decltype(R)::value]; }); });
      staged_store<K6>(lds, tmp, f_out + cfg0 * K6, lane, nvalid);
    }
  }
  if (c_out != nullptr) staged_store<N>(lds, c, c_out + cfg0 * N, lane, nvalid);
}

// rnea_bpass (:600-621): f [B,6,n] in -> c [B,n] and f accumulated IN PLACE, as the reference does.
// (debug / per-pass surface of README.md:19; one configuration per lane, plain strided reads)
template <class T>
__global__ __launch_bounds__(64) void rnea_bpass_kernel(const T* __restrict__ q, T* __restrict__ f_io, long long B,
                                                        T* __restrict__ c_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x;
  const long long cfg0 = (long long)blockIdx.x * 64;
  const long long rem = B - cfg0;
  const int nvalid = rem < 64 ? (int)rem : 64;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);
  constexpr int K6 = 6 * N;
  JTrig<T> tr[N];
  T qv[N];
  sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; qv[j] = q[b * N + j]; });
  sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j] = make_trig<j>(qv[j]); });
  T c[N];
  if constexpr (6 * N * sizeof(T) > 6 * 16 * sizeof(double)) {
    // Big robots (30 bodies in fp64: 6 n doubles = 360 registers, scratch): f never lives in registers as a whole.  The
    // block's [64][6 n] tensor comes in through the LDS image (coalesced), every lane accumulates child -> parent on its
    // own image row (stride odd: conflict-free), and the image leaves as it is.
    constexpr int KP = odd_pad<K6>();
    for (int g = lane; g < nvalid * K6; g += 64) {
      const int cfg = g / K6;
      lds[cfg * KP + (g - cfg * K6)] = f_io[cfg0 * K6 + g];
    }
    __syncthreads();
    T* mine = lds + lane * KP;
    sfor_down<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int p = PARENT[j];
      T fj[6];
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; fj[r] = mine[r * N + j]; });
      c[j] = S_dot<j>(fj);
      if constexpr (p >= 0) {
        T t[6];
        xform_T<j>(tr[j], fj, t);
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; mine[r * N + p] += t[r]; });
      }
    });
    __syncthreads();
    flush_tile<K6>(lds, f_io + cfg0 * K6, lane, nvalid);
    staged_store<N>(lds, c, c_out + cfg0 * N, lane, nvalid);
  } else {
    T f[N][6];
    sfor<0, N>([&](auto J) { sfor<0, 6>([&](auto R) { constexpr int j = decltype(J)::value, r = decltype(R)::value; f[j][r] = f_io[b * K6 + r * N + j]; }); });
    sfor_down<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int p = PARENT[j];
      c[j] = S_dot<j>(f[j]);
      if constexpr (p >= 0) {
        T t[6];
        xform_T<j>(tr[j], f[j], t);
        sfor<0, 6>([&](auto R) { f[p][decltype(R)::value] += t[decltype(R)::value]; });
      }
    });
    __syncthreads();   // every lane has read its f before the tile is written back
    T tmp[K6];
    sfor<0, N>([&](auto J) { sfor<0, 6>([&](auto R) { tmp[decltype(R)::value * N + decltype(J)::value] = f[decltype(J)::value][decltype(R)::value]; }); });
    staged_store<K6>(lds, tmp, f_io + cfg0 * K6, lane, nvalid);
    staged_store<N>(lds, c, c_out + cfg0 * N, lane, nvalid);
  }
}

#endif  // RBD_NEED_RNEA

// ---------------------------------------------------------------------------------------------
// rnea_grad: (q, qd, qdd) -> c [B,n] (optional), dc_du = [dc_dq | dc_dqd]  [B, n, 2n]  (:1345-1368)
// ---------------------------------------------------------------------------------------------
constexpr int GRAD_ROW = 2 * N;               // one output row: [dc_dq[i,:] | dc_dqd[i,:]]
constexpr int GRAD_TILE = 2 * N * N;          // floats per configuration
// Independent roots (e.g. the four legs of the quadruped) are processed one after the other; when
// every root's subtree occupies a contiguous index range (DFS numbering) each root is a "group"
// whose output rows are staged and flushed on their own, so the LDS tile only has to hold the
// largest group.  Otherwise the whole robot is one group.
constexpr int root_of(int j) {
  while (PARENT[j] != -1) j = PARENT[j];
  return j;
}
constexpr int last_of_root(int r) {
  int m = r;
  for (int j = 0; j < N; ++j)
    if (root_of(j) == r) m = j;
  return m;
}
constexpr bool roots_contiguous() {
  for (int j = 0; j < N; ++j) {
    const int r = root_of(j);
    for (int k = r; k <= last_of_root(r); ++k)
      if (root_of(k) != r) return false;
  }
  return true;
}
constexpr bool GRAD_PER_ROOT = roots_contiguous();
constexpr bool grp_head(int rt) { return GRAD_PER_ROOT ? PARENT[rt] == -1 : rt == 0; }
constexpr bool grp_has(int rt, int j) { return GRAD_PER_ROOT ? root_of(j) == rt : true; }
constexpr int grp_row0(int rt) { return GRAD_PER_ROOT ? rt : 0; }
constexpr int grp_rows(int rt) { return GRAD_PER_ROOT ? last_of_root(rt) - rt + 1 : N; }
constexpr int grp_first() {
  for (int r = 0; r < N; ++r)
    if (grp_head(r)) return r;
  return 0;
}
constexpr int grp_next(int rt) {   // next group head after rt, -1 if none
  for (int r = rt + 1; r < N; ++r)
    if (grp_head(r)) return r;
  return -1;
}
constexpr int grp_index(int rt) {   // ordinal of group head rt
  int k = 0;
  for (int x = 0; x < rt; ++x) k += grp_head(x) ? 1 : 0;
  return k;
}
constexpr int n_groups() {
  int k = 0;
  for (int x = 0; x < N; ++x) k += grp_head(x) ? 1 : 0;
  return k;
}

// ---- segments: big sibling subtrees of a root subtree ("limbs": Atlas' two arms under the third back joint)
// and what is left of it ("stem").  Limbs are independent of each other between their parent's forward
// quantities and what they hand back to it, so kernels for small batches give each its own wave.
constexpr int LIMB_MIN = 4;
// evaluated ONCE into a table (the queries below are used inside nested constexpr loops; recomputed from the
// parent array every time they exceed the compiler's constant-evaluation step limit at n = 30)
struct LimbTable {
  int limb_of[N];      // head of the limb body j belongs to, -1: a stem body
  bool head[N];
};
constexpr LimbTable make_limb_table() {
  LimbTable t{};
  int size[N] = {};
  for (int j = N - 1; j >= 0; --j) {        // parents precede children: one reverse pass gives the subtree sizes
    size[j] += 1;
    if (PARENT[j] >= 0) size[PARENT[j]] += size[j];
  }
  bool cand[N] = {};
  for (int j = 0; j < N; ++j) {
    int big = 0;
    if (PARENT[j] >= 0)
      for (int x = 0; x < N; ++x) big += (PARENT[x] == PARENT[j] && size[x] >= LIMB_MIN) ? 1 : 0;
    cand[j] = PARENT[j] >= 0 && size[j] >= LIMB_MIN && big >= 2;
  }
  for (int j = 0; j < N; ++j) {             // outermost candidates only: a limb is not split again
    const int p = PARENT[j];
    const int up = p >= 0 ? t.limb_of[p] : -1;
    t.head[j] = cand[j] && up == -1;
    t.limb_of[j] = up != -1 ? up : (t.head[j] ? j : -1);
  }
  return t;
}
constexpr LimbTable LIMBS = make_limb_table();
constexpr bool limb_head(int j) { return LIMBS.head[j]; }
constexpr int limb_of(int j) { return LIMBS.limb_of[j]; }
constexpr bool seg_head(int j) { return PARENT[j] == -1 || limb_head(j); }
constexpr bool seg_has(int h, int j) { return limb_head(h) ? limb_of(j) == h : (root_of(j) == h && limb_of(j) == -1); }
constexpr int seg_index(int h) {
  int k = 0;
  for (int x = 0; x < h; ++x) k += seg_head(x) ? 1 : 0;
  return k;
}
constexpr int n_segs() {
  int k = 0;
  for (int x = 0; x < N; ++x) k += seg_head(x) ? 1 : 0;
  return k;
}
constexpr int n_limbs() {
  int k = 0;
  for (int x = 0; x < N; ++x) k += limb_head(x) ? 1 : 0;
  return k;
}
constexpr int limb_index(int h) {
  int k = 0;
  for (int x = 0; x < h; ++x) k += limb_head(x) ? 1 : 0;
  return k;
}
constexpr bool stem_has_limbs(int h) {      // h: a root
  for (int x = 0; x < N; ++x)
    if (limb_head(x) && root_of(x) == h) return true;
  return false;
}
constexpr int limbs_of_root(int rt) {
  int k = 0;
  for (int x = 0; x < N; ++x) k += (limb_head(x) && root_of(x) == rt) ? 1 : 0;
  return k;
}
constexpr int max_limbs_per_root() {
  int m = 0;
  for (int x = 0; x < N; ++x)
    if (PARENT[x] == -1 && limbs_of_root(x) > m) m = limbs_of_root(x);
  return m;
}
constexpr int kth_limb(int rt, int k) {     // head of the k-th limb (k = 0, 1, ...) of root subtree rt, -1 if none
  for (int x = 0; x < N; ++x)
    if (limb_head(x) && root_of(x) == rt) {
      if (k == 0) return x;
      --k;
    }
  return -1;
}
constexpr int grad_max_rows() {
  int m = 0;
  for (int r = 0; r < N; ++r)
    if (grp_head(r) && grp_rows(r) > m) m = grp_rows(r);
  return m;
}
// LDS stride between configurations: == 2 (mod 32) keeps the 16 configurations of a 32-lane
// LDS group on distinct even banks; the odd lane's +N offset lands on the odd banks when N is odd.
constexpr int grad_tile_stride() {
  int s = grad_max_rows() * GRAD_ROW;
  while (s % 32 != 2) ++s;
  return s;
}
constexpr int GRAD_TS = grad_tile_stride();
// Configurations per block (two lanes each): 32 = one full wave when the LDS tiles fit, otherwise
// the largest power of two whose tiles fit in 160 KiB (big robots: partially filled wave, slow but
// correct -- DESIGN.md "next").
template <class T>
constexpr int grad_cfgs() {
  int c = 32;
  while (c > 1 && (long long)c * GRAD_TS * (long long)sizeof(T) > 160 * 1024) c /= 2;
  return c;
}

// dc[i, c] is structurally non-zero only for related (ancestor/descendant) bodies.  Up to 72 such
// pairs per lane are accumulated in registers; bigger robots accumulate in the LDS tile instead
// (plain read-add-write: every address has exactly one owner lane; LDS float atomics measured ~4x
// slower than that).
constexpr int related_pairs() {
  int k = 0;
  for (int i = 0; i < N; ++i)
    for (int c = 0; c < N; ++c) k += related(i, c) ? 1 : 0;
  return k;
}
constexpr bool GRAD_ACC_IN_REGS = related_pairs() <= 72;

#ifdef RBD_NEED_RNEA
// ---------------------------------------------------------------------------------------------
// rnea for robots with several independent root subtrees ("groups", e.g. Atlas: torso + arms, left leg,
// right leg; the quadruped's four legs): one WAVE per group inside a block of 64 configurations.  A launch of
// B = 16 384 Atlas configurations is 256 single-wave blocks with one lane per configuration -- one wave
// per CU, its whole 30-body forward / backward recursion a single dependent instruction stream.  The
// groups never exchange anything (:576, :614: independent roots), so wave g runs the recursions of group g
// only: the serial length drops to the largest group (18 of 30 bodies).  The waves share the LDS
// images of the (6, NB) outputs -- every wave fills its group's columns -- so the rows still leave as
// whole coalesced rows (giving each group its own BLOCK instead had turned the stores into 24-72-byte
// segments and was slower).
// ---------------------------------------------------------------------------------------------
constexpr int RG_WAVES = n_groups();
template <class T>
constexpr bool rnea_groups_ok() {
  return GRAD_PER_ROOT && RG_WAVES > 1 && RG_WAVES <= 4 && 2ull * 64 * odd_pad<6 * N>() * sizeof(T) <= 150 * 1024;
}

template <class T, bool HAS_QDD>
__global__ __launch_bounds__(64 * RG_WAVES) void rnea_groups_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                                   const T* __restrict__ qdd, T grav, long long B,
                                                                   T* __restrict__ c_out, T* __restrict__ v_out,
                                                                   T* __restrict__ a_out, T* __restrict__ f_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int K6 = 6 * N, KP6 = odd_pad<K6>(), KPN = odd_pad<N>(), NT = 64 * RG_WAVES;
  T* tileV = reinterpret_cast<T*>(smem_raw);          // [64][KP6]  v, later the accumulated f
  T* tileA = tileV + 64 * KP6;                        // [64][KP6]  a, later c ([64][KPN])
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const long long cfg0 = (long long)blockIdx.x * 64;
  const long long rem = B - cfg0;
  const int nvalid = rem < 64 ? (int)rem : 64;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);
  T* myV = tileV + lane * KP6;
  T* myA = tileA + lane * KP6;

  JTrig<T> tr[N];
  T f[N][6];
  const T zero6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  // ---- forward pass of this wave's group (:569-596); v, a go straight to the shared images ------------
  sfor<0, N>([&](auto Rt) {
    constexpr int rt = decltype(Rt)::value;
    if constexpr (grp_head(rt)) {
      if (wave == grp_index(rt)) {
        T qv[N], qdv[N], qddv[N];
        sfor<0, N>([&](auto J) {
          constexpr int j = decltype(J)::value;
          if constexpr (grp_has(rt, j)) {
            qv[j] = q[b * N + j];
            qdv[j] = qd[b * N + j];
            if constexpr (HAS_QDD) qddv[j] = qdd[b * N + j]; else qddv[j] = T(0);
          }
        });
        sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; if constexpr (grp_has(rt, j)) tr[j] = make_trig<j>(qv[j]); });
        T v[N][6], a[N][6];
        sfor<0, N>([&](auto J) {
          constexpr int j = decltype(J)::value;
          constexpr int p = PARENT[j];
          if constexpr (grp_has(rt, j)) {
            T xv[6], xa[6];
            if constexpr (p < 0)
              rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddv[j], grav, zero6, zero6, xv, xa, v[j], a[j], f[j]);
            else
              rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddv[j], grav, v[p], a[p], xv, xa, v[j], a[j], f[j]);
            sfor<0, 6>([&](auto R) {       // reference layout (6, NB): element [r][j]
              constexpr int r = decltype(R)::value;
              myV[r * N + j] = v[j][r];
              myA[r * N + j] = a[j][r];
            });
            pin6(f[j]);                    // keep the bodies in program order (bounds the live v / a set)
          }
        });
      }
    }
  });
  __syncthreads();
  flush_tile_nt<K6, NT>(tileV, v_out + cfg0 * K6, tid, nvalid);
  flush_tile_nt<K6, NT>(tileA, a_out + cfg0 * K6, tid, nvalid);
  __syncthreads();
  // ---- backward pass of the group (:607-619): c, accumulated f ------------------------------------------
  T* myC = tileA + lane * KPN;
  sfor<0, N>([&](auto Rt) {
    constexpr int rt = decltype(Rt)::value;
    if constexpr (grp_head(rt)) {
      if (wave == grp_index(rt)) {
        sfor_down<0, N>([&](auto J) {
          constexpr int j = decltype(J)::value;
          constexpr int p = PARENT[j];
          if constexpr (grp_has(rt, j)) {
            myC[j] = S_dot<j>(f[j]);
            if constexpr (p >= 0) {
              T t[6];
              xform_T<j>(tr[j], f[j], t);
              sfor<0, 6>([&](auto R) { f[p][decltype(R)::value] += t[decltype(R)::value]; });
            }
            sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; myV[r * N + j] = f[j][r]; });
          }
        });
      }
    }
  });
  __syncthreads();
  flush_tile_nt<K6, NT>(tileV, f_out + cfg0 * K6, tid, nvalid);
  if (c_out != nullptr) flush_tile_nt<N, NT>(tileA, c_out + cfg0 * N, tid, nvalid);
}


// ---------------------------------------------------------------------------------------------
// rnea with one wave per SEGMENT: the group waves above still run a group's whole recursion as one serial
// stream (Atlas torso: 18 forward + 18 backward body steps).  Big sibling subtrees of a group ("limbs": the two
// arms under the third back joint) are independent of each other between the moment their parent's v, a
// exist and the moment their root force goes back to it (:576-581, :618-619), so each limb gets its own
// wave of the block and the group's remaining bodies ("stem") another:
//   phase 1   stems: forward of the SPINE (the bodies above a limb)     |  limb-less groups: their first steps
//   phase 2   limbs: forward (parent's v, a from the shared LDS image)  |  stems: the rest (Atlas' neck), forward
//             and backward  |  limb-less groups: the rest of forward, then backward steps
//   phase 3   limbs: backward, X^T f of the limb's root parked for the parent  |  other waves: leftover backward
//             steps, then stream v out
//   phase 4   stems: parked limb forces added, backward of the spine    |  other waves stream a out
//   phase 5   every wave: f and c out
// Every wave loads its inputs and computes its sin / cos before the first barrier.
// Atlas: 3 + 7 + 7 + 3 = 20 serial body steps instead of 36, and two thirds of the output leave while the
// backward passes run: at B = 16 384 (one block per CU, so no other block to overlap with) the launch was
// 6 us of recursion followed by 6 us of stores at HBM speed.  v, a, f have an LDS image each.
// ---------------------------------------------------------------------------------------------
constexpr int RS_WAVES = n_segs();
// rank of segment h among the waves that stream v out in phase 3 (every segment but the limbs; STEMS = false)
// or a out in phase 4 (every segment but the stems that wait for limbs; STEMS = true)
template <bool STEMS>
constexpr int seg_rank(int h) {
  int k = 0;
  for (int x = 0; x < h; ++x) {
    if (!seg_head(x)) continue;
    const bool out = STEMS ? (PARENT[x] == -1 && stem_has_limbs(x)) : limb_head(x);
    k += out ? 0 : 1;
  }
  return k;
}
constexpr int n_busy_stems() {             // stems that wait for limbs: the waves that work in phase 3
  int k = 0;
  for (int x = 0; x < N; ++x) k += (PARENT[x] == -1 && stem_has_limbs(x)) ? 1 : 0;
  return k;
}
// the stem's spine: the stem bodies above a limb (everything the limbs wait for)
constexpr bool rs_spine(int j) {
  if (limb_of(j) != -1) return false;
  for (int x = 0; x < N; ++x)
    if (limb_head(x) && is_anc_or_self(j, PARENT[x])) return true;
  return false;
}
constexpr int rs_budget1() {                // longest spine: the length of phase 1
  int m = 0;
  for (int r = 0; r < N; ++r)
    if (PARENT[r] == -1) {
      int k = 0;
      for (int j = 0; j < N; ++j) k += (root_of(j) == r && rs_spine(j)) ? 1 : 0;
      m = k > m ? k : m;
    }
  return m;
}
constexpr int rs_budget2() {                // longest limb: the length of phase 2
  int m = 0;
  for (int h = 0; h < N; ++h)
    if (limb_head(h)) {
      int k = 0;
      for (int j = 0; j < N; ++j) k += limb_of(j) == h ? 1 : 0;
      m = k > m ? k : m;
    }
  return m;
}
constexpr int rs_rows(int h) {              // bodies of segment h
  int k = 0;
  for (int j = 0; j < N; ++j) k += seg_has(h, j) ? 1 : 0;
  return k;
}
constexpr int rs_body(int h, int k) {       // k-th body of segment h
  for (int j = 0; j < N; ++j)
    if (seg_has(h, j)) {
      if (k == 0) return j;
      --k;
    }
  return -1;
}
constexpr int RS_PARK = 7;                  // 6 scalars per (limb, configuration), odd stride
template <class T>
constexpr size_t rnea_segs_lds() {
  return sizeof(T) * 64 * ((size_t)3 * odd_pad<6 * N>() + odd_pad<N>() + (size_t)(n_limbs() > 0 ? n_limbs() : 1) * RS_PARK);
}
template <class T>
constexpr bool rnea_segs_ok() { return n_limbs() >= 2 && RS_WAVES <= 8 && rnea_segs_lds<T>() <= 156 * 1024; }

template <class T, bool HAS_QDD>
__global__ __launch_bounds__(64 * RS_WAVES) void rnea_segments_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                                     const T* __restrict__ qdd, T grav, long long B,
                                                                     T* __restrict__ c_out, T* __restrict__ v_out,
                                                                     T* __restrict__ a_out, T* __restrict__ f_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int K6 = 6 * N, KP6 = odd_pad<K6>(), KPN = odd_pad<N>(), NT = 64 * RS_WAVES;
  T* tileV = reinterpret_cast<T*>(smem_raw);          // [64][KP6]  v
  T* tileA = tileV + 64 * KP6;                        // [64][KP6]  a
  T* tileF = tileA + 64 * KP6;                        // [64][KP6]  accumulated f
  T* tileC = tileF + 64 * KP6;                        // [64][KPN]  c
  T* park = tileC + 64 * KPN;                         // [limb][64][RS_PARK]  X^T f of a limb's root, for its parent
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const long long cfg0 = (long long)blockIdx.x * 64;
  const long long rem = B - cfg0;
  const int nvalid = rem < 64 ? (int)rem : 64;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);
  T* myV = tileV + lane * KP6;
  T* myA = tileA + lane * KP6;
  T* myF = tileF + lane * KP6;
  T* myC = tileC + lane * KPN;

  JTrig<T> tr[N];
  T qdv[N], qddv[N];
  T v[N][6], a[N][6], f[N][6];
  const T zero6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  // inputs and sin / cos of segment H's joints: done by every wave before the first barrier (the limbs' waves
  // have nothing else to do in phase 1)
  auto load_inputs = [&](auto H) {
    constexpr int h = decltype(H)::value;
    T qv[N];
    sfor<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (seg_has(h, j)) {
        qv[j] = q[b * N + j];
        qdv[j] = qd[b * N + j];
        if constexpr (HAS_QDD) qddv[j] = qdd[b * N + j]; else qddv[j] = T(0);
      }
    });
    sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; if constexpr (seg_has(h, j)) tr[j] = make_trig<j>(qv[j]); });
  };
  // forward step of body J of segment H (:569-596); a limb's root takes its parent's v, a from the images
  auto fwd_body = [&](auto H, auto J) {
    constexpr int h = decltype(H)::value, j = decltype(J)::value;
    constexpr int p = PARENT[j];
    T xv[6], xa[6];
    if constexpr (p < 0) {
      rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddv[j], grav, zero6, zero6, xv, xa, v[j], a[j], f[j]);
    } else if constexpr (!seg_has(h, p)) {          // the limb's root: its parent lives in the stem's wave
      T vp[6], ap[6];
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; vp[r] = myV[r * N + p]; ap[r] = myA[r * N + p]; });
      rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddv[j], grav, vp, ap, xv, xa, v[j], a[j], f[j]);
    } else {
      rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddv[j], grav, v[p], a[p], xv, xa, v[j], a[j], f[j]);
    }
    sfor<0, 6>([&](auto R) {       // reference layout (6, NB): element [r][j]
      constexpr int r = decltype(R)::value;
      myV[r * N + j] = v[j][r];
      myA[r * N + j] = a[j][r];
    });
    pin6(f[j]);                    // keep the bodies in program order (bounds the live v / a set)
  };
  // backward step of body J of segment H (:607-619)
  auto bwd_body = [&](auto H, auto J) {
    constexpr int h = decltype(H)::value, j = decltype(J)::value;
    constexpr int p = PARENT[j];
    myC[j] = S_dot<j>(f[j]);
    if constexpr (p >= 0) {
      T t[6];
      xform_T<j>(tr[j], f[j], t);
      if constexpr (seg_has(h, p)) {
        sfor<0, 6>([&](auto R) { f[p][decltype(R)::value] += t[decltype(R)::value]; });
      } else {                                        // limb root -> parked for the stem
        constexpr int li = limb_index(h);           // (bound to a constant: a constexpr call in a run-time expression is not folded)
        T* pk = park + (li * 64 + lane) * RS_PARK;
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; pk[r] = t[r]; });
      }
    }
    sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; myF[r * N + j] = f[j][r]; });
  };
  // Every wave walks its own segment's timeline (same number of barriers on every path): kept as ONE branch
  // per wave so that a segment's registers (its f and sin / cos across the phases) are not live through the
  // code of the other segments -- phase-by-phase dispatch had put the whole block's state into one wave's
  // register file (256 VGPRs and spills instead of 111).
  //   stem   1: forward of its SPINE (the bodies above a limb)   2: the other stem bodies, forward and backward
  //          3: streams v   4: parked limb forces, backward of the spine
  //   limb   1: --   2: forward   3: backward, root force parked   4: streams a
  //   other  its 2 x rows steps (forward, then backward) spread over phases 1-3 with the stem's / limbs' step
  //          counts as budgets (all forward steps by the end of phase 2), then streams v;  4: streams a
  constexpr int NT3 = 64 * (RS_WAVES - n_limbs()), NT4 = 64 * (RS_WAVES - n_busy_stems());
  constexpr int B1 = rs_budget1(), B2 = rs_budget2();
  sfor<0, N>([&](auto H) {
    constexpr int h = decltype(H)::value;
    if constexpr (seg_head(h)) {
      constexpr int si = seg_index(h);
      constexpr bool limb = limb_head(h);
      constexpr bool stem = !limb && stem_has_limbs(h);
      constexpr int rank3 = seg_rank<false>(h), rank4 = seg_rank<true>(h);   // among the waves that stream in phase 3 / 4
      if (wave == si) {
        load_inputs(H);
        if constexpr (limb) {
          __syncthreads();
          sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; if constexpr (seg_has(h, j)) fwd_body(H, J); });
          __syncthreads();
          sfor_down<0, N>([&](auto J) { constexpr int j = decltype(J)::value; if constexpr (seg_has(h, j)) bwd_body(H, J); });
          __syncthreads();
          flush_tile_nt<K6, NT4>(tileA, a_out + cfg0 * K6, rank4 * 64 + lane, nvalid);
        } else if constexpr (stem) {
          sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; if constexpr (seg_has(h, j) && rs_spine(j)) fwd_body(H, J); });
          __syncthreads();
          sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; if constexpr (seg_has(h, j) && !rs_spine(j)) fwd_body(H, J); });
          sfor_down<0, N>([&](auto J) { constexpr int j = decltype(J)::value; if constexpr (seg_has(h, j) && !rs_spine(j)) bwd_body(H, J); });
          __syncthreads();
          flush_tile_nt<K6, NT3>(tileV, v_out + cfg0 * K6, rank3 * 64 + lane, nvalid);
          __syncthreads();
          sfor_down<0, N>([&](auto L) {       // descending, the order in which the reference's loop adds them (:618)
            constexpr int l = decltype(L)::value;
            if constexpr (limb_head(l) && root_of(l) == h) {
              constexpr int li = limb_index(l), pl = PARENT[l];
              const T* pk = park + (li * 64 + lane) * RS_PARK;
              sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; f[pl][r] += pk[r]; });
            }
          });
          sfor_down<0, N>([&](auto J) { constexpr int j = decltype(J)::value; if constexpr (seg_has(h, j) && rs_spine(j)) bwd_body(H, J); });
        } else {
          // a limb-less root subtree: step k < rows is the forward step of its k-th body, step rows + k the backward
          // step of its (rows - 1 - k)-th body
          constexpr int rows = rs_rows(h);
          constexpr int e1 = B1 < rows ? B1 : rows;                                  // steps [0, e1) in phase 1
          constexpr int e2 = (e1 + B2 > rows ? e1 + B2 : rows) < 2 * rows ? (e1 + B2 > rows ? e1 + B2 : rows) : 2 * rows;   // [e1, e2) in phase 2
          auto steps = [&](auto A, auto E) {
            constexpr int a0 = decltype(A)::value, e0 = decltype(E)::value;
            sfor<a0, e0>([&](auto K) {
              constexpr int k = decltype(K)::value;
              if constexpr (k < rows) fwd_body(H, std::integral_constant<int, rs_body(h, k)>{});
              else bwd_body(H, std::integral_constant<int, rs_body(h, 2 * rows - 1 - k)>{});
            });
          };
          steps(std::integral_constant<int, 0>{}, std::integral_constant<int, e1>{});
          __syncthreads();
          steps(std::integral_constant<int, e1>{}, std::integral_constant<int, e2>{});
          __syncthreads();
          steps(std::integral_constant<int, e2>{}, std::integral_constant<int, 2 * rows>{});
          flush_tile_nt<K6, NT3>(tileV, v_out + cfg0 * K6, rank3 * 64 + lane, nvalid);
          __syncthreads();
          flush_tile_nt<K6, NT4>(tileA, a_out + cfg0 * K6, rank4 * 64 + lane, nvalid);
        }
      }
    }
  });
  __syncthreads();
  // ---- phase 5: f and c --------------------------------------------------------------------------------------
  flush_tile_nt<K6, NT>(tileF, f_out + cfg0 * K6, tid, nvalid);
  if (c_out != nullptr) flush_tile_nt<N, NT>(tileC, c_out + cfg0 * N, tid, nvalid);
}

#endif  // RBD_NEED_RNEA (group waves)

#ifdef RBD_NEED_GRAD
template <class T>
RBD_DEV T from_odd_lane(T x);
template <>
RBD_DEV float from_odd_lane<float>(float x) {
  // DPP quad_perm [1,1,3,3]: every lane reads the odd lane of its pair
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xF5, 0xF, 0xF, true));
}
template <>
RBD_DEV double from_odd_lane<double>(double x) {
  unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  int lo = __builtin_amdgcn_mov_dpp((int)(u & 0xffffffffu), 0xF5, 0xF, 0xF, true);
  int hi = __builtin_amdgcn_mov_dpp((int)(u >> 32), 0xF5, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// DPP quad_perm [0,0,2,2]: every lane reads the even lane of its pair
RBD_DEV float from_even_lane(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xA0, 0xF, 0xF, true));
}
RBD_DEV double from_even_lane(double x) {
  unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  int lo = __builtin_amdgcn_mov_dpp((int)(u & 0xffffffffu), 0xA0, 0xF, 0xF, true);
  int hi = __builtin_amdgcn_mov_dpp((int)(u >> 32), 0xA0, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// The two lanes of a configuration need the same sin/cos of every joint: for two consecutive bodies of
// the same joint type the even lane evaluates the first, the odd lane the second (one sincos call in
// the instruction stream instead of two) and the results are exchanged with DPP.
template <int G, int J0, class T>
RBD_DEV void pair_trig(bool isqd, const T (&qv)[N], JTrig<T> (&tr)[N]) {
  if constexpr (J0 < N) {
    if constexpr (!grp_has(G, J0)) {
      pair_trig<G, J0 + 1>(isqd, qv, tr);
    } else if constexpr (J0 + 1 < N && grp_has(G, J0 + 1) && JTYPE[J0] == JTYPE[J0 + 1]) {
      const JTrig<T> mine = make_trig<J0>(sel(isqd, qv[J0 + 1], qv[J0]));
      tr[J0].s = from_even_lane(mine.s); tr[J0].c = from_even_lane(mine.c);
      tr[J0 + 1].s = from_odd_lane(mine.s); tr[J0 + 1].c = from_odd_lane(mine.c);
      pair_trig<G, J0 + 2>(isqd, qv, tr);
    } else {
      tr[J0] = make_trig<J0>(qv[J0]);
      pair_trig<G, J0 + 1>(isqd, qv, tr);
    }
  }
}

// Occupancy request (waves per SIMD) for the register allocator.  fp32 needs ~178 VGPRs at n = 7:
// asking for 3 waves (168) makes the allocator spill ~17 values to scratch, which measured +45 % HBM
// traffic (PMC) for no sustained speed-up; 2 waves run spill-free.
// Robots whose LDS tile leaves room for one block per CU only (Atlas: 139 KB) get the whole register
// file: asking for 2 waves there bought nothing and spilled 140 values.
template <class T>
constexpr int grad_min_waves() {
#ifdef GRAD_MIN_WAVES
  return GRAD_MIN_WAVES;
#else
  if ((long long)grad_cfgs<T>() * GRAD_TS * (long long)sizeof(T) > 80 * 1024) return 1;
  return sizeof(T) == 4 ? 2 : 1;
#endif
}


// FDG = forward_dynamics_grad epilogue (:1376-1384): the accumulated dc_du block of each lane is
// multiplied by -Minv (read from `minv_in`, [B, n, n], prefetched into LDS at kernel start) before
// it is parked, so  [qdd_dq | qdd_dqd] = -Minv [dc_dq | dc_dqd]  costs no extra HBM round trip.
// Only available when the accumulators live in registers (GRAD_ACC_IN_REGS).
template <class T, bool HAS_QDD, bool FDG>
__global__ __launch_bounds__(2 * grad_cfgs<T>(), grad_min_waves<T>()) void rnea_grad_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                       const T* __restrict__ qdd, T grav, int use_damping,
                                                       long long B, T* __restrict__ c_out,
                                                       T* __restrict__ dcdu, const T* __restrict__ minv_in, int split) {
  static_assert(!FDG || GRAD_ACC_IN_REGS, "fused -Minv epilogue needs register accumulators");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x;
  const int slot = lane >> 1;
  const bool isqd = (lane & 1) != 0;
  constexpr int CFGS = grad_cfgs<T>();
  constexpr int NT = 2 * CFGS;
  // split == 1: every group of the robot in this block, one after the other; split == number of
  // groups: the block handles group blockIdx.x % split of configurations blockIdx.x / split (the
  // group is the FAST index, so that the blocks that write neighbouring rows of a configuration run
  // close in time and L2 can merge their partial cache lines)
  // XCD-aware: blocks x and x + 8 share an XCD (observed dispatch order; speed only, never correctness), and each XCD has
  // its own L2.  The `split` blocks of one batch of configurations read the same input rows and write neighbouring
  // rows of the same matrices, so they are given indices 8 apart: the inputs are fetched into ONE L2 instead of
  // `split` (quadruped fp64 B = 65 536: FETCH_SIZE x 2 was 75.8 MB for 18.9 MB of inputs) and partial lines of
  // neighbouring rows meet in the same L2.  Index within the XCD = (configuration block, group), group fastest.
#ifdef RBD_EXP_NO_XCD_MAP
  const long long cblk = blockIdx.x / split;
  const int gsel_x = (int)(blockIdx.x % split);
#else
  const long long kx = blockIdx.x >> 3;
  const long long cblk = split > 1 ? (kx / split) * 8 + (blockIdx.x & 7) : blockIdx.x;
  const int gsel_x = split > 1 ? (int)(kx % split) : 0;
  if (cblk * CFGS >= B) return;          // (the grid is rounded up to a multiple of 8 configuration blocks; no barrier has been passed)
#endif
  const long long cfg0 = cblk * CFGS;
  const long long rem = B - cfg0;
  const int nvalid = rem < CFGS ? (int)rem : CFGS;
  const long long b = cfg0 + (slot < nvalid ? slot : nvalid - 1);
  T* mtile = tile + CFGS * GRAD_TS;      // FDG: [CFGS][N*N] copy of Minv; with one block per group: [CFGS][FDG_MS] copy of that group's rows
  constexpr int FDG_MS = (grad_max_rows() * N) | 1;      // odd stride: the lanes' rows start in different banks
  if constexpr (FDG) {
    // one block for all groups: the block's whole Minv rows; one block per group (split): only that group's rows, below
    if (split <= 1) {
      const T* msrc = minv_in + cfg0 * (N * N);
      // batches of 16 loads in flight, then their LDS writes (not load -> write per iteration: one HBM latency each)
      constexpr int NL = (CFGS * N * N + NT - 1) / NT;
      sfor<0, (NL + 15) / 16>([&](auto C_) {
        constexpr int c0 = decltype(C_)::value * 16, c1 = c0 + 16 < NL ? c0 + 16 : NL;
        T mb[16];
        sfor<c0, c1>([&](auto I_) { constexpr int i_ = decltype(I_)::value; const int g = lane + i_ * NT; mb[i_ - c0] = msrc[g < nvalid * N * N ? g : 0]; });
        sfor<c0, c1>([&](auto I_) { constexpr int i_ = decltype(I_)::value; const int g = lane + i_ * NT; if (g < CFGS * N * N) mtile[g] = mb[i_ - c0]; });
      });
    }
  }

  // Independent roots (the quadruped's four legs) are processed one after the other -- loads, RNEA
  // passes, gradient sweep, park -- so that only one root's state is live at a time.  A robot with a
  // single root runs this body once.
  JTrig<T> tr[N];
  T qv[N], qdv[N], qddv[N];
  T f[N][6], c[N];
  T acc[N][N];
  T v[N][6], a[N][6];
  T dv[N][MAXDEPTH][6], da[N][MAXDEPTH][6];
  const T zero6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
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
  // split > 1: one block per (configurations, group) -- the groups are independent problems, so
  // giving each its own blocks multiplies the wave count and divides a wave's serial length
  const int gsel = split > 1 ? gsel_x : -1;
  sfor<0, N>([&](auto Rt) {
   constexpr int rt = decltype(Rt)::value;
   if constexpr (grp_head(rt)) {
   constexpr int gi = grp_index(rt);
   if (gsel < 0 || gsel == gi) {
   constexpr int row0 = grp_row0(rt);
   constexpr int rows = grp_rows(rt);
   T* my = tile + slot * GRAD_TS + (isqd ? N : 0) - row0 * GRAD_ROW;   // my[i * 2N + c], rows of this group
  // All of a group's input loads are issued together and before its first sincos (whose
  // range-reduction branch would otherwise fence each load behind the previous joint's trig: n
  // serialized HBM round trips); the NEXT group's loads are issued before this group's passes, so
  // their latency hides behind a whole group of arithmetic.
  if (gsel >= 0 || rt == grp_first()) load_group(Rt);
  if constexpr (FDG) {
    if (gsel >= 0) {          // this block's group only: rows row0 .. row0 + rows - 1 of every configuration's Minv (rows * N contiguous scalars)
      constexpr int RN = rows * N;
      const T* msrc = minv_in + cfg0 * (N * N) + row0 * N;
      // every load issued before the first LDS write (a load -> write loop costs one HBM latency per iteration: 18 of them here)
      constexpr int NL = (CFGS * RN + NT - 1) / NT;
      T mb[NL];
      sfor<0, NL>([&](auto I_) {
        constexpr int i_ = decltype(I_)::value;
        const int g = lane + i_ * NT;
        const int cfg = g / RN, r2 = g - cfg * RN;
        mb[i_] = msrc[g < nvalid * RN ? cfg * (N * N) + r2 : 0];
      });
      sfor<0, NL>([&](auto I_) {
        constexpr int i_ = decltype(I_)::value;
        const int g = lane + i_ * NT;
        const int cfg = g / RN, r2 = g - cfg * RN;
        if (g < CFGS * RN) mtile[cfg * FDG_MS + r2] = mb[i_];
      });
    }
  }
  pair_trig<rt, 0>(isqd, qv, tr);
  if constexpr (grp_next(rt) >= 0) {
    if (gsel < 0) load_group(std::integral_constant<int, grp_next(rt) >= 0 ? grp_next(rt) : 0>{});
  }

  // ---- pass 1: RNEA forward + backward -> c and the ACCUMULATED forces f (:569-619) -----------
  {
    T v1[N][6], a1[N][6];
    sfor<0, N>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int p = PARENT[j];
      if constexpr (grp_has(rt, j)) {
        T xv[6], xa[6];
        if constexpr (p < 0)
          rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddv[j], grav, zero6, zero6, xv, xa, v1[j], a1[j], f[j]);
        else
          rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddv[j], grav, v1[p], a1[p], xv, xa, v1[j], a1[j], f[j]);
      }
    });
  }
  sfor_down<0, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int p = PARENT[j];
    if constexpr (grp_has(rt, j)) {
      c[j] = S_dot<j>(f[j]);
      if constexpr (p >= 0) {
        T t[6];
        xform_T<j>(tr[j], f[j], t);
        sfor<0, 6>([&](auto R) { f[p][decltype(R)::value] += t[decltype(R)::value]; });
      }
    }
  });
  if (c_out != nullptr && !isqd && slot < nvalid) {
    sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; if constexpr (grp_has(rt, j)) c_out[b * N + j] = c[j]; });
  }
  // ---- pass 2: forward gradient sweep (:1139-1185, :1210-1252 fused; backward passes folded in)
  // v, a are recomputed here from laundered inputs instead of being kept from pass 1.
  // Column slot s of body j = its ancestor-or-self at depth s.  dv/da[j][s] are this lane's
  // derivative columns (dq columns on even lanes, dqd columns on odd lanes).  acc[i][c] collects
  // dc[i, c] for related (i, c); everything is statically indexed => registers.
  sfor<0, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int p = PARENT[j];
    constexpr int d = DEPTH[j];
    if constexpr (grp_has(rt, j)) {
    const JTrig<T> g{launder(tr[j].s), launder(tr[j].c)};
    const T qdj = launder(qdv[j]);
    const T qddj = HAS_QDD ? launder(qddv[j]) : T(0);
    T xv[6], xa[6];
    {
      T fdead[6];
      if constexpr (p < 0)
        rnea_fwd_body<j, HAS_QDD>(g, qdj, qddj, grav, zero6, zero6, xv, xa, v[j], a[j], fdead);
      else
        rnea_fwd_body<j, HAS_QDD>(g, qdj, qddj, grav, v[p], a[p], xv, xa, v[j], a[j], fdead);
    }
    // inherited columns: dv = X dv_p ; da = X da_p + qd_j crm(dv) S           (:1158,:1163,:1170)
    sfor<0, d>([&](auto Sx) {
      constexpr int s = decltype(Sx)::value;
      xform<j>(g, dv[p][s], dv[j][s]);
      xform<j>(g, da[p][s], da[j][s]);
      add_mxS<j>(dv[j][s], qdj, da[j][s]);
    });
    // own column: dq:  dv = crm(X v_p) S (0 at a root, :1157-1159);  da = qd crm(dv) S + crm(X a_p) S (:1170-1175)
    //             dqd: dv = S (:1231);                                da = qd crm(S) S (= 0) + crm(v_j) S (:1243)
    {
      T sdq[6], sS[6], e1[6], e2[6];
      mxS<j>(xv, T(1), sdq);
      sfor<0, 6>([&](auto R) { sS[decltype(R)::value] = T(0); });
      add_S<j>(T(1), sS);
      mxS<j>(xa, T(1), e1);
      mxS<j>(v[j], T(1), e2);
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        dv[j][d][r] = sel(isqd, sS[r], sdq[r]);
        da[j][d][r] = sel(isqd, e2[r], e1[r]);
      });
      add_mxS<j>(dv[j][d], qdj, da[j][d]);
    }
    // Phi[i, j] = dv_dqd[:, i, j] lives on the odd lane of the pair
    T phi[d + 1][6];
    sfor<0, d + 1>([&](auto Sx) {
      constexpr int s = decltype(Sx)::value;
      sfor<0, 6>([&](auto R) { phi[s][decltype(R)::value] = from_odd_lane(dv[j][s][decltype(R)::value]); });
    });
    T Iv[6];
    cmatvec<MatI, j>(v[j], Iv);
    // extra dq term of the backward pass (:1292-1294): column j gains X_j^T fxS(S_j, f_j) at the
    // parent, i.e. rows i that are STRICT ancestors see  Phi[i, j]^T (-crm(f_j) S_j).
    T w[6];
    mxS<j>(f[j], T(-1), w);
    sfor<0, d + 1>([&](auto Sc) {
      constexpr int sc = decltype(Sc)::value;
      constexpr int col = anc_at(j, sc);
      // df = I da + crf(dv) (I v) + crf(v) (I dv)                             (:1179-1185)
      T df[6], Idv[6];
      cmatvec<MatI, j>(da[j][sc], df);
      cmatvec<MatI, j>(dv[j][sc], Idv);
      fxv<true>(dv[j][sc], Iv, df);
      fxv<true>(v[j], Idv, df);
      // dc[i, c] += Phi[i]^T df[c] for all ancestor-or-self rows i of j
      sfor<0, d + 1>([&](auto Si) {
        constexpr int si = decltype(Si)::value;
        constexpr int row = anc_at(j, si);
        T val = dot6(phi[si], df);
        if constexpr (sc == d && si < d) {
          const T ex = dot6(phi[si], w);
          val += sel(isqd, T(0), ex);
        }
        if constexpr (GRAD_ACC_IN_REGS) {
          if constexpr (si == d || sc == d) acc[row][col] = val;   // first touch of (row, col)
          else acc[row][col] += val;
        } else {
          if constexpr (si == d || sc == d) my[row * GRAD_ROW + col] = val;
          else my[row * GRAD_ROW + col] += val;
        }
      });
    });
    }  // grp_has(rt, j)
  });

  // ---- this group is complete: finish its rows in the LDS image and stream them out -----------
  if constexpr (FDG) {
    // out[i][c] = - sum_k Minv[i][k] dc[k][c]  (:1382-1383); Minv and dc are block-diagonal over groups
    __syncthreads();                       // mtile is complete (written by other lanes at the start)
    // (split: the tile holds rows row0 .. of this group only, stride FDG_MS; mt[i * N + k] below then addresses row i - row0)
    const T* mt = gsel >= 0 ? mtile + (slot < nvalid ? slot : 0) * FDG_MS - row0 * N : mtile + (slot < nvalid ? slot : 0) * (N * N);
    sfor<0, N>([&](auto C) {
      constexpr int cc = decltype(C)::value;
      if constexpr (!grp_has(rt, cc)) {
        sfor<row0, row0 + rows>([&](auto I) { my[decltype(I)::value * GRAD_ROW + cc] = T(0); });
      } else {
        T colv[N];
        sfor<0, N>([&](auto K) {
          constexpr int k = decltype(K)::value;
          if constexpr (related(k, cc)) colv[k] = acc[k][cc] + ((k == cc) ? sel(use_damping != 0 && isqd, T(DAMPING[k]), T(0)) : T(0));
          else colv[k] = T(0);
        });
        sfor<row0, row0 + rows>([&](auto I) {
          constexpr int i = decltype(I)::value;
          T o = T(0);
          sfor<0, N>([&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (related(k, cc)) o = fma_(-mt[i * N + k], colv[k], o);
          });
          my[i * GRAD_ROW + cc] = o;
        });
      }
    });
  } else {
    sfor<row0, row0 + rows>([&](auto I) {
      sfor<0, N>([&](auto C) {
        constexpr int i = decltype(I)::value, cc = decltype(C)::value;
        if constexpr (!related(i, cc)) {
          my[i * GRAD_ROW + cc] = T(0);                                     // structural zero
        } else if constexpr (GRAD_ACC_IN_REGS) {
          if constexpr (i == cc) my[i * GRAD_ROW + cc] = acc[i][cc] + sel(use_damping != 0 && isqd, T(DAMPING[i]), T(0));  // :1336-1341
          else my[i * GRAD_ROW + cc] = acc[i][cc];
        } else if constexpr (i == cc) {
          my[i * GRAD_ROW + cc] += sel(use_damping != 0 && isqd, T(DAMPING[i]), T(0));
        }
      });
    });
  }
  __syncthreads();
#ifdef RBD_GRAD_EXP_NOFLUSH       // timing experiment: the group's rows stay in the tile (results are wrong)
  if (use_damping == 12345)
#endif
  {
    constexpr int RW = rows * GRAD_ROW;                         // elements of this group per configuration
    T* gdst = dcdu + cfg0 * GRAD_TILE + row0 * GRAD_ROW;
    constexpr int VE = 16 / sizeof(T);
    bool done = false;
    if constexpr (RW == GRAD_TILE && GRAD_TS == GRAD_TILE && (CFGS * GRAD_TILE) % VE == 0) {
      // LDS image == HBM image: 16-byte copies (block base is a multiple of CFGS * GRAD_TILE elements)
      typedef T V __attribute__((ext_vector_type(VE)));
      if (nvalid == CFGS) {
        const V* src = reinterpret_cast<const V*>(tile);
        V* dst = reinterpret_cast<V*>(gdst);
#pragma unroll 4
        for (int g = lane; g < CFGS * GRAD_TILE / VE; g += NT) dst[g] = src[g];
        done = true;
      }
    }
    if constexpr (RW % VE == 0 && GRAD_TS % VE == 0 && GRAD_TILE % VE == 0 && (row0 * GRAD_ROW) % VE == 0) {
      // rows of one group among several, or a ragged tile: still 16-byte pieces (a configuration's segment of RW
      // scalars and its LDS row both start on 16-byte boundaries)
      if (!done) {
        typedef T V __attribute__((ext_vector_type(VE)));
        constexpr int RV = RW / VE;
#ifndef RBD_GRAD_EXP_OLD_FLUSH
        if constexpr ((CFGS * RV) % NT == 0) {
          if (nvalid == CFGS) {   // full tile: no division per piece, batched reads, buffer stores (rbd_spatial.h)
            flush_cfg_rows_full<T, CFGS, RW, GRAD_TS, GRAD_TILE, NT>(tile, gdst, lane);
            done = true;
          }
        }
#endif
      }
      if (!done) {
        typedef T V __attribute__((ext_vector_type(VE)));
        constexpr int RV = RW / VE;
#pragma unroll 4
        for (int g = lane; g < nvalid * RV; g += NT) {
          const int cfg = g / RV;
          const int r = g - cfg * RV;
          *reinterpret_cast<V*>(gdst + cfg * GRAD_TILE + r * VE) = *reinterpret_cast<const V*>(tile + cfg * GRAD_TS + r * VE);
        }
        done = true;
      }
    }
    if (!done) {
#pragma unroll 4
      for (int g = lane; g < nvalid * RW; g += NT) {
        const int cfg = g / RW;
        const int rem = g - cfg * RW;
        gdst[cfg * GRAD_TILE + rem] = tile[cfg * GRAD_TS + rem];
      }
    }
  }
  if constexpr (GRAD_PER_ROOT && rows != N) __syncthreads();   // the next group reuses the tile
   }  // this block's group
   }  // grp_head(rt)
  });
}

}  // namespace rbdk
#include "rbd_idsva.h"
#include "rbd_idsva_pipe.h"
#include "rbd_idsva_tree.h"
#if defined(RBD_TU_GRAD_F64) || defined(RBD_TU_GRADN_F64)
#include "rbd_idsva_tree_ws.h"   // fp64 only: the fp32 units never see it (their cached objects stay valid when it changes)
#define RBD_HAVE_TWS 1
#endif
#ifdef RBD_NEED_GRAD
#include "rbd_grad_cols.h"
#endif
namespace rbdk {
#ifdef RBD_NO_IDSVA
constexpr bool GRAD_USE_IDSVA = false;
#else
constexpr bool GRAD_USE_IDSVA = GRAD_IDSVA_OK;
#endif
// Trees whose column-recursion accumulators do not fit registers (Atlas) take the chain-by-chain
// world-frame kernel (rbd_idsva_tree.h) in fp32; every eligible robot can be put on it with
// rbd_set_option(RBD_OPT_GRAD_KERNEL, RBD_GRAD_KERNEL_TREE) (tests, experiments).
constexpr bool GRAD_TREE_DEFAULT = GRAD_TREE_OK && !GRAD_USE_IDSVA && !GRAD_ACC_IN_REGS;
// The one-lane chain kernel in fp64 (round 3): one wave per SIMD -- 476-486 VGPRs for a 7-body chain, no scratch, since the
// backward sweep recomputes the rotations instead of keeping them -- and 1.58x the two-lane column kernel (iiwa,
// B = 1 048 576: 310 vs 489 us, identical to 1.5e-15).  Chains of up to 7 bodies: an 8th would not fit 512 VGPRs.
// RBD_NO_F64_CHAIN restores the column kernel (experiments).  The fused forward_dynamics_grad epilogue (FDG) stays fp32.
template <class T>
#ifdef RBD_NO_F64_CHAIN
constexpr bool grad_chain_kernel() { return GRAD_USE_IDSVA && sizeof(T) == 4; }
#else
constexpr bool grad_chain_kernel() { return GRAD_USE_IDSVA && (sizeof(T) == 4 || grad_max_rows() <= 7); }
#endif
template <class T>
constexpr bool grad_chain_fdg() { return GRAD_USE_IDSVA && sizeof(T) == 4; }
#endif  // RBD_NEED_GRAD

// ---------------------------------------------------------------------------------------------
// minv: q -> Minv [B, n, n]                                                    (:630-806)
//
// Two launches.  Phase A (one configuration per lane) runs the articulated-inertia recursion
// of the backward pass (:697-700, :728-733) -- the only part that is serial per configuration --
// and leaves {U_i, 1/D_i, sin q_i, cos q_i} per body in a workspace laid out [body][config][12].
// Phase B gives every COLUMN of Minv / F its own lane (columns are independent: :702-726,
// :771-776): a column's F vector climbs its unique root path in the backward sweep and the
// forward sweep recomputes F per body, so the reference's (n, 6, n) F tensor never exists.
// ---------------------------------------------------------------------------------------------
constexpr int MINV_WS = 12;   // scalars per (body, configuration) in the workspace
constexpr int s_index(int i) { return (JTYPE[i] == 0 ? 0 : 3) + AXIS[i]; }

#ifdef RBD_NEED_MINV

template <class T>
struct BodyCfg {
  T U[6];
  T Dinv;   // 1 / D   (the reference's `Dinv` array holds D itself, :698)
  T s, c;
};

template <class T>
RBD_DEV void ws_store(T* ws, long long B, int i, long long b, const BodyCfg<T>& bc) {
  constexpr int VE = 16 / sizeof(T);
  typedef T V __attribute__((ext_vector_type(VE)));
  V* dst = reinterpret_cast<V*>(ws + ((long long)i * B + b) * MINV_WS);
  const T flat[MINV_WS] = {bc.U[0], bc.U[1], bc.U[2], bc.U[3], bc.U[4], bc.U[5], bc.Dinv, bc.s, bc.c, T(0), T(0), T(0)};
  sfor<0, MINV_WS / VE>([&](auto K) {
    constexpr int k = decltype(K)::value;
    V x;
    sfor<0, VE>([&](auto E) { x[decltype(E)::value] = flat[k * VE + decltype(E)::value]; });
    dst[k] = x;
  });
}
// read one 12-scalar record from the LDS copy (three / six 16-byte reads)
template <class T>
RBD_DEV void ws_read_lds(const T* recs, int i, T (&flat)[MINV_WS]) {
  constexpr int VE = 16 / sizeof(T);
  typedef T V __attribute__((ext_vector_type(VE)));
  const V* src = reinterpret_cast<const V*>(recs + i * MINV_WS);
  sfor<0, MINV_WS / VE>([&](auto K) {
    constexpr int k = decltype(K)::value;
    const V x = src[k];
    sfor<0, VE>([&](auto E) { flat[k * VE + decltype(E)::value] = x[decltype(E)::value]; });
  });
}

template <class T>
__global__ __launch_bounds__(64) void minv_ia_kernel(const T* __restrict__ q, long long B, T* __restrict__ ws) {
  const int lane = threadIdx.x;
  const long long b0 = (long long)blockIdx.x * 64 + lane;
  const bool valid = b0 < B;
  const long long b = valid ? b0 : B - 1;
  JTrig<T> tr[N];
  T qv[N];
  sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; qv[j] = q[b * N + j]; });
  sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j] = make_trig<j>(qv[j]); });
  T IA[N][6][6];
  sfor<0, N>([&](auto J) {
    sfor<0, 6>([&](auto R) {
      sfor<0, 6>([&](auto C) {
        constexpr int j = decltype(J)::value, r = decltype(R)::value, c = decltype(C)::value;
        IA[j][r][c] = T(IM[j][r * 6 + c]);
      });
    });
  });
  sfor_down<0, N>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int p = PARENT[i];
    constexpr int si = s_index(i);
    BodyCfg<T> bc;
    sfor<0, 6>([&](auto R) { bc.U[decltype(R)::value] = IA[i][decltype(R)::value][si]; });   // U = IA S   (:697)
    bc.Dinv = T(1) / bc.U[si];                                                                 // D = S^T U (:698,:700)
    bc.s = tr[i].s; bc.c = tr[i].c;
    if (valid) ws_store(ws, B, i, b, bc);
    if constexpr (p >= 0) {
      // Ia = IA - U U^T / D  (:728-731);  IA_p += X^T Ia X  (:732-733)
      T A[6][6];   // A = X^T Ia, built column by column
      sfor<0, 6>([&](auto C) {
        constexpr int c = decltype(C)::value;
        T col[6], y[6];
        const T uc = bc.U[c] * bc.Dinv;
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; col[r] = fma_(-bc.U[r], uc, IA[i][r][c]); });
        xform_T<i>(tr[i], col, y);
        sfor<0, 6>([&](auto R) { A[decltype(R)::value][c] = y[decltype(R)::value]; });
      });
      sfor<0, 6>([&](auto R) {   // (A X)[r][:] = X^T A[r][:]^T
        constexpr int r = decltype(R)::value;
        T y[6];
        xform_T<i>(tr[i], A[r], y);
        sfor<0, 6>([&](auto C) { IA[p][r][decltype(C)::value] += y[decltype(C)::value]; });
      });
    }
  });
}

#ifndef MINV_COLS_MIN_WAVES
#define MINV_COLS_MIN_WAVES 4
#endif
// Phase B works per GROUP (root subtree): Minv is block-diagonal over groups, so the columns of a
// group only ever meet the bodies of that group.  Every group gets its own blocks (lanes per
// configuration = the group's column count rounded up to 8 / 16 / 32 / 64), which cuts the serial
// body loop of a wave from n to the group's size (Atlas: 30 -> 18 / 6 / 6).
// lanes per configuration = the group's column count exactly (Atlas' torso group: 3 x 18 of 64 lanes
// instead of 2 x 32; the lanes beyond CPB * rows repeat the last column and store nothing)
constexpr int minv_lc(int rows) { return rows <= 64 ? rows : 64; }
constexpr int minv_cpb(int rt) { return 64 / minv_lc(grp_rows(rt)); }              // configurations per wave
// LDS tile stride between the configurations of a block: a multiple of 4 scalars (+4) when the group's rows
// can leave as 16-byte pieces, otherwise odd
constexpr bool minv_vec_flush(int rt) { return (grp_rows(rt) * N) % 4 == 0 && (N * N) % 4 == 0 && (grp_row0(rt) * N) % 4 == 0; }
// the same for 16-byte pieces of T (4 floats / 2 doubles): a configuration's segment of the group's rows starts and ends on 16-byte boundaries
template <class T>
constexpr bool minv_piece_flush(int rt) {
  constexpr int VE = 16 / (int)sizeof(T);
  return (grp_rows(rt) * N) % VE == 0 && (N * N) % VE == 0 && (grp_row0(rt) * N) % VE == 0;
}
constexpr int minv_ts(int rt) { return minv_vec_flush(rt) ? grp_rows(rt) * N + 4 : (grp_rows(rt) * N) | 1; }
// the column phase's LDS tile holds only the group's OWN columns ([rows][rows] per configuration): the other
// groups' columns are structural zeros, generated at the flush (9.1 -> 6.5 KB per Atlas torso block: 6
// instead of 4 waves per SIMD)
constexpr int minv_tso(int rt) { return grp_rows(rt) * grp_rows(rt) + ((grp_rows(rt) * grp_rows(rt)) % 2 == 0 ? 1 : 0); }
// Waves per block of the column phase.  Root subtrees with limbs (Atlas' torso) give every column CLASS its own
// wave -- class 0: the stem's columns, class k: the columns of the k-th limb -- because a class needs only part
// of the bodies: backward sweep = the bodies with a class column in their subtree (a limb's own bodies and
// its stem ancestors), forward sweep = the bodies up to the class' last column (rows i <= j of the upper
// triangle; what lies below is mirrored).  Atlas torso, per 9 configurations: 8 + 21 + 28 wave-steps (stem,
// left arm, right arm) instead of 3 x 36.  The waves of a block share the records and the tile.  In limb-less
// groups the waves of a block are independent (each its own configurations and LDS region).
constexpr int MINV_COLS_W = GRAD_PER_ROOT ? 1 + max_limbs_per_root() : 1;
constexpr int mcl_limbs(int rt) { return GRAD_PER_ROOT ? limbs_of_root(rt) : 0; }
constexpr bool mcl_has(int rt, int cls, int j) {     // column j belongs to class cls of root subtree rt
  if (root_of(j) != rt) return false;
  return cls == 0 ? limb_of(j) == -1 : limb_of(j) == kth_limb(rt, cls - 1);
}
constexpr int mcl_count(int rt, int cls) {
  int k = 0;
  for (int j = 0; j < N; ++j) k += mcl_has(rt, cls, j) ? 1 : 0;
  return k;
}
constexpr int mcl_col(int rt, int cls, int k) {      // k-th column of the class
  for (int j = 0; j < N; ++j)
    if (mcl_has(rt, cls, j)) {
      if (k == 0) return j;
      --k;
    }
  return -1;
}
constexpr int mcl_index(int rt, int cls, int i) {    // position of column / body i within the class
  int k = 0;
  for (int j = 0; j < i; ++j) k += mcl_has(rt, cls, j) ? 1 : 0;
  return k;
}
constexpr int limb_rank_in_group(int h) {            // ordinal of limb h among the limbs of its root subtree
  int k = 0;
  for (int x = 0; x < h; ++x) k += (limb_head(x) && root_of(x) == root_of(h)) ? 1 : 0;
  return k;
}
constexpr int mcl_max(int rt, int cls) {
  int m = -1;
  for (int j = 0; j < N; ++j)
    if (mcl_has(rt, cls, j)) m = j;
  return m;
}
constexpr bool mcl_bwd(int rt, int cls, int i) {     // body i takes part in the class' backward sweep
  if (root_of(i) != rt) return false;
  for (int j = 0; j < N; ++j)
    if (mcl_has(rt, cls, j) && is_anc_or_self(i, j)) return true;
  return false;
}
constexpr bool mcl_fwd(int rt, int cls, int i) { return root_of(i) == rt && i <= mcl_max(rt, cls); }
constexpr int mcl_cpb(int rt) {                      // configurations per block of a group with limbs
  int m = 64;
  for (int c = 0; c <= mcl_limbs(rt); ++c) {
    const int x = 64 / mcl_count(rt, c);
    m = x < m ? x : m;
  }
  return m;
}
// configurations per BLOCK and LDS scalars per block of group rt
constexpr int minv_cfgs_per_block(int rt) { return mcl_limbs(rt) > 0 ? mcl_cpb(rt) : minv_cpb(rt) * MINV_COLS_W; }
constexpr size_t minv_wave_scalars(int rt) { return (size_t)minv_cpb(rt) * minv_tso(rt) + (size_t)minv_cpb(rt) * grp_rows(rt) * MINV_WS; }
constexpr size_t minv_block_scalars(int rt) {
  return mcl_limbs(rt) > 0 ? (size_t)mcl_cpb(rt) * minv_tso(rt) + (size_t)mcl_cpb(rt) * grp_rows(rt) * MINV_WS
                           : minv_wave_scalars(rt) * MINV_COLS_W;
}
template <class T>
constexpr size_t minv_cols_lds_bytes() {
  size_t m = 0;
  for (int rt = 0; rt < N; ++rt)
    if (grp_head(rt)) {
      const size_t x = sizeof(T) * ((minv_block_scalars(rt) + 3) / 4 * 4);
      m = x > m ? x : m;
    }
  return m;
}
// groups of at most 8 bodies are finished by minv_ia8_kernel when it runs as phase A (rbd_minv_ia8.h)
constexpr bool minv_small_grp(int rt) { return GRAD_PER_ROOT && grp_rows(rt) <= 8; }
inline long long minv_cols_blocks(long long B, bool skip_small) {
  long long nb = 0;
  for (int rt = 0; rt < N; ++rt)
    if (grp_head(rt) && !(skip_small && minv_small_grp(rt))) nb += (B + minv_cfgs_per_block(rt) - 1) / minv_cfgs_per_block(rt);
  return nb;
}
constexpr unsigned long long subtree_mask(int i) {
  unsigned long long m = 0;
  for (int j = 0; j < N; ++j) m |= is_anc_or_self(i, j) ? (1ull << j) : 0ull;
  return m;
}

template <class T, int RT>
RBD_DEV void minv_cols_group(const T* __restrict__ ws, long long B, int dense, T* __restrict__ Minv,
                             const T* __restrict__ u_in, const T* __restrict__ c_in, T* __restrict__ qdd_out,
                             long long blk, unsigned char* smem_raw) {
  constexpr int row0 = grp_row0(RT), rows = grp_rows(RT);
  constexpr int LC = minv_lc(rows), CPB = 64 / LC, TS = minv_tso(RT);
  // the waves of a block are independent here: wave w works on sub-block blk * W + w in its own LDS region
  const int wave = threadIdx.x >> 6;
  constexpr int WAVE_SCALARS = (int)minv_wave_scalars(RT);   // (bound to a constant: a constexpr call in a run-time expression is not folded)
  T* wsl = reinterpret_cast<T*>(smem_raw) + wave * WAVE_SCALARS;   // [CPB][rows][MINV_WS] per-body records
  T* tile = wsl + CPB * rows * MINV_WS;                    // [CPB][TS] the group's own block [rows][rows] of Minv
  blk = blk * MINV_COLS_W + wave;
  const int lane = threadIdx.x & 63;
  const int slot0 = lane / LC;
  const bool spare = slot0 >= CPB;       // lanes beyond CPB * rows
  const int slot = spare ? CPB - 1 : slot0;
  const int jl = spare ? LC - 1 : lane - slot0 * LC;   // this lane's column within the group
  const int j = row0 + jl;               // ... and in the matrix
  const long long cfg0 = blk * CPB;
  const long long rem = B - cfg0;
  const int nvalid = rem < CPB ? (rem > 0 ? (int)rem : 0) : CPB;   // 0: a trailing wave without configurations

  // stage the block's {U, 1/D, sin, cos} records in LDS once (both sweeps read them; every lane of a
  // configuration reads the same record => LDS broadcast instead of 2 x n dependent L2 round trips)
  {
    constexpr int VE = 16 / sizeof(T);
    constexpr int VPB = MINV_WS / VE;                      // 16-byte pieces per record
    typedef T V __attribute__((ext_vector_type(VE)));
    V* dst = reinterpret_cast<V*>(wsl);
#pragma unroll 2
    for (int idx = lane; idx < CPB * rows * VPB; idx += 64) {
      const int piece = idx % VPB;
      const int rec = idx / VPB;
      const int body = row0 + rec % rows;
      const int cs = rec / rows;
      const long long bb = nvalid > 0 ? cfg0 + (cs < nvalid ? cs : nvalid - 1) : B - 1;
      dst[idx] = reinterpret_cast<const V*>(ws + ((long long)body * B + bb) * MINV_WS)[piece];
    }
  }
  __syncthreads();
  const T* myws = wsl + (slot * rows - row0) * MINV_WS;   // myws + i * MINV_WS = record of body i

  T mcol[N];
  T Fj[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  // Records are read one body ahead of their use; pin6() at the end of each body is an ordering
  // point that keeps the compiler from hoisting ALL record reads to the top of the kernel (which
  // costs 9 n VGPRs and spills: the instruction selector otherwise schedules every LDS read first).
  T rec[N][MINV_WS];
  ws_read_lds(myws, row0 + rows - 1, rec[row0 + rows - 1]);
  // ---- backward sweep (:665-726), column j -----------------------------------------------------
  sfor_down<row0, row0 + rows>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int p = PARENT[i];
    constexpr unsigned long long mask = subtree_mask(i);
    if constexpr (i > row0) ws_read_lds(myws, i - 1, rec[i - 1]);
    BodyCfg<T> bc;
    sfor<0, 6>([&](auto R) { bc.U[decltype(R)::value] = rec[i][decltype(R)::value]; });
    bc.Dinv = rec[i][6]; bc.s = rec[i][7]; bc.c = rec[i][8];
    const bool insub = ((mask >> j) & 1ull) != 0;
    T m = sel(j == i, bc.Dinv, -(bc.Dinv * S_dot<i>(Fj)));         // :700, :702-708
    m = sel(insub, m, T(0));
    mcol[i] = m;
    if constexpr (p >= 0) {
      T t[6], y[6];
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; t[r] = fma_(bc.U[r], m, Fj[r]); });   // :721-723
      const JTrig<T> g{bc.s, bc.c};
      xform_T<i>(g, t, y);                                                                                  // :724-726
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Fj[r] = sel(insub, y[r], Fj[r]); });
    }
    pin6(Fj);
  });
  // ---- forward sweep (:760-781), column j ------------------------------------------------------
  T Ff[N][6];
  // (the memory clobbers of pin6() also stop the compiler from forwarding the backward sweep's
  // record reads to the forward sweep, which would keep all 9 n values live in between)
  const T* myws2 = myws;
  T rec2[N][MINV_WS];
  ws_read_lds(myws2, row0, rec2[row0]);
  sfor<row0, row0 + rows>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int p = PARENT[i];
    constexpr int si = s_index(i);
    if constexpr (i + 1 < row0 + rows) ws_read_lds(myws2, i + 1, rec2[i + 1]);
    if constexpr (p < 0) {
      sfor<0, 6>([&](auto R) { Ff[i][decltype(R)::value] = T(0); });
      Ff[i][si] = mcol[i];                                                                        // :781
    } else {
      BodyCfg<T> bc;
      sfor<0, 6>([&](auto R) { bc.U[decltype(R)::value] = rec2[i][decltype(R)::value]; });
      bc.Dinv = rec2[i][6]; bc.s = rec2[i][7]; bc.c = rec2[i][8];
      const JTrig<T> g{bc.s, bc.c};
      xform<i>(g, Ff[p], Ff[i]);
      const T m = fma_(-bc.Dinv, dot6(bc.U, Ff[i]), mcol[i]);                                     // :771-773
      mcol[i] = m;
      Ff[i][si] += m;                                                                             // :774-776
    }
    pin6(Ff[i]);
  });
  // ---- symmetrise (:799-804) through LDS, then stream the group's rows out -----------------------
  T* myt = tile + slot * TS;                               // myt[(i - row0) * rows + (c - row0)]
  if (!spare) {
    sfor<row0, row0 + rows>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if (i <= j) myt[(i - row0) * rows + jl] = mcol[i];
      if (i < j) myt[jl * rows + (i - row0)] = sel(dense != 0, mcol[i], T(0));
    });
  }
  __syncthreads();
  if (qdd_out != nullptr) {
    // forward_dynamics epilogue (:1371-1374): qdd = Minv (u - c); lane j owns row j of the dense tile.
    // (u - c) is parked in the record area, which both sweeps have finished reading.
    T* tau = wsl + slot * rows - row0;                     // tau[k], k in the group
    if (!spare && slot < nvalid) tau[j] = u_in[(cfg0 + slot) * N + j] - c_in[(cfg0 + slot) * N + j];
    __syncthreads();
    if (!spare && slot < nvalid) {
      T o = T(0);
      sfor<row0, row0 + rows>([&](auto K) { constexpr int k = decltype(K)::value; o = fma_(myt[jl * rows + (k - row0)], tau[k], o); });
      qdd_out[(cfg0 + slot) * N + j] = o;
    }
  }
  if (Minv != nullptr) {
    constexpr int RW = rows * N;
    T* gdst = Minv + cfg0 * (N * N) + row0 * N;
    // element e of a configuration's RW = rows * N contiguous scalars is (row e / N, column e % N): the
    // group's own columns come from the tile, every other column is a structural zero
    auto elem = [&](int cfg, int e) -> T {
      const int r = e / N;
      const int c = e - r * N - row0;
      const bool own = c >= 0 && c < rows;
      const T x = tile[cfg * TS + r * rows + (own ? c : 0)];
      return own ? x : T(0);
    };
    if constexpr (minv_piece_flush<T>(RT)) {
      // 16-byte pieces: the segment of a configuration is 16-byte aligned at both ends
      minv_own_rows_flush<T, row0, rows, CPB, TS, 64>(tile, gdst, lane, nvalid);
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

// next body below i (backward order) / above i (forward order) that class CLS of group RT visits, -1 if none
constexpr int mcl_next_bwd(int rt, int cls, int i) {
  for (int x = i - 1; x >= 0; --x)
    if (mcl_bwd(rt, cls, x)) return x;
  return -1;
}
constexpr int mcl_first_bwd(int rt, int cls) { return mcl_next_bwd(rt, cls, N); }
constexpr int mcl_next_fwd(int rt, int cls, int i) {
  for (int x = i + 1; x < N; ++x)
    if (mcl_fwd(rt, cls, x)) return x;
  return -1;
}
constexpr int mcl_first_fwd(int rt, int cls) { return mcl_next_fwd(rt, cls, -1); }

// The two sweeps of ONE column class of a group with limbs (wave CLS of the block): same arithmetic as
// minv_cols_group, restricted at compile time to the bodies the class needs; fills the class' columns of
// the shared tile (and their mirror images).
template <class T, int RT, int CLS, int CPB = mcl_cpb(RT)>
RBD_DEV void minv_cols_class(const T* wsl, T* tile, int dense, int lane, int& slot_out, int& j_out, bool& spare_out) {
  constexpr int row0 = grp_row0(RT), rows = grp_rows(RT);
  constexpr int LC = mcl_count(RT, CLS), TS = minv_tso(RT);
  static_assert(CPB * LC <= 64, "a class' columns of the block's configurations must fit one wave");
  const int slot0 = lane / LC;
  const bool spare = slot0 >= CPB;       // lanes beyond CPB * LC
  const int slot = spare ? CPB - 1 : slot0;
  const int kc = spare ? LC - 1 : lane - slot0 * LC;     // this lane's column within the class
  constexpr int col0 = mcl_col(RT, CLS, 0);   // (bound to a constant first: a constexpr call in a run-time expression is not folded)
  int j = col0;
  sfor<1, LC>([&](auto K) { constexpr int k = decltype(K)::value; constexpr int col = mcl_col(RT, CLS, k); j = sel(kc == k, col, j); });
  slot_out = slot; j_out = j; spare_out = spare;
  const T* myws = wsl + (slot * rows - row0) * MINV_WS;   // myws + i * MINV_WS = record of body i

  T mcol[N];
  sfor<row0, row0 + rows>([&](auto I) { mcol[decltype(I)::value] = T(0); });   // bodies the backward sweep skips
  T Fj[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  T rec[N][MINV_WS];
  constexpr int b0 = mcl_first_bwd(RT, CLS);
  ws_read_lds(myws, b0, rec[b0]);
  // ---- backward sweep (:665-726), column j: the bodies with a class column in their subtree -----------------
  sfor_down<row0, row0 + rows>([&](auto I) {
    constexpr int i = decltype(I)::value;
    if constexpr (mcl_bwd(RT, CLS, i)) {
      constexpr int p = PARENT[i];
      constexpr unsigned long long mask = subtree_mask(i);
      constexpr int nx = mcl_next_bwd(RT, CLS, i);
      if constexpr (nx >= 0) ws_read_lds(myws, nx, rec[nx]);
      BodyCfg<T> bc;
      sfor<0, 6>([&](auto R) { bc.U[decltype(R)::value] = rec[i][decltype(R)::value]; });
      bc.Dinv = rec[i][6]; bc.s = rec[i][7]; bc.c = rec[i][8];
      const bool insub = ((mask >> j) & 1ull) != 0;
      T m = sel(j == i, bc.Dinv, -(bc.Dinv * S_dot<i>(Fj)));         // :700, :702-708
      m = sel(insub, m, T(0));
      mcol[i] = m;
      if constexpr (p >= 0) {
        T t[6], y[6];
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; t[r] = fma_(bc.U[r], m, Fj[r]); });   // :721-723
        const JTrig<T> g{bc.s, bc.c};
        xform_T<i>(g, t, y);                                                                                  // :724-726
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Fj[r] = sel(insub, y[r], Fj[r]); });
      }
      pin6(Fj);
    }
  });
  // ---- forward sweep (:760-781), column j: rows i <= the class' last column ---------------------------------
  T Ff[N][6];
  const T* myws2 = myws;
  T rec2[N][MINV_WS];
  constexpr int f0 = mcl_first_fwd(RT, CLS);
  ws_read_lds(myws2, f0, rec2[f0]);
  sfor<row0, row0 + rows>([&](auto I) {
    constexpr int i = decltype(I)::value;
    if constexpr (mcl_fwd(RT, CLS, i)) {
      constexpr int p = PARENT[i];
      constexpr int si = s_index(i);
      constexpr int nx = mcl_next_fwd(RT, CLS, i);
      if constexpr (nx >= 0) ws_read_lds(myws2, nx, rec2[nx]);
      if constexpr (p < 0) {
        sfor<0, 6>([&](auto R) { Ff[i][decltype(R)::value] = T(0); });
        Ff[i][si] = mcol[i];                                                                        // :781
      } else {
        BodyCfg<T> bc;
        sfor<0, 6>([&](auto R) { bc.U[decltype(R)::value] = rec2[i][decltype(R)::value]; });
        bc.Dinv = rec2[i][6]; bc.s = rec2[i][7]; bc.c = rec2[i][8];
        const JTrig<T> g{bc.s, bc.c};
        xform<i>(g, Ff[p], Ff[i]);
        const T m = fma_(-bc.Dinv, dot6(bc.U, Ff[i]), mcol[i]);                                     // :771-773
        mcol[i] = m;
        Ff[i][si] += m;                                                                             // :774-776
      }
      pin6(Ff[i]);
    }
  });
  // ---- the class' columns of the tile, mirrored (:799-804) ------------------------------------------------------
  T* myt = tile + slot * TS;                               // myt[(i - row0) * rows + (c - row0)]
  const int jl = j - row0;
  if (!spare) {
    sfor<row0, row0 + rows>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (mcl_fwd(RT, CLS, i)) {
        if (i <= j) myt[(i - row0) * rows + jl] = mcol[i];
        if (i < j) myt[jl * rows + (i - row0)] = sel(dense != 0, mcol[i], T(0));
      }
    });
  }
}

// ---- round 4: the column phase WITHOUT the forward sweep.  Minv = Psi^T D^-1 Psi (tools/check_minv_factorisation.py: the
// operator factorisation minv_bpass + minv_fpass evaluate recursively, :630-783, exact at 2e-16 on every golden robot):
//     Minv[i, j] = sum over k in anc(i) & anc(j) of  D_k m[k][i] m[k][j],     m[k][j] = minv_bpass's Minv[k, j], m[k][k] = 1 / D_k
// so a column needs its BACKWARD sweep only (along its root path); the entries then follow from the table of all columns'
// m -- which is the tile's upper triangle -- with a handful of FMAs per entry instead of one six-vector transform per
// (column, body): the right arm's class of the 30-body robot 79 FMAs + LDS reads per lane instead of 28 dependent body steps.
// Three steps with a block barrier between them (the table must be complete before it is read, and read before the final
// values overwrite it):  _bwd -> table,  _fin -> registers,  _put -> tile (mirrored, :799-804).
// TRI: the tile is the packed upper triangle (stride minv_tst), not the [rows][rows] square -- the table and the final values
// then share their place and the mirror image is made by the flush (rbd_spatial.h: minv_own_rows_flush<..., TRI>).
constexpr int minv_tst(int rt) { return (grp_rows(rt) * (grp_rows(rt) + 1) / 2) | 1; }
template <class T, int RT, int CLS, int CPB = mcl_cpb(RT), bool TRI = false>
RBD_DEV void minv_cols_class_bwd(const T* wsl, T* tile, int lane, int& slot_out, int& j_out, bool& spare_out, T (&wj)[N]) {
  constexpr int row0 = grp_row0(RT), rows = grp_rows(RT);
  constexpr int LC = mcl_count(RT, CLS), TS = TRI ? minv_tst(RT) : minv_tso(RT);
  static_assert(CPB * LC <= 64, "a class' columns of the block's configurations must fit one wave");
  const int slot0 = lane / LC;
  const bool spare = slot0 >= CPB;       // lanes beyond CPB * LC
  const int slot = spare ? CPB - 1 : slot0;
  const int kc = spare ? LC - 1 : lane - slot0 * LC;     // this lane's column within the class
  constexpr int col0 = mcl_col(RT, CLS, 0);
  int j = col0;
  sfor<1, LC>([&](auto K) { constexpr int k = decltype(K)::value; constexpr int col = mcl_col(RT, CLS, k); j = sel(kc == k, col, j); });
  slot_out = slot; j_out = j; spare_out = spare;
  const T* myws = wsl + (slot * rows - row0) * MINV_WS;   // myws + i * MINV_WS = record of body i
  T mcol[N];
  sfor<row0, row0 + rows>([&](auto I) { mcol[decltype(I)::value] = T(0); wj[decltype(I)::value] = T(0); });
  T Fj[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  T rec[N][MINV_WS];
  constexpr int b0 = mcl_first_bwd(RT, CLS);
  ws_read_lds(myws, b0, rec[b0]);
  sfor_down<row0, row0 + rows>([&](auto I) {
    constexpr int i = decltype(I)::value;
    if constexpr (mcl_bwd(RT, CLS, i)) {
      constexpr int p = PARENT[i];
      constexpr unsigned long long mask = subtree_mask(i);
      constexpr int nx = mcl_next_bwd(RT, CLS, i);
      if constexpr (nx >= 0) ws_read_lds(myws, nx, rec[nx]);
      BodyCfg<T> bc;
      sfor<0, 6>([&](auto R) { bc.U[decltype(R)::value] = rec[i][decltype(R)::value]; });
      bc.Dinv = rec[i][6]; bc.s = rec[i][7]; bc.c = rec[i][8];
      const bool insub = ((mask >> j) & 1ull) != 0;
      T m = sel(j == i, bc.Dinv, -(bc.Dinv * S_dot<i>(Fj)));         // :700, :702-708
      m = sel(insub, m, T(0));
      mcol[i] = m;
      wj[i] = bc.U[s_index(i)] * m;                                  // D_i m[i][j]   (D_i = S^T U_i, :698)
      if constexpr (p >= 0) {
        T t[6], y[6];
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; t[r] = fma_(bc.U[r], m, Fj[r]); });   // :721-723
        const JTrig<T> g{bc.s, bc.c};
        xform_T<i>(g, t, y);                                                                                  // :724-726
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Fj[r] = sel(insub, y[r], Fj[r]); });
      }
      pin6(Fj);
    }
  });
  // the table: column j of the tile's upper triangle, zeros included (rows that are no ancestors of j)
  T* myt = tile + slot * TS;
  const int jl = j - row0;
  if (!spare) {
    sfor<row0, row0 + rows>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (mcl_fwd(RT, CLS, i)) {
        if (i <= j) myt[TRI ? tri_off(i - row0, jl, rows) : (i - row0) * rows + jl] = mcol[i];
      }
    });
  }
}
template <class T, int RT, int CLS, bool TRI = false>
RBD_DEV void minv_cols_class_fin(const T* tile, int slot, const T (&wj)[N], T (&acc)[N]) {
  constexpr int row0 = grp_row0(RT), rows = grp_rows(RT), TS = TRI ? minv_tst(RT) : minv_tso(RT);
  const T* myt = tile + slot * TS;
  sfor<row0, row0 + rows>([&](auto I) {
    constexpr int i = decltype(I)::value;
    if constexpr (mcl_fwd(RT, CLS, i)) {
      T a = T(0);
      sfor<row0, i + 1>([&](auto K) {
        constexpr int k = decltype(K)::value;
        // k contributes to row i only if it is an ancestor-or-self of i, and to this class only if the class' backward
        // sweep visits it (else w_j[k] = 0): both known at compile time
        if constexpr (mcl_bwd(RT, CLS, k) && is_anc_or_self(k, i))
          a = fma_(wj[k], myt[TRI ? tri_off(k - row0, i - row0, rows) : (k - row0) * rows + (i - row0)], a);
      });
      acc[i] = a;
    }
  });
}
template <class T, int RT, int CLS, bool TRI = false>
RBD_DEV void minv_cols_class_put(T* tile, int dense, int slot, int j, bool spare, const T (&acc)[N]) {
  constexpr int row0 = grp_row0(RT), rows = grp_rows(RT), TS = TRI ? minv_tst(RT) : minv_tso(RT);
  T* myt = tile + slot * TS;
  const int jl = j - row0;
  if (!spare) {
    sfor<row0, row0 + rows>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (mcl_fwd(RT, CLS, i)) {
        if constexpr (TRI) {
          if (i <= j) myt[tri_off(i - row0, jl, rows)] = acc[i];
        } else {
          if (i <= j) myt[(i - row0) * rows + jl] = acc[i];
          if (i < j) myt[jl * rows + (i - row0)] = sel(dense != 0, acc[i], T(0));
        }
      }
    });
  }
}

// A group with limbs: one block = mcl_cpb(RT) configurations, one wave per column class, shared records and tile.
template <class T, int RT>
RBD_DEV void minv_cols_limbs(const T* __restrict__ ws, long long B, int dense, T* __restrict__ Minv,
                             const T* __restrict__ u_in, const T* __restrict__ c_in, T* __restrict__ qdd_out,
                             long long blk, unsigned char* smem_raw) {
  constexpr int row0 = grp_row0(RT), rows = grp_rows(RT);
  constexpr int CPB = mcl_cpb(RT), TS = minv_tso(RT), NT = 64 * MINV_COLS_W, NL = mcl_limbs(RT);
  T* wsl = reinterpret_cast<T*>(smem_raw);                 // [CPB][rows][MINV_WS] per-body records
  T* tile = wsl + CPB * rows * MINV_WS;                    // [CPB][TS] the group's own block [rows][rows] of Minv
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const long long cfg0 = blk * CPB;
  const long long rem = B - cfg0;
  const int nvalid = rem < CPB ? (int)rem : CPB;
  {
    constexpr int VE = 16 / sizeof(T);
    constexpr int VPB = MINV_WS / VE;                      // 16-byte pieces per record
    typedef T V __attribute__((ext_vector_type(VE)));
    V* dst = reinterpret_cast<V*>(wsl);
#pragma unroll 2
    for (int idx = tid; idx < CPB * rows * VPB; idx += NT) {
      const int piece = idx % VPB;
      const int rec = idx / VPB;
      const int body = row0 + rec % rows;
      const int cs = rec / rows;
      const long long bb = cfg0 + (cs < nvalid ? cs : nvalid - 1);
      dst[idx] = reinterpret_cast<const V*>(ws + ((long long)body * B + bb) * MINV_WS)[piece];
    }
  }
  __syncthreads();
  int slot = 0, j = row0;
  bool spare = true;                                       // a wave without a class in this group stays "spare"
#ifdef RBD_MINV_EXP_FWD_SWEEP      // the reference's forward sweep per column (rounds 1-3), for A/B timing
  sfor<0, MINV_COLS_W>([&](auto W) {
    constexpr int w = decltype(W)::value;
    if constexpr (w <= NL) {
      if (wave == w) minv_cols_class<T, RT, w>(wsl, tile, dense, lane, slot, j, spare);
    }
  });
  __syncthreads();
#else
  T wj[N], accv[N];
  sfor<0, MINV_COLS_W>([&](auto W) {
    constexpr int w = decltype(W)::value;
    if constexpr (w <= NL) { if (wave == w) minv_cols_class_bwd<T, RT, w>(wsl, tile, lane, slot, j, spare, wj); }
  });
  __syncthreads();
  sfor<0, MINV_COLS_W>([&](auto W) {
    constexpr int w = decltype(W)::value;
    if constexpr (w <= NL) { if (wave == w) minv_cols_class_fin<T, RT, w>(tile, slot, wj, accv); }
  });
  __syncthreads();
  sfor<0, MINV_COLS_W>([&](auto W) {
    constexpr int w = decltype(W)::value;
    if constexpr (w <= NL) { if (wave == w) minv_cols_class_put<T, RT, w>(tile, dense, slot, j, spare, accv); }
  });
  __syncthreads();
#endif

  if (qdd_out != nullptr) {
    // forward_dynamics epilogue (:1371-1374): qdd = Minv (u - c); lane j owns row j of the dense tile.
    T* tau = wsl + slot * rows - row0;                     // tau[k], k in the group (the record area is free now)
    if (!spare && slot < nvalid) tau[j] = u_in[(cfg0 + slot) * N + j] - c_in[(cfg0 + slot) * N + j];
    __syncthreads();
    if (!spare && slot < nvalid) {
      const T* myt = tile + slot * TS + (j - row0) * rows;
      T o = T(0);
      sfor<row0, row0 + rows>([&](auto K) { constexpr int k = decltype(K)::value; o = fma_(myt[k - row0], tau[k], o); });
      qdd_out[(cfg0 + slot) * N + j] = o;
    }
  }
  if (Minv != nullptr) {
    constexpr int RW = rows * N;
    T* gdst = Minv + cfg0 * (N * N) + row0 * N;
    auto elem = [&](int cfg, int e) -> T {
      const int r = e / N;
      const int c = e - r * N - row0;
      const bool own = c >= 0 && c < rows;
      const T x = tile[cfg * TS + r * rows + (own ? c : 0)];
      return own ? x : T(0);
    };
    if constexpr (minv_piece_flush<T>(RT)) {
      minv_own_rows_flush<T, row0, rows, CPB, TS, NT>(tile, gdst, tid, nvalid);
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
__global__ __launch_bounds__(64 * MINV_COLS_W, MINV_COLS_MIN_WAVES) void minv_cols_kernel(const T* __restrict__ ws, long long B, int dense,
                                                       T* __restrict__ Minv, const T* __restrict__ u_in,
                                                       const T* __restrict__ c_in, T* __restrict__ qdd_out, int skip_small) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // 1-D grid: the blocks of group 0, then those of group 1, ... (each group has its own
  // configurations-per-block count)
  long long blk = blockIdx.x;
  bool done = false;
  sfor<0, N>([&](auto Rt) {
    constexpr int rt = decltype(Rt)::value;
    if constexpr (grp_head(rt)) {
      constexpr int cpb = minv_cfgs_per_block(rt);   // constexpr on purpose: as a plain call the tree walk ran at run time
      constexpr bool small = minv_small_grp(rt);
      const long long nb = (small && skip_small != 0) ? 0 : (B + cpb - 1) / cpb;
      if (!done) {
        if (blk < nb) {
          if constexpr (mcl_limbs(rt) > 0) {
            minv_cols_limbs<T, rt>(ws, B, dense, Minv, u_in, c_in, qdd_out, blk, smem_raw);
          } else {
            minv_cols_group<T, rt>(ws, B, dense, Minv, u_in, c_in, qdd_out, blk, smem_raw);
          }
          done = true;
        } else {
          blk -= nb;
        }
      }
    }
  });
}

#endif  // RBD_NEED_MINV

}  // namespace rbdk
#ifdef RBD_NEED_FD
// out[b] = -Minv[b] dc_du[b]  ([n, n] x [n, 2n]) for robots whose gradient kernel cannot fold the product into its
// epilogue: neg_mm_kernel<T, N> (C consecutive configurations per block, in and out as flat 16-byte copies through LDS)
#include "rbd_negmm.h"
#endif  // RBD_NEED_FD
#include "rbd_minv_lane.h"     // also defines MINV_LANE_OK, which sizes the workspaces (every unit)
#ifdef RBD_NEED_MINV
#include "rbd_minv_ia8.h"
#include "rbd_minv_fused.h"
#endif
#if defined(RBD_TU_MINV_F32) || defined(RBD_TU_MINV_F64)
#include "rbd_crba.h"
#endif
#ifdef RBD_NEED_FD
#include "rbd_aba.h"
#include "rbd_fd_chain.h"
#endif
#ifdef RBD_NEED_PASS
#include "rbd_passes.h"
#endif
namespace rbdk {
#ifdef RBD_NO_MINV_LANE
template <class T>
constexpr bool minv_use_lane() { return false; }
#else
template <class T>
constexpr bool minv_use_lane() { return minv_lane_ok<T>(); }
#endif
// scalars of HBM workspace per configuration (none where the one-lane kernel serves BOTH precisions; a robot it serves in
// fp32 only keeps the size the fp64 path needs -- rbd_minv_workspace_bytes does not know the caller's kernel -- and the fp32
// launch ignores the buffer)
constexpr size_t MINV_WS_PER_CFG = (minv_use_lane<float>() && minv_use_lane<double>()) ? 0 : (size_t)N * MINV_WS;
}  // namespace rbdk

// =============================================================================================
// C-ABI (include/rbd_hip.h)
// =============================================================================================
#include "../../include/rbd_hip.h"
#include "rbd_host.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>
constexpr int RBD_MAX_DEVICES = 16;


// thread-local message buffer behind rbd_last_error(); one instance, owned by the COMMON unit
extern "C" __attribute__((visibility("hidden"))) char* rbd_err_buf(void);
#ifdef RBD_TU_COMMON
extern "C" char* rbd_err_buf(void) {
  static thread_local char buf[512] = "";
  return buf;
}
#endif
constexpr size_t RBD_ERR_LEN = 512;

// Tuning options (rbd_set_option): one instance, owned by the COMMON unit; relaxed atomics -- an option
// only ever selects between kernels that compute the same result.
extern "C" __attribute__((visibility("hidden"))) std::atomic<int>* rbd_option_slot(int option);
#ifdef RBD_TU_COMMON
extern "C" std::atomic<int>* rbd_option_slot(int option) {
  static std::atomic<int> slots[RBD_OPT_COUNT_];
  return option >= 0 && option < RBD_OPT_COUNT_ ? &slots[option] : nullptr;
}
#endif
static inline int rbd_option(int option) { return rbd_option_slot(option)->load(std::memory_order_relaxed); }
// the batch size kernel selection looks at: the call's own, unless the caller has declared the global batch it is a shard of
static inline int64_t rbd_select_batch(int64_t B) {
  const int g = rbd_option(RBD_OPT_SELECT_BATCH);
  return g > 0 ? (int64_t)g : B;
}

// The forward-dynamics units reuse the kernels of the RNEA / MINV / GRAD units through these entry
// points instead of instantiating the same templates a second time (Atlas: the fp64 gradient kernel
// alone costs 200 s of compile time).  rbd_minv_fd_* = rbd_minv_* plus the fused qdd = Minv (u - c).
extern "C" {
// qdd = None gradient launches (GRADN units) for the GRAD / FD units
__attribute__((visibility("hidden"))) int rbd_grad_noqdd_f32(const float* q, const float* qd, float gravity, int use_damping, int64_t B, float* c, float* dc_du, void* stream);
__attribute__((visibility("hidden"))) int rbd_grad_noqdd_f64(const double* q, const double* qd, double gravity, int use_damping, int64_t B, double* c, double* dc_du, void* stream);
__attribute__((visibility("hidden"))) int rbd_grad_cols_noqdd_f32(const float* q, const float* qd, float gravity, int use_damping, int64_t B, float* c, float* v, float* a, float* f, float* dc_du, void* stream);
__attribute__((visibility("hidden"))) int rbd_grad_cols_noqdd_f64(const double* q, const double* qd, double gravity, int use_damping, int64_t B, double* c, double* v, double* a, double* f, double* dc_du, void* stream);
__attribute__((visibility("hidden"))) int rbd_minv_fd_f32(const float* q, int64_t B, float* Minv, void* workspace, size_t wsb,
                                                          void* stream, const float* u, const float* c, float* qdd, const float* qd, float gravity);
__attribute__((visibility("hidden"))) int rbd_minv_fd_f64(const double* q, int64_t B, double* Minv, void* workspace, size_t wsb,
                                                          void* stream, const double* u, const double* c, double* qdd, const double* qd, double gravity);
}

namespace {
int fail(int code, const char* msg) {
  std::snprintf(rbd_err_buf(), RBD_ERR_LEN, "%s", msg);
  return code;
}
int hip_fail(hipError_t e, const char* where) {
  std::snprintf(rbd_err_buf(), RBD_ERR_LEN, "%s: %s", where, hipGetErrorString(e));
  return (int)e > 0 ? (int)e : 1;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is needed once per kernel (and device), not per
// launch: the granted sizes are remembered.
template <class K>
int ensure_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> granted[RBD_MAX_DEVICES];
  int dev = 0;
  (void)hipGetDevice(&dev);
  const void* key = reinterpret_cast<const void*>(kernel);
  const int slot = dev >= 0 && dev < RBD_MAX_DEVICES ? dev : 0;
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = granted[slot].find(key);
    if (it != granted[slot].end() && it->second >= bytes) return 0;
  }
  hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  std::lock_guard<std::mutex> g(mu);
  granted[slot][key] = bytes;
  return 0;
}

// Blocks of `threads` threads and `lds` bytes that are resident at once on the current device
// (occupancy per CU x CUs), remembered per kernel and device: the grid of the tile-walking kernels.
template <class K>
int resident_blocks(K kernel, int threads, size_t lds, int* out) {
  static std::mutex mu;
  static std::unordered_map<const void*, int> known[RBD_MAX_DEVICES];
  int dev = 0;
  (void)hipGetDevice(&dev);
  const void* key = reinterpret_cast<const void*>(kernel);
  const int slot = dev >= 0 && dev < RBD_MAX_DEVICES ? dev : 0;
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = known[slot].find(key);
    if (it != known[slot].end()) { *out = it->second; return 0; }
  }
  int per_cu = 0, cus = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds);
  if (e != hipSuccess) return hip_fail(e, "hipOccupancyMaxActiveBlocksPerMultiprocessor");
  e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return hip_fail(e, "hipDeviceGetAttribute(MultiprocessorCount)");
  const int n = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
  std::lock_guard<std::mutex> g(mu);
  known[slot][key] = n;
  *out = n;
  return 0;
}
}  // namespace

// Library-owned scratch of the workspace gradient kernel (rbd_idsva_tree_ws.h): one buffer per (device, stream), keyed
// by the STREAM's device (hipStreamGetDevice; the null stream belongs to the calling thread's current device).
// Launches on one stream are ordered and share it; launches on different streams get different buffers.
// Lifetime rule: a buffer that was ever handed out is NEVER freed or moved by a later call -- launches in flight, bound
// launches and captured hipGraphs may hold its address.  The callers ask for a size that depends on the kernel's
// occupancy, not on B, so a (device, stream) normally sees one allocation; should a later call need more (another
// kernel of the library), a larger buffer is allocated NEXT TO the old one, which is retired, not released.  Only
// rbd_release_workspaces() frees (after hipDeviceSynchronize, by contract with no call of this library in flight and
// no graph that contains one still alive).  The first call on a stream allocates (hipMalloc is not capturable: run a
// call once before capturing it into a graph, as for every kernel that needs hipFuncSetAttribute); no call ever
// synchronises the device.  One pool, owned by the COMMON unit.
extern "C" __attribute__((visibility("hidden"))) int rbd_stream_workspace(void* stream, size_t bytes, void** out);
#ifdef RBD_TU_COMMON
namespace {
struct RbdWsBuf { void* p = nullptr; size_t n = 0; };
struct RbdWsEntry { RbdWsBuf cur; std::vector<RbdWsBuf> retired; };
std::mutex& rbd_ws_mutex() { static std::mutex mu; return mu; }
std::unordered_map<const void*, RbdWsEntry>* rbd_ws_pool() {
  static std::unordered_map<const void*, RbdWsEntry> pool[RBD_MAX_DEVICES];
  return pool;
}
int rbd_ws_device_of(void* stream, int* dev) {
  int d = 0;
  hipError_t e = stream ? hipStreamGetDevice((hipStream_t)stream, &d) : hipGetDevice(&d);
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipGetDevice(&d); }
  if (e != hipSuccess) return (int)e;
  *dev = d;
  return 0;
}
}  // namespace
extern "C" int rbd_stream_workspace(void* stream, size_t bytes, void** out) {
  int dev = 0;
  if (rbd_ws_device_of(stream, &dev) != 0) return hip_fail(hipErrorInvalidDevice, "rbd workspace: device of the stream");
  if (dev < 0 || dev >= RBD_MAX_DEVICES) return fail(RBD_ERR_UNSUPPORTED, "rbd workspace: device index beyond RBD_MAX_DEVICES");
  std::lock_guard<std::mutex> g(rbd_ws_mutex());
  RbdWsEntry& en = rbd_ws_pool()[dev][stream];
  if (en.cur.n < bytes) {
    int cur_dev = dev;
    (void)hipGetDevice(&cur_dev);
    if (cur_dev != dev) (void)hipSetDevice(dev);            // allocate on the stream's device
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (cur_dev != dev) (void)hipSetDevice(cur_dev);
    if (e != hipSuccess) return hip_fail(e, "rbd workspace hipMalloc");
    if (en.cur.p) en.retired.push_back(en.cur);               // may still be referenced: kept, not freed
    en.cur.p = p;
    en.cur.n = bytes;
  }
  *out = en.cur.p;
  return 0;
}
extern "C" int rbd_release_workspaces(void) {
  std::lock_guard<std::mutex> g(rbd_ws_mutex());
  int cur_dev = 0;
  const bool have_dev = hipGetDevice(&cur_dev) == hipSuccess;
  if (!have_dev) (void)hipGetLastError();
  int rc = 0;
  for (int dev = 0; dev < RBD_MAX_DEVICES; ++dev) {
    auto& m = rbd_ws_pool()[dev];
    if (m.empty()) continue;
    hipError_t e = hipSetDevice(dev);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess && rc == 0) rc = hip_fail(e, "rbd_release_workspaces: hipDeviceSynchronize");
    for (auto& kv : m) {
      if (kv.second.cur.p) (void)hipFree(kv.second.cur.p);
      for (auto& b : kv.second.retired) (void)hipFree(b.p);
    }
    m.clear();
  }
  if (have_dev) (void)hipSetDevice(cur_dev);
  return rc;
}
#endif

namespace {

#ifdef RBD_NEED_RNEA
template <class T>
int rnea_launch(const T* q, const T* qd, const T* qdd, T gravity, int64_t B, T* c, T* v, T* a, T* f,
                void* stream, int fpass_only = 0) {
  using namespace rbdk;
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_rnea: B < 0");
  if (B == 0) return 0;
  if (!q || !qd || (!c && !fpass_only)) return fail(RBD_ERR_ARG, "rbd_rnea: q, qd and c must be non-null");
  if (fpass_only && !(v && a && f)) return fail(RBD_ERR_ARG, "rbd_rnea_fpass: v, a, f must be non-null");
  const bool vaf = v || a || f;
  if (vaf && !(v && a && f)) return fail(RBD_ERR_ARG, "rbd_rnea: v, a, f must be all null or all non-null");
  if (((reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(f)) & 15u) != 0)
    return fail(RBD_ERR_ARG, "rbd_rnea: output buffers must be 16-byte aligned");
  const int64_t blocks = (B + 63) / 64;
  if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_rnea: B too large");
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = rnea_lds_bytes<T>(true);
  const size_t lds_c = rnea_lds_bytes<T>(false);
  int rc;
  if constexpr (rnea_segs_ok<T>()) {
    // one wave per segment (stem / limb / limb-less group): Atlas fp32 B = 16 384
    const int ropt = rbd_option(RBD_OPT_RNEA_KERNEL);
    if (vaf && !fpass_only && ropt != RBD_RNEA_KERNEL_BATCH && ropt != RBD_RNEA_KERNEL_GROUPS) {
      constexpr size_t ldss = rnea_segs_lds<T>();
      if (qdd) {
        auto k = rnea_segments_kernel<T, true>;
        if ((rc = ensure_lds(k, ldss)) != 0) return rc;
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * RS_WAVES), ldss, s, q, qd, qdd, gravity, (long long)B, c, v, a, f);
      } else {
        auto k = rnea_segments_kernel<T, false>;
        if ((rc = ensure_lds(k, ldss)) != 0) return rc;
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * RS_WAVES), ldss, s, q, qd, qdd, gravity, (long long)B, c, v, a, f);
      }
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return hip_fail(e, "rbd_rnea (segment waves) launch");
      return 0;
    }
  }
  if constexpr (rnea_groups_ok<T>() && !(RBD_FAST_STAGE && rnea_segs_ok<T>())) {   // (first-use build: not next to the segment kernel AUTO picks)
    // one wave per independent root group: faster than one lane per configuration at every batch size
    // measured (Atlas fp32: 17.4 -> 13.5 us at B = 16 384, 235 -> 167 us at B = 262 144; quadruped fp32
    // B = 1M: 199 -> 174 us = 6.4 TB/s)
    const int ropt = rbd_option(RBD_OPT_RNEA_KERNEL);
    if (vaf && !fpass_only && ropt != RBD_RNEA_KERNEL_BATCH) {
      const size_t ldsg = 2 * sizeof(T) * 64 * (size_t)odd_pad<6 * N>();
      if (qdd) {
        auto k = rnea_groups_kernel<T, true>;
        if ((rc = ensure_lds(k, ldsg)) != 0) return rc;
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * RG_WAVES), ldsg, s, q, qd, qdd, gravity, (long long)B, c, v, a, f);
      } else {
        auto k = rnea_groups_kernel<T, false>;
        if ((rc = ensure_lds(k, ldsg)) != 0) return rc;
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * RG_WAVES), ldsg, s, q, qd, qdd, gravity, (long long)B, c, v, a, f);
      }
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return hip_fail(e, "rbd_rnea (group waves) launch");
      return 0;
    }
  }
#define RBD_LAUNCH_RNEA(HQ, VAF, LDS)                                                              \
  do {                                                                                             \
    auto k = rnea_kernel<T, HQ, VAF>;                                                              \
    if ((rc = ensure_lds(k, LDS)) != 0) return rc;                                                 \
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), LDS, s, q, qd, qdd, gravity,           \
                       (long long)B, c, v, a, f, fpass_only);                                      \
  } while (0)
  if (qdd) { if (vaf) RBD_LAUNCH_RNEA(true, true, lds); else RBD_LAUNCH_RNEA(true, false, lds_c); }
  else     { if (vaf) RBD_LAUNCH_RNEA(false, true, lds); else RBD_LAUNCH_RNEA(false, false, lds_c); }
#undef RBD_LAUNCH_RNEA
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rbd_rnea launch");
  return 0;
}

template <class T>
int rnea_bpass_launch(const T* q, T* f, int64_t B, T* c, void* stream) {
  using namespace rbdk;
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_rnea_bpass: B < 0");
  if (B == 0) return 0;
  if (!q || !f || !c) return fail(RBD_ERR_ARG, "rbd_rnea_bpass: q, f and c must be non-null");
  const int64_t blocks = (B + 63) / 64;
  if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_rnea_bpass: B too large");
  const size_t lds = sizeof(T) * 64 * (size_t)odd_pad<6 * N>();
  auto k = rnea_bpass_kernel<T>;
  int rc;
  if ((rc = ensure_lds(k, lds)) != 0) return rc;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, q, f, (long long)B, c);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rbd_rnea_bpass launch");
  return 0;
}

// the kernel rbd_rnea launches when v, a, f are requested (c alone is always the one-lane kernel)
template <class T>
int rnea_kernel_name(int64_t, char* buf, size_t len) {
  using namespace rbdk;
  const char* t = sizeof(T) == 4 ? "float" : "double";
  const int ropt = rbd_option(RBD_OPT_RNEA_KERNEL);
  bool done = false;
  if constexpr (rnea_segs_ok<T>()) {
    if (!done && ropt != RBD_RNEA_KERNEL_BATCH && ropt != RBD_RNEA_KERNEL_GROUPS) { std::snprintf(buf, len, "rnea_segments_kernel<%s>", t); done = true; }
  }
  if constexpr (rnea_groups_ok<T>()) {
    if (!done && ropt != RBD_RNEA_KERNEL_BATCH) { std::snprintf(buf, len, "rnea_groups_kernel<%s>", t); done = true; }
  }
  if (!done) std::snprintf(buf, len, "rnea_kernel<%s>", t);
  return 0;
}

#endif  // RBD_NEED_RNEA

#ifdef RBD_NEED_GRAD
// launches exactly one instantiation (the FD translation units use this to avoid compiling the
// variants they never call)
template <class T, bool HAS_QDD, bool FDG>
int rnea_grad_launch1(const T* q, const T* qd, const T* qdd, T gravity, int use_damping, int64_t B,
                      T* c, T* dc_du, void* stream, const T* minv_in) {
  using namespace rbdk;
  constexpr int CFGS = grad_cfgs<T>();
  const int64_t blocks = (B + CFGS - 1) / CFGS;
  if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_rnea_grad: B too large");
  if ((reinterpret_cast<uintptr_t>(dc_du) & 15u) != 0) return fail(RBD_ERR_ARG, "rbd_rnea_grad: dc_du must be 16-byte aligned");
  // (with one block per group the Minv tile holds that group's rows only: 55 -> 28 KB for the quadruped in fp64, i.e. five
  // 64-thread blocks per CU instead of two)
  constexpr bool SPLIT = GRAD_PER_ROOT && n_groups() > 1;
  const size_t lds = sizeof(T) * ((size_t)CFGS * GRAD_TS + (FDG ? (size_t)CFGS * (SPLIT ? (size_t)((grad_max_rows() * N) | 1) : (size_t)(N * N)) : 0));
  if (lds > 160 * 1024) return fail(RBD_ERR_UNSUPPORTED, "rbd_rnea_grad: output tile does not fit LDS for this robot size");
  auto k = rnea_grad_kernel<T, HAS_QDD, FDG>;
  int rc;
  if ((rc = ensure_lds(k, lds)) != 0) return rc;
  // independent root subtrees get their own blocks -- also with the fused -Minv epilogue (round 4): Minv and dc_du are
  // block-diagonal over the groups, so a group's block stages that group's Minv rows only (the quadruped's fp64
  // forward_dynamics_grad gradient leg: 151 us with every leg in one block, serial)
  const int split = (GRAD_PER_ROOT && n_groups() > 1) ? n_groups() : 1;
#ifdef RBD_EXP_NO_XCD_MAP
  const int64_t grid = blocks * split;
#else
  const int64_t grid = split > 1 ? ((blocks + 7) / 8 * 8) * split : blocks;     // whole XCD rounds (the kernel's block -> (configurations, group) map)
#endif
  if (grid > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_rnea_grad: B too large");
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(2 * CFGS), lds, (hipStream_t)stream, q, qd, qdd, gravity,
                     use_damping, (long long)B, c, dc_du, minv_in, split);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rbd_rnea_grad launch");
  return 0;
}

// One-lane chain kernel (rbd_idsva.h): the grid is what is resident at once, every block walks tiles.
template <class T, bool HAS_QDD, bool FDG>
int idsva_launch(const T* q, const T* qd, const T* qdd, T gravity, int use_damping, int64_t B,
                 T* c, T* dc_du, void* stream, const T* minv_in) {
  using namespace rbdk;
  const int64_t tiles = (B + 63) / 64;
  if (tiles > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_rnea_grad: B too large");
  if ((reinterpret_cast<uintptr_t>(dc_du) & 15u) != 0) return fail(RBD_ERR_ARG, "rbd_rnea_grad: dc_du must be 16-byte aligned");
  const size_t lds = sizeof(T) * (size_t)64 * IDS_TS;
  if (lds > 160 * 1024) return fail(RBD_ERR_UNSUPPORTED, "rbd_rnea_grad: output tile does not fit LDS for this robot size");
  int rc, resident = 0;
#ifndef RBD_EXP_NO_PIPE
  // one chain, fp32, plain rnea_grad: the software-pipelined tile loop (rbd_idsva_pipe.h)
  if constexpr (!FDG && IDS_PIPE_OK && sizeof(T) == 4) {
    auto kp = rnea_grad_idsva_pipe_kernel<T, HAS_QDD>;
    if ((rc = ensure_lds(kp, lds)) != 0) return rc;
    if ((rc = resident_blocks(kp, 64, lds, &resident)) != 0) return rc;
    const int64_t blocks = tiles < resident ? tiles : resident;
    hipLaunchKernelGGL(kp, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, q, qd, qdd, gravity, use_damping,
                       (long long)B, c, dc_du, (const T*)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rbd_rnea_grad launch");
    return 0;
  } else
#endif
  {   // (an else branch: the kernel below is not even instantiated where the pipelined one serves the robot)
  auto k = rnea_grad_idsva_kernel<T, HAS_QDD, FDG>;
  if ((rc = ensure_lds(k, lds)) != 0) return rc;
  if ((rc = resident_blocks(k, 64, lds, &resident)) != 0) return rc;
#ifdef RBD_EXP_IDS_GRID_TILES
  resident = 0x7fffffff;                   // experiment: one block per tile (no tile walking)
#endif
#ifdef RBD_EXP_IDS_GRID_SCALE
  resident = resident * RBD_EXP_IDS_GRID_SCALE / 8;   // experiment: k/8 of the resident blocks
#endif
  const int64_t blocks = tiles < resident ? tiles : resident;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, q, qd, qdd, gravity, use_damping,
                     (long long)B, c, dc_du, minv_in);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rbd_rnea_grad launch");
  return 0;
  }
}

// Small batches: one lane per (configuration, derivative column) (rbd_grad_cols.h).  Chosen when the
// batch-parallel kernels would leave most of the chip idle: at most two of its waves per SIMD.
constexpr int64_t GRAD_COLS_MAX_WAVES = 2048;
template <class T>
inline bool grad_use_cols(int64_t B) {
  using namespace rbdk;
  if (RBD_FAST_STAGE || !grad_cols_ok<T>()) return false;
  const int opt = rbd_option(RBD_OPT_GRAD_KERNEL);
  if (opt == RBD_GRAD_KERNEL_COLS) return true;
  if (opt != RBD_GRAD_KERNEL_AUTO) return false;
  return (rbd_select_batch(B) + GC_CPW - 1) / GC_CPW <= GRAD_COLS_MAX_WAVES;
}
template <class T, bool HAS_QDD>
int grad_cols_launch(const T* q, const T* qd, const T* qdd, T gravity, int use_damping, int64_t B,
                     T* c, T* v, T* a, T* f, T* dc_du, void* stream) {
  using namespace rbdk;
  if constexpr (RBD_FAST_STAGE || !grad_cols_ok<T>()) {
    return fail(RBD_ERR_UNSUPPORTED, "rbd_rnea_grad: the column kernel is not built for this robot size");
  } else {
    const int64_t blocks = (B + GC_CPW - 1) / GC_CPW;
    if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_rnea_grad: B too large");
    hipLaunchKernelGGL((rnea_grad_cols_kernel<T, HAS_QDD>), dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, q, qd, qdd,
                       gravity, use_damping, (long long)B, c, v, a, f, dc_du);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rbd_rnea_grad (column kernel) launch");
    return 0;
  }
}

// fp64 trees on the workspace kernel (rbd_idsva_tree_ws.h).  Default where the fp32 default is the tree kernel
// (Atlas: the two-lane column kernel spilled 388 registers there and is no longer built in fp64); with
// RBD_OPT_GRAD_KERNEL = TREE for every other fp64 robot whose root path does not fit the register plan.
#ifndef RBD_HAVE_TWS
template <class T>
constexpr bool tws_built() { return false; }
template <class T>
constexpr bool tws_only() { return false; }
template <class T, bool HAS_QDD>
int tree_ws_launch(const T*, const T*, const T*, T, int, int64_t, T*, T*, void*) {
  return fail(RBD_ERR_UNSUPPORTED, "rbd_rnea_grad: the workspace tree kernel is not part of this unit");
}
#else
template <class T>
constexpr bool tws_built() {
  using namespace rbdk;
#ifdef RBD_TWS_FORCE                                          // experiments: the workspace kernel for every eligible robot
  return tws_ok<T>();
#else
  constexpr bool reg_plan = N <= 12 && rbdm::MAXDEPTH <= 5;   // served by rnea_grad_tree_kernel<double>
  if constexpr (!tws_ok<T>() || reg_plan) return false;
  else return GRAD_TREE_DEFAULT || !RBD_FAST_STAGE;
#endif
}
template <class T>
#ifdef RBD_TWS_FORCE
constexpr bool tws_only() { return tws_built<T>(); }
#else
constexpr bool tws_only() { return tws_built<T>() && rbdk::GRAD_TREE_DEFAULT; }
#endif
template <class T, bool HAS_QDD>
int tree_ws_launch(const T* q, const T* qd, const T* qdd, T gravity, int use_damping, int64_t B, T* c, T* dc_du, void* stream) {
  using namespace rbdk;
  if constexpr (!tws_built<T>()) {
    return fail(RBD_ERR_UNSUPPORTED, "rbd_rnea_grad: the workspace tree kernel is not built for this robot");
  } else {
    constexpr size_t lds = tws_lds_bytes<T>();
    auto k = rnea_grad_tree_ws_kernel<T, HAS_QDD>;
    int rc, resident = 0;
    if ((rc = ensure_lds(k, lds)) != 0) return rc;
    if ((rc = resident_blocks(k, 64 * TWS_W, lds, &resident)) != 0) return rc;
    const int yroots = TWS_MULTI ? 1 : tree_n_roots();
    // one launch covers what is resident at once; larger batches walk the same workspace chunk by chunk
    int64_t xblocks = resident / yroots;
    if (xblocks < 1) xblocks = 1;
    const int64_t need = (B + 63) / 64;
    if (xblocks > need) xblocks = need;
    const int64_t rows = xblocks * 64;
    // sized by what is resident at once (never by B), one region per (x, root) block of the single-wave layout: a
    // (device, stream) sees ONE allocation for this kernel, whatever batch sizes follow (rbd_stream_workspace)
    const int64_t xres = resident / yroots > 0 ? resident / yroots : 1;
    void* ws = nullptr;
    if ((rc = rbd_stream_workspace(stream, (size_t)xres * 64 * yroots * TWS_SLOTS * sizeof(T), &ws)) != 0) return rc;
    T* pws = reinterpret_cast<T*>(ws);
    T* ews = pws + (size_t)64 * TWS_PATH_SLOTS;   // [block][slot][lane]: a block's entry slots follow its path slots
    for (int64_t r0 = 0; r0 < B; r0 += rows) {
      const int64_t nb = B - r0 < rows ? B - r0 : rows;
      hipLaunchKernelGGL(k, dim3((unsigned)((nb + 63) / 64), yroots), dim3(64 * TWS_W), lds, (hipStream_t)stream, q + r0 * N, qd + r0 * N,
                         qdd ? qdd + r0 * N : nullptr, gravity, use_damping, (long long)nb, c ? c + r0 * N : nullptr,
                         dc_du + r0 * (2 * N * N), pws, ews);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return hip_fail(e, "rbd_rnea_grad (workspace tree kernel) launch");
    }
    return 0;
  }
}
#endif  // RBD_HAVE_TWS

// One instantiation per (T, HAS_QDD): the forward-dynamics units only ever need HAS_QDD = true.
template <class T, bool HAS_QDD>
int rnea_grad_launch_q(const T* q, const T* qd, const T* qdd, T gravity, int use_damping, int64_t B,
                       T* c, T* dc_du, void* stream) {
  using namespace rbdk;
  if (grad_use_cols<T>(B)) return grad_cols_launch<T, HAS_QDD>(q, qd, qdd, gravity, use_damping, B, c, nullptr, nullptr, nullptr, dc_du, stream);
  // fp32 robots whose default is the tree kernel never build the column kernel (Atlas: 404 VGPRs of
  // code nobody runs); fp64 x big tree would need > 512 VGPRs and is not built either.
  if constexpr (tws_built<T>()) {
    if (tws_only<T>() || rbd_option(RBD_OPT_GRAD_KERNEL) == RBD_GRAD_KERNEL_TREE)
      return tree_ws_launch<T, HAS_QDD>(q, qd, qdd, gravity, use_damping, B, c, dc_du, stream);
  }
  constexpr bool TREE_ONLY = GRAD_TREE_DEFAULT && sizeof(T) == 4;
  constexpr bool TREE_BUILT = RBD_FAST_STAGE ? TREE_ONLY : GRAD_TREE_OK && (sizeof(T) == 4 || (N <= 12 && rbdm::MAXDEPTH <= 5));   // fp64: the root path's S / psid / psidd (36 registers per body) must fit 512 VGPRs without scratch
  if constexpr (TREE_BUILT) {
    constexpr size_t lds = tree_lds_bytes<T>();
    static_assert(!TREE_ONLY || lds <= 160 * 1024, "tree kernel is the only gradient kernel of this robot but does not fit LDS");
    const bool use_tree = TREE_ONLY || rbd_option(RBD_OPT_GRAD_KERNEL) == RBD_GRAD_KERNEL_TREE;
    if (use_tree && lds <= 160 * 1024) {
      const int64_t blocks = (B + 63) / 64;
      if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_rnea_grad: B too large");
      int rc;
      auto k = rnea_grad_tree_kernel<T, HAS_QDD>;
      if ((rc = ensure_lds(k, lds)) != 0) return rc;
      hipLaunchKernelGGL(k, dim3((unsigned)blocks, TREE_MULTI ? 1 : tree_n_roots()), dim3(64 * TREE_W), lds, (hipStream_t)stream, q, qd, qdd, gravity, use_damping, (long long)B, c, dc_du);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return hip_fail(e, "rbd_rnea_grad (tree kernel) launch");
      return 0;
    }
  }
  if constexpr (TREE_ONLY || tws_only<T>()) {
    return fail(RBD_ERR_UNSUPPORTED, "rbd_rnea_grad: no kernel");   // unreachable (static_assert / return above)
  } else if constexpr (grad_chain_kernel<T>()) {
    // one lane per configuration, tile-walking blocks (rbd_idsva.h)
    return idsva_launch<T, HAS_QDD, false>(q, qd, qdd, gravity, use_damping, B, c, dc_du, stream, nullptr);
  } else {
    return rnea_grad_launch1<T, HAS_QDD, false>(q, qd, qdd, gravity, use_damping, B, c, dc_du, stream, nullptr);
  }
}

template <class T>
int rnea_grad_launch(const T* q, const T* qd, const T* qdd, T gravity, int use_damping, int64_t B,
                     T* c, T* dc_du, void* stream) {
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_rnea_grad: B < 0");
  if (B == 0) return 0;
  if (!q || !qd || !dc_du) return fail(RBD_ERR_ARG, "rbd_rnea_grad: q, qd and dc_du must be non-null");
  if (((reinterpret_cast<uintptr_t>(dc_du) | reinterpret_cast<uintptr_t>(c)) & 15u) != 0)
    return fail(RBD_ERR_ARG, "rbd_rnea_grad: output buffers must be 16-byte aligned");
  if (qdd) return rnea_grad_launch_q<T, true>(q, qd, qdd, gravity, use_damping, B, c, dc_du, stream);
  // qdd = None (:589): the HAS_QDD = false kernels live in the GRADN unit
  if constexpr (sizeof(T) == 4) return rbd_grad_noqdd_f32((const float*)q, (const float*)qd, (float)gravity, use_damping, B, (float*)c, (float*)dc_du, stream);
  else return rbd_grad_noqdd_f64((const double*)q, (const double*)qd, (double)gravity, use_damping, B, (double*)c, (double*)dc_du, stream);
}

// rnea + rnea_grad: (c, v, a, f, dc_du).  One launch when the column kernel serves the batch, otherwise the
// rnea kernel of the RNEA unit followed by the gradient kernel on the same stream.
template <class T>
int rnea_with_grad_launch(const T* q, const T* qd, const T* qdd, T gravity, int use_damping, int64_t B,
                          T* c, T* v, T* a, T* f, T* dc_du, void* stream) {
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_rnea_with_grad: B < 0");
  if (B == 0) return 0;
  if (!q || !qd || !c || !v || !a || !f || !dc_du) return fail(RBD_ERR_ARG, "rbd_rnea_with_grad: q, qd, c, v, a, f, dc_du must be non-null");
  if (((reinterpret_cast<uintptr_t>(dc_du) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(a) |
        reinterpret_cast<uintptr_t>(f)) & 15u) != 0)
    return fail(RBD_ERR_ARG, "rbd_rnea_with_grad: output buffers must be 16-byte aligned");
  if (grad_use_cols<T>(B)) {
    if (qdd) return grad_cols_launch<T, true>(q, qd, qdd, gravity, use_damping, B, c, v, a, f, dc_du, stream);
    if constexpr (sizeof(T) == 4) return rbd_grad_cols_noqdd_f32((const float*)q, (const float*)qd, (float)gravity, use_damping, B, (float*)c, (float*)v, (float*)a, (float*)f, (float*)dc_du, stream);
    else return rbd_grad_cols_noqdd_f64((const double*)q, (const double*)qd, (double)gravity, use_damping, B, (double*)c, (double*)v, (double*)a, (double*)f, (double*)dc_du, stream);
  }
  int rc;
  if constexpr (sizeof(T) == 4) rc = rbd_rnea_f32((const float*)q, (const float*)qd, (const float*)qdd, (float)gravity, B, (float*)c, (float*)v, (float*)a, (float*)f, stream);
  else rc = rbd_rnea_f64((const double*)q, (const double*)qd, (const double*)qdd, (double)gravity, B, (double*)c, (double*)v, (double*)a, (double*)f, stream);
  if (rc != 0) return rc;
  return rnea_grad_launch<T>(q, qd, qdd, gravity, use_damping, B, nullptr, dc_du, stream);
}

// name of the kernel rnea_grad_launch<T> would run (HAS_QDD = true) under the current options
template <class T>
int grad_kernel_name(int64_t B, char* buf, size_t len) {
  using namespace rbdk;
  const char* t = sizeof(T) == 4 ? "float" : "double";
  if (grad_use_cols<T>(B)) { std::snprintf(buf, len, "rnea_grad_cols_kernel<%s,true>", t); return 0; }
  constexpr bool TREE_ONLY = GRAD_TREE_DEFAULT && sizeof(T) == 4;
  constexpr bool TREE_BUILT = RBD_FAST_STAGE ? TREE_ONLY : GRAD_TREE_OK && (sizeof(T) == 4 || (N <= 12 && rbdm::MAXDEPTH <= 5));   // fp64: the root path's S / psid / psidd (36 registers per body) must fit 512 VGPRs without scratch
  bool tree = TREE_ONLY;
  if constexpr (TREE_BUILT) tree = tree || (rbd_option(RBD_OPT_GRAD_KERNEL) == RBD_GRAD_KERNEL_TREE && tree_lds_bytes<T>() <= 160 * 1024);
  bool tws = false;
  if constexpr (tws_built<T>()) tws = tws_only<T>() || rbd_option(RBD_OPT_GRAD_KERNEL) == RBD_GRAD_KERNEL_TREE;
  if (tws) std::snprintf(buf, len, "rnea_grad_tree_ws_kernel<%s,true>", t);
  else if (tree) std::snprintf(buf, len, "rnea_grad_tree_kernel<%s,true>", t);
  else if (grad_chain_kernel<T>()) {
#ifndef RBD_EXP_NO_PIPE
    if (IDS_PIPE_OK && sizeof(T) == 4) std::snprintf(buf, len, "rnea_grad_idsva_pipe_kernel<%s,true,false>", t);
    else
#endif
    std::snprintf(buf, len, "rnea_grad_idsva_kernel<%s,true,false>", t);
  }
  else std::snprintf(buf, len, "rnea_grad_kernel<%s,true,false>", t);
  return 0;
}

#endif  // RBD_NEED_GRAD

#ifdef RBD_NEED_MINV
template <class T>
int minv_launch(const T* q, int64_t B, int output_dense, T* Minv, void* workspace, size_t wsb, void* stream,
                const T* u = nullptr, const T* cbias = nullptr, T* qdd = nullptr, const T* qd = nullptr, T gravity = T(0)) {
  using namespace rbdk;
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_minv: B < 0");
  if (B == 0) return 0;
  if (!q || (!Minv && !qdd)) return fail(RBD_ERR_ARG, "rbd_minv: q and Minv must be non-null");
  if ((reinterpret_cast<uintptr_t>(Minv) & 15u) != 0) return fail(RBD_ERR_ARG, "rbd_minv: Minv must be 16-byte aligned");
  if constexpr (minv_use_lane<T>()) {
    // fused one-lane-per-configuration kernel (rbd_minv_lane.h): no workspace
    const int64_t blocks = (B + 63) / 64;
    if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_minv: B too large");
    const size_t lds = sizeof(T) * (size_t)64 * MINV_LANE_TS;
    auto k = minv_lane_kernel<T>;
    int rc;
    if ((rc = ensure_lds(k, lds)) != 0) return rc;
    // (qd given instead of c: the kernel computes the bias force itself, rbd_minv_lane.h)
    if (qdd && !cbias && !qd) return fail(RBD_ERR_ARG, "rbd_minv (forward dynamics): c or qd must be given");
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, q, (long long)B, output_dense, Minv, u, cbias, qdd,
                       cbias ? (const T*)nullptr : qd, gravity);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rbd_minv launch");
    return 0;
  } else {
  if (qdd && !cbias) return fail(RBD_ERR_ARG, "rbd_minv (forward dynamics): this robot's kernels need the bias force c");
  hipStream_t s = (hipStream_t)stream;
  const int pa = rbd_option(RBD_OPT_MINV_PHASE_A);
  // phase A: one lane per configuration when that alone fills the chip (>= 4 waves per SIMD),
  // otherwise eight lanes per configuration (rbd_minv_ia8.h), which also finishes the groups of <= 8 bodies
  // a robot whose big groups have limbs: everything in one launch (rbd_minv_fused.h); measured on Atlas against the
  // two launches: 11.5 vs 17.3 us at B = 4 096, 27.0 vs 33.8 at 16 384, 174 vs 233 at 131 072, 785 vs 928 at 524 288
  if constexpr (MINV_FUSED_OK && mf_lds_bytes<T>() <= 160 * 1024) {
    if (pa == RBD_MINV_PHASE_A_FUSED || pa == RBD_MINV_PHASE_A_AUTO) {
      const int64_t nbf = mf_blocks(B);
      if (nbf > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_minv: B too large");
      constexpr size_t ldsf = mf_lds_bytes<T>();
      auto kf = minv_fused_kernel<T>;
      int rcf;
      if ((rcf = ensure_lds(kf, ldsf)) != 0) return rcf;
      hipLaunchKernelGGL(kf, dim3((unsigned)nbf), dim3(64 * MF_W), ldsf, s, q, (long long)B, output_dense, Minv, u, cbias, qdd);
      hipError_t ef = hipGetLastError();
      if (ef != hipSuccess) return hip_fail(ef, "rbd_minv (fused) launch");
      return 0;
    }
  }
  // the two-launch path is the only one that goes through the HBM workspace (rbd_minv_workspace_bytes reports 0
  // when the one-launch kernel is selected)
  const size_t need = (size_t)B * MINV_WS_PER_CFG * sizeof(T);
  if (!workspace || wsb < need) return fail(RBD_ERR_WORKSPACE, "rbd_minv: workspace missing or smaller than rbd_minv_workspace_bytes()");
  if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) return fail(RBD_ERR_WORKSPACE, "rbd_minv: workspace must be 16-byte aligned");
  T* ws = reinterpret_cast<T*>(workspace);
  const bool lane_a = pa == RBD_MINV_PHASE_A_LANE || (pa != RBD_MINV_PHASE_A_IA8 && rbd_select_batch(B) >= 64 * 1024 * 4);
  const int64_t blocksA = (B + 63) / 64, blocksB = minv_cols_blocks(B, !lane_a);
  if (blocksB > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_minv: B too large");
  if (lane_a) {
    hipLaunchKernelGGL(minv_ia_kernel<T>, dim3((unsigned)blocksA), dim3(64), 0, s, q, (long long)B, ws);
  } else {
    hipLaunchKernelGGL(minv_ia8_kernel<T>, dim3((unsigned)((B + 7) / 8), n_groups()), dim3(64), 0, s, q, (long long)B, ws, 1,
                       output_dense, Minv, u, cbias, qdd);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rbd_minv phase A launch");
  if (blocksB > 0) {
    constexpr size_t lds = minv_cols_lds_bytes<T>();
    auto k = minv_cols_kernel<T>;
    int rc;
    if ((rc = ensure_lds(k, lds)) != 0) return rc;
    hipLaunchKernelGGL(k, dim3((unsigned)blocksB), dim3(64 * MINV_COLS_W), lds, s, (const T*)ws, (long long)B, output_dense, Minv, u, cbias,
                       qdd, lane_a ? 0 : 1);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rbd_minv phase B launch");
  return 0;
  }
}

// name of the dominant kernel minv_launch<T> would run for B rows under the current options
template <class T>
int minv_kernel_name(int64_t B, char* buf, size_t len) {
  const char* t = sizeof(T) == 4 ? "float" : "double";
  const int pa = rbd_option(RBD_OPT_MINV_PHASE_A);
  bool fused = false;
  if constexpr (!rbdk::minv_use_lane<T>()) {
    if constexpr (rbdk::MINV_FUSED_OK && rbdk::mf_lds_bytes<T>() <= 160 * 1024)
      fused = pa == RBD_MINV_PHASE_A_FUSED || pa == RBD_MINV_PHASE_A_AUTO;
  }
  if (rbdk::minv_use_lane<T>()) std::snprintf(buf, len, "minv_lane_kernel<%s>", t);
  else if (fused) std::snprintf(buf, len, "minv_fused_kernel<%s>", t);
  else std::snprintf(buf, len, "minv_cols_kernel<%s>", t);
  return 0;
}

// does minv_launch<T> go through the HBM workspace under the current options?
template <class T>
int minv_needs_workspace() {
  if constexpr (rbdk::minv_use_lane<T>()) return 0;
  if constexpr (rbdk::MINV_FUSED_OK && rbdk::mf_lds_bytes<T>() <= 160 * 1024) {
    const int pa = rbd_option(RBD_OPT_MINV_PHASE_A);
    if (pa == RBD_MINV_PHASE_A_FUSED || pa == RBD_MINV_PHASE_A_AUTO) return 0;
  }
  return 1;
}

#endif  // RBD_NEED_MINV

#if defined(RBD_TU_MINV_F32) || defined(RBD_TU_MINV_F64)
template <class T>
int crba_launch(const T* q, int64_t B, T* H, void* stream) {
  using namespace rbdk;
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_crba: B < 0");
  if (B == 0) return 0;
  if (!q || !H) return fail(RBD_ERR_ARG, "rbd_crba: q and H must be non-null");
  const int64_t blocks = (B + 63) / 64;
  if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_crba: B too large");
  const size_t lds = crba_tile_fits<T>() ? sizeof(T) * (size_t)64 * CRBA_TS : 0;
  auto k = crba_kernel<T>;
  int rc;
  if ((rc = ensure_lds(k, lds)) != 0) return rc;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, q, (long long)B, H);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rbd_crba launch");
  return 0;
}

// ---- forward dynamics: compositions of the three kernels with fused epilogues ----
#endif

constexpr size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
#ifdef RBD_NEED_FD
template <class T>
int aba_launch(const T* q, const T* qd, const T* tau, T gravity, int64_t B, T* qdd, void* stream) {
  using namespace rbdk;
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_aba: B < 0");
  if (B == 0) return 0;
  if (!q || !qd || !tau || !qdd) return fail(RBD_ERR_ARG, "rbd_aba: q, qd, tau and qdd must be non-null");
  constexpr int lanes = ABA_PARK ? aba_lanes<T>() : 64;
  const int64_t blocks = (B + lanes - 1) / lanes;
  if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_aba: B too large");
  constexpr size_t lds = aba_lds_bytes<T>();
  if (lds > 160 * 1024) return fail(RBD_ERR_UNSUPPORTED, "rbd_aba: per-body state does not fit LDS for this robot size");
  if (int rc = ensure_lds(aba_kernel<T>, lds)) return rc;
  hipLaunchKernelGGL(aba_kernel<T>, dim3((unsigned)blocks, ABA_PARK ? n_groups() : 1), dim3(64), lds, (hipStream_t)stream, q, qd, tau,
                     gravity, (long long)B, qdd);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rbd_aba launch");
  return 0;
}

#endif  // RBD_NEED_FD

template <class T>
struct FdWorkspace {
  size_t off_minv_ws, off_c, off_minv, off_qdd, off_dcdu, total;
  explicit FdWorkspace(int64_t B) {
    using namespace rbdk;
    size_t o = 0;
    off_minv_ws = o; o += align16((size_t)B * MINV_WS_PER_CFG * sizeof(T));
    off_c = o;       o += align16((size_t)B * N * sizeof(T));
    // (one-chain fp32 robots keep the packed upper triangle of Minv here, in whole tiles: rbd_fd_chain.h)
    const size_t dense = (size_t)B * N * N, packed = (size_t)((B + 63) / 64) * 64 * (N * (N + 1) / 2);
    off_minv = o;    o += align16((dense > packed ? dense : packed) * sizeof(T));
    off_qdd = o;     o += align16((size_t)B * N * sizeof(T));
    off_dcdu = o;    o += GRAD_ACC_IN_REGS ? 0 : align16((size_t)B * 2 * N * N * sizeof(T));
    total = o;
  }
};

#ifdef RBD_NEED_FD
template <class T>
int fd_launch(const T* q, const T* qd, const T* u, T gravity, int64_t B, T* qdd, T* dqdd_du, bool want_grad,
              void* workspace, size_t wsb, void* stream) {
  using namespace rbdk;
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_forward_dynamics: B < 0");
  if (B == 0) return 0;
  if (!q || !qd || !u) return fail(RBD_ERR_ARG, "rbd_forward_dynamics: q, qd, u must be non-null");
  if (want_grad ? !dqdd_du : !qdd) return fail(RBD_ERR_ARG, "rbd_forward_dynamics: output pointer is null");
  // refused HERE, before the first of the three launches (the gradient launcher's own check would come after two of them)
  if (((reinterpret_cast<uintptr_t>(qdd) | reinterpret_cast<uintptr_t>(dqdd_du)) & 15u) != 0)
    return fail(RBD_ERR_ARG, "rbd_forward_dynamics: output buffers must be 16-byte aligned");
  // qdd alone: the articulated-body sweep gives Minv (u - c) (:1372-1374) without forming Minv or c
  // (one launch, no workspace; 47 vs 76 us for the 7-DoF arm at B = 1M)
  if (!want_grad) return aba_launch<T>(q, qd, u, gravity, B, qdd, stream);
  const FdWorkspace<T> L(B);
  if (!workspace || wsb < L.total) return fail(RBD_ERR_WORKSPACE, "rbd_forward_dynamics: workspace missing or smaller than rbd_fd_workspace_bytes()");
  if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) return fail(RBD_ERR_WORKSPACE, "rbd_forward_dynamics: workspace must be 16-byte aligned");
  char* w = reinterpret_cast<char*>(workspace);
  T* c = reinterpret_cast<T*>(w + L.off_c);
  T* Mi = reinterpret_cast<T*>(w + L.off_minv);
  T* qdd_buf = qdd ? qdd : reinterpret_cast<T*>(w + L.off_qdd);
  int rc;
  if constexpr (fd_chain_ok<T>() && grad_chain_kernel<T>()) {
    // one chain (rbd_fd_chain.h): fd_pre_kernel (bias force, Minv, qdd in one lane; Minv's upper triangle to a lane-major
    // workspace), then the world-frame chain gradient kernel with the -Minv product on its finished entries -- the
    // software-pipelined kernel in fp32 where it applies
    const int64_t tiles = (B + 63) / 64;
    if (tiles > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_forward_dynamics_grad: B too large");
    hipLaunchKernelGGL(fd_pre_kernel<T>, dim3((unsigned)tiles), dim3(64), 0, (hipStream_t)stream, q, qd, u, gravity, (long long)B, qdd_buf, Mi);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rbd_forward_dynamics_grad (fd_pre_kernel) launch");
    const size_t lds = sizeof(T) * (size_t)64 * IDS_TS;
    int resident = 0;
#ifndef RBD_EXP_NO_PIPE
    if constexpr (IDS_PIPE_OK && sizeof(T) == 4) {
      auto kp = rnea_grad_idsva_pipe_kernel<T, true, true>;
      if ((rc = ensure_lds(kp, lds)) != 0) return rc;
      if ((rc = resident_blocks(kp, 64, lds, &resident)) != 0) return rc;
      const int64_t blocks = tiles < resident ? tiles : resident;
      hipLaunchKernelGGL(kp, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, q, qd, (const T*)qdd_buf, gravity, 0, (long long)B,
                         (T*)nullptr, dqdd_du, (const T*)Mi);
    } else
#endif
    {
      auto k = rnea_grad_idsva_kernel<T, true, true>;
      if ((rc = ensure_lds(k, lds)) != 0) return rc;
      if ((rc = resident_blocks(k, 64, lds, &resident)) != 0) return rc;
      const int64_t blocks = tiles < resident ? tiles : resident;
      hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, q, qd, (const T*)qdd_buf, gravity, 0, (long long)B,
                         (T*)nullptr, dqdd_du, (const T*)Mi);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rbd_forward_dynamics_grad (gradient kernel) launch");
    return 0;
  } else {
  // c = rnea(q, qd) with qdd = None (:1372): the c-only kernel of the RNEA unit -- or, where the one-lane minv kernel serves the
  // robot, no launch at all: that kernel computes the bias force of its groups from qd itself (rbd_minv_lane.h)
  constexpr bool bias_in_minv = minv_use_lane<T>();
  if constexpr (!bias_in_minv) {
    if constexpr (sizeof(T) == 4) rc = rbd_rnea_f32((const float*)q, (const float*)qd, nullptr, (float)gravity, B, (float*)c, nullptr, nullptr, nullptr, stream);
    else rc = rbd_rnea_f64((const double*)q, (const double*)qd, nullptr, (double)gravity, B, (double*)c, nullptr, nullptr, nullptr, stream);
    if (rc != 0) return rc;
  }
  const T* cb = bias_in_minv ? nullptr : c;
  // qdd = Minv (u - c) (:1373-1374), fused into the last phase of minv (MINV unit)
  if constexpr (sizeof(T) == 4) rc = rbd_minv_fd_f32((const float*)q, B, (float*)Mi, w + L.off_minv_ws, (size_t)B * MINV_WS_PER_CFG * sizeof(T), stream, (const float*)u, (const float*)cb, (float*)qdd_buf, (const float*)qd, (float)gravity);
  else rc = rbd_minv_fd_f64((const double*)q, B, (double*)Mi, w + L.off_minv_ws, (size_t)B * MINV_WS_PER_CFG * sizeof(T), stream, (const double*)u, (const double*)cb, (double*)qdd_buf, (const double*)qd, (double)gravity);
  if (rc != 0) return rc;
  if (!want_grad) return 0;
  // [qdd_dq | qdd_dqd] = -Minv rnea_grad(q, qd, qdd) (:1378-1383)
  if constexpr (GRAD_ACC_IN_REGS) {
    return rnea_grad_launch1<T, true, true>(q, qd, qdd_buf, gravity, 0, B, nullptr, dqdd_du, stream, Mi);
  } else {
    T* dc = reinterpret_cast<T*>(w + L.off_dcdu);
    // plain rnea_grad of the GRAD unit, then the -Minv product
    if constexpr (sizeof(T) == 4) rc = rbd_rnea_grad_f32((const float*)q, (const float*)qd, (const float*)qdd_buf, (float)gravity, 0, B, nullptr, (float*)dc, stream);
    else rc = rbd_rnea_grad_f64((const double*)q, (const double*)qd, (const double*)qdd_buf, (double)gravity, 0, B, nullptr, (double*)dc, stream);
    if (rc != 0) return rc;
    constexpr int MMC = negmm_cfgs<T, N>();
    const int64_t ablocks = (B + MMC - 1) / MMC;
    if (ablocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_forward_dynamics_grad: B too large");
    hipLaunchKernelGGL((neg_mm_kernel<T, N>), dim3((unsigned)ablocks), dim3(negmm_threads<T, N>()), 0, (hipStream_t)stream,
                       (const T*)Mi, (const T*)dc, (long long)B, dqdd_du);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rbd_forward_dynamics_grad apply launch");
    return 0;
  }
  }
}
#endif  // RBD_NEED_FD

#if defined(RBD_TU_PASS_F32) || defined(RBD_TU_PASS_F64)
// ---- per-pass entry points (rbd_passes.h) ------------------------------------------------------------
int pass_blocks(int64_t B, const char* who, unsigned* blocks) {
  if (B < 0) { std::snprintf(rbd_err_buf(), RBD_ERR_LEN, "%s: B < 0", who); return RBD_ERR_ARG; }
  const int64_t nb = (B + 63) / 64;
  if (nb > 0x7fffffffLL) { std::snprintf(rbd_err_buf(), RBD_ERR_LEN, "%s: B too large", who); return RBD_ERR_ARG; }
  *blocks = (unsigned)nb;
  return 0;
}
int pass_done(const char* who) {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, who);
}
template <class T, bool DQ>
int grad_fpass_launch(const T* q, const T* qd, const T* v, const T* a, T gravity, int64_t B, T* dv, T* da, T* df, void* stream) {
  const char* who = DQ ? "rbd_rnea_grad_fpass_dq" : "rbd_rnea_grad_fpass_dqd";
  unsigned blocks;
  if (int rc = pass_blocks(B, who, &blocks)) return rc;
  if (B == 0) return 0;
  if (!q || !qd || !v || (DQ && !a) || !dv || !da || !df) {
    std::snprintf(rbd_err_buf(), RBD_ERR_LEN, "%s: null pointer argument", who);
    return RBD_ERR_ARG;
  }
  hipLaunchKernelGGL((rbdk::grad_fpass_kernel<T, DQ>), dim3(blocks), dim3(64), 0, (hipStream_t)stream, q, qd, v, a, gravity,
                     (long long)B, dv, da, df);
  return pass_done(who);
}
template <class T, bool DQ>
int grad_bpass_launch(const T* q, const T* f, T* df, int use_damping, int64_t B, T* dc, void* stream) {
  const char* who = DQ ? "rbd_rnea_grad_bpass_dq" : "rbd_rnea_grad_bpass_dqd";
  unsigned blocks;
  if (int rc = pass_blocks(B, who, &blocks)) return rc;
  if (B == 0) return 0;
  if (!q || (DQ && !f) || !df || !dc) {
    std::snprintf(rbd_err_buf(), RBD_ERR_LEN, "%s: null pointer argument", who);
    return RBD_ERR_ARG;
  }
  hipLaunchKernelGGL((rbdk::grad_bpass_kernel<T, DQ>), dim3(blocks), dim3(64), 0, (hipStream_t)stream, q, f, df, use_damping,
                     (long long)B, dc);
  return pass_done(who);
}
template <class T>
int minv_bpass_launch(const T* q, int64_t B, T* Minv, T* F, T* U, T* D, void* stream) {
  unsigned blocks;
  if (int rc = pass_blocks(B, "rbd_minv_bpass", &blocks)) return rc;
  if (B == 0) return 0;
  if (!q || !Minv || !F || !U || !D) return fail(RBD_ERR_ARG, "rbd_minv_bpass: null pointer argument");
  hipLaunchKernelGGL(rbdk::minv_bpass_kernel<T>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, q, (long long)B, Minv, F, U, D);
  return pass_done("rbd_minv_bpass");
}
template <class T>
int minv_fpass_launch(const T* q, int64_t B, T* Minv, T* F, const T* U, const T* D, void* stream) {
  unsigned blocks;
  if (int rc = pass_blocks(B, "rbd_minv_fpass", &blocks)) return rc;
  if (B == 0) return 0;
  if (!q || !Minv || !F || !U || !D) return fail(RBD_ERR_ARG, "rbd_minv_fpass: null pointer argument");
  hipLaunchKernelGGL(rbdk::minv_fpass_kernel<T>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, q, (long long)B, Minv, F, U, D);
  return pass_done("rbd_minv_fpass");
}
#endif
}  // namespace

extern "C" {

// kernel names: every family unit answers for its own kernels (the selection logic lives there)
__attribute__((visibility("hidden"))) int rbd_grad_kernel_name_f32(int64_t B, char* buf, size_t len);
__attribute__((visibility("hidden"))) int rbd_grad_kernel_name_f64(int64_t B, char* buf, size_t len);
__attribute__((visibility("hidden"))) int rbd_minv_kernel_name_f32(int64_t B, char* buf, size_t len);
__attribute__((visibility("hidden"))) int rbd_minv_kernel_name_f64(int64_t B, char* buf, size_t len);
__attribute__((visibility("hidden"))) int rbd_minv_needs_ws_f32(void);
__attribute__((visibility("hidden"))) int rbd_minv_needs_ws_f64(void);
__attribute__((visibility("hidden"))) int rbd_rnea_kernel_name_f32(int64_t B, char* buf, size_t len);
__attribute__((visibility("hidden"))) int rbd_rnea_kernel_name_f64(int64_t B, char* buf, size_t len);
#ifdef RBD_TU_RNEA_F32
int rbd_rnea_kernel_name_f32(int64_t B, char* buf, size_t len) { return rnea_kernel_name<float>(B, buf, len); }
#endif
#ifdef RBD_TU_RNEA_F64
int rbd_rnea_kernel_name_f64(int64_t B, char* buf, size_t len) { return rnea_kernel_name<double>(B, buf, len); }
#endif
#ifdef RBD_TU_GRAD_F32
int rbd_grad_kernel_name_f32(int64_t B, char* buf, size_t len) { return grad_kernel_name<float>(B, buf, len); }
#endif
#ifdef RBD_TU_GRAD_F64
int rbd_grad_kernel_name_f64(int64_t B, char* buf, size_t len) { return grad_kernel_name<double>(B, buf, len); }
#endif
#ifdef RBD_TU_MINV_F32
int rbd_minv_kernel_name_f32(int64_t B, char* buf, size_t len) { return minv_kernel_name<float>(B, buf, len); }
int rbd_minv_needs_ws_f32(void) { return minv_needs_workspace<float>(); }
#endif
#ifdef RBD_TU_MINV_F64
int rbd_minv_kernel_name_f64(int64_t B, char* buf, size_t len) { return minv_kernel_name<double>(B, buf, len); }
int rbd_minv_needs_ws_f64(void) { return minv_needs_workspace<double>(); }
#endif

#ifdef RBD_TU_COMMON
int rbd_abi_version(void) { return RBD_ABI_VERSION; }
const char* rbd_last_error(void) { return rbd_err_buf(); }

int rbd_set_option(int option, int value) {
  std::atomic<int>* s = rbd_option_slot(option);
  if (!s) return fail(RBD_ERR_ARG, "rbd_set_option: unknown option");
  if (value < 0 || (option != RBD_OPT_SELECT_BATCH && value > (option == RBD_OPT_RNEA_KERNEL ? 2 : 3)))
    return fail(RBD_ERR_ARG, "rbd_set_option: value out of range");
  s->store(value, std::memory_order_relaxed);
  return 0;
}
int rbd_get_option(int option) {
  std::atomic<int>* s = rbd_option_slot(option);
  return s ? s->load(std::memory_order_relaxed) : RBD_ERR_ARG;
}
int rbd_kernel_name(int op, int elem_size, int64_t B, char* buf, size_t len) {
  if (!buf || len == 0 || (elem_size != 4 && elem_size != 8)) return fail(RBD_ERR_ARG, "rbd_kernel_name: bad arguments");
  switch (op) {
    case RBD_OP_RNEA:
      return elem_size == 4 ? rbd_rnea_kernel_name_f32(B, buf, len) : rbd_rnea_kernel_name_f64(B, buf, len);
    case RBD_OP_RNEA_GRAD:
      return elem_size == 4 ? rbd_grad_kernel_name_f32(B, buf, len) : rbd_grad_kernel_name_f64(B, buf, len);
    case RBD_OP_MINV:
      return elem_size == 4 ? rbd_minv_kernel_name_f32(B, buf, len) : rbd_minv_kernel_name_f64(B, buf, len);
    default:
      return fail(RBD_ERR_ARG, "rbd_kernel_name: unknown op");
  }
}

int rbd_model_info(rbd_model_info_t* out) {
  if (!out) return fail(RBD_ERR_ARG, "rbd_model_info: out is null");
  std::memset(out, 0, sizeof(*out));
  out->abi_version = RBD_ABI_VERSION;
  out->n = rbdm::N;
  out->max_depth = rbdm::MAXDEPTH;
  out->hash = RBD_MODEL_HASH;
  out->floating_base = rbdm::FLOATING_BASE ? 1 : 0;
  out->nv = rbdm::NV;
  std::snprintf(out->name, sizeof(out->name), "%s", RBD_MODEL_NAME);
  for (int i = 0; i < rbdm::N && i < RBD_MAX_BODIES; ++i) {
    out->parent[i] = rbdm::PARENT[i];
    out->joint_type[i] = rbdm::JTYPE[i];
    out->joint_axis[i] = rbdm::AXIS[i];
  }
  return 0;
}
size_t rbd_minv_workspace_bytes(int64_t B, int elem_size) {
  if (B <= 0 || (elem_size != 4 && elem_size != 8)) return 0;
  if (rbdm::FLOATING_BASE) return 0;
  // none when the kernel selected by the current RBD_OPT_MINV_PHASE_A runs without it (the one-lane kernel, the
  // one-launch kernel): query again after changing that option
  if (!(elem_size == 4 ? rbd_minv_needs_ws_f32() : rbd_minv_needs_ws_f64())) return 0;
  return (size_t)B * rbdk::MINV_WS_PER_CFG * (size_t)elem_size;
}
size_t rbd_fd_workspace_bytes(int64_t B, int elem_size) {
  if (B <= 0) return 0;
  if (rbdm::FLOATING_BASE) {      // c [B, NV] | Minv [B, NV, NV] | qdd [B, NV] | dc_du [B, NV, 2 NV]  (rbd_fb_kernels.hip; the
    if (elem_size != 4 && elem_size != 8) return 0;   // last two serve forward_dynamics_grad only)
    return align16((size_t)B * rbdm::NV * elem_size) + align16((size_t)B * rbdm::NV * rbdm::NV * elem_size) +
           align16((size_t)B * rbdm::NV * elem_size) + align16((size_t)B * rbdm::NV * 2 * rbdm::NV * elem_size);
  }
  if (elem_size == 4) return FdWorkspace<float>(B).total;
  if (elem_size == 8) return FdWorkspace<double>(B).total;
  return 0;
}
#endif
#ifdef RBD_TU_RNEA_F32
int rbd_rnea_f32(const float* q, const float* qd, const float* qdd, float gravity, int64_t B,
                 float* c, float* v, float* a, float* f, void* stream) {
  RbdStreamDevice sd_(stream); return rnea_launch<float>(q, qd, qdd, gravity, B, c, v, a, f, stream);
}
#endif
#ifdef RBD_TU_RNEA_F32
int rbd_rnea_fpass_f32(const float* q, const float* qd, const float* qdd, float gravity, int64_t B,
                       float* v, float* a, float* f, void* stream) {
  RbdStreamDevice sd_(stream); return rnea_launch<float>(q, qd, qdd, gravity, B, nullptr, v, a, f, stream, 1);
}
int rbd_rnea_bpass_f32(const float* q, float* f, int64_t B, float* c, void* stream) {
  RbdStreamDevice sd_(stream); return rnea_bpass_launch<float>(q, f, B, c, stream);
}
#endif
#ifdef RBD_TU_RNEA_F64
int rbd_rnea_fpass_f64(const double* q, const double* qd, const double* qdd, double gravity, int64_t B,
                       double* v, double* a, double* f, void* stream) {
  RbdStreamDevice sd_(stream); return rnea_launch<double>(q, qd, qdd, gravity, B, nullptr, v, a, f, stream, 1);
}
int rbd_rnea_bpass_f64(const double* q, double* f, int64_t B, double* c, void* stream) {
  RbdStreamDevice sd_(stream); return rnea_bpass_launch<double>(q, f, B, c, stream);
}
#endif
#ifdef RBD_TU_RNEA_F64
int rbd_rnea_f64(const double* q, const double* qd, const double* qdd, double gravity, int64_t B,
                 double* c, double* v, double* a, double* f, void* stream) {
  RbdStreamDevice sd_(stream); return rnea_launch<double>(q, qd, qdd, gravity, B, c, v, a, f, stream);
}
#endif
#ifdef RBD_TU_GRADN_F32
int rbd_grad_noqdd_f32(const float* q, const float* qd, float gravity, int use_damping, int64_t B, float* c, float* dc_du, void* stream) {
  RbdStreamDevice sd_(stream); return rnea_grad_launch_q<float, false>(q, qd, nullptr, gravity, use_damping, B, c, dc_du, stream);
}
int rbd_grad_cols_noqdd_f32(const float* q, const float* qd, float gravity, int use_damping, int64_t B, float* c, float* v, float* a, float* f, float* dc_du, void* stream) {
  RbdStreamDevice sd_(stream); return grad_cols_launch<float, false>(q, qd, nullptr, gravity, use_damping, B, c, v, a, f, dc_du, stream);
}
#endif
#ifdef RBD_TU_GRADN_F64
int rbd_grad_noqdd_f64(const double* q, const double* qd, double gravity, int use_damping, int64_t B, double* c, double* dc_du, void* stream) {
  RbdStreamDevice sd_(stream); return rnea_grad_launch_q<double, false>(q, qd, nullptr, gravity, use_damping, B, c, dc_du, stream);
}
int rbd_grad_cols_noqdd_f64(const double* q, const double* qd, double gravity, int use_damping, int64_t B, double* c, double* v, double* a, double* f, double* dc_du, void* stream) {
  RbdStreamDevice sd_(stream); return grad_cols_launch<double, false>(q, qd, nullptr, gravity, use_damping, B, c, v, a, f, dc_du, stream);
}
#endif
#ifdef RBD_TU_GRAD_F32
int rbd_rnea_grad_f32(const float* q, const float* qd, const float* qdd, float gravity,
                      int use_damping, int64_t B, float* c, float* dc_du, void* stream) {
  RbdStreamDevice sd_(stream); return rnea_grad_launch<float>(q, qd, qdd, gravity, use_damping, B, c, dc_du, stream);
}
#endif
#ifdef RBD_TU_GRAD_F64
int rbd_rnea_grad_f64(const double* q, const double* qd, const double* qdd, double gravity,
                      int use_damping, int64_t B, double* c, double* dc_du, void* stream) {
  RbdStreamDevice sd_(stream); return rnea_grad_launch<double>(q, qd, qdd, gravity, use_damping, B, c, dc_du, stream);
}
int rbd_rnea_with_grad_f64(const double* q, const double* qd, const double* qdd, double gravity, int use_damping, int64_t B,
                           double* c, double* v, double* a, double* f, double* dc_du, void* stream) {
  RbdStreamDevice sd_(stream); return rnea_with_grad_launch<double>(q, qd, qdd, gravity, use_damping, B, c, v, a, f, dc_du, stream);
}
#endif
#ifdef RBD_TU_GRAD_F32
int rbd_rnea_with_grad_f32(const float* q, const float* qd, const float* qdd, float gravity, int use_damping, int64_t B,
                           float* c, float* v, float* a, float* f, float* dc_du, void* stream) {
  RbdStreamDevice sd_(stream); return rnea_with_grad_launch<float>(q, qd, qdd, gravity, use_damping, B, c, v, a, f, dc_du, stream);
}
#endif
#ifdef RBD_TU_MINV_F32
int rbd_crba_f32(const float* q, int64_t B, float* H, void* stream) { RbdStreamDevice sd_(stream); return crba_launch<float>(q, B, H, stream); }
#endif
#ifdef RBD_TU_MINV_F64
int rbd_crba_f64(const double* q, int64_t B, double* H, void* stream) { RbdStreamDevice sd_(stream); return crba_launch<double>(q, B, H, stream); }
#endif
#ifdef RBD_TU_MINV_F32
int rbd_minv_f32(const float* q, int64_t B, int output_dense, float* Minv, void* workspace,
                 size_t workspace_bytes, void* stream) {
  RbdStreamDevice sd_(stream); return minv_launch<float>(q, B, output_dense, Minv, workspace, workspace_bytes, stream);
}
int rbd_minv_fd_f32(const float* q, int64_t B, float* Minv, void* workspace, size_t wsb, void* stream,
                    const float* u, const float* c, float* qdd, const float* qd, float gravity) {
  RbdStreamDevice sd_(stream); return minv_launch<float>(q, B, 1, Minv, workspace, wsb, stream, u, c, qdd, qd, gravity);
}
#endif
#ifdef RBD_TU_FD_F32
int rbd_aba_f32(const float* q, const float* qd, const float* tau, float gravity, int64_t B, float* qdd, void* stream) {
  RbdStreamDevice sd_(stream); return aba_launch<float>(q, qd, tau, gravity, B, qdd, stream);
}
int rbd_forward_dynamics_f32(const float* q, const float* qd, const float* u, float gravity, int64_t B,
                             float* qdd, void* workspace, size_t workspace_bytes, void* stream) {
  RbdStreamDevice sd_(stream); return fd_launch<float>(q, qd, u, gravity, B, qdd, nullptr, false, workspace, workspace_bytes, stream);
}
int rbd_forward_dynamics_grad_f32(const float* q, const float* qd, const float* u, float gravity, int64_t B,
                                  float* qdd, float* dqdd_du, void* workspace, size_t workspace_bytes, void* stream) {
  RbdStreamDevice sd_(stream); return fd_launch<float>(q, qd, u, gravity, B, qdd, dqdd_du, true, workspace, workspace_bytes, stream);
}
#endif
#ifdef RBD_TU_FD_F64
int rbd_aba_f64(const double* q, const double* qd, const double* tau, double gravity, int64_t B, double* qdd, void* stream) {
  RbdStreamDevice sd_(stream); return aba_launch<double>(q, qd, tau, gravity, B, qdd, stream);
}
int rbd_forward_dynamics_f64(const double* q, const double* qd, const double* u, double gravity, int64_t B,
                             double* qdd, void* workspace, size_t workspace_bytes, void* stream) {
  RbdStreamDevice sd_(stream); return fd_launch<double>(q, qd, u, gravity, B, qdd, nullptr, false, workspace, workspace_bytes, stream);
}
int rbd_forward_dynamics_grad_f64(const double* q, const double* qd, const double* u, double gravity, int64_t B,
                                  double* qdd, double* dqdd_du, void* workspace, size_t workspace_bytes, void* stream) {
  RbdStreamDevice sd_(stream); return fd_launch<double>(q, qd, u, gravity, B, qdd, dqdd_du, true, workspace, workspace_bytes, stream);
}
#endif
#ifdef RBD_TU_MINV_F64
int rbd_minv_f64(const double* q, int64_t B, int output_dense, double* Minv, void* workspace,
                 size_t workspace_bytes, void* stream) {
  RbdStreamDevice sd_(stream); return minv_launch<double>(q, B, output_dense, Minv, workspace, workspace_bytes, stream);
}
int rbd_minv_fd_f64(const double* q, int64_t B, double* Minv, void* workspace, size_t wsb, void* stream,
                    const double* u, const double* c, double* qdd, const double* qd, double gravity) {
  RbdStreamDevice sd_(stream); return minv_launch<double>(q, B, 1, Minv, workspace, wsb, stream, u, c, qdd, qd, gravity);
}
#endif

#ifdef RBD_TU_PASS_F32
int rbd_rnea_grad_fpass_dq_f32(const float* q, const float* qd, const float* v, const float* a, float gravity, int64_t B,
                               float* dv_dq, float* da_dq, float* df_dq, void* stream) {
  RbdStreamDevice sd_(stream); return grad_fpass_launch<float, true>(q, qd, v, a, gravity, B, dv_dq, da_dq, df_dq, stream);
}
int rbd_rnea_grad_fpass_dqd_f32(const float* q, const float* qd, const float* v, int64_t B, float* dv_dqd, float* da_dqd,
                                float* df_dqd, void* stream) {
  RbdStreamDevice sd_(stream); return grad_fpass_launch<float, false>(q, qd, v, nullptr, float(0), B, dv_dqd, da_dqd, df_dqd, stream);
}
int rbd_rnea_grad_bpass_dq_f32(const float* q, const float* f, float* df_dq, int64_t B, float* dc_dq, void* stream) {
  RbdStreamDevice sd_(stream); return grad_bpass_launch<float, true>(q, f, df_dq, 0, B, dc_dq, stream);
}
int rbd_rnea_grad_bpass_dqd_f32(const float* q, float* df_dqd, int use_damping, int64_t B, float* dc_dqd, void* stream) {
  RbdStreamDevice sd_(stream); return grad_bpass_launch<float, false>(q, nullptr, df_dqd, use_damping, B, dc_dqd, stream);
}
int rbd_minv_bpass_f32(const float* q, int64_t B, float* Minv, float* F, float* U, float* Dinv, void* stream) {
  RbdStreamDevice sd_(stream); return minv_bpass_launch<float>(q, B, Minv, F, U, Dinv, stream);
}
int rbd_minv_fpass_f32(const float* q, int64_t B, float* Minv, float* F, const float* U, const float* Dinv, void* stream) {
  RbdStreamDevice sd_(stream); return minv_fpass_launch<float>(q, B, Minv, F, U, Dinv, stream);
}
#endif
#ifdef RBD_TU_PASS_F64
int rbd_rnea_grad_fpass_dq_f64(const double* q, const double* qd, const double* v, const double* a, double gravity, int64_t B,
                               double* dv_dq, double* da_dq, double* df_dq, void* stream) {
  RbdStreamDevice sd_(stream); return grad_fpass_launch<double, true>(q, qd, v, a, gravity, B, dv_dq, da_dq, df_dq, stream);
}
int rbd_rnea_grad_fpass_dqd_f64(const double* q, const double* qd, const double* v, int64_t B, double* dv_dqd, double* da_dqd,
                                double* df_dqd, void* stream) {
  RbdStreamDevice sd_(stream); return grad_fpass_launch<double, false>(q, qd, v, nullptr, double(0), B, dv_dqd, da_dqd, df_dqd, stream);
}
int rbd_rnea_grad_bpass_dq_f64(const double* q, const double* f, double* df_dq, int64_t B, double* dc_dq, void* stream) {
  RbdStreamDevice sd_(stream); return grad_bpass_launch<double, true>(q, f, df_dq, 0, B, dc_dq, stream);
}
int rbd_rnea_grad_bpass_dqd_f64(const double* q, double* df_dqd, int use_damping, int64_t B, double* dc_dqd, void* stream) {
  RbdStreamDevice sd_(stream); return grad_bpass_launch<double, false>(q, nullptr, df_dqd, use_damping, B, dc_dqd, stream);
}
int rbd_minv_bpass_f64(const double* q, int64_t B, double* Minv, double* F, double* U, double* Dinv, void* stream) {
  RbdStreamDevice sd_(stream); return minv_bpass_launch<double>(q, B, Minv, F, U, Dinv, stream);
}
int rbd_minv_fpass_f64(const double* q, int64_t B, double* Minv, double* F, const double* U, const double* Dinv, void* stream) {
  RbdStreamDevice sd_(stream); return minv_fpass_launch<double>(q, B, Minv, F, U, Dinv, stream);
}
#endif

// ---- stubs (-DRBD_TU_STUBS with -DRBD_STUB_<unit> per missing unit): a FAMILY library holds COMMON, the units of one
// family and these, so that it links and loads like a full library; an entry point of another family says so ------
#ifdef RBD_TU_STUBS
#define RBD_STUB_BODY(name) { return fail(RBD_ERR_NOT_BUILT, name ": not part of this family library (rbdreference_amd.build: first-use build)"); }
#define RBD_STUBS_RNEA(SFX, T)                                                                                                   \
  int rbd_rnea_kernel_name_##SFX(int64_t, char*, size_t) RBD_STUB_BODY("rbd_kernel_name(RBD_OP_RNEA)")                           \
  int rbd_rnea_##SFX(const T*, const T*, const T*, T, int64_t, T*, T*, T*, T*, void*) RBD_STUB_BODY("rbd_rnea")                  \
  int rbd_rnea_fpass_##SFX(const T*, const T*, const T*, T, int64_t, T*, T*, T*, void*) RBD_STUB_BODY("rbd_rnea_fpass")          \
  int rbd_rnea_bpass_##SFX(const T*, T*, int64_t, T*, void*) RBD_STUB_BODY("rbd_rnea_bpass")
#define RBD_STUBS_GRAD(SFX, T)                                                                                                   \
  int rbd_grad_kernel_name_##SFX(int64_t, char*, size_t) RBD_STUB_BODY("rbd_kernel_name(RBD_OP_RNEA_GRAD)")                      \
  int rbd_rnea_grad_##SFX(const T*, const T*, const T*, T, int, int64_t, T*, T*, void*) RBD_STUB_BODY("rbd_rnea_grad")           \
  int rbd_rnea_with_grad_##SFX(const T*, const T*, const T*, T, int, int64_t, T*, T*, T*, T*, T*, void*) RBD_STUB_BODY("rbd_rnea_with_grad")
#define RBD_STUBS_GRADN(SFX, T)                                                                                                  \
  int rbd_grad_noqdd_##SFX(const T*, const T*, T, int, int64_t, T*, T*, void*) RBD_STUB_BODY("rbd_rnea_grad (qdd = NULL)")      \
  int rbd_grad_cols_noqdd_##SFX(const T*, const T*, T, int, int64_t, T*, T*, T*, T*, T*, void*) RBD_STUB_BODY("rbd_rnea_with_grad (qdd = NULL)")
#define RBD_STUBS_MINV(SFX, T)                                                                                                   \
  int rbd_minv_kernel_name_##SFX(int64_t, char*, size_t) RBD_STUB_BODY("rbd_kernel_name(RBD_OP_MINV)")                           \
  int rbd_minv_needs_ws_##SFX(void) { return 1; }                                                                                \
  int rbd_crba_##SFX(const T*, int64_t, T*, void*) RBD_STUB_BODY("rbd_crba")                                                     \
  int rbd_minv_##SFX(const T*, int64_t, int, T*, void*, size_t, void*) RBD_STUB_BODY("rbd_minv")                                 \
  int rbd_minv_fd_##SFX(const T*, int64_t, T*, void*, size_t, void*, const T*, const T*, T*, const T*, T) RBD_STUB_BODY("rbd_minv")
#define RBD_STUBS_FD(SFX, T)                                                                                                     \
  int rbd_aba_##SFX(const T*, const T*, const T*, T, int64_t, T*, void*) RBD_STUB_BODY("rbd_aba")                                \
  int rbd_forward_dynamics_##SFX(const T*, const T*, const T*, T, int64_t, T*, void*, size_t, void*) RBD_STUB_BODY("rbd_forward_dynamics") \
  int rbd_forward_dynamics_grad_##SFX(const T*, const T*, const T*, T, int64_t, T*, T*, void*, size_t, void*) RBD_STUB_BODY("rbd_forward_dynamics_grad")
#define RBD_STUBS_PASS(SFX, T)                                                                                                   \
  int rbd_rnea_grad_fpass_dq_##SFX(const T*, const T*, const T*, const T*, T, int64_t, T*, T*, T*, void*) RBD_STUB_BODY("rbd_rnea_grad_fpass_dq") \
  int rbd_rnea_grad_fpass_dqd_##SFX(const T*, const T*, const T*, int64_t, T*, T*, T*, void*) RBD_STUB_BODY("rbd_rnea_grad_fpass_dqd") \
  int rbd_rnea_grad_bpass_dq_##SFX(const T*, const T*, T*, int64_t, T*, void*) RBD_STUB_BODY("rbd_rnea_grad_bpass_dq")           \
  int rbd_rnea_grad_bpass_dqd_##SFX(const T*, T*, int, int64_t, T*, void*) RBD_STUB_BODY("rbd_rnea_grad_bpass_dqd")              \
  int rbd_minv_bpass_##SFX(const T*, int64_t, T*, T*, T*, T*, void*) RBD_STUB_BODY("rbd_minv_bpass")                             \
  int rbd_minv_fpass_##SFX(const T*, int64_t, T*, T*, const T*, const T*, void*) RBD_STUB_BODY("rbd_minv_fpass")
#ifdef RBD_STUB_RNEA_F32
RBD_STUBS_RNEA(f32, float)
#endif
#ifdef RBD_STUB_RNEA_F64
RBD_STUBS_RNEA(f64, double)
#endif
#ifdef RBD_STUB_GRAD_F32
RBD_STUBS_GRAD(f32, float)
#endif
#ifdef RBD_STUB_GRAD_F64
RBD_STUBS_GRAD(f64, double)
#endif
#ifdef RBD_STUB_GRADN_F32
RBD_STUBS_GRADN(f32, float)
#endif
#ifdef RBD_STUB_GRADN_F64
RBD_STUBS_GRADN(f64, double)
#endif
#ifdef RBD_STUB_MINV_F32
RBD_STUBS_MINV(f32, float)
#endif
#ifdef RBD_STUB_MINV_F64
RBD_STUBS_MINV(f64, double)
#endif
#ifdef RBD_STUB_FD_F32
RBD_STUBS_FD(f32, float)
#endif
#ifdef RBD_STUB_FD_F64
RBD_STUBS_FD(f64, double)
#endif
#ifdef RBD_STUB_PASS_F32
RBD_STUBS_PASS(f32, float)
#endif
#ifdef RBD_STUB_PASS_F64
RBD_STUBS_PASS(f64, double)
#endif
#endif  // RBD_TU_STUBS

}  // extern "C"
