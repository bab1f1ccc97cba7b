// rbd_fb_minv.h -- FLOATING-BASE minv with one WAVE per subtree of the base (blocks of 64 configurations x FBW_W waves).
//
// RBDReference.minv for a floating base (/root/reference/RBDReference.py:630-806 with the branches :652-691, :761-779):
// articulated inertias leaf -> base, the base as ONE 6 x 6 block inv(IA_0), per joint column the F vector climbs
// its root path, the base block gives rows 0..5, the forward sweep every other row.  minv_fb_kernel (rbd_fb.h) gives
// a configuration four LANES, each of which runs the whole articulated-inertia recursion; here the subtrees hanging
// off the base -- independent of each other until the base is built (:728-733 only ever adds into the parent) --
// each get a WAVE, one configuration per lane:
//   phase A   wave w: sin / cos, U = IA S, 1 / D and the articulated inertia X^T Ia X handed to the base, for the
//             bodies of ITS subtrees only; {U, 1/D, sin, cos} go to lane-private LDS records (every wave's forward
//             sweep needs every body's), the contribution to IA_0 to a per-wave slot          -- block barrier --
//   phase B   every wave: IA_0 = I_0 + sum of the contributions, inv(IA_0) (Gauss-Jordan, redundantly: cheaper than
//             a second barrier); then the joint columns of ITS bodies: backward sweep along the column's root path
//             (own records), rows 0..5 from the base block, forward sweep over ALL bodies (records from LDS)
//   output    the matrices of 16 configurations at a time through a [16][nv * nv] LDS image (each wave writes its
//             columns and their mirror entries, wave 0 the base block), flat 16-byte copies of whole lines.
// Same arithmetic per column as minv_fb_kernel, so the two agree to rounding; 1.8x fewer instructions per
// configuration (the recursion is done once, not four times).
// forward dynamics (:1371-1374): with u, c the kernel also emits qdd = Minv (u - c) -- every wave multiplies the columns it
// holds in registers (and their mirror entries) with tau, the partial sums meet in the lane-private LDS slots the records
// leave free -- and with Minv == nullptr it skips the image and the 4 nv^2-byte matrix altogether (round 3 ran a separate
// kernel that re-read the dense Minv for this product: 21 us of a 58 us forward_dynamics at B = 65 536).
#pragma once
#include "rbd_fb.h"
#include "rbd_fb_world.h"      // FBW_W, fbw_wave_of (the wave <-> subtree assignment of the gradient kernel)

namespace rbdk {

constexpr int FBM_REC = 9;                                  // {U[6], 1/D, sin, cos} per body
constexpr int FBM_IA0 = 21;                                 // symmetric 6 x 6 contribution to IA_0, upper triangle
constexpr int FBM_Q = 16;                                   // configurations per output image
constexpr int FBM_IMG = FBM_Q * NV * NV;                    // scalars of the image
constexpr int FBM_PRIV = FBM_REC * N + FBM_IA0 * FBW_W;     // lane-private scalars (slot * 64 + lane)
// forward dynamics with the bias force computed here: f_0 contributions [W][6] and c of the joints [NV], lane-private, in the
// IMAGE's space (the image is first written after the column phase; a block's LDS stays at two blocks per CU)
constexpr int FBM_BIAS_SLOTS = 6 * FBW_W + NV;
static_assert(64 * FBM_BIAS_SLOTS <= FBM_IMG, "bias-force slots live in the output image's space");
template <class T>
constexpr size_t minv_fbm_lds_bytes() { return sizeof(T) * ((size_t)FBM_IMG + (size_t)64 * FBM_PRIV); }
template <class T>
constexpr bool minv_fbm_ok() { return N >= 2 && minv_fbm_lds_bytes<T>() <= 160 * 1024 && (FBM_Q * NV * NV) % (16 / sizeof(T)) == 0; }
constexpr int fbm_col_ord(int jb) {                         // ordinal of body jb among the bodies of its wave
  int k = 0;
  for (int x = 1; x < jb; ++x) k += fbw_wave_of(x) == fbw_wave_of(jb) ? 1 : 0;
  return k;
}
constexpr int fbm_max_cols() {
  int m = 0;
  for (int jb = 1; jb < N; ++jb) m = fbm_col_ord(jb) + 1 > m ? fbm_col_ord(jb) + 1 : m;
  return m;
}
constexpr int fbm_sym_index(int r, int c) {                 // r <= c -> 0..20
  return r * 6 - r * (r - 1) / 2 + (c - r);
}

template <class T>
__global__ __launch_bounds__(64 * FBW_W, 1) void minv_fbm_kernel(const T* __restrict__ q, long long B, int dense, T* __restrict__ Minv,
                                                                  const T* __restrict__ u_in = nullptr, const T* __restrict__ c_in = nullptr,
                                                                  T* __restrict__ qdd_out = nullptr, const T* __restrict__ qd_in = nullptr, T grav = T(0)) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* img = reinterpret_cast<T*>(smem_raw);                                   // [FBM_Q][NV * NV]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  T* priv = reinterpret_cast<T*>(smem_raw) + FBM_IMG + lane;                 // priv[slot * 64]
  const long long cfg0 = (long long)blockIdx.x * 64;
  const long long rem = B - cfg0;
  const int nvalid = rem < 64 ? (int)rem : 64;
  const long long b = cfg0 + (lane < nvalid ? lane : nvalid - 1);
  const T* qb = q + b * NV;
  auto rec = [&](int i, int k) -> T& { return priv[(FBM_REC * i + k) * 64]; };
  auto ia0 = [&](int w, int k) -> T& { return priv[(FBM_REC * N + FBM_IA0 * w + k) * 64]; };

  // forward dynamics with qd instead of c: the bias force c = rnea(q, qd, qdd = None) (:559-621 with the floating-base lines
  // :585, :591, :612) is computed here, subtree by subtree next to the articulated inertias -- every wave the base's v_0, a_0
  // (cheap), its own bodies' v, a, f and joint entries c_j, and what its subtrees hand to f_0; the pieces meet in lane-private
  // LDS slots at the barrier phase B waits at anyway.  (The c-only rnea launch this replaces: 10 of forward_dynamics' 28 us.)
  const bool own_bias = qdd_out != nullptr && qd_in != nullptr;  // (uniform over the launch)
  auto f0s = [&](int w, int r) -> T& { return img[(6 * w + r) * 64 + lane]; };
  auto cjs = [&](int j) -> T& { return img[(6 * FBW_W + j) * 64 + lane]; };
  const T* qdb = own_bias ? qd_in + b * NV : q;                  // (never read without own_bias)
  T v0[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, a0[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, f0acc[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  if (own_bias) {
    T E[3][3];
    fb_base_E(qb[3], qb[4], qb[5], E);
    sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; v0[r] = qdb[r]; });
    a0[3] = -(grav * E[0][2]); a0[4] = -(grav * E[1][2]); a0[5] = -(grav * E[2][2]);   // X_0 a_grav (rbd_fb.h: rnea_fb_kernel)
    if (wave == 0) {                                             // the base's own force, once
      T Iv[6], Ia[6];
      cmatvec<MatI, 0>(v0, Iv);
      cmatvec<MatI, 0>(a0, Ia);
      sfor<0, 6>([&](auto R) { f0acc[decltype(R)::value] = Ia[decltype(R)::value]; });
      fxv<true>(v0, Iv, f0acc);
    }
  }
  // ---- phase A: this wave's subtrees ---------------------------------------------------------------------------
  {
    T acc0[6][6];                                             // what this wave's subtrees hand to the base
    sfor<0, 6>([&](auto R) { sfor<0, 6>([&](auto C) { acc0[decltype(R)::value][decltype(C)::value] = T(0); }); });
    sfor<1, N>([&](auto RT) {
      constexpr int rt = decltype(RT)::value;
      if constexpr (PARENT[rt] == 0) {
        if (wave == fbw_wave_of_child(rt)) {
          // the bodies of the subtree under rt, leaf -> rt; IA only for them
          T IA[N][6][6];
          JTrig<T> tr[N];
          sfor<1, N>([&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (fbw_child_root(j) == rt) {
              tr[j] = make_trig<j>(qb[j + 5]);
              sfor<0, 6>([&](auto R) { sfor<0, 6>([&](auto C) { constexpr int r = decltype(R)::value, c = decltype(C)::value; IA[j][r][c] = T(IM[j][r * 6 + c]); }); });
            }
          });
          if (own_bias) {
            T vb[N][6], ab[N][6], fbd[N][6];
            sfor<1, N>([&](auto J) {
              constexpr int j = decltype(J)::value;
              if constexpr (fbw_child_root(j) == rt) {
                constexpr int p = PARENT[j];
                T xv[6], xa[6];
                if constexpr (p == 0) rnea_fwd_body<j, false>(tr[j], qdb[j + 5], T(0), grav, v0, a0, xv, xa, vb[j], ab[j], fbd[j]);
                else rnea_fwd_body<j, false>(tr[j], qdb[j + 5], T(0), grav, vb[p], ab[p], xv, xa, vb[j], ab[j], fbd[j]);
              }
            });
            sfor_down<1, N>([&](auto J) {
              constexpr int j = decltype(J)::value;
              if constexpr (fbw_child_root(j) == rt) {
                constexpr int p = PARENT[j];
                cjs(j + 5) = S_dot<j>(fbd[j]);                                          // :612
                T t[6];
                xform_T<j>(tr[j], fbd[j], t);                                            // :618-619
                sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; if constexpr (p == 0) f0acc[r] += t[r]; else fbd[p][r] += t[r]; });
              }
            });
          }
          sfor_down<1, N>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (fbw_child_root(i) == rt) {
              constexpr int p = PARENT[i];
              constexpr int si = fb_s_index(i);
              T U[6];
              sfor<0, 6>([&](auto R) { U[decltype(R)::value] = IA[i][decltype(R)::value][si]; });   // :697
              const T Dinv = T(1) / U[si];                                                          // :698, :700
              sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; rec(i, r) = U[r]; });
              rec(i, 6) = Dinv; rec(i, 7) = tr[i].s; rec(i, 8) = tr[i].c;
              T A[6][6];   // A = X^T Ia, Ia = IA - U U^T / D   (:728-731)
              sfor<0, 6>([&](auto C) {
                constexpr int c = decltype(C)::value;
                T col[6], y[6];
                const T uc = U[c] * Dinv;
                sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; col[r] = fma_(-U[r], uc, IA[i][r][c]); });
                xform_T<i>(tr[i], col, y);
                sfor<0, 6>([&](auto R) { A[decltype(R)::value][c] = y[decltype(R)::value]; });
              });
              sfor<0, 6>([&](auto R) {   // (A X)[r][:] = X^T A[r][:]^T   (:732-733)
                constexpr int r = decltype(R)::value;
                T y[6];
                xform_T<i>(tr[i], A[r], y);
                sfor<0, 6>([&](auto C) {
                  constexpr int c = decltype(C)::value;
                  if constexpr (p == 0) acc0[r][c] += y[c]; else IA[p][r][c] += y[c];
                });
              });
            }
          });
        }
      }
    });
    sfor<0, 6>([&](auto R) { sfor<0, 6>([&](auto C) { constexpr int r = decltype(R)::value, c = decltype(C)::value; if constexpr (c >= r) ia0(wave, fbm_sym_index(r, c)) = acc0[r][c]; }); });
    if (own_bias) { sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; f0s(wave, r) = f0acc[r]; }); }
  }
  __syncthreads();
  // ---- phase B: the base block (every wave, redundantly) -----------------------------------------------------------
  T fb6[6][6];
  {
    T IA0[6][6];
    sfor<0, 6>([&](auto R) {
      sfor<0, 6>([&](auto C) {
        constexpr int r = decltype(R)::value, c = decltype(C)::value;
        if constexpr (c >= r) {
          T x = T(IM[0][r * 6 + c]);
          sfor<0, FBW_W>([&](auto W_) { x += ia0(decltype(W_)::value, fbm_sym_index(r, c)); });
          IA0[r][c] = x; IA0[c][r] = x;
        }
      });
    });
    fb_inv6(IA0, fb6);                                            // fb_Dinv = inv(S^T IA_0 S), S = eye(6)  (:681-683)
  }
  // ---- the joint columns of this wave's bodies: kept in registers until their quarter of the block is flushed ------
  // [k-th column of this wave][row]: indexed by the column's ORDINAL within its wave -- indexed by body, every wave
  // would hold registers for every other wave's columns too (the branches below are per wave at run time)
  T colv[fbm_max_cols()][NV];
  const bool fd = qdd_out != nullptr;                            // (uniform over the launch)
  T tau[NV], qacc[NV];
  sfor<0, NV>([&](auto R) {
    constexpr int r = decltype(R)::value;
    T cr = T(0);
    if (own_bias) {
      if constexpr (r < 6) { sfor<0, FBW_W>([&](auto W_) { cr += f0s(decltype(W_)::value, r); }); }      // c[0:6] = f_0 (S = eye(6), :612)
      else cr = cjs(r);
    } else if (fd) {
      cr = c_in[b * NV + r];
    }
    tau[r] = fd ? u_in[b * NV + r] - cr : T(0);
    qacc[r] = T(0);
  });
  if (fd && wave == 0) {                                         // the base block times tau[0:6]
    sfor<0, 6>([&](auto R) { sfor<0, 6>([&](auto C) { constexpr int r = decltype(R)::value, c = decltype(C)::value; qacc[r] = fma_(fb6[r][c], tau[c], qacc[r]); }); });
  }
  sfor<1, N>([&](auto JB) {
    constexpr int jb = decltype(JB)::value;
    if (wave == fbw_wave_of(jb)) {
      T mcol[N];
      T Fj[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
      // backward sweep (:665-726): the column's F vector climbs the root path of body jb
      sfor_down<1, N>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (is_anc_or_self(i, jb)) {
          JTrig<T> g; g.s = rec(i, 7); g.c = rec(i, 8);
          T U[6];
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; U[r] = rec(i, r); });
          const T Dinv = rec(i, 6);
          const T m = (i == jb) ? Dinv : -(Dinv * S_dot<i>(Fj));          // :700, :702-708
          mcol[i] = m;
          T t[6], y[6];
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; t[r] = fma_(U[r], m, Fj[r]); });   // :721-723
          xform_T<i>(g, t, y);                                                                             // :724-726
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Fj[r] = y[r]; });
        } else {
          mcol[i] = T(0);
        }
      });
      // the base rows of the column: Minv[0:6, j] = -inv(IA_0) F_0[:, j]   (:686-691)
      T Ff[N][6];
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        T o = T(0);
        sfor<0, 6>([&](auto K) { constexpr int k = decltype(K)::value; o = fma_(-fb6[r][k], Fj[k], o); });
        Ff[0][r] = o;                                                     // F_0[:, j] = S Minv[0:6, j], S = eye(6)  (:779)
      });
      // forward sweep (:760-776) over every body
      sfor<1, N>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int p = PARENT[i];
        constexpr int si = fb_s_index(i);
        JTrig<T> g; g.s = rec(i, 7); g.c = rec(i, 8);
        T U[6];
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; U[r] = rec(i, r); });
        xform<i>(g, Ff[p], Ff[i]);
        const T m = fma_(-rec(i, 6), dot6(U, Ff[i]), mcol[i]);            // :771-773
        mcol[i] = m;
        Ff[i][si] += m;                                                   // :774-776
      });
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; colv[fbm_col_ord(jb)][r] = Ff[0][r]; });
      sfor<1, N>([&](auto I) { constexpr int i = decltype(I)::value; colv[fbm_col_ord(jb)][i + 5] = mcol[i]; });
      if (fd) {     // column j = jb + 5 of the symmetric matrix: rows r <= j, and their mirror entries in row j
        constexpr int j = jb + 5;
        sfor<0, NV>([&](auto R) {
          constexpr int r = decltype(R)::value;
          if constexpr (r <= j) {
            qacc[r] = fma_(colv[fbm_col_ord(jb)][r], tau[j], qacc[r]);
            if constexpr (r < j) qacc[j] = fma_(colv[fbm_col_ord(jb)][r], tau[r], qacc[j]);
          }
        });
      }
    }
  });
  if (fd) {
    static_assert(FBW_W * NV <= FBM_PRIV, "qdd partial sums: lane-private slots");
    __syncthreads();                                             // every wave has read the last record: the slots are free
    sfor<0, NV>([&](auto R) { constexpr int r = decltype(R)::value; priv[(wave * NV + r) * 64] = qacc[r]; });
    __syncthreads();
    if (wave == 0 && lane < nvalid) {
      sfor<0, NV>([&](auto R) {
        constexpr int r = decltype(R)::value;
        T o = qacc[r];
        sfor<1, FBW_W>([&](auto W_) { o += priv[(decltype(W_)::value * NV + r) * 64]; });
        qdd_out[b * NV + r] = o;
      });
    }
  }
  if (Minv == nullptr) return;                                   // forward_dynamics: the matrix itself is not wanted
  // ---- output: 16 configurations at a time ---------------------------------------------------------------------------
  constexpr int VE = 16 / sizeof(T);
  typedef T V __attribute__((ext_vector_type(VE)));
  for (int qt = 0; qt < 64 / FBM_Q; ++qt) {
    if (qt * FBM_Q >= nvalid) break;                             // (uniform over the block)
    __syncthreads();                                             // the previous image has been read
    if ((lane >> 4) == qt) {
      T* Mb = img + (lane & 15) * (NV * NV);
      if (wave == 0) {
        // the base block (:685): symmetric by mirroring its upper part
        sfor<0, 6>([&](auto R) {
          sfor<0, 6>([&](auto C) {
            constexpr int r = decltype(R)::value, c = decltype(C)::value;
            if constexpr (c >= r) {
              Mb[r * NV + c] = fb6[r][c];
              if constexpr (c > r) Mb[c * NV + r] = dense ? fb6[r][c] : T(0);
            }
          });
        });
      }
      sfor<1, N>([&](auto JB) {
        constexpr int jb = decltype(JB)::value;
        if (wave == fbw_wave_of(jb)) {
          constexpr int j = jb + 5;
          // rows r <= j of column j, mirrored below the diagonal
          sfor<0, NV>([&](auto R) {
            constexpr int r = decltype(R)::value;
            if constexpr (r <= j) {
              Mb[r * NV + j] = colv[fbm_col_ord(jb)][r];
              if constexpr (r < j) Mb[j * NV + r] = dense ? colv[fbm_col_ord(jb)][r] : T(0);
            }
          });
        }
      });
    }
    __syncthreads();
    const int nq = nvalid - qt * FBM_Q < FBM_Q ? nvalid - qt * FBM_Q : FBM_Q;
    T* gdst = Minv + (cfg0 + qt * FBM_Q) * (NV * NV);
    if (nq == FBM_Q) {
      constexpr int NVEC = FBM_Q * NV * NV / VE;
      for (int g = threadIdx.x; g < NVEC; g += 64 * FBW_W) reinterpret_cast<V*>(gdst)[g] = reinterpret_cast<const V*>(img)[g];
    } else {
      for (int g = threadIdx.x; g < nq * NV * NV; g += 64 * FBW_W) gdst[g] = img[g];
    }
  }
}

}  // namespace rbdk
