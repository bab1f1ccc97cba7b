// rbd_fb_passes.h -- the reference's individual passes for FLOATING-BASE robots (README.md:19: the
// accelerator-testing surface), with the reference's layouts and in-place behaviour:
//   rnea_grad_fpass_dq / _dqd  (/root/reference/RBDReference.py:1127-1187, :1189-1255)  -> dv, da, df [B, 6, nv, NB]
//   rnea_grad_bpass_dq / _dqd  (:1257-1297, :1299-1343)   -> dc [B, nv, nv], df accumulated child -> parent IN PLACE
//   minv_bpass                 (:630-735, floating-base branches :652-691)  -> Minv, F [B, nv, 6, nv], U [B, nv, 6], Dinv
//   minv_fpass                 (:737-783, :779)           Minv rows updated in place, F rebuilt (indexed by BODY id)
//   forward_dynamics_grad      (:1376-1384)               composition + one batched product
// (rnea_fpass / rnea_bpass are modes of rnea_fbw_kernel, rbd_fb_world.h.)  These kernels exist to be compared pass
// by pass with the reference: they move its O(n^2) six-vectors through HBM by definition and store element by
// element; one lane per (configuration, derivative / matrix column), bodies unrolled, columns a run-time loop.
#pragma once
#include "rbd_negmm.h"
#include "rbd_fb.h"

namespace rbdk {

constexpr int FBP_L = 8;                 // lanes per configuration (they share the columns)
constexpr int fbp_child_root(int j) {     // the child of the base that body j >= 1 hangs under
  while (PARENT[j] != 0) j = PARENT[j];
  return j;
}
constexpr int FBP_C = 64 / FBP_L;        // configurations per block

// ---- rnea_grad forward passes: column `col` of (dv, da, df) for every body -----------------------------------
// v [B, 6, NB] and (dq only) a [B, 6, NB] are INPUTS, as in the reference's signature.
template <class T, bool ISQD>
__global__ __launch_bounds__(64, 1) void fb_grad_fpass_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                              const T* __restrict__ v_in, const T* __restrict__ a_in,
                                                              T grav, long long B, T* __restrict__ dv_out,
                                                              T* __restrict__ da_out, T* __restrict__ df_out) {
  const int lane = threadIdx.x;
  const int sub = lane % FBP_L, slot = lane / FBP_L;
  const long long b = (long long)blockIdx.x * FBP_C + slot;
  if (b >= B) return;
  const T* qb = q + b * NV; const T* qdb = qd + b * NV;
  const T* vb = v_in + b * (6 * N);
  const T* ab = ISQD ? nullptr : a_in + b * (6 * N);
  JTrig<T> tr[N];
  T qdv[N];
  sfor<1, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j] = make_trig<j>(qb[j + 5]); qdv[j] = qdb[j + 5]; });
  // v [6][NB] of the configuration is an INPUT: read where it is used (every column re-reads it from L1 / L2) instead
  // of living in 6 NB registers next to the column's dv and da (fp64: 512 VGPRs and scratch otherwise)
  auto vload = [&](int j, T (&x)[6]) { sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; x[r] = vb[r * N + j]; }); };
  T qd0[6];
  sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; qd0[r] = qdb[r]; });
  T ag0[6];                                             // X_0 a_grav
  {
    T E[3][3];
    fb_base_E(qb[3], qb[4], qb[5], E);
    ag0[0] = ag0[1] = ag0[2] = T(0);
    ag0[3] = -(grav * E[0][2]); ag0[4] = -(grav * E[1][2]); ag0[5] = -(grav * E[2][2]);
  }
  const long long ob = b * (6LL * NV * N);              // [b][r][c][i] at ob + (r * NV + c) * N + i
#pragma clang loop unroll(disable)
  for (int col = sub; col < NV; col += FBP_L) {
    sfor<1, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j].s = launder(tr[j].s); tr[j].c = launder(tr[j].c); });
    T dv[N][6], da[N][6];
    auto emit = [&](auto I, const T (&d)[6]) {
      constexpr int i = decltype(I)::value;
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        dv_out[ob + (r * NV + col) * N + i] = dv[i][r];
        da_out[ob + (r * NV + col) * N + i] = da[i][r];
        df_out[ob + (r * NV + col) * N + i] = d[r];
      });
    };
    {
      // the base.  dq: dv = 0, da = crm(X_0 a_grav) e_col (:1175).  dqd: dv = e_col (:1231),
      // da = crm(dv) qd[0:6] (:1236-1238) + crm(v_0) e_col (:1243)
      T e[6];
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; e[r] = col == r ? T(1) : T(0); });
      T t0[6], t1[6], t2[6], v0[6];
      vload(0, v0);
      crm_mul(ag0, e, t0);
      crm_mul(e, qd0, t1);
      crm_mul(v0, e, t2);
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        dv[0][r] = ISQD ? e[r] : T(0);
        da[0][r] = ISQD ? t1[r] + t2[r] : t0[r];
      });
      T Iv[6], Idv[6], d[6];
      cmatvec<MatI, 0>(v0, Iv);
      cmatvec<MatI, 0>(dv[0], Idv);
      cmatvec<MatI, 0>(da[0], d);
      fxv<true>(dv[0], Iv, d);
      fxv<true>(v0, Idv, d);
      emit(std::integral_constant<int, 0>{}, d);
    }
    sfor<1, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      const bool own = col == i + 5;
      xform<i>(tr[i], dv[p], dv[i]);          // (:1158 / :1230)
      xform<i>(tr[i], da[p], da[i]);          // (:1163 / :1234)
      T xv[6], sdq[6], sS[6], e1[6], e2[6], vp[6], vi[6];
      vload(p, vp); vload(i, vi);
      xform<i>(tr[i], vp, xv);
      mxS<i>(xv, T(1), sdq);                  // crm(X v_p) S   (:1159)
      sfor<0, 6>([&](auto R) { sS[decltype(R)::value] = T(0); });
      add_S<i>(T(1), sS);                     // S              (:1231)
      if constexpr (!ISQD) {
        T ap[6], xa[6];
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; ap[r] = ab[r * N + p]; });
        xform<i>(tr[i], ap, xa);
        mxS<i>(xa, T(1), e1);                 // crm(X a_p) S   (:1173)
      } else {
        sfor<0, 6>([&](auto R) { e1[decltype(R)::value] = T(0); });
      }
      mxS<i>(vi, T(1), e2);                   // crm(v_i) S     (:1243)
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; dv[i][r] += sel(own, ISQD ? sS[r] : sdq[r], T(0)); });
      add_mxS<i>(dv[i], qdv[i], da[i]);       // (:1170 / :1240)
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; da[i][r] += sel(own, ISQD ? e2[r] : e1[r], T(0)); });
      T Iv[6], Idv[6], d[6];
      cmatvec<MatI, i>(vi, Iv);
      cmatvec<MatI, i>(dv[i], Idv);
      cmatvec<MatI, i>(da[i], d);
      fxv<true>(dv[i], Iv, d);                // (:1179-1185 / :1247-1252)
      fxv<true>(vi, Idv, d);
      emit(I, d);
    });
  }
}

// ---- rnea_grad backward passes: column `col` of dc, df accumulated in place --------------------------------------
template <class T, bool ISQD>
__global__ __launch_bounds__(64, 1) void fb_grad_bpass_kernel(const T* __restrict__ q, const T* __restrict__ f_in,
                                                              T* __restrict__ df_io, int use_damping, long long B,
                                                              T* __restrict__ dc_out) {
  const int lane = threadIdx.x;
  const int sub = lane % FBP_L, slot = lane / FBP_L;
  const long long b = (long long)blockIdx.x * FBP_C + slot;
  if (b >= B) return;
  const T* qb = q + b * NV;
  const T* fb = ISQD ? nullptr : f_in + b * (6 * N);
  JTrig<T> tr[N];
  sfor<1, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j] = make_trig<j>(qb[j + 5]); });
  const long long ob = b * (6LL * NV * N);
  T* dcb = dc_out + b * (NV * NV);
#pragma clang loop unroll(disable)
  for (int col = sub; col < NV; col += FBP_L) {
    sfor<1, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j].s = launder(tr[j].s); tr[j].c = launder(tr[j].c); });
    sfor_down<1, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      T d[6], x[6], y[6];
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; d[r] = df_io[ob + (r * NV + col) * N + i]; });
      T o = S_dot<i>(d);                                                  // (:1284 / :1325)
      dcb[(i + 5) * NV + col] = o;
      if constexpr (!ISQD) {
        T fi[6], w[6];
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; fi[r] = fb[r * N + i]; });
        mxS<i>(fi, T(-1), w);                                             // fxS(S, f) = -crm(f) S  (:1292-1294)
        const bool ex = col == i + 5;
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; x[r] = d[r] + sel(ex, w[r], T(0)); });
      } else {
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; x[r] = d[r]; });
      }
      xform_T<i>(tr[i], x, y);                                            // (:1291 / :1331)
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; df_io[ob + (r * NV + col) * N + p] += y[r]; });
    });
    sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; dcb[r * NV + col] = df_io[ob + (r * NV + col) * N + 0]; });   // S = eye(6) (:1282)
    if constexpr (ISQD) {
      if (use_damping) {
        // (:1336-1341) literally: the base adds its damping to a 5 x 5 block, body ind >= 1 to entry (ind, ind)
        if constexpr (DAMPING[0] != 0.0) {
          if (col < 5) sfor<0, 5>([&](auto R) { constexpr int r = decltype(R)::value; dcb[r * NV + col] += T(DAMPING[0]); });
        }
        sfor<1, N>([&](auto I) {
          constexpr int i = decltype(I)::value;
          if constexpr (DAMPING[i] != 0.0) { if (col == i) dcb[i * NV + i] += T(DAMPING[i]); }
        });
      }
    }
  }
}

// ---- minv_bpass: (Minv, F, U, Dinv); Minv and F must be zero on entry (the launch code clears them) ------------------
template <class T>
__global__ __launch_bounds__(64, 1) void fb_minv_bpass_kernel(const T* __restrict__ q, long long B, T* __restrict__ Minv,
                                                              T* __restrict__ F, T* __restrict__ U_out, T* __restrict__ D_out) {
  // {U[6], 1/D} of every body, per configuration of the block: the column sweeps of the configuration's four lanes read
  // them from here (same address: an LDS broadcast) instead of each lane holding 7 NB registers across the column loop
  __shared__ T recs[64 / FB_MINV_L][N][8];
  __shared__ T fbs[64 / FB_MINV_L][36];
  const int sub = threadIdx.x % FB_MINV_L, slot = threadIdx.x / FB_MINV_L;
  const long long b0 = (long long)blockIdx.x * (64 / FB_MINV_L) + slot;
  const bool live = b0 < B;
  const long long b = live ? b0 : B - 1;                // lanes beyond the batch repeat the last configuration and store nothing
  const T* qb = q + b * NV;
  T* Mb = Minv + b * (NV * NV);
  T* Fb = F + b * (6LL * NV * NV);                      // [mi][r][c] at (mi * 6 + r) * NV + c
  JTrig<T> tr[N];
  sfor<1, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j] = make_trig<j>(qb[j + 5]); });
  {
    // articulated inertias, one subtree of the base after the other (only that subtree's IA is live)
    T acc0[6][6];
    sfor<0, 6>([&](auto R) { sfor<0, 6>([&](auto C) { constexpr int r = decltype(R)::value, c = decltype(C)::value; acc0[r][c] = T(IM[0][r * 6 + c]); }); });
    sfor<1, N>([&](auto RT) {
      constexpr int rt = decltype(RT)::value;
      if constexpr (PARENT[rt] == 0) {
        T IA[N][6][6];
        sfor<1, N>([&](auto J) {
          constexpr int j = decltype(J)::value;
          if constexpr (fbp_child_root(j) == rt)
            sfor<0, 6>([&](auto R) { sfor<0, 6>([&](auto C) { constexpr int r = decltype(R)::value, c = decltype(C)::value; IA[j][r][c] = T(IM[j][r * 6 + c]); }); });
        });
        sfor_down<1, N>([&](auto I) {
          constexpr int i = decltype(I)::value;
          if constexpr (fbp_child_root(i) == rt) {
            constexpr int p = PARENT[i];
            constexpr int si = fb_s_index(i);
            T U[6];
            sfor<0, 6>([&](auto R) { U[decltype(R)::value] = IA[i][decltype(R)::value][si]; });
            const T D = U[si];
            const T Dinv = T(1) / D;
            if (sub == 0) {
              sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; recs[slot][i][r] = U[r]; });
              recs[slot][i][6] = Dinv;
              if (live) {
                sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; U_out[(b * NV + i + 5) * 6 + r] = U[r]; });   // :697
                D_out[b * NV + i + 5] = D;                                                                                 // :698 (holds D)
              }
            }
            T A[6][6];
            sfor<0, 6>([&](auto C) {
              constexpr int c = decltype(C)::value;
              T col[6], y[6];
              const T uc = U[c] * Dinv;
              sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; col[r] = fma_(-U[r], uc, IA[i][r][c]); });
              xform_T<i>(tr[i], col, y);
              sfor<0, 6>([&](auto R) { A[decltype(R)::value][c] = y[decltype(R)::value]; });
            });
            sfor<0, 6>([&](auto R) {
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
    });
    T fb6[6][6];
    fb_inv6(acc0, fb6);
    if (sub == 0) {
      sfor<0, 6>([&](auto R) { sfor<0, 6>([&](auto C) { constexpr int r = decltype(R)::value, c = decltype(C)::value; fbs[slot][r * 6 + c] = fb6[r][c]; }); });
      if (live) {
        // U[0:6] = IA_0 S = IA_0 (:680); Dinv[0:6] stays 0 (the reference never writes it); the base block of Minv (:685)
        sfor<0, 6>([&](auto R) {
          sfor<0, 6>([&](auto C) {
            constexpr int r = decltype(R)::value, c = decltype(C)::value;
            U_out[(b * NV + r) * 6 + c] = acc0[r][c];
            Mb[r * NV + c] = fb6[r][c];
          });
          D_out[b * NV + decltype(R)::value] = T(0);
        });
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma clang loop unroll(disable)
  for (int jb = 1 + sub; jb < N; jb += FB_MINV_L) {
    const int j = jb + 5;
    T Fj[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    sfor_down<1, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr unsigned long long mask = fb_subtree_mask(i);
      const bool insub = ((mask >> jb) & 1ull) != 0;
      T U[6];
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; U[r] = recs[slot][i][r]; });
      const T Dinv = recs[slot][i][6];
      T m = sel(jb == i, Dinv, -(Dinv * S_dot<i>(Fj)));                // :700, :702-708
      T t[6], y[6];
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; t[r] = fma_(U[r], m, Fj[r]); });   // :721-723
      xform_T<i>(tr[i], t, y);                                                                              // :724-726
      if (insub) {
        if (live) {
          Mb[(i + 5) * NV + j] = m;
          sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Fb[((i + 5) * 6 + r) * NV + j] = t[r]; });
        }
        sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Fj[r] = y[r]; });
      }
    });
    if (live) {
      // what reached the base sits in the base's F slot, matrix index 5 (:724 with parent_ind + 5); its rows :686-691
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Fb[(5 * 6 + r) * NV + j] = Fj[r]; });
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        T o = T(0);
        sfor<0, 6>([&](auto K) { constexpr int k = decltype(K)::value; o = fma_(-fbs[slot][r * 6 + k], Fj[k], o); });
        Mb[r * NV + j] = o;
      });
    }
  }
}

// ---- minv_fpass: one lane per (configuration, column); Minv rows updated in place, F rebuilt by BODY id ------------------
template <class T>
__global__ __launch_bounds__(64, 1) void fb_minv_fpass_kernel(const T* __restrict__ q, long long B, T* __restrict__ Minv,
                                                              T* __restrict__ F, const T* __restrict__ U_in, const T* __restrict__ D_in) {
  const int lane = threadIdx.x;
  const int sub = lane % FBP_L, slot = lane / FBP_L;
  const long long b = (long long)blockIdx.x * FBP_C + slot;
  if (b >= B) return;
  const T* qb = q + b * NV;
  T* Mb = Minv + b * (NV * NV);
  T* Fb = F + b * (6LL * NV * NV);
  JTrig<T> tr[N];
  T U[N][6], Dinv[N];
  sfor<1, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    tr[j] = make_trig<j>(qb[j + 5]);
    sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; U[j][r] = U_in[(b * NV + j + 5) * 6 + r]; });
    Dinv[j] = T(1) / D_in[b * NV + j + 5];
  });
#pragma clang loop unroll(disable)
  for (int c = sub; c < NV; c += FBP_L) {
    sfor<1, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j].s = launder(tr[j].s); tr[j].c = launder(tr[j].c); });
    T Ff[N][6];
    sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Ff[0][r] = Mb[r * NV + c]; Fb[(0 * 6 + r) * NV + c] = Ff[0][r]; });   // :779
    sfor<1, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int p = PARENT[i];
      constexpr int si = fb_s_index(i);
      xform<i>(tr[i], Ff[p], Ff[i]);
      const T m = fma_(-Dinv[i], dot6(U[i], Ff[i]), Mb[(i + 5) * NV + c]);   // :771-773
      Mb[(i + 5) * NV + c] = m;
      Ff[i][si] += m;                                                          // :774-776
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; Fb[(i * 6 + r) * NV + c] = Ff[i][r]; });
    });
  }
}

// ---- out [B, NV, 2 NV] = -Minv dc_du   (:1381-1384): neg_mm_kernel<T, NV> of rbd_negmm.h (flat 16-byte copies through LDS) ----
}  // namespace rbdk
