// rbd_grad_cols.h -- rnea + rnea_grad for SMALL batches: one lane per (configuration, derivative column).
//
// The batch-parallel gradient kernels give a configuration one lane (rbd_idsva.h) or two
// (rnea_grad_kernel): at B = 4096 that is 64 waves on a chip with 1 024 SIMDs, each wave running the
// whole ~4.8 k-instruction evaluation serially (10-12 us, all of it latency).  Here the 2n derivative
// columns of a configuration -- independent recursions in the reference (RBDReference.py:1164-1185,
// :1235-1252, :1291, :1331) -- get a lane each: 16 lanes per 7-DoF configuration, 4 configurations per
// wave, 1 024 waves at B = 4096, and a lane's instruction stream is the reference's own algorithm for
// ONE column:
//     rnea_fpass / rnea_bpass           (:559-621)   -> v, a, accumulated f, c   (same in every lane)
//     rnea_grad_fpass_dq | _dqd, column c  (:1127-1255)  -> dv, da, df of column c for every body
//     rnea_grad_bpass_dq | _dqd, column c  (:1257-1343)  -> dc[:, c]
// A column is seeded at body c and propagated with X to every later body, so bodies outside subtree(c)
// receive X * 0 = 0 exactly as the reference's dense column updates do: any tree, revolute and
// prismatic joints, the literal fxS term (:1292) -- the only condition is that a lane's state fits its registers.
// One launch returns everything BASELINE configs[1] asks for: c, v, a, f (RBDReference.rnea) and dc_du.
// Used when the batch is too small to fill the chip with the batch-parallel kernels (the launch
// code decides from B), or on request (rbd_set_option).
#pragma once
#include "rbd_spatial.h"

namespace rbdk {

constexpr int gc_lanes() { return 2 * N <= 8 ? 8 : 2 * N <= 16 ? 16 : 2 * N <= 32 ? 32 : 64; }
// a lane keeps the accumulated f and the df of every body for its backward pass (12 n scalars) beside
// the running v, a, dv, da: built for the robots where that fits the 512-register file of a lone wave
// (a dense 14-body tree already spills 280 bytes in fp32)
template <class T>
constexpr bool grad_cols_ok() { return N <= (sizeof(T) == 4 ? 12 : 6); }
constexpr int GC_L = gc_lanes();            // lanes per configuration
constexpr int GC_CPW = 64 / GC_L;           // configurations per wave
// LDS: per configuration slot the image of (v, a, f, c) = 18 n + n scalars, padded to 16 bytes
constexpr int GC_VAF = 18 * N + N;
constexpr int GC_VAF_PAD = (GC_VAF + 3) / 4 * 4;

template <class T, bool HAS_QDD>
__global__ __launch_bounds__(64, 1) void rnea_grad_cols_kernel(const T* __restrict__ q, const T* __restrict__ qd,
                                                               const T* __restrict__ qdd, T grav, int use_damping,
                                                               long long B, T* __restrict__ c_out, T* __restrict__ v_out,
                                                               T* __restrict__ a_out, T* __restrict__ f_out,
                                                               T* __restrict__ dcdu) {
  __shared__ __attribute__((aligned(16))) T img[GC_CPW * GC_VAF_PAD];
  const int lane = threadIdx.x;
  const int sub = lane % GC_L;                // column slot of this lane
  const int slot = lane / GC_L;               // configuration within the wave
  const long long cfg0 = (long long)blockIdx.x * GC_CPW;
  const long long cfg = cfg0 + slot;
  const bool valid = cfg < B;
  const long long b = valid ? cfg : B - 1;
  const int col = sub < 2 * N ? sub : 2 * N - 1;      // spare lanes repeat the last column (and store nothing)
  const bool isqd = col >= N;                          // d/dqd column (else d/dq)
  const int c = isqd ? col - N : col;

  // every lane of a configuration reads the same 3 n scalars (one address per group: a broadcast)
  T qv[N], qdv[N], qddv[N];
  sfor<0, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    qv[j] = q[b * N + j];
    qdv[j] = qd[b * N + j];
    if constexpr (HAS_QDD) qddv[j] = qdd[b * N + j]; else qddv[j] = T(0);
  });
  JTrig<T> tr[N];
  sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; tr[j] = make_trig<j>(qv[j]); });

  // ---- rnea (:559-628): v, a, local f; then the accumulated f and c --------------------------------
  T v[N][6], a[N][6], f[N][6];
  const T zero6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  sfor<0, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int p = PARENT[j];
    T xv[6], xa[6];
    if constexpr (p < 0)
      rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddv[j], grav, zero6, zero6, xv, xa, v[j], a[j], f[j]);
    else
      rnea_fwd_body<j, HAS_QDD>(tr[j], qdv[j], qddv[j], grav, v[p], a[p], xv, xa, v[j], a[j], f[j]);
  });
  T cr[N];
  sfor_down<0, N>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int p = PARENT[j];
    cr[j] = S_dot<j>(f[j]);
    if constexpr (p >= 0) {
      T t[6];
      xform_T<j>(tr[j], f[j], t);
      sfor<0, 6>([&](auto R) { f[p][decltype(R)::value] += t[decltype(R)::value]; });
    }
  });

  // ---- forward pass of column c (:1139-1185 | :1210-1252) ------------------------------------------
  T dv[N][6], da[N][6], df[N][6];
  sfor<0, N>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int p = PARENT[i];
    if constexpr (p >= 0) {
      xform<i>(tr[i], dv[p], dv[i]);          // (:1158 / :1230)
      xform<i>(tr[i], da[p], da[i]);          // (:1163 / :1234)
    } else {
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; dv[i][r] = T(0); da[i][r] = T(0); });
    }
    // own column (i == c):  dq:  dv += crm(X v_p) S (0 at a root, :1157-1159);  da += crm(X a_p) S (X a0 at a root, :1172-1175)
    //                       dqd: dv += S (:1231);                                da += crm(v_i) S (:1243)
    {
      T xv[6], xa[6];
      if constexpr (p >= 0) {
        xform<i>(tr[i], v[p], xv);
        xform<i>(tr[i], a[p], xa);
      } else {
        const T a0[6] = {T(0), T(0), T(0), T(0), T(0), -grav};
        sfor<0, 6>([&](auto R) { xv[decltype(R)::value] = T(0); });
        xform<i>(tr[i], a0, xa);
      }
      T sdq[6], sS[6], e1[6], e2[6];
      mxS<i>(xv, T(1), sdq);
      sfor<0, 6>([&](auto R) { sS[decltype(R)::value] = T(0); });
      add_S<i>(T(1), sS);
      mxS<i>(xa, T(1), e1);
      mxS<i>(v[i], T(1), e2);
      const bool own = (c == i);
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        dv[i][r] += sel(own, sel(isqd, sS[r], sdq[r]), T(0));
      });
      add_mxS<i>(dv[i], qdv[i], da[i]);       // da[:,c,i] += qd_i crm(dv[:,c,i]) S  (:1164-1170 / :1235-1240)
      sfor<0, 6>([&](auto R) {
        constexpr int r = decltype(R)::value;
        da[i][r] += sel(own, sel(isqd, e2[r], e1[r]), T(0));
      });
    }
    // df = I da + crf(dv)(I v) + crf(v)(I dv)  (:1179-1185 / :1247-1252)
    T Iv[6], Idv[6];
    cmatvec<MatI, i>(v[i], Iv);
    cmatvec<MatI, i>(dv[i], Idv);
    cmatvec<MatI, i>(da[i], df[i]);
    fxv<true>(dv[i], Iv, df[i]);
    fxv<true>(v[i], Idv, df[i]);
  });

  // ---- backward pass of column c (:1264-1294 | :1306-1341) -----------------------------------------
  T dc[N];
  sfor_down<0, N>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int p = PARENT[i];
    T d = S_dot<i>(df[i]);                                    // dc[i, c] = S^T df[:, c, i]  (:1284 / :1325)
    d += sel(use_damping != 0 && isqd && c == i, T(DAMPING[i]), T(0));   // (:1336-1341)
    dc[i] = d;
    if constexpr (p >= 0) {
      // dq column i of body i: df[:, i, p] += X^T fxS(S, f_i),  fxS(S, f) = -crm(f) S  (:166-168, :1292-1294)
      T w[6], x[6], y[6];
      mxS<i>(f[i], T(-1), w);
      const bool ex = (c == i) && !isqd;
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; x[r] = df[i][r] + sel(ex, w[r], T(0)); });
      xform_T<i>(tr[i], x, y);                                // df[:, c, p] += X^T df[:, c, i]  (:1291 / :1331)
      sfor<0, 6>([&](auto R) { constexpr int r = decltype(R)::value; df[p][r] += y[r]; });
    }
  });

  // ---- dc_du [B, n, 2n]: for every row i the 2n lanes of a configuration hold 2n consecutive scalars --
  if (valid && sub < 2 * N) {
    T* o = dcdu + cfg * (2 * N * N) + col;
    sfor<0, N>([&](auto I) { constexpr int i = decltype(I)::value; o[i * 2 * N] = dc[i]; });
  }
  // ---- c, v, a, f (identical in every lane of a configuration): lane 0 of the group parks them in
  // LDS in the reference's (6, NB) layouts, the wave streams the images of its configurations out ------
  const bool want_vaf = v_out != nullptr;
  if (want_vaf || c_out != nullptr) {
    if (sub == 0) {
      T* im = img + slot * GC_VAF_PAD;
      sfor<0, 6>([&](auto R) {
        sfor<0, N>([&](auto J) {
          constexpr int r = decltype(R)::value, j = decltype(J)::value;
          im[r * N + j] = v[j][r];
          im[6 * N + r * N + j] = a[j][r];
          im[12 * N + r * N + j] = f[j][r];
        });
      });
      sfor<0, N>([&](auto J) { constexpr int j = decltype(J)::value; im[18 * N + j] = cr[j]; });
    }
    __syncthreads();
    const long long rem = B - cfg0;
    const int nv = rem < GC_CPW ? (int)rem : GC_CPW;
    if (want_vaf) {
      for (int g = lane; g < nv * 6 * N; g += 64) {
        const int s = g / (6 * N), e = g - s * (6 * N);
        const T* im = img + s * GC_VAF_PAD;
        v_out[cfg0 * (6 * N) + g] = im[e];
        a_out[cfg0 * (6 * N) + g] = im[6 * N + e];
        f_out[cfg0 * (6 * N) + g] = im[12 * N + e];
      }
    }
    if (c_out != nullptr) {
      for (int g = lane; g < nv * N; g += 64) {
        const int s = g / N, e = g - s * N;
        c_out[cfg0 * N + g] = img[s * GC_VAF_PAD + 18 * N + e];
      }
    }
  }
}

}  // namespace rbdk
