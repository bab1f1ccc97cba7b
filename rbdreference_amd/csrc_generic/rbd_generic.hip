// rbd_generic.hip -- librbd_generic.so: the MODEL-HANDLE library of include/rbd_generic.h.
//
// One configuration per lane; the robot is a run-time table in device memory (wave-uniform indices -> scalar loads);
// per-lane state lives in arrays indexed by body, i.e. in private memory.  The arithmetic is the reference's own
// recursion in body coordinates on dense 6x6 operands (/root/reference/RBDReference.py:559-628, :630-806, :1127-1368,
// :1371-1384), reorganised per derivative COLUMN so that a lane never holds the (6, n, NB) tensors:
//   - rnea_grad: column c of dv / da / df lives only on the bodies of subtree(c) (forward recursions :1157-1185,
//     :1229-1252) and, in the backward sweep, on the ancestors of c (:1284-1294, :1325-1341);
//   - minv: column c of F climbs the root path of c in the backward sweep (:702-726) and is rebuilt body by body in
//     the forward sweep (:771-781); only rows <= c of the column are kept (the upper triangle, mirrored for
//     output_dense, :799-804).
// This is the first-use path (no compiler, no wait); the specialised per-robot kernels (rbd_kernels.hip) are the fast path.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/rbd_generic.h"
#include "../csrc/rbd_sincos.h"

#define GDEV __device__ __forceinline__
#ifndef RBD_G_EXTRA_LDS_DEFAULT
#define RBD_G_EXTRA_LDS_DEFAULT 0
#endif

namespace rbdg {

// Resident waves per CU of the lane-per-configuration kernels.  Their per-lane state lives in private (scratch) memory, 1.4-37 KB
// per lane, and a launch's scratch footprint is (resident waves) x 64 x that: with every wave slot the LDS tile allows (9 per
// CU) the 7-body gradient kernel's footprint is 426 MB -- more than the 256 MB last-level cache, so every private access is an
// HBM access.  Extra dynamic LDS per block caps the residency (RBD_G_EXTRA_LDS bytes; default chosen from measurements).
inline size_t extra_lds() {
  static const size_t v = [] { const char* e = std::getenv("RBD_G_EXTRA_LDS"); return e ? (size_t)std::atol(e) : (size_t)RBD_G_EXTRA_LDS_DEFAULT; }();
  return v;
}


constexpr int MB = RBD_G_MAX_BODIES;

template <class T>
struct DevModel {
  int n;                        // bodies
  int fb;                       // 1: body 0 is the floating base (6-DoF joint, S = eye(6), q[0:6] = px py pz rx ry rz)
  int parent[MB];
  int jtype[MB];
  unsigned long long anc[MB];   // bit j of anc[i]: j == i or j is an ancestor of i
  T S[MB][6];
  T X0[MB][36], Xs[MB][36], Xc[MB][36];
  T I[MB][36];
  T damping[MB];
  // world-frame gradient kernel (g_rnea_grad_world_kernel): valid when every joint is revolute with S = (axis; 0) and
  // every inertia has rigid-body form; then mass / centre of mass / rotational inertia about it, per body
  int world_ok;
  T mass[MB], com[MB][3], Ic[MB][6], sa[MB][3];
};

// own sin / cos (csrc/rbd_sincos.h): fast path + branch-free wide path behind a wave-uniform branch, no libm / ocml
// routine (their lane-masked if / else bodies are the round-3 fault's cause, see that header)
GDEV void sincos_g(float x, float* s, float* c) { rbdsc::sincos_(x, s, c); }
GDEV void sincos_g(double x, double* s, double* c) { rbdsc::sincos_(x, s, c); }

// (f1, f2) of X(q) = X0 + Xs f1 + Xc f2
template <class T>
GDEV void joint_fun(int jt, T q, T& f1, T& f2) {
  if (jt == 0) {
    sincos_g(q, &f1, &f2);
  } else {
    f1 = q;
    f2 = T(0);
  }
}
template <class T>
GDEV void build_X(const DevModel<T>* __restrict__ m, int i, T f1, T f2, T (&X)[36]) {
#pragma unroll
  for (int k = 0; k < 36; ++k) X[k] = fma(m->Xc[i][k], f2, fma(m->Xs[i][k], f1, m->X0[i][k]));
}
template <class T>
GDEV void mv(const T (&X)[36], const T* v, T (&o)[6]) {          // o = X v
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    T acc = X[r * 6] * v[0];
#pragma unroll
    for (int c = 1; c < 6; ++c) acc = fma(X[r * 6 + c], v[c], acc);
    o[r] = acc;
  }
}
template <class T>
GDEV void mtv(const T (&X)[36], const T* f, T (&o)[6]) {         // o = X^T f
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    T acc = X[c] * f[0];
#pragma unroll
    for (int r = 1; r < 6; ++r) acc = fma(X[r * 6 + c], f[r], acc);
    o[c] = acc;
  }
}
template <class T>
GDEV void mvI(const DevModel<T>* __restrict__ m, int i, const T* v, T (&o)[6]) {   // o = I_i v
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    T acc = m->I[i][r * 6] * v[0];
#pragma unroll
    for (int c = 1; c < 6; ++c) acc = fma(m->I[i][r * 6 + c], v[c], acc);
    o[r] = acc;
  }
}
template <class T>
GDEV void cross3g(const T* a, const T* b, T* o) {
  o[0] = fma(a[1], b[2], -(a[2] * b[1]));
  o[1] = fma(a[2], b[0], -(a[0] * b[2]));
  o[2] = fma(a[0], b[1], -(a[1] * b[0]));
}
// o = crm(v) x   (cross_operator, RBDReference.py:9-21; mxS(S, v) = crm(v) S, :56-59)
template <class T>
GDEV void crm_mul(const T* v, const T* x, T (&o)[6]) {
  T t0[3], t1[3], t2[3];
  cross3g(v, x, t0);
  cross3g(v + 3, x, t1);
  cross3g(v, x + 3, t2);
  o[0] = t0[0]; o[1] = t0[1]; o[2] = t0[2];
  o[3] = t1[0] + t2[0]; o[4] = t1[1] + t2[1]; o[5] = t1[2] + t2[2];
}
// o = crf(v) f = -crm(v)^T f   (fxv, :149-164)
template <class T>
GDEV void crf_mul(const T* v, const T* f, T (&o)[6]) {
  T t0[3], t1[3], t2[3];
  cross3g(v, f, t0);
  cross3g(v + 3, f + 3, t1);
  cross3g(v, f + 3, t2);
  o[0] = t0[0] + t1[0]; o[1] = t0[1] + t1[1]; o[2] = t0[2] + t1[2];
  o[3] = t2[0]; o[4] = t2[1]; o[5] = t2[2];
}
template <class T>
GDEV T dot6g(const T* a, const T* b) {
  T acc = a[0] * b[0];
#pragma unroll
  for (int r = 1; r < 6; ++r) acc = fma(a[r], b[r], acc);
  return acc;
}

// world -> base transform of the 6-DoF base joint: plux(Rz(rz) Ry(ry) Rx(rx), p) with the coordinate-transform
// rotations (robot.floating_base_X; RBDReference.py:634-637).  bt = {sx, cx, sy, cy, sz, cz, px, py, pz}
template <class T>
GDEV void base_X(const T (&bt)[9], T (&X)[36]) {
  const T sx = bt[0], cx = bt[1], sy = bt[2], cy = bt[3], sz = bt[4], cz = bt[5];
  // E = Rz Ry Rx,  Rx = [[1,0,0],[0,c,s],[0,-s,c]], Ry = [[c,0,-s],[0,1,0],[s,0,c]], Rz = [[c,s,0],[-s,c,0],[0,0,1]]
  const T A[3][3] = {{cy, sy * sx, -sy * cx}, {T(0), cx, sx}, {sy, -cy * sx, cy * cx}};   // Ry Rx
  T E[3][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    E[0][c] = fma(cz, A[0][c], sz * A[1][c]);
    E[1][c] = fma(-sz, A[0][c], cz * A[1][c]);
    E[2][c] = A[2][c];
  }
  const T p[3] = {bt[6], bt[7], bt[8]};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    // -E p^x : row r = -(E[r] x p)^T ... (E p^x)[r][c] = sum_k E[r][k] (p^x)[k][c];  p^x = [[0,-p2,p1],[p2,0,-p0],[-p1,p0,0]]
    const T b0 = -(E[r][1] * p[2] - E[r][2] * p[1]);
    const T b1 = -(E[r][2] * p[0] - E[r][0] * p[2]);
    const T b2 = -(E[r][0] * p[1] - E[r][1] * p[0]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      X[r * 6 + c] = E[r][c];
      X[r * 6 + 3 + c] = T(0);
      X[(3 + r) * 6 + 3 + c] = E[r][c];
    }
    X[(3 + r) * 6 + 0] = b0; X[(3 + r) * 6 + 1] = b1; X[(3 + r) * 6 + 2] = b2;
  }
}


// ---- coalesced output: a lane's results wait in private memory, the wave writes them out through an LDS tile ----------
// A lane owns a configuration, and a configuration's outputs are contiguous in HBM: written as they are formed they
// are 4-byte stores 2 n^2 (or 6 n, n^2) elements apart between neighbouring lanes -- every 64-byte sector is written
// sixteen times, partially.  Instead the values collect in a private array and leave KCH per configuration at a time:
// tile[lane][k] <- private, then consecutive lanes store consecutive elements of one configuration (256-byte runs).
#ifndef RBD_G_KCH
#define RBD_G_KCH 64
#endif
constexpr int KCH = RBD_G_KCH;
constexpr int KCHP = KCH + 1;                       // odd row stride: the lane-major tile writes hit 64 different banks
template <int NMAX>
constexpr bool stage_outputs() { return NMAX <= 32; }   // (64 bodies: 2 * 69^2 values per lane would not fit private memory)
GDEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
template <class T, class F>
GDEV void flush_coalesced(F get, int count, T* __restrict__ gdst, long long cfg_stride, int nvalid, T* tile, int lane) {
  for (int k0 = 0; k0 < count; k0 += KCH) {
    const int kc = count - k0 < KCH ? count - k0 : KCH;
    for (int k = 0; k < kc; ++k) tile[lane * KCHP + k] = get(k0 + k);
    wave_sync();
    for (int idx = lane; idx < nvalid * kc; idx += 64) {
      const int cfg = idx / kc, k = idx - cfg * kc;
      gdst[cfg * cfg_stride + k0 + k] = tile[cfg * KCHP + k];
    }
    wave_sync();
  }
}
// this lane's configuration: the block's last wave repeats its last row in the lanes beyond the batch (all 64 lanes
// take part in the flush; only `nvalid` configurations are stored)
#define RBDG_LANE_CONFIG()                                                              \
  const long long b0 = (long long)blockIdx.x * 64;                                      \
  const int lane = threadIdx.x;                                                         \
  const int nvalid = (int)(B - b0 < 64 ? B - b0 : 64);                                  \
  const long long b = b0 + (lane < nvalid ? lane : nvalid - 1)

template <class T, int NMAX>
struct RneaState {
  T v[NMAX][6], a[NMAX][6], f[NMAX][6];
  T f1[NMAX], f2[NMAX], qd[NMAX];
  T bt[9], qd6[6];              // floating base: trig / position of q[0:6], the base twist qd[0:6]
};
// X_i(q) of body i from the state (floating base: body 0 from bt)
template <bool FB, class T, int NMAX>
GDEV void body_X(const DevModel<T>* __restrict__ m, int i, const RneaState<T, NMAX>& st, T (&X)[36]) {
  if (FB && i == 0) base_X(st.bt, X);
  else build_X(m, i, st.f1[i], st.f2[i], X);
}

// rnea_fpass + rnea_bpass of one configuration (:559-621); leaves v, a, the ACCUMULATED f, (f1, f2), qd in st.
// vo / ao / fo / co: this configuration's output rows (nullable); v, a, f are [6][nb], c has nv entries.
// FB: body 0 is the floating base -- X_0 from q[0:6], vJ = qd[0:6] (S = eye(6), :585), qdd[0:6] (:591), c[0:6] = f_0
// (:612), body i >= 1 owns index i + 5.
template <bool FB, class T, int NMAX>
GDEV void rnea_config(const DevModel<T>* __restrict__ m, int n, const T* q, const T* qd, const T* qdd, T grav,
                      RneaState<T, NMAX>& st, T* co, T* vo, T* ao, T* fo, bool fpass_only = false) {
  constexpr int OFF = FB ? 5 : 0;
  for (int i = 0; i < n; ++i) {
    T X[36], v[6], a[6], vJ[6], t[6];
    const int p = m->parent[i];
    if (FB && i == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { sincos_g(q[3 + k], &st.bt[2 * k], &st.bt[2 * k + 1]); st.bt[6 + k] = q[k]; }
      base_X(st.bt, X);
#pragma unroll
      for (int r = 0; r < 6; ++r) { vJ[r] = qd[r]; st.qd6[r] = vJ[r]; }
      st.f1[0] = T(0); st.f2[0] = T(1); st.qd[0] = T(0);
    } else {
      T f1, f2;
      joint_fun(m->jtype[i], q[i + OFF], f1, f2);
      st.f1[i] = f1; st.f2[i] = f2;
      const T qdi = qd[i + OFF];
      st.qd[i] = qdi;
      build_X(m, i, f1, f2, X);
#pragma unroll
      for (int r = 0; r < 6; ++r) vJ[r] = m->S[i][r] * qdi;                                // :586
    }
    if (p < 0) {                                            // v_p = 0, a_p = [0,0,0,0,0,-GRAVITY]  (:565-566, :576-581)
#pragma unroll
      for (int r = 0; r < 6; ++r) { v[r] = T(0); a[r] = X[r * 6 + 5] * (-grav); }
    } else {
      mv(X, st.v[p], v);
      mv(X, st.a[p], a);
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) v[r] += vJ[r];                                               // :587
    crm_mul(v, vJ, t);                                                                      // :588
#pragma unroll
    for (int r = 0; r < 6; ++r) a[r] += t[r];
    if (qdd != nullptr) {                                                                   // :589-593
      if (FB && i == 0) {
#pragma unroll
        for (int r = 0; r < 6; ++r) a[r] += qdd[r];
      } else {
        const T qddi = qdd[i + OFF];
#pragma unroll
        for (int r = 0; r < 6; ++r) a[r] = fma(m->S[i][r], qddi, a[r]);
      }
    }
    T Iv[6], Ia[6], w[6];
    mvI(m, i, v, Iv);
    mvI(m, i, a, Ia);
    crf_mul(v, Iv, w);                                                                      // :595-596
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      st.v[i][r] = v[r]; st.a[i][r] = a[r]; st.f[i][r] = Ia[r] + w[r];
      if (vo) vo[r * n + i] = v[r];
      if (ao) ao[r * n + i] = a[r];
      if (fpass_only && fo) fo[r * n + i] = Ia[r] + w[r];                                    // rnea_fpass returns the LOCAL force (:595-598)
    }
  }
  if (fpass_only) return;
  for (int i = n - 1; i >= 0; --i) {
    T f[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) f[r] = st.f[i][r];
    if (co) {                                                                               // :612
      if (FB && i == 0) {
#pragma unroll
        for (int r = 0; r < 6; ++r) co[r] = f[r];
      } else {
        co[i + OFF] = dot6g(m->S[i], f);
      }
    }
    if (fo) {
#pragma unroll
      for (int r = 0; r < 6; ++r) fo[r * n + i] = f[r];
    }
    const int p = m->parent[i];
    if (p >= 0) {
      T X[36], t[6];
      build_X(m, i, st.f1[i], st.f2[i], X);
      mtv(X, f, t);                                                                         // :618-619
#pragma unroll
      for (int r = 0; r < 6; ++r) st.f[p][r] += t[r];
    }
  }
}

template <class T, int NMAX, bool FB>
__global__ void __launch_bounds__(64) g_rnea_kernel(const DevModel<T>* __restrict__ m, const T* __restrict__ q,
                                                    const T* __restrict__ qd, const T* __restrict__ qdd, T grav,
                                                    long long B, T* __restrict__ c, T* __restrict__ v,
                                                    T* __restrict__ a, T* __restrict__ f, int stage_rt) {
  RBDG_LANE_CONFIG();
  const int n = m->n, nv = n + (FB ? 5 : 0);
  RneaState<T, NMAX> st;
  if (stage_outputs<NMAX>() && stage_rt) {
    __shared__ T tile[64 * KCHP];
    T cb[NMAX + 5], vb[6 * NMAX], ab[6 * NMAX], fb[6 * NMAX];
    rnea_config<FB, T, NMAX>(m, n, q + b * nv, qd + b * nv, qdd ? qdd + b * nv : nullptr, grav, st, cb,
                             v ? vb : nullptr, a ? ab : nullptr, f ? fb : nullptr);
    flush_coalesced([&](int k) { return cb[k]; }, nv, c + b0 * nv, nv, nvalid, tile, lane);
    if (v) flush_coalesced([&](int k) { return vb[k]; }, 6 * n, v + b0 * 6 * n, 6 * n, nvalid, tile, lane);
    if (a) flush_coalesced([&](int k) { return ab[k]; }, 6 * n, a + b0 * 6 * n, 6 * n, nvalid, tile, lane);
    if (f) flush_coalesced([&](int k) { return fb[k]; }, 6 * n, f + b0 * 6 * n, 6 * n, nvalid, tile, lane);
  } else {
    rnea_config<FB, T, NMAX>(m, n, q + b * nv, qd + b * nv, qdd ? qdd + b * nv : nullptr, grav, st, c + b * nv,
                             v ? v + b * 6 * n : nullptr, a ? a + b * 6 * n : nullptr, f ? f + b * 6 * n : nullptr);
  }
}

// df = I da + crf(dv)(I v) + crf(v)(I dv)   (:1179-1185, :1247-1252)
template <class T>
GDEV void df_of(const DevModel<T>* __restrict__ m, int i, const T* v, const T (&Iv)[6], const T (&dv)[6], const T (&da)[6],
                T* out) {
  T Ida[6], Idv[6], w1[6], w2[6];
  mvI(m, i, da, Ida);
  mvI(m, i, dv, Idv);
  crf_mul(dv, Iv, w1);
  crf_mul(v, Idv, w2);
#pragma unroll
  for (int r = 0; r < 6; ++r) out[r] = Ida[r] + w1[r] + w2[r];
}

// Velocity damping exactly as the reference adds it (:1336-1341): matrix index = BODY id `ind`; for a floating base
// the base contributes a whole 5 x 5 block and body ind >= 1 lands on row / column `ind` (not ind + 5).
template <bool FB, class T>
GDEV T damping_at(const DevModel<T>* __restrict__ m, int n, int row, int col) {
  if (!FB) return row == col ? m->damping[row] : T(0);
  T d = T(0);
  if (row < 5 && col < 5) d += m->damping[0];
  if (row == col && row >= 1 && row < n) d += m->damping[row];
  return d;
}

template <class T, int NMAX, bool FB>
__global__ void __launch_bounds__(64) g_rnea_grad_kernel(const DevModel<T>* __restrict__ m, const T* __restrict__ q,
                                                         const T* __restrict__ qd, const T* __restrict__ qdd, T grav,
                                                         int use_damping, long long B, T* __restrict__ c,
                                                         T* __restrict__ dc, int stage_rt) {
  RBDG_LANE_CONFIG();
  constexpr int OFF = FB ? 5 : 0;
  constexpr bool STAGEC = stage_outputs<NMAX>();
  const bool STAGE = STAGEC && stage_rt;
  const int n = m->n, nv = n + OFF;
  RneaState<T, NMAX> st;
  T cb[STAGEC ? NMAX + 5 : 1];
  T obuf[STAGEC ? 2 * (NMAX + OFF) * (NMAX + OFF) : 1];
  rnea_config<FB, T, NMAX>(m, n, q + b * nv, qd + b * nv, qdd ? qdd + b * nv : nullptr, grav, st,
                           c ? (STAGE ? cb : c + b * nv) : nullptr, (T*)nullptr, (T*)nullptr, (T*)nullptr);   // :1353
  T dvq[NMAX][6], daq[NMAX][6], dfq[NMAX][6], dvd[NMAX][6], dad[NMAX][6], dfd[NMAX][6];
  T* row = STAGE ? obuf : dc + b * 2 * nv * nv;
  for (int col = 0; col < nv; ++col) {
    const int c0 = FB ? (col < 6 ? 0 : col - 5) : col;      // the body that owns the column
    // ---- forward recursions of the column over subtree(c0) ----
    for (int i = c0; i < n; ++i) {
      if (!((m->anc[i] >> c0) & 1ull)) continue;
      T X[36], S[6];
      body_X<FB>(m, i, st, X);
      const int p = m->parent[i];
      T vq[6], aq[6], vd[6], ad[6], t[6];
      if (FB && i == 0) {                                   // base column `col` (S = e_col); :1175, :1231, :1236-1243
        T xa[6], e[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) { xa[r] = X[r * 6 + 5] * (-grav); e[r] = r == col ? T(1) : T(0); vq[r] = T(0); vd[r] = e[r]; }
        crm_mul(xa, e, aq);                                 // da[:, col, 0] = crm(X a_grav) S
        crm_mul(vd, st.qd6, t);                             // sum_ii qd_ii crm(dv) S[ii]
        crm_mul(st.v[0], e, ad);                            // + crm(v) S
#pragma unroll
        for (int r = 0; r < 6; ++r) ad[r] += t[r];
      } else {
#pragma unroll
        for (int r = 0; r < 6; ++r) S[r] = m->S[i][r];
        const T qdi = st.qd[i];
        if (i == c0) {
          T xa[6];
          if (p >= 0) {
            T xv[6];
            mv(X, st.v[p], xv);
            crm_mul(xv, S, vq);                             // dv[:,i,i] += crm(X v_p) S     (:1157-1159)
            mv(X, st.a[p], xa);
          } else {
#pragma unroll
            for (int r = 0; r < 6; ++r) { vq[r] = T(0); xa[r] = X[r * 6 + 5] * (-grav); }
          }
          crm_mul(vq, S, t);                                // da[:,c,i] += qd_i crm(dv[:,c,i]) S   (:1164-1170)
          crm_mul(xa, S, aq);                               // da[:,i,i] += crm(X a_p) S            (:1172-1175)
#pragma unroll
          for (int r = 0; r < 6; ++r) aq[r] = fma(qdi, t[r], aq[r]);
#pragma unroll
          for (int r = 0; r < 6; ++r) vd[r] = S[r];         // dv[:,i,i] += S                       (:1231)
          crm_mul(vd, S, t);                                // (:1235-1240)
          crm_mul(st.v[i], S, ad);                          // da[:,i,i] += crm(v_i) S              (:1243)
#pragma unroll
          for (int r = 0; r < 6; ++r) ad[r] = fma(qdi, t[r], ad[r]);
        } else {
          mv(X, dvq[p], vq);                                // (:1157, :1162-1163)
          mv(X, daq[p], aq);
          crm_mul(vq, S, t);
#pragma unroll
          for (int r = 0; r < 6; ++r) aq[r] = fma(qdi, t[r], aq[r]);
          mv(X, dvd[p], vd);                                // (:1229-1234)
          mv(X, dad[p], ad);
          crm_mul(vd, S, t);
#pragma unroll
          for (int r = 0; r < 6; ++r) ad[r] = fma(qdi, t[r], ad[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) { dvq[i][r] = vq[r]; daq[i][r] = aq[r]; dvd[i][r] = vd[r]; dad[i][r] = ad[r]; }
      T Iv[6];
      mvI(m, i, st.v[i], Iv);
      df_of(m, i, st.v[i], Iv, vq, aq, dfq[i]);
      df_of(m, i, st.v[i], Iv, vd, ad, dfd[i]);
    }
    for (int x = m->parent[c0]; x >= 0; x = m->parent[x]) {
#pragma unroll
      for (int r = 0; r < 6; ++r) { dfq[x][r] = T(0); dfd[x][r] = T(0); }
    }
    // ---- backward sweep of the column over subtree(c0) and the ancestors of c0 ----
    for (int i = n - 1; i >= 0; --i) {
      const bool rel = (((m->anc[i] >> c0) & 1ull) != 0) || (((m->anc[c0] >> i) & 1ull) != 0);
      T gq[6], gd[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) { gq[r] = rel ? dfq[i][r] : T(0); gd[r] = rel ? dfd[i][r] : T(0); }
      if (FB && i == 0) {                                   // dc[0:6, :] = df[:, :, 0]   (:1282, :1325 with S = eye(6))
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          row[r * 2 * nv + col] = gq[r];
          row[r * 2 * nv + nv + col] = gd[r] + (use_damping ? damping_at<FB>(m, n, r, col) : T(0));
        }
        continue;
      }
      T eq = T(0), ed = T(0);
      if (rel) {
        eq = dot6g(m->S[i], gq);                            // dc_dq[i,:] = S^T df[:,:,i]   (:1284)
        ed = dot6g(m->S[i], gd);                            // (:1325)
        const int p = m->parent[i];
        if (p >= 0) {
          T X[36], t[6];
          build_X(m, i, st.f1[i], st.f2[i], X);
          if (i == c0 && (!FB || col >= 6)) {               // df[:,i,p] += X^T fxS(S, f_i), fxS = -crm(f_i) S  (:1292-1294)
            T g[6];
            crm_mul(st.f[i], m->S[i], g);
#pragma unroll
            for (int r = 0; r < 6; ++r) gq[r] -= g[r];
          }
          mtv(X, gq, t);                                    // df[:,:,p] += X^T df[:,:,i]   (:1291)
#pragma unroll
          for (int r = 0; r < 6; ++r) dfq[p][r] += t[r];
          mtv(X, gd, t);                                    // (:1331)
#pragma unroll
          for (int r = 0; r < 6; ++r) dfd[p][r] += t[r];
        }
      }
      if (use_damping) ed += damping_at<FB>(m, n, i + OFF, col);   // (:1336-1341)
      row[(i + OFF) * 2 * nv + col] = eq;
      row[(i + OFF) * 2 * nv + nv + col] = ed;
    }
  }
  if (STAGE) {
    __shared__ T tile[64 * KCHP];
    if (c) flush_coalesced([&](int k) { return cb[k]; }, nv, c + b0 * nv, nv, nvalid, tile, lane);
    flush_coalesced([&](int k) { return obuf[k]; }, 2 * nv * nv, dc + b0 * 2 * nv * nv, 2LL * nv * nv, nvalid, tile, lane);
  }
}


// ---- world-frame gradient kernel for all-revolute robots ---------------------------------------------------------------
// The first-order identities of the reference's IDSVA scheme (/root/reference/RBDReference.py:1413-1484), the ones the
// specialised kernels evaluate (rbd_world.h), with the robot as a run-time table: ~500 operations per body and ~40 per
// (body, ancestor) pair instead of ~300 per pair, and a third of the private-memory traffic of the column recursion
// above.  They reproduce RBDReference.rnea_grad (:1345-1368) for revolute joints (the reference's fxS term is the true
// derivative only there, :1292-1294), so robots with a prismatic joint keep g_rnea_grad_kernel.
template <class T>
struct GInertia { T m, h[3], I[6]; };           // rigid inertia about the world origin: mass, m c, Ibar (xx xy xz yy yz zz)
template <class T>
struct GSym { T TL[6], G[3]; };                  // symmetric part of the body-level Coriolis matrix: [[TL, G^x], [G^x^T, 0]]
template <class T>
struct GComp { GInertia<T> IC; GSym<T> SC; T pm[6], f[6]; };
constexpr int GCOMP = 31;

template <class T>
GDEV void gin_apply(const GInertia<T>& R, const T* x, T (&y)[6]) {
  const T* w = x; const T* u = x + 3;
  T hu[3], hw[3];
  cross3g(R.h, u, hu);
  cross3g(R.h, w, hw);
  y[0] = fma(R.I[0], w[0], fma(R.I[1], w[1], fma(R.I[2], w[2], hu[0])));
  y[1] = fma(R.I[1], w[0], fma(R.I[3], w[1], fma(R.I[4], w[2], hu[1])));
  y[2] = fma(R.I[2], w[0], fma(R.I[4], w[1], fma(R.I[5], w[2], hu[2])));
  y[3] = fma(R.m, u[0], -hw[0]); y[4] = fma(R.m, u[1], -hw[1]); y[5] = fma(R.m, u[2], -hw[2]);
}
template <class T>
GDEV void gsym_apply(const GSym<T>& S, const T* x, T (&y)[6]) {
  const T* a = x; const T* b = x + 3;
  T gb[3], ga[3];
  cross3g(S.G, b, gb);
  cross3g(S.G, a, ga);
  y[0] = fma(S.TL[0], a[0], fma(S.TL[1], a[1], fma(S.TL[2], a[2], gb[0])));
  y[1] = fma(S.TL[1], a[0], fma(S.TL[3], a[1], fma(S.TL[4], a[2], gb[1])));
  y[2] = fma(S.TL[2], a[0], fma(S.TL[4], a[1], fma(S.TL[5], a[2], gb[2])));
  y[3] = -ga[0]; y[4] = -ga[1]; y[5] = -ga[2];
}

template <class T, int NMAX>
__global__ void __launch_bounds__(64) g_rnea_grad_world_kernel(const DevModel<T>* __restrict__ m, const T* __restrict__ q,
                                                               const T* __restrict__ qd, const T* __restrict__ qdd, T grav,
                                                               int use_damping, long long B, T* __restrict__ c_out,
                                                               T* __restrict__ dc, int stage_rt) {
  RBDG_LANE_CONFIG();
  constexpr bool STAGEC = stage_outputs<NMAX>();
  const bool STAGE = STAGEC && stage_rt;
  const int n = m->n;
  T cb[STAGEC ? NMAX : 1];
  T obuf[STAGEC ? 2 * NMAX * NMAX : 1];
  T Sv[NMAX][6], Pd[NMAX][6], Pdd[NMAX][6], vw[NMAX][6], aw[NMAX][6], Rw[NMAX][9], pw[NMAX][3];
  T Cm[NMAX][GCOMP];
  // ---- root -> leaves: world kinematics (:1413-1434) and every body's own inertia terms (:1436-1440) ----
  for (int i = 0; i < n; ++i) {
    T f1, f2, X[36];
    joint_fun(0, q[b * n + i], f1, f2);
    build_X(m, i, f1, f2, X);
    const int p = m->parent[i];
    const T qdi = qd[b * n + i], qddi = qdd ? qdd[b * n + i] : T(0);
    // X = [[E, 0], [-E r^x, E]]: R = R_p E^T, origin = p_p + R_p r with r^x = -E^T (lower-left block)
    T r3[3];
    r3[0] = -(X[0 * 6 + 2] * X[3 * 6 + 1] + X[1 * 6 + 2] * X[4 * 6 + 1] + X[2 * 6 + 2] * X[5 * 6 + 1]);
    r3[1] = -(X[0 * 6 + 0] * X[3 * 6 + 2] + X[1 * 6 + 0] * X[4 * 6 + 2] + X[2 * 6 + 0] * X[5 * 6 + 2]);
    r3[2] = -(X[0 * 6 + 1] * X[3 * 6 + 0] + X[1 * 6 + 1] * X[4 * 6 + 0] + X[2 * 6 + 1] * X[5 * 6 + 0]);
    T R[9], pp[3];
    if (p < 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int c = 0; c < 3; ++c) R[a * 3 + c] = X[c * 6 + a];
        pp[a] = r3[a];
      }
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const T ra0 = Rw[p][a * 3], ra1 = Rw[p][a * 3 + 1], ra2 = Rw[p][a * 3 + 2];
#pragma unroll
        for (int c = 0; c < 3; ++c) R[a * 3 + c] = fma(ra0, X[c * 6 + 0], fma(ra1, X[c * 6 + 1], ra2 * X[c * 6 + 2]));
        pp[a] = fma(ra0, r3[0], fma(ra1, r3[1], fma(ra2, r3[2], pw[p][a])));
      }
    }
    T S[6], psid[6], psidd[6], v[6], a[6];
#pragma unroll
    for (int r = 0; r < 3; ++r) S[r] = fma(R[r * 3], m->sa[i][0], fma(R[r * 3 + 1], m->sa[i][1], R[r * 3 + 2] * m->sa[i][2]));
    cross3g(pp, S, S + 3);
    if (p < 0) {            // v_p = 0, a_p = [0,0,0,0,0,-GRAVITY]  (:1417-1420)
      T ap[6] = {T(0), T(0), T(0), T(0), T(0), -grav};
      crm_mul(ap, S, psidd);
#pragma unroll
      for (int r = 0; r < 6; ++r) { psid[r] = T(0); v[r] = S[r] * qdi; a[r] = fma(S[r], qddi, ap[r]); }
    } else {
      T t1[6], t2[6], vp[6], ap[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) { vp[r] = vw[p][r]; ap[r] = aw[p][r]; }
      crm_mul(vp, S, psid);                                 // psid  = v_p x S                 (:1431)
      crm_mul(ap, S, t1);                                   // psidd = a_p x S + v_p x psid    (:1432)
      crm_mul(vp, psid, t2);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        psidd[r] = t1[r] + t2[r];
        v[r] = fma(S[r], qdi, vp[r]);                                       // (:1433)
        a[r] = fma(S[r], qddi, fma(psid[r], qdi, ap[r]));                   // (:1430, :1434)
      }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) { Sv[i][r] = S[r]; Pd[i][r] = psid[r]; Pdd[i][r] = psidd[r]; vw[i][r] = v[r]; aw[i][r] = a[r]; }
#pragma unroll
    for (int k = 0; k < 9; ++k) Rw[i][k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) pw[i][k] = pp[k];
    // the body's own terms about the world origin
    GComp<T> L;
    T cw[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) cw[r] = fma(R[r * 3], m->com[i][0], fma(R[r * 3 + 1], m->com[i][1], fma(R[r * 3 + 2], m->com[i][2], pp[r])));
    L.IC.m = m->mass[i];
#pragma unroll
    for (int r = 0; r < 3; ++r) L.IC.h[r] = L.IC.m * cw[r];
    {
      const T i0 = m->Ic[i][0], i1 = m->Ic[i][1], i2 = m->Ic[i][2], i3 = m->Ic[i][3], i4 = m->Ic[i][4], i5 = m->Ic[i][5];
      T A[9];                                               // A = R Ic
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        A[r * 3 + 0] = fma(R[r * 3], i0, fma(R[r * 3 + 1], i1, R[r * 3 + 2] * i2));
        A[r * 3 + 1] = fma(R[r * 3], i1, fma(R[r * 3 + 1], i3, R[r * 3 + 2] * i4));
        A[r * 3 + 2] = fma(R[r * 3], i2, fma(R[r * 3 + 1], i4, R[r * 3 + 2] * i5));
      }
      const T cc = fma(cw[0], cw[0], fma(cw[1], cw[1], cw[2] * cw[2]));
      const int IR[6] = {0, 0, 0, 1, 1, 2}, IC_[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const int r = IR[e], c = IC_[e];
        T x = fma(A[r * 3], R[c * 3], fma(A[r * 3 + 1], R[c * 3 + 1], A[r * 3 + 2] * R[c * 3 + 2]));
        const T mcc = L.IC.h[r] * cw[c];
        x += (r == c) ? fma(L.IC.m, cc, -mcc) : -mcc;
        L.IC.I[e] = x;
      }
    }
    T Ia[6];
    gin_apply(L.IC, v, L.pm);
    gin_apply(L.IC, a, Ia);
    crf_mul(v, L.pm, L.f);                                  // f = I a + v x* (I v)   (:1440)
#pragma unroll
    for (int r = 0; r < 6; ++r) L.f[r] += Ia[r];
    {
      const T* w = v; const T* u = v + 3;
      const T If[9] = {L.IC.I[0], L.IC.I[1], L.IC.I[2], L.IC.I[1], L.IC.I[3], L.IC.I[4], L.IC.I[2], L.IC.I[4], L.IC.I[5]};
      T K[9];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const T col[3] = {If[c], If[3 + c], If[6 + c]};
        T o[3];
        cross3g(w, col, o);
        K[c] = o[0]; K[3 + c] = o[1]; K[6 + c] = o[2];
      }
      const T uh2 = T(2) * fma(u[0], L.IC.h[0], fma(u[1], L.IC.h[1], u[2] * L.IC.h[2]));
      const int IR[6] = {0, 0, 0, 1, 1, 2}, IC_[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const int r = IR[e], c = IC_[e];
        T x = K[r * 3 + c] + K[c * 3 + r];
        x = fma(-L.IC.h[r], u[c], fma(-u[r], L.IC.h[c], x));
        if (r == c) x += uh2;
        L.SC.TL[e] = x;
      }
      T g[3];
      cross3g(w, L.IC.h, g);
#pragma unroll
      for (int r = 0; r < 3; ++r) L.SC.G[r] = fma(L.IC.m, u[r], g[r]);
    }
    Cm[i][0] = L.IC.m;
#pragma unroll
    for (int r = 0; r < 3; ++r) { Cm[i][1 + r] = L.IC.h[r]; Cm[i][16 + r] = L.SC.G[r]; }
#pragma unroll
    for (int r = 0; r < 6; ++r) { Cm[i][4 + r] = L.IC.I[r]; Cm[i][10 + r] = L.SC.TL[r]; Cm[i][19 + r] = L.pm[r]; Cm[i][25 + r] = L.f[r]; }
  }
  // ---- leaves -> root: composites (:1446-1448), c, t-vectors (:1481-1484), every (body, ancestor) pair ----
  T* row = STAGE ? obuf : dc + b * 2 * n * n;
  for (int j = n - 1; j >= 0; --j) {
    GComp<T> C;
    C.IC.m = Cm[j][0];
#pragma unroll
    for (int r = 0; r < 3; ++r) { C.IC.h[r] = Cm[j][1 + r]; C.SC.G[r] = Cm[j][16 + r]; }
#pragma unroll
    for (int r = 0; r < 6; ++r) { C.IC.I[r] = Cm[j][4 + r]; C.SC.TL[r] = Cm[j][10 + r]; C.pm[r] = Cm[j][19 + r]; C.f[r] = Cm[j][25 + r]; }
    const int p = m->parent[j];
    if (p >= 0) {
#pragma unroll
      for (int k = 0; k < GCOMP; ++k) Cm[p][k] += Cm[j][k];
    }
    T S[6], psid[6], psidd[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) { S[r] = Sv[j][r]; psid[r] = Pd[j][r]; psidd[r] = Pdd[j][r]; }
    if (c_out) { const T cj = dot6g(S, C.f); if (STAGE) cb[j] = cj; else c_out[b * n + j] = cj; }
    T t1[6], t2[6], t3[6], t4[6];
    {
      T y3[6], s1[6], z1[6], zf[6], y2[6], s2[6], z2[6];
      gin_apply(C.IC, S, t1);
      gin_apply(C.IC, psidd, y3);
      gsym_apply(C.SC, S, s1);
      crf_mul(S, C.pm, z1);
      crf_mul(S, C.f, zf);
      gin_apply(C.IC, psid, y2);
      gsym_apply(C.SC, psid, s2);
      crf_mul(psid, C.pm, z2);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        t4[r] = s1[r] - z1[r];
        t3[r] = (s2[r] + z2[r]) + (y3[r] + zf[r]);
        t2[r] = fma(T(2), y2[r], s1[r] + z1[r]);
      }
    }
    const unsigned long long anc = m->anc[j];
    for (int c = 0; c < n; ++c) {
      if ((anc >> c) & 1ull) {                              // c == j or an ancestor of j
        T Sc[6], Pc[6], Pcc[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) { Sc[r] = Sv[c][r]; Pc[r] = Pd[c][r]; Pcc[r] = Pdd[c][r]; }
        T dq = dot6g(t4, Pc) + dot6g(t1, Pcc);
        T dqd = fma(T(2), dot6g(t1, Pc), dot6g(t4, Sc));
        if (c == j && use_damping) dqd += m->damping[j];    // (:1336-1341)
        row[j * 2 * n + c] = dq;
        row[j * 2 * n + n + c] = dqd;
        if (c != j) {
          row[c * 2 * n + j] = dot6g(Sc, t3);
          row[c * 2 * n + n + j] = dot6g(Sc, t2);
        }
      } else if (!((m->anc[c] >> j) & 1ull)) {              // unrelated bodies: structural zeros
        row[j * 2 * n + c] = T(0);
        row[j * 2 * n + n + c] = T(0);
      }
    }
  }
  if (STAGE) {
    __shared__ T tile[64 * KCHP];
    if (c_out) flush_coalesced([&](int k) { return cb[k]; }, n, c_out + b0 * n, n, nvalid, tile, lane);
    flush_coalesced([&](int k) { return obuf[k]; }, 2 * n * n, dc + b0 * 2 * n * n, 2LL * n * n, nvalid, tile, lane);
  }
}

// in-place inverse of a symmetric positive definite 6 x 6 (Gauss-Jordan, no pivoting needed)
template <class T>
GDEV void inv6_spd(T (&A)[36]) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const T d = T(1) / A[k * 6 + k];
#pragma unroll
    for (int c = 0; c < 6; ++c) A[k * 6 + c] = (c == k) ? d : A[k * 6 + c] * d;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      if (r == k) continue;
      const T f = A[r * 6 + k];
#pragma unroll
      for (int c = 0; c < 6; ++c) A[r * 6 + c] = (c == k) ? -(f * d) : fma(-f, A[k * 6 + c], A[r * 6 + c]);
    }
  }
}

template <class T, int NMAX, bool FB>
__global__ void __launch_bounds__(64) g_minv_kernel(const DevModel<T>* __restrict__ m, const T* __restrict__ q, long long B,
                                                    int dense, T* __restrict__ Minv, int stage_rt) {
  RBDG_LANE_CONFIG();
  constexpr int OFF = FB ? 5 : 0;
  constexpr bool STAGEC = stage_outputs<NMAX>();
  const bool STAGE = STAGEC && stage_rt;
  const int n = m->n, nv = n + OFF;
  T obuf[STAGEC ? (NMAX + OFF) * (NMAX + OFF) : 1];
  T IA[NMAX][36];
  T U[NMAX][6], Di[NMAX], f1[NMAX], f2[NMAX];
  for (int i = 0; i < n; ++i) {
    if (FB && i == 0) { f1[0] = T(0); f2[0] = T(1); }
    else joint_fun(m->jtype[i], q[b * nv + i + OFF], f1[i], f2[i]);
#pragma unroll
    for (int k = 0; k < 36; ++k) IA[i][k] = m->I[i][k];      // IA = deepcopy(I)   (:662)
  }
  // articulated inertias (:697-700, :728-733); the floating base itself has no parent: IA_0 is inverted below
  for (int i = n - 1; i >= (FB ? 1 : 0); --i) {
    T A[36], u[6], S[6];
#pragma unroll
    for (int k = 0; k < 36; ++k) A[k] = IA[i][k];
#pragma unroll
    for (int r = 0; r < 6; ++r) S[r] = m->S[i][r];
    mv(A, S, u);
    const T D = dot6g(S, u);
    const T di = T(1) / D;
    Di[i] = di;
#pragma unroll
    for (int r = 0; r < 6; ++r) U[i][r] = u[r];
    const int p = m->parent[i];
    if (p >= 0) {
      T X[36];
      build_X(m, i, f1[i], f2[i], X);
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) A[r * 6 + cc] = fma(-(u[r] * di), u[cc], A[r * 6 + cc]);   // Ia = IA - U U^T / D
      T W[36];                                               // W = Ia X
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
          T acc = A[r * 6] * X[cc];
#pragma unroll
          for (int k = 1; k < 6; ++k) acc = fma(A[r * 6 + k], X[k * 6 + cc], acc);
          W[r * 6 + cc] = acc;
        }
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {                     // IA_p += X^T W
          T acc = X[r] * W[cc];
#pragma unroll
          for (int k = 1; k < 6; ++k) acc = fma(X[k * 6 + r], W[k * 6 + cc], acc);
          IA[p][r * 6 + cc] += acc;
        }
    }
  }
  T* out = STAGE ? obuf : Minv + b * nv * nv;
  T fbi[36];                                                 // floating base: inv(S^T IA_0 S) = inv(IA_0)   (:679-683)
  if (FB) {
#pragma unroll
    for (int k = 0; k < 36; ++k) fbi[k] = IA[0][k];
    inv6_spd(fbi);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int cc = 0; cc < 6; ++cc)                          // Minv[0:6, 0:6] = fb  (:685); the stored block is exactly symmetric
        out[r * nv + cc] = cc >= r ? fbi[r * 6 + cc] : (dense ? fbi[cc * 6 + r] : T(0));
  }
  T Mc[NMAX], F[NMAX][6];
  for (int c0 = FB ? 1 : 0; c0 < n; ++c0) {                  // the column of body c0 (matrix index c0 + OFF)
    for (int i = 0; i <= c0; ++i) Mc[i] = T(0);
    T F0[6];                                                 // floating base: the base rows of the column
    {                                                        // backward sweep of the column: its root path (:700-726)
      T Fv[6];
      const T m0 = Di[c0];
      Mc[c0] = m0;
#pragma unroll
      for (int r = 0; r < 6; ++r) Fv[r] = U[c0][r] * m0;
      for (int i = c0, p = m->parent[c0]; p >= 0; i = p, p = m->parent[p]) {
        T X[36], Fp[6];
        build_X(m, i, f1[i], f2[i], X);
        mtv(X, Fv, Fp);
        if (FB && p == 0) {                                  // Minv[0:6, sub] -= fb F[base slot][:, sub]   (:686-691)
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            T acc = fbi[r * 6] * Fp[0];
#pragma unroll
            for (int k = 1; k < 6; ++k) acc = fma(fbi[r * 6 + k], Fp[k], acc);
            F0[r] = -acc;
          }
        } else {
          const T mp = -(Di[p] * dot6g(m->S[p], Fp));
          Mc[p] = mp;
#pragma unroll
          for (int r = 0; r < 6; ++r) Fv[r] = fma(U[p][r], mp, Fp[r]);
        }
      }
    }
    if (FB) {                                                // F[0] = S Minv[0:6, :]   (:779)
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        F[0][r] = F0[r];
        out[r * nv + c0 + OFF] = F0[r];
        out[(c0 + OFF) * nv + r] = dense ? F0[r] : T(0);
      }
    }
    for (int i = FB ? 1 : 0; i <= c0; ++i) {                 // forward sweep, rows <= c0 (:771-781)
      const int p = m->parent[i];
      T mm = Mc[i];
      if (p >= 0) {
        T X[36], xf[6], Fp[6], Ui[6];
        build_X(m, i, f1[i], f2[i], X);
#pragma unroll
        for (int r = 0; r < 6; ++r) { Fp[r] = F[p][r]; Ui[r] = U[i][r]; }
        mv(X, Fp, xf);
        mm = fma(-Di[i], dot6g(Ui, xf), mm);
#pragma unroll
        for (int r = 0; r < 6; ++r) F[i][r] = fma(m->S[i][r], mm, xf[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 6; ++r) F[i][r] = m->S[i][r] * mm;
      }
      out[(i + OFF) * nv + c0 + OFF] = mm;
      if (i != c0) out[(c0 + OFF) * nv + i + OFF] = dense ? mm : T(0);      // (:799-804)
    }
  }
  if (STAGE) {
    __shared__ T tile[64 * KCHP];
    flush_coalesced([&](int k) { return obuf[k]; }, nv * nv, Minv + b0 * nv * nv, (long long)nv * nv, nvalid, tile, lane);
  }
}

// qdd = Minv (u - c)   (:1371-1374): one thread per (configuration, row)
template <class T>
__global__ void __launch_bounds__(256) g_fd_apply_kernel(int n, long long B, const T* __restrict__ Minv,
                                                         const T* __restrict__ u, const T* __restrict__ c,
                                                         T* __restrict__ qdd) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * n) return;
  const long long b = idx / n;
  const T* Mr = Minv + idx * n;
  T acc = T(0);
  for (int j = 0; j < n; ++j) acc = fma(Mr[j], u[b * n + j] - c[b * n + j], acc);
  qdd[idx] = acc;
}
// out = -Minv dc   (:1383-1384): one thread per output element
template <class T>
__global__ void __launch_bounds__(256) g_neg_mm_kernel(int n, long long B, const T* __restrict__ Minv,
                                                       const T* __restrict__ dc, T* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = 2LL * n * n;
  if (idx >= B * per) return;
  const long long b = idx / per;
  const int rem = (int)(idx - b * per);
  const int i = rem / (2 * n), k = rem - i * 2 * n;
  const T* Mr = Minv + (b * n + i) * n;
  const T* D = dc + b * per + k;
  T acc = T(0);
  for (int j = 0; j < n; ++j) acc = fma(Mr[j], D[j * 2 * n], acc);
  out[idx] = -acc;
}

// ---- host side ----------------------------------------------------------------------------------------------------
std::atomic<int> g_grad_kernel_option{0};
std::atomic<int> g_stage_option{0};          // 0 auto, 1 never, 2 always (rbd_g_set_output_staging)
// Staging the outputs (flush_coalesced) pays when the launch is throughput-bound: it trades partial-sector stores for a
// serial epilogue per wave.  Measured (tools/time_generic.py, MI355X): 7-body arm at B = 1 M: rnea 1 486 -> 754 us,
// minv 1 608 -> 1 139; 30-body humanoid at B = 16 384 (256 waves): rnea 138 -> 203 us, i.e. worse; at B = 65 536
// (1 024 waves) staged wins again (576 -> 391 us; 7-body arm 80 -> 64 us).  AUTO: stage from one wave per SIMD upwards
// (tools/time_generic_staging.py, profiles/r03_generic_staging_sweep.txt).
inline int stage_for(long long B) {
  const int o = g_stage_option.load(std::memory_order_relaxed);
  if (o == 1) return 0;
  if (o == 2) return 1;
  return (B + 63) / 64 >= 1024 ? 1 : 0;
}

char* err_buf() {
  static thread_local char buf[512] = "";
  return buf;
}
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  snprintf(err_buf(), 512, "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}


// =====================================================================================================================
// The per-pass surface the reference designates for accelerator testing (README.md:19), fixed base: every pass as the
// LITERAL recurrence, one configuration per lane, with the pass's own output tensors as working storage (a lane reads
// back what it wrote for the parent body: rows of different configurations never meet).  Layouts as the reference
// returns them, batch outermost: v, a, f [B, 6, n]; dv, da, df [B, 6, n, n] (element [b, r, c, i]); F [B, n, 6, n].
// These kernels exist to be compared pass by pass with the reference; they move its O(n^2) six-vectors through HBM by
// definition.
// =====================================================================================================================
template <class T, int NMAX>
__global__ void __launch_bounds__(64) g_rnea_fpass_kernel(const DevModel<T>* __restrict__ m, const T* __restrict__ q, const T* __restrict__ qd,
                                                          const T* __restrict__ qdd, T grav, long long B, T* __restrict__ v, T* __restrict__ a,
                                                          T* __restrict__ f) {
  const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int n = m->n;
  RneaState<T, NMAX> st;
  rnea_config<false, T, NMAX>(m, n, q + b * n, qd + b * n, qdd ? qdd + b * n : nullptr, grav, st, nullptr, v + b * 6 * n, a + b * 6 * n,
                              f + b * 6 * n, true);
}
// rnea_bpass (:600-621): c = S^T f, f accumulated child -> parent IN PLACE
template <class T>
__global__ void __launch_bounds__(64) g_rnea_bpass_kernel(const DevModel<T>* __restrict__ m, const T* __restrict__ q, T* __restrict__ f,
                                                          long long B, T* __restrict__ c) {
  const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int n = m->n;
  T* fb = f + b * 6 * n;
  for (int i = n - 1; i >= 0; --i) {
    T fi[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) fi[r] = fb[r * n + i];
    c[b * n + i] = dot6g(m->S[i], fi);                                                      // :612
    const int p = m->parent[i];
    if (p >= 0) {
      T f1, f2, X[36], t[6];
      joint_fun(m->jtype[i], q[b * n + i], f1, f2);
      build_X(m, i, f1, f2, X);
      mtv(X, fi, t);                                                                        // :618-619
#pragma unroll
      for (int r = 0; r < 6; ++r) fb[r * n + p] += t[r];
    }
  }
}
// rnea_grad_fpass_dq (DQ, :1127-1187) / rnea_grad_fpass_dqd (:1189-1255)
template <class T, bool DQ>
__global__ void __launch_bounds__(64) g_grad_fpass_kernel(const DevModel<T>* __restrict__ m, const T* __restrict__ q, const T* __restrict__ qd,
                                                          const T* __restrict__ v, const T* __restrict__ a, T grav, long long B,
                                                          T* __restrict__ dv, T* __restrict__ da, T* __restrict__ df) {
  const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int n = m->n;
  const T* vb = v + b * 6 * n;
  const T* ab = DQ ? a + b * 6 * n : nullptr;
  T* dvb = dv + b * 6LL * n * n;
  T* dab = da + b * 6LL * n * n;
  T* dfb = df + b * 6LL * n * n;
  for (int i = 0; i < n; ++i) {
    const int p = m->parent[i];
    T f1, f2, X[36], S[6], vi[6], Iv[6], seed_v[6], seed_a[6];
    joint_fun(m->jtype[i], q[b * n + i], f1, f2);
    build_X(m, i, f1, f2, X);
    const T qdi = qd[b * n + i];
#pragma unroll
    for (int r = 0; r < 6; ++r) { S[r] = m->S[i][r]; vi[r] = vb[r * n + i]; }
    mvI(m, i, vi, Iv);
    // what column i receives at body i: dq -- crm(X v_p) S into dv (:1159), crm(X a_p | X a_grav) S into da (:1173-1175);
    // dqd -- S into dv (:1231), crm(v_i) S into da (:1243)
    if (DQ) {
      T vp[6], ap[6], xv[6], xa[6];
      if (p >= 0) {
#pragma unroll
        for (int r = 0; r < 6; ++r) { vp[r] = vb[r * n + p]; ap[r] = ab[r * n + p]; }
        mv(X, vp, xv);
        mv(X, ap, xa);
        crm_mul(xv, S, seed_v);
      } else {
#pragma unroll
        for (int r = 0; r < 6; ++r) { xa[r] = X[r * 6 + 5] * (-grav); seed_v[r] = T(0); }
      }
      crm_mul(xa, S, seed_a);
    } else {
#pragma unroll
      for (int r = 0; r < 6; ++r) seed_v[r] = S[r];
      crm_mul(vi, S, seed_a);
    }
    for (int c = 0; c < n; ++c) {
      T dvc[6], dac[6], t[6], dfc[6];
      if (p >= 0) {
        T dvp[6], dap[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) { dvp[r] = dvb[(r * n + c) * n + p]; dap[r] = dab[(r * n + c) * n + p]; }
        mv(X, dvp, dvc);                                                                    // :1158 / :1230
        mv(X, dap, dac);                                                                    // :1163 / :1234
      } else {
#pragma unroll
        for (int r = 0; r < 6; ++r) { dvc[r] = T(0); dac[r] = T(0); }
      }
      if (c == i) {
#pragma unroll
        for (int r = 0; r < 6; ++r) dvc[r] += seed_v[r];
      }
      crm_mul(dvc, S, t);                                                                   // :1164-1170 / :1235-1240
#pragma unroll
      for (int r = 0; r < 6; ++r) dac[r] = fma(qdi, t[r], dac[r]);
      if (c == i) {
#pragma unroll
        for (int r = 0; r < 6; ++r) dac[r] += seed_a[r];
      }
      df_of(m, i, vi, Iv, dvc, dac, dfc);                                                   // :1177-1185 / :1245-1252
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        dvb[(r * n + c) * n + i] = dvc[r];
        dab[(r * n + c) * n + i] = dac[r];
        dfb[(r * n + c) * n + i] = dfc[r];
      }
    }
  }
}
// rnea_grad_bpass_dq (DQ, :1257-1297: f is the ACCUMULATED rnea force, the literal fxS term) / rnea_grad_bpass_dqd
// (:1299-1343, damping); df accumulated child -> parent IN PLACE
template <class T, bool DQ>
__global__ void __launch_bounds__(64) g_grad_bpass_kernel(const DevModel<T>* __restrict__ m, const T* __restrict__ q, const T* __restrict__ f,
                                                          T* __restrict__ df, int damp, long long B, T* __restrict__ dc) {
  const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int n = m->n;
  T* dfb = df + b * 6LL * n * n;
  T* dcb = dc + b * (long long)n * n;
  for (int i = n - 1; i >= 0; --i) {
    const int p = m->parent[i];
    T f1, f2, X[36], S[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) S[r] = m->S[i][r];
    if (p >= 0) {
      joint_fun(m->jtype[i], q[b * n + i], f1, f2);
      build_X(m, i, f1, f2, X);
    }
    for (int c = 0; c < n; ++c) {
      T dfc[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) dfc[r] = dfb[(r * n + c) * n + i];
      dcb[i * n + c] = dot6g(S, dfc) + ((!DQ && damp && c == i) ? m->damping[i] : T(0));    // :1284 / :1325, :1336-1341
      if (p >= 0) {
        T t[6];
        mtv(X, dfc, t);                                                                     // :1291 / :1331
#pragma unroll
        for (int r = 0; r < 6; ++r) dfb[(r * n + c) * n + p] += t[r];
      }
    }
    if (DQ && p >= 0) {                                                                     // :1292-1294: df[:, i, p] += X^T fxS(S, f_i), fxS = -crm(f) S
      T fi[6], w[6], t[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) fi[r] = f[b * 6 * n + r * n + i];
      crm_mul(fi, S, w);
#pragma unroll
      for (int r = 0; r < 6; ++r) w[r] = -w[r];
      mtv(X, w, t);
#pragma unroll
      for (int r = 0; r < 6; ++r) dfb[(r * n + i) * n + p] += t[r];
    }
  }
}
// minv_bpass (:630-735) -> (Minv, F, U, Dinv = D)
template <class T, int NMAX>
__global__ void __launch_bounds__(64) g_minv_bpass_kernel(const DevModel<T>* __restrict__ m, const T* __restrict__ q, long long B,
                                                          T* __restrict__ Minv, T* __restrict__ F, T* __restrict__ U, T* __restrict__ Dinv) {
  const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int n = m->n;
  T* Mb = Minv + b * (long long)n * n;
  T* Fb = F + b * 6LL * n * n;                       // F[i][r][s] = Fb[(i * 6 + r) * n + s]
  T IA[NMAX][36];
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 36; ++k) IA[i][k] = m->I[i][k];                                     // :662
    for (int s2 = 0; s2 < n; ++s2) Mb[i * n + s2] = T(0);
    for (int k = 0; k < 6 * n; ++k) Fb[i * 6 * n + k] = T(0);
  }
  for (int i = n - 1; i >= 0; --i) {
    const int p = m->parent[i];
    T A[36], S[6], u[6];
#pragma unroll
    for (int k = 0; k < 36; ++k) A[k] = IA[i][k];
#pragma unroll
    for (int r = 0; r < 6; ++r) S[r] = m->S[i][r];
    mv(A, S, u);                                                                            // :697
    const T D = dot6g(S, u);                                                                // :698
    const T Di = T(1) / D;
#pragma unroll
    for (int r = 0; r < 6; ++r) U[(b * n + i) * 6 + r] = u[r];
    Dinv[b * n + i] = D;
    Mb[i * n + i] = Di;                                                                     // :700
    T f1, f2, X[36];
    if (p >= 0) {
      joint_fun(m->jtype[i], q[b * n + i], f1, f2);
      build_X(m, i, f1, f2, X);
    }
    for (int s2 = 0; s2 < n; ++s2) {
      if (!((m->anc[s2] >> i) & 1ull)) continue;                                            // s2 in subtree(i) (incl. i)
      T Fi[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) Fi[r] = Fb[(i * 6 + r) * n + s2];
      const T ms = Mb[i * n + s2] - Di * dot6g(S, Fi);                                      // :702-708
      Mb[i * n + s2] = ms;
      if (p >= 0) {                                                                         // :720-726
        T t[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) { Fi[r] = fma(u[r], ms, Fi[r]); Fb[(i * 6 + r) * n + s2] = Fi[r]; }
        mtv(X, Fi, t);
#pragma unroll
        for (int r = 0; r < 6; ++r) Fb[(p * 6 + r) * n + s2] += t[r];
      }
    }
    if (p >= 0) {                                                                           // :728-733
      T Ia[36], XtIa[36];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) Ia[r * 6 + cc] = A[r * 6 + cc] - u[r] * (u[cc] * Di);
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
          T acc = T(0);
#pragma unroll
          for (int k = 0; k < 6; ++k) acc = fma(X[k * 6 + r], Ia[k * 6 + cc], acc);
          XtIa[r * 6 + cc] = acc;
        }
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
          T acc = T(0);
#pragma unroll
          for (int k = 0; k < 6; ++k) acc = fma(XtIa[r * 6 + k], X[k * 6 + cc], acc);
          IA[p][r * 6 + cc] += acc;
        }
    }
  }
}
// minv_fpass (:737-783): whole rows of Minv updated in place (:771), F rebuilt (:774-781); Dinv holds D
template <class T>
__global__ void __launch_bounds__(64) g_minv_fpass_kernel(const DevModel<T>* __restrict__ m, const T* __restrict__ q, long long B,
                                                          T* __restrict__ Minv, T* __restrict__ F, const T* __restrict__ U,
                                                          const T* __restrict__ Dinv) {
  const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int n = m->n;
  T* Mb = Minv + b * (long long)n * n;
  T* Fb = F + b * 6LL * n * n;
  for (int i = 0; i < n; ++i) {
    const int p = m->parent[i];
    T S[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) S[r] = m->S[i][r];
    if (p >= 0) {
      T f1, f2, X[36], u[6], UX[6];
      joint_fun(m->jtype[i], q[b * n + i], f1, f2);
      build_X(m, i, f1, f2, X);
#pragma unroll
      for (int r = 0; r < 6; ++r) u[r] = U[(b * n + i) * 6 + r];
      mtv(X, u, UX);                                                                        // U^T X
      const T Di = T(1) / Dinv[b * n + i];
      for (int c = 0; c < n; ++c) {
        T Fp[6], Fi[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) Fp[r] = Fb[(p * 6 + r) * n + c];
        const T ms = Mb[i * n + c] - Di * dot6g(UX, Fp);                                    // :771-773
        Mb[i * n + c] = ms;
        mv(X, Fp, Fi);
#pragma unroll
        for (int r = 0; r < 6; ++r) Fb[(i * 6 + r) * n + c] = fma(S[r], ms, Fi[r]);         // :774-776
      }
    } else {
      for (int c = 0; c < n; ++c) {
        const T ms = Mb[i * n + c];
#pragma unroll
        for (int r = 0; r < 6; ++r) Fb[(i * 6 + r) * n + c] = S[r] * ms;                    // :781
      }
    }
  }
}
// crba (fixed-base branch, :1091-1124): composite inertias child -> parent, then fh = IC_i S_i up the root path
template <class T, int NMAX>
__global__ void __launch_bounds__(64) g_crba_kernel(const DevModel<T>* __restrict__ m, const T* __restrict__ q, long long B, T* __restrict__ H) {
  const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int n = m->n;
  T* Hb = H + b * (long long)n * n;
  T IC[NMAX][36], f1[NMAX], f2[NMAX];
  for (int i = 0; i < n; ++i) {
    joint_fun(m->jtype[i], q[b * n + i], f1[i], f2[i]);
#pragma unroll
    for (int k = 0; k < 36; ++k) IC[i][k] = m->I[i][k];
    for (int c = 0; c < n; ++c) Hb[i * n + c] = T(0);
  }
  for (int i = n - 1; i >= 0; --i) {
    const int p = m->parent[i];
    if (p < 0) continue;
    T X[36], A[36], XtA[36];
    build_X(m, i, f1[i], f2[i], X);
#pragma unroll
    for (int k = 0; k < 36; ++k) A[k] = IC[i][k];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) acc = fma(X[k * 6 + r], A[k * 6 + cc], acc);
        XtA[r * 6 + cc] = acc;
      }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) acc = fma(XtA[r * 6 + k], X[k * 6 + cc], acc);
        IC[p][r * 6 + cc] += acc;                                                           // :1101
      }
  }
  for (int i = 0; i < n; ++i) {
    T A[36], S[6], fh[6];
#pragma unroll
    for (int k = 0; k < 36; ++k) A[k] = IC[i][k];
#pragma unroll
    for (int r = 0; r < 6; ++r) S[r] = m->S[i][r];
    mv(A, S, fh);                                                                           // :1109
    Hb[i * n + i] = dot6g(S, fh);                                                           // :1110
    int j = i;
    while (m->parent[j] >= 0) {                                                             // :1113-1121
      T X[36], t[6];
      build_X(m, j, f1[j], f2[j], X);
      mtv(X, fh, t);
#pragma unroll
      for (int r = 0; r < 6; ++r) fh[r] = t[r];
      j = m->parent[j];
      const T h = dot6g(m->S[j], fh);
      Hb[i * n + j] = h;
      Hb[j * n + i] = h;
    }
  }
}

}  // namespace rbdg

struct rbd_model {
  int device;
  int n;        // bodies
  int fb;       // floating base
  int nv;       // velocities: n, or n + 5 with a floating base
  int world_ok; // the world-frame gradient kernel serves this robot
  rbdg::DevModel<float>* d32;
  rbdg::DevModel<double>* d64;
};

namespace rbdg {

template <class T>
const DevModel<T>* dev_of(const rbd_model* m);
template <>
const DevModel<float>* dev_of<float>(const rbd_model* m) { return m->d32; }
template <>
const DevModel<double>* dev_of<double>(const rbd_model* m) { return m->d64; }

template <class T>
void fill(DevModel<T>& d, const rbd_model_desc* s) {
  std::memset(&d, 0, sizeof(d));
  d.n = s->n;
  d.fb = s->floating_base ? 1 : 0;
  for (int i = 0; i < s->n; ++i) {
    const bool base = d.fb && i == 0;                 // its joint is the 6-DoF base joint: S, X0, Xs, Xc are not read
    d.parent[i] = s->parent[i];
    d.jtype[i] = base ? 0 : s->joint_type[i];
    d.anc[i] = (1ull << i) | (s->parent[i] >= 0 ? d.anc[s->parent[i]] : 0ull);
    for (int r = 0; r < 6; ++r) d.S[i][r] = base ? T(0) : (T)s->S[i * 6 + r];
    for (int k = 0; k < 36; ++k) {
      d.X0[i][k] = base ? T(k % 7 == 0 ? 1 : 0) : (T)s->X0[i * 36 + k];
      d.Xs[i][k] = base ? T(0) : (T)s->Xs[i * 36 + k];
      d.Xc[i][k] = (!base && s->joint_type[i] == 0) ? (T)s->Xc[i * 36 + k] : T(0);
      d.I[i][k] = (T)s->I[i * 36 + k];
    }
    d.damping[i] = (T)s->damping[i];
  }
  // world-frame kernel: revolute joints with S = (axis; 0), rigid-body inertias
  bool ok = !d.fb;
  for (int i = 0; i < s->n && ok; ++i) {
    const double* I = s->I + i * 36;
    const double* S = s->S + i * 6;
    const double mass = I[3 * 6 + 3];
    double scale = 0;
    for (int k = 0; k < 36; ++k) scale = std::fmax(scale, std::fabs(I[k]));
    const double tol = 1e-12 * std::fmax(scale, 1.0);
    ok = s->joint_type[i] == 0 && S[3] == 0 && S[4] == 0 && S[5] == 0 && mass > 0;
    // [[Io, h^x], [h^x^T, m 1]] with h^x = [[0,-h2,h1],[h2,0,-h0],[-h1,h0,0]]
    const double h0 = I[2 * 6 + 4], h1 = I[0 * 6 + 5], h2 = I[1 * 6 + 3];
    const double U[9] = {0, -h2, h1, h2, 0, -h0, -h1, h0, 0};
    for (int r = 0; r < 3 && ok; ++r)
      for (int c = 0; c < 3 && ok; ++c) {
        ok = std::fabs(I[r * 6 + 3 + c] - U[r * 3 + c]) <= tol && std::fabs(I[(3 + c) * 6 + r] - U[r * 3 + c]) <= tol &&
             std::fabs(I[(3 + r) * 6 + 3 + c] - (r == c ? mass : 0.0)) <= tol && std::fabs(I[r * 6 + c] - I[c * 6 + r]) <= tol;
      }
    if (!ok) break;
    const double c0 = h0 / mass, c1 = h1 / mass, c2 = h2 / mass, cc = c0 * c0 + c1 * c1 + c2 * c2;
    const double cv[3] = {c0, c1, c2};
    d.mass[i] = (T)mass;
    for (int r = 0; r < 3; ++r) { d.com[i][r] = (T)cv[r]; d.sa[i][r] = (T)S[r]; }
    const int IR[6] = {0, 0, 0, 1, 1, 2}, IC_[6] = {0, 1, 2, 1, 2, 2};
    for (int e = 0; e < 6; ++e) {                          // Ic = Io - m (|c|^2 1 - c c^T)
      const int r = IR[e], c = IC_[e];
      d.Ic[i][e] = (T)(I[r * 6 + c] - mass * ((r == c ? cc : 0.0) - cv[r] * cv[c]));
    }
  }
  d.world_ok = ok ? 1 : 0;
}

int check_call(const rbd_model* m, long long B, const char* who) {
  if (m == nullptr) return fail(RBD_G_ERR_ARG, "%s: null model", who);
  if (B < 0) return fail(RBD_G_ERR_ARG, "%s: B = %lld", who, B);
  if (B > (1LL << 31) * 64 - 64) return fail(RBD_G_ERR_ARG, "%s: B = %lld exceeds the grid", who, B);
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, who);
  if (dev != m->device) return fail(RBD_G_ERR_ARG, "%s: the model lives on device %d, the current device is %d", who, m->device, dev);
  return 0;
}

#define RBDG_DISPATCH(m, CALL)                                        \
  do {                                                               \
    const int n_ = (m)->n;                                           \
    if ((m)->fb) {                                                   \
      if (n_ <= 8) { CALL(8, true); }                                \
      else if (n_ <= 16) { CALL(16, true); }                         \
      else if (n_ <= 32) { CALL(32, true); }                         \
      else { CALL(64, true); }                                       \
    } else {                                                         \
      if (n_ <= 8) { CALL(8, false); }                               \
      else if (n_ <= 16) { CALL(16, false); }                        \
      else if (n_ <= 32) { CALL(32, false); }                        \
      else { CALL(64, false); }                                      \
    }                                                                \
  } while (0)

template <class T>
int rnea_host(const rbd_model* m, const T* q, const T* qd, const T* qdd, T grav, long long B, T* c, T* v, T* a, T* f, void* stream) {
  if (int rc = check_call(m, B, "rbd_g_rnea")) return rc;
  if (!q || !qd || !c) return fail(RBD_G_ERR_ARG, "rbd_g_rnea: q, qd, c must not be null");
  if (B == 0) return 0;
  const unsigned grid = (unsigned)((B + 63) / 64);
#define CALL(NM, FB) hipLaunchKernelGGL((g_rnea_kernel<T, NM, FB>), dim3(grid), dim3(64), extra_lds(), (hipStream_t)stream, dev_of<T>(m), q, qd, qdd, grav, B, c, v, a, f, stage_for(B))
  RBDG_DISPATCH(m, CALL);
#undef CALL
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_g_rnea launch");
}
template <class T>
int grad_host(const rbd_model* m, const T* q, const T* qd, const T* qdd, T grav, int damp, long long B, T* c, T* dc, void* stream) {
  if (int rc = check_call(m, B, "rbd_g_rnea_grad")) return rc;
  if (!q || !qd || !dc) return fail(RBD_G_ERR_ARG, "rbd_g_rnea_grad: q, qd, dc_du must not be null");
  if (m->fb && m->n < 6)      // the reference's own pass indexes bodies 0..5 for the base and raises IndexError (:1168)
    return fail(RBD_G_ERR_UNSUPPORTED, "rbd_g_rnea_grad: a floating-base robot needs >= 6 bodies (RBDReference.py:1168 raises for fewer)");
  if (B == 0) return 0;
  const unsigned grid = (unsigned)((B + 63) / 64);
  const int opt = g_grad_kernel_option.load(std::memory_order_relaxed);
  if (m->world_ok && opt != RBD_G_GRAD_KERNEL_COLUMNS) {
    const int n_ = m->n;
#define WCALL(NM) hipLaunchKernelGGL((g_rnea_grad_world_kernel<T, NM>), dim3(grid), dim3(64), extra_lds(), (hipStream_t)stream, dev_of<T>(m), q, qd, qdd, grav, damp, B, c, dc, stage_for(B))
    if (n_ <= 8) { WCALL(8); } else if (n_ <= 16) { WCALL(16); } else if (n_ <= 32) { WCALL(32); } else { WCALL(64); }
#undef WCALL
    hipError_t ew = hipGetLastError();
    return ew == hipSuccess ? 0 : hip_fail(ew, "rbd_g_rnea_grad (world-frame kernel) launch");
  }
  if (opt == RBD_G_GRAD_KERNEL_WORLD) return fail(RBD_G_ERR_UNSUPPORTED, "rbd_g_rnea_grad: the world-frame kernel needs a fixed base, revolute joints and rigid-body inertias");
#define CALL(NM, FB) hipLaunchKernelGGL((g_rnea_grad_kernel<T, NM, FB>), dim3(grid), dim3(64), extra_lds(), (hipStream_t)stream, dev_of<T>(m), q, qd, qdd, grav, damp, B, c, dc, stage_for(B))
  RBDG_DISPATCH(m, CALL);
#undef CALL
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_g_rnea_grad launch");
}
template <class T>
int minv_host(const rbd_model* m, const T* q, long long B, int dense, T* Minv, void* stream) {
  if (int rc = check_call(m, B, "rbd_g_minv")) return rc;
  if (!q || !Minv) return fail(RBD_G_ERR_ARG, "rbd_g_minv: q, Minv must not be null");
  if (B == 0) return 0;
  const unsigned grid = (unsigned)((B + 63) / 64);
#define CALL(NM, FB) hipLaunchKernelGGL((g_minv_kernel<T, NM, FB>), dim3(grid), dim3(64), extra_lds(), (hipStream_t)stream, dev_of<T>(m), q, B, dense, Minv, stage_for(B))
  RBDG_DISPATCH(m, CALL);
#undef CALL
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_g_minv launch");
}

// ---- per-pass surface + crba (fixed base only: the reference's own crba raises for a floating base, and its floating-base
//      pass layouts are served by the per-robot libraries) --------------------------------------------------------------------
int pass_check(const rbd_model* m, long long B, const char* who) {
  if (int rc = check_call(m, B, who)) return rc;
  if (m->fb) return fail(RBD_G_ERR_UNSUPPORTED, "%s: fixed-base robots only in the model-handle library (floating-base passes: the robot's own library)", who);
  return 0;
}
#define RBDG_NM_DISPATCH(m, CALL)                 \
  do {                                           \
    const int n_ = (m)->n;                       \
    if (n_ <= 8) { CALL(8); }                    \
    else if (n_ <= 16) { CALL(16); }             \
    else if (n_ <= 32) { CALL(32); }             \
    else { CALL(64); }                           \
  } while (0)
template <class T>
int rnea_fpass_host(const rbd_model* m, const T* q, const T* qd, const T* qdd, T grav, long long B, T* v, T* a, T* f, void* stream) {
  if (int rc = pass_check(m, B, "rbd_g_rnea_fpass")) return rc;
  if (!q || !qd || !v || !a || !f) return fail(RBD_G_ERR_ARG, "rbd_g_rnea_fpass: q, qd, v, a, f must not be null");
  if (B == 0) return 0;
  const unsigned grid = (unsigned)((B + 63) / 64);
#define CALL(NM) hipLaunchKernelGGL((g_rnea_fpass_kernel<T, NM>), dim3(grid), dim3(64), 0, (hipStream_t)stream, dev_of<T>(m), q, qd, qdd, grav, B, v, a, f)
  RBDG_NM_DISPATCH(m, CALL);
#undef CALL
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_g_rnea_fpass launch");
}
template <class T>
int rnea_bpass_host(const rbd_model* m, const T* q, T* f, long long B, T* c, void* stream) {
  if (int rc = pass_check(m, B, "rbd_g_rnea_bpass")) return rc;
  if (!q || !f || !c) return fail(RBD_G_ERR_ARG, "rbd_g_rnea_bpass: q, f, c must not be null");
  if (B == 0) return 0;
  hipLaunchKernelGGL((g_rnea_bpass_kernel<T>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, dev_of<T>(m), q, f, B, c);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_g_rnea_bpass launch");
}
template <class T, bool DQ>
int grad_fpass_host(const rbd_model* m, const T* q, const T* qd, const T* v, const T* a, T grav, long long B, T* dv, T* da, T* df, void* stream) {
  const char* who = DQ ? "rbd_g_rnea_grad_fpass_dq" : "rbd_g_rnea_grad_fpass_dqd";
  if (int rc = pass_check(m, B, who)) return rc;
  if (!q || !qd || !v || (DQ && !a) || !dv || !da || !df) return fail(RBD_G_ERR_ARG, "%s: null argument", who);
  if (B == 0) return 0;
  hipLaunchKernelGGL((g_grad_fpass_kernel<T, DQ>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, dev_of<T>(m), q, qd, v, a, grav, B, dv, da, df);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_g_rnea_grad_fpass launch");
}
template <class T, bool DQ>
int grad_bpass_host(const rbd_model* m, const T* q, const T* f, T* df, int damp, long long B, T* dc, void* stream) {
  const char* who = DQ ? "rbd_g_rnea_grad_bpass_dq" : "rbd_g_rnea_grad_bpass_dqd";
  if (int rc = pass_check(m, B, who)) return rc;
  if (!q || (DQ && !f) || !df || !dc) return fail(RBD_G_ERR_ARG, "%s: null argument", who);
  if (B == 0) return 0;
  hipLaunchKernelGGL((g_grad_bpass_kernel<T, DQ>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, dev_of<T>(m), q, f, df, damp, B, dc);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_g_rnea_grad_bpass launch");
}
template <class T>
int minv_bpass_host(const rbd_model* m, const T* q, long long B, T* Minv, T* F, T* U, T* Dinv, void* stream) {
  if (int rc = pass_check(m, B, "rbd_g_minv_bpass")) return rc;
  if (!q || !Minv || !F || !U || !Dinv) return fail(RBD_G_ERR_ARG, "rbd_g_minv_bpass: null argument");
  if (B == 0) return 0;
  const unsigned grid = (unsigned)((B + 63) / 64);
#define CALL(NM) hipLaunchKernelGGL((g_minv_bpass_kernel<T, NM>), dim3(grid), dim3(64), 0, (hipStream_t)stream, dev_of<T>(m), q, B, Minv, F, U, Dinv)
  RBDG_NM_DISPATCH(m, CALL);
#undef CALL
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_g_minv_bpass launch");
}
template <class T>
int minv_fpass_host(const rbd_model* m, const T* q, long long B, T* Minv, T* F, const T* U, const T* Dinv, void* stream) {
  if (int rc = pass_check(m, B, "rbd_g_minv_fpass")) return rc;
  if (!q || !Minv || !F || !U || !Dinv) return fail(RBD_G_ERR_ARG, "rbd_g_minv_fpass: null argument");
  if (B == 0) return 0;
  hipLaunchKernelGGL((g_minv_fpass_kernel<T>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, dev_of<T>(m), q, B, Minv, F, U, Dinv);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_g_minv_fpass launch");
}
template <class T>
int crba_host(const rbd_model* m, const T* q, long long B, T* H, void* stream) {
  if (int rc = pass_check(m, B, "rbd_g_crba")) return rc;
  if (!q || !H) return fail(RBD_G_ERR_ARG, "rbd_g_crba: q, H must not be null");
  if (B == 0) return 0;
  const unsigned grid = (unsigned)((B + 63) / 64);
#define CALL(NM) hipLaunchKernelGGL((g_crba_kernel<T, NM>), dim3(grid), dim3(64), 0, (hipStream_t)stream, dev_of<T>(m), q, B, H)
  RBDG_NM_DISPATCH(m, CALL);
#undef CALL
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_g_crba launch");
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
size_t fd_ws(int n, long long B, int esz, int grad) {
  if (B <= 0) return 0;
  size_t s = align256((size_t)B * n * esz) + align256((size_t)B * n * n * esz);
  if (grad) s += align256((size_t)B * 2 * n * n * esz);
  return s;
}
template <class T>
int fd_host(const rbd_model* m, const T* q, const T* qd, const T* u, T grav, long long B, T* qdd, T* dout, bool grad,
            void* ws, size_t wsb, void* stream) {
  const char* who = grad ? "rbd_g_forward_dynamics_grad" : "rbd_g_forward_dynamics";
  if (int rc = check_call(m, B, who)) return rc;
  if (!q || !qd || !u || !qdd || (grad && !dout)) return fail(RBD_G_ERR_ARG, "%s: null argument", who);
  if (B == 0) return 0;
  const int n = m->nv;
  const size_t need = fd_ws(n, B, (int)sizeof(T), grad ? 1 : 0);
  if (!ws || wsb < need) return fail(RBD_G_ERR_WORKSPACE, "%s: workspace of %zu bytes needed, %zu given", who, need, wsb);
  char* w = (char*)ws;
  T* c = (T*)w;
  T* Mi = (T*)(w + align256((size_t)B * n * sizeof(T)));
  T* dc = (T*)((char*)Mi + align256((size_t)B * n * n * sizeof(T)));
  if (int rc = rnea_host<T>(m, q, qd, nullptr, grav, B, c, nullptr, nullptr, nullptr, stream)) return rc;   // c(q, qd)  (:1372)
  if (int rc = minv_host<T>(m, q, B, 1, Mi, stream)) return rc;                                              // (:1373)
  if ((B * 2LL * n * n + 255) / 256 > 0x7fffffffLL) return fail(RBD_G_ERR_ARG, "%s: B = %lld exceeds the grid", who, (long long)B);
  {
    const long long tot = B * n;
    hipLaunchKernelGGL((g_fd_apply_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, B, Mi, u, c, qdd);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, who);
  }
  if (!grad) return 0;
  if (int rc = grad_host<T>(m, q, qd, qdd, grav, 0, B, nullptr, dc, stream)) return rc;                      // (:1380)
  {
    const long long tot = B * 2LL * n * n;
    hipLaunchKernelGGL((g_neg_mm_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, B, Mi, dc, dout);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, who);
  }
  return 0;
}

}  // namespace rbdg

extern "C" {

int rbd_g_abi_version(void) { return RBD_G_ABI_VERSION; }
const char* rbd_g_last_error(void) { return rbdg::err_buf(); }

int rbd_model_create(const rbd_model_desc* s, int device, rbd_model** out) {
  using namespace rbdg;
  if (!s || !out) return fail(RBD_G_ERR_ARG, "rbd_model_create: null argument");
  *out = nullptr;
  if (s->abi_version != RBD_G_ABI_VERSION) return fail(RBD_G_ERR_ARG, "rbd_model_create: abi_version %d, library has %d", s->abi_version, RBD_G_ABI_VERSION);
  if (s->n < 1 || s->n > MB) return fail(RBD_G_ERR_UNSUPPORTED, "rbd_model_create: n = %d, supported 1..%d", s->n, MB);
  if (!s->parent || !s->joint_type || !s->S || !s->X0 || !s->Xs || !s->Xc || !s->I || !s->damping)
    return fail(RBD_G_ERR_ARG, "rbd_model_create: null array in the description");
  for (int i = 0; i < s->n; ++i) {
    if (s->parent[i] < -1 || s->parent[i] >= i) return fail(RBD_G_ERR_ARG, "rbd_model_create: parent[%d] = %d must be -1 or precede the body", i, s->parent[i]);
    if (s->floating_base && i > 0 && s->parent[i] < 0) return fail(RBD_G_ERR_ARG, "rbd_model_create: floating base: body %d must descend from body 0", i);
    if (s->floating_base && i == 0) {
      bool ok0 = std::isfinite(s->damping[0]);
      for (int k = 0; k < 36; ++k) ok0 = ok0 && std::isfinite(s->I[k]);
      if (!ok0) return fail(RBD_G_ERR_ARG, "rbd_model_create: body 0 has a non-finite constant");
      continue;
    }
    if (s->joint_type[i] != 0 && s->joint_type[i] != 1) return fail(RBD_G_ERR_UNSUPPORTED, "rbd_model_create: joint_type[%d] = %d", i, s->joint_type[i]);
    bool ok = std::isfinite(s->damping[i]);
    for (int r = 0; r < 6; ++r) ok = ok && std::isfinite(s->S[i * 6 + r]);
    for (int k = 0; k < 36; ++k)
      ok = ok && std::isfinite(s->X0[i * 36 + k]) && std::isfinite(s->Xs[i * 36 + k]) && std::isfinite(s->Xc[i * 36 + k]) && std::isfinite(s->I[i * 36 + k]);
    if (!ok) return fail(RBD_G_ERR_ARG, "rbd_model_create: body %d has a non-finite constant", i);
  }
  int prev = -1;
  hipError_t e = hipGetDevice(&prev);
  if (e != hipSuccess) return hip_fail(e, "rbd_model_create");
  e = hipSetDevice(device);
  if (e != hipSuccess) return hip_fail(e, "rbd_model_create: hipSetDevice");
  rbd_model* m = new (std::nothrow) rbd_model{device, s->n, s->floating_base ? 1 : 0, s->n + (s->floating_base ? 5 : 0), 0, nullptr, nullptr};
  DevModel<float>* h32 = new (std::nothrow) DevModel<float>;
  DevModel<double>* h64 = new (std::nothrow) DevModel<double>;
  int rc = 0;
  if (!m || !h32 || !h64) rc = fail(RBD_G_ERR_ARG, "rbd_model_create: out of host memory");
  if (rc == 0) {
    fill(*h32, s);
    fill(*h64, s);
    m->world_ok = h64->world_ok;
    if ((e = hipMalloc((void**)&m->d32, sizeof(*h32))) != hipSuccess || (e = hipMalloc((void**)&m->d64, sizeof(*h64))) != hipSuccess ||
        (e = hipMemcpy(m->d32, h32, sizeof(*h32), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(m->d64, h64, sizeof(*h64), hipMemcpyHostToDevice)) != hipSuccess)
      rc = hip_fail(e, "rbd_model_create: upload");
  }
  delete h32;
  delete h64;
  if (rc != 0 && m) {
    if (m->d32) (void)hipFree(m->d32);
    if (m->d64) (void)hipFree(m->d64);
    delete m;
    m = nullptr;
  }
  (void)hipSetDevice(prev);
  *out = m;
  return rc;
}
void rbd_model_destroy(rbd_model* m) {
  if (!m) return;
  int prev = -1;
  if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(m->device) == hipSuccess) {
    (void)hipFree(m->d32);
    (void)hipFree(m->d64);
    (void)hipSetDevice(prev);
  }
  delete m;
}
int rbd_model_n(const rbd_model* m) { return m ? m->n : 0; }
int rbd_model_nv(const rbd_model* m) { return m ? m->nv : 0; }
int rbd_g_set_grad_kernel(int which) {
  if (which < 0 || which > 2) return rbdg::fail(RBD_G_ERR_ARG, "rbd_g_set_grad_kernel: %d", which);
  rbdg::g_grad_kernel_option.store(which, std::memory_order_relaxed);
  return 0;
}
int rbd_g_set_output_staging(int which) {
  if (which < 0 || which > 2) return rbdg::fail(RBD_G_ERR_ARG, "rbd_g_set_output_staging: %d", which);
  rbdg::g_stage_option.store(which, std::memory_order_relaxed);
  return 0;
}
int rbd_g_grad_kernel_of(const rbd_model* m) {
  if (!m) return -1;
  const int opt = rbdg::g_grad_kernel_option.load(std::memory_order_relaxed);
  return (m->world_ok && opt != RBD_G_GRAD_KERNEL_COLUMNS) ? RBD_G_GRAD_KERNEL_WORLD : RBD_G_GRAD_KERNEL_COLUMNS;
}

int rbd_g_rnea_f32(const rbd_model* m, const float* q, const float* qd, const float* qdd, float g, int64_t B, float* c, float* v, float* a, float* f, void* st) {
  return rbdg::rnea_host<float>(m, q, qd, qdd, g, B, c, v, a, f, st);
}
int rbd_g_rnea_f64(const rbd_model* m, const double* q, const double* qd, const double* qdd, double g, int64_t B, double* c, double* v, double* a, double* f, void* st) {
  return rbdg::rnea_host<double>(m, q, qd, qdd, g, B, c, v, a, f, st);
}
int rbd_g_rnea_grad_f32(const rbd_model* m, const float* q, const float* qd, const float* qdd, float g, int damp, int64_t B, float* c, float* dc, void* st) {
  return rbdg::grad_host<float>(m, q, qd, qdd, g, damp, B, c, dc, st);
}
int rbd_g_rnea_grad_f64(const rbd_model* m, const double* q, const double* qd, const double* qdd, double g, int damp, int64_t B, double* c, double* dc, void* st) {
  return rbdg::grad_host<double>(m, q, qd, qdd, g, damp, B, c, dc, st);
}
int rbd_g_minv_f32(const rbd_model* m, const float* q, int64_t B, int dense, float* Mi, void* st) { return rbdg::minv_host<float>(m, q, B, dense, Mi, st); }
int rbd_g_minv_f64(const rbd_model* m, const double* q, int64_t B, int dense, double* Mi, void* st) { return rbdg::minv_host<double>(m, q, B, dense, Mi, st); }

size_t rbd_g_fd_workspace_bytes(const rbd_model* m, int64_t B, int elem_size, int with_grad) {
  if (!m || (elem_size != 4 && elem_size != 8)) return 0;
  return rbdg::fd_ws(m->nv, B, elem_size, with_grad);
}
int rbd_g_forward_dynamics_f32(const rbd_model* m, const float* q, const float* qd, const float* u, float g, int64_t B, float* qdd, void* ws, size_t wsb, void* st) {
  return rbdg::fd_host<float>(m, q, qd, u, g, B, qdd, nullptr, false, ws, wsb, st);
}
int rbd_g_forward_dynamics_f64(const rbd_model* m, const double* q, const double* qd, const double* u, double g, int64_t B, double* qdd, void* ws, size_t wsb, void* st) {
  return rbdg::fd_host<double>(m, q, qd, u, g, B, qdd, nullptr, false, ws, wsb, st);
}
int rbd_g_forward_dynamics_grad_f32(const rbd_model* m, const float* q, const float* qd, const float* u, float g, int64_t B, float* qdd, float* d, void* ws, size_t wsb, void* st) {
  return rbdg::fd_host<float>(m, q, qd, u, g, B, qdd, d, true, ws, wsb, st);
}
int rbd_g_forward_dynamics_grad_f64(const rbd_model* m, const double* q, const double* qd, const double* u, double g, int64_t B, double* qdd, double* d, void* ws, size_t wsb, void* st) {
  return rbdg::fd_host<double>(m, q, qd, u, g, B, qdd, d, true, ws, wsb, st);
}

#define RBDG_PASS_API(SFX, T)                                                                                                                    \
  int rbd_g_rnea_fpass_##SFX(const rbd_model* m, const T* q, const T* qd, const T* qdd, T g, int64_t B, T* v, T* a, T* f, void* st) {              \
    return rbdg::rnea_fpass_host<T>(m, q, qd, qdd, g, B, v, a, f, st);                                                                           \
  }                                                                                                                                              \
  int rbd_g_rnea_bpass_##SFX(const rbd_model* m, const T* q, T* f, int64_t B, T* c, void* st) { return rbdg::rnea_bpass_host<T>(m, q, f, B, c, st); } \
  int rbd_g_rnea_grad_fpass_dq_##SFX(const rbd_model* m, const T* q, const T* qd, const T* v, const T* a, T g, int64_t B, T* dv, T* da, T* df, void* st) { \
    return rbdg::grad_fpass_host<T, true>(m, q, qd, v, a, g, B, dv, da, df, st);                                                                 \
  }                                                                                                                                              \
  int rbd_g_rnea_grad_fpass_dqd_##SFX(const rbd_model* m, const T* q, const T* qd, const T* v, int64_t B, T* dv, T* da, T* df, void* st) {        \
    return rbdg::grad_fpass_host<T, false>(m, q, qd, v, nullptr, T(0), B, dv, da, df, st);                                                       \
  }                                                                                                                                              \
  int rbd_g_rnea_grad_bpass_dq_##SFX(const rbd_model* m, const T* q, const T* f, T* df, int64_t B, T* dc, void* st) {                             \
    return rbdg::grad_bpass_host<T, true>(m, q, f, df, 0, B, dc, st);                                                                            \
  }                                                                                                                                              \
  int rbd_g_rnea_grad_bpass_dqd_##SFX(const rbd_model* m, const T* q, T* df, int damp, int64_t B, T* dc, void* st) {                              \
    return rbdg::grad_bpass_host<T, false>(m, q, nullptr, df, damp, B, dc, st);                                                                  \
  }                                                                                                                                              \
  int rbd_g_minv_bpass_##SFX(const rbd_model* m, const T* q, int64_t B, T* Mi, T* F, T* U, T* D, void* st) { return rbdg::minv_bpass_host<T>(m, q, B, Mi, F, U, D, st); } \
  int rbd_g_minv_fpass_##SFX(const rbd_model* m, const T* q, int64_t B, T* Mi, T* F, const T* U, const T* D, void* st) {                          \
    return rbdg::minv_fpass_host<T>(m, q, B, Mi, F, U, D, st);                                                                                   \
  }                                                                                                                                              \
  int rbd_g_crba_##SFX(const rbd_model* m, const T* q, int64_t B, T* H, void* st) { return rbdg::crba_host<T>(m, q, B, H, st); }
RBDG_PASS_API(f32, float)
RBDG_PASS_API(f64, double)

}  // extern "C"
