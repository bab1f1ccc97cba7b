/* rbd_oracle.c -- CPU oracle in plain C (float64) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A second, independent restatement of the reference's rnea / rnea_grad / minv passes
 * (/root/reference/RBDReference.py:559-806, 1127-1368), written the way the reference writes them:
 * dense 6x6 products, (6, n, NB) derivative tensors, one configuration at a time.  It exists for
 * two reasons: (1) a cross-check of the numpy oracle (oracle/rbd_oracle.py) that shares no code with
 * it; (2) a compiled CPU baseline for bench.py's `cpu_baseline` leg (OpenMP over configurations,
 * every host core), since a numpy port says more about the Python interpreter than about the CPU.
 * Only tests/, __graft_entry__.smoke()/build() and bench.py may build, load or call it.
 *
 * Pinning: tests/test_c_oracle.py holds it to the golden vectors generated from the real reference
 * (tests/golden) at 1e-12 and to the numpy oracle on random inputs.
 *
 * Model (filled by oracle/c_oracle.py from the same black-box getter sampling the numpy oracle uses):
 *   X_i(q) = X0_i + Xs_i * s + Xc_i * c   with (s, c) = (sin q, cos q) for revolute joints and
 *   (q, 0) for prismatic ones;  S_i (6), I_i (6x6), parent[i], damping[i].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int n;
  const int32_t* parent;      /* [n] */
  const int32_t* prismatic;   /* [n] */
  const double* S;            /* [n][6] */
  const double* I;            /* [n][36] */
  const double* X0;           /* [n][36] */
  const double* Xs;           /* [n][36] */
  const double* Xc;           /* [n][36] */
  const double* damping;      /* [n] */
} rbdo_model;

static void xmat(const rbdo_model* m, int i, double q, double* X) {
  const double s = m->prismatic[i] ? q : sin(q), c = m->prismatic[i] ? 0.0 : cos(q);
  for (int k = 0; k < 36; ++k) X[k] = m->X0[i * 36 + k] + m->Xs[i * 36 + k] * s + m->Xc[i * 36 + k] * c;
}
static void mv(const double* A, const double* x, double* y) { /* y = A x (6x6) */
  for (int r = 0; r < 6; ++r) {
    double a = 0;
    for (int c = 0; c < 6; ++c) a += A[r * 6 + c] * x[c];
    y[r] = a;
  }
}
static void mtv(const double* A, const double* x, double* y) { /* y = A^T x */
  for (int c = 0; c < 6; ++c) {
    double a = 0;
    for (int r = 0; r < 6; ++r) a += A[r * 6 + c] * x[r];
    y[c] = a;
  }
}
/* cross_operator (RBDReference.py:9-21) applied: y = crm(v) x */
static void crm_mul(const double* v, const double* x, double* y) {
  y[0] = -v[2] * x[1] + v[1] * x[2];
  y[1] = v[2] * x[0] - v[0] * x[2];
  y[2] = -v[1] * x[0] + v[0] * x[1];
  y[3] = -v[5] * x[1] + v[4] * x[2] - v[2] * x[4] + v[1] * x[5];
  y[4] = v[5] * x[0] - v[3] * x[2] + v[2] * x[3] - v[0] * x[5];
  y[5] = -v[4] * x[0] + v[3] * x[1] - v[1] * x[3] + v[0] * x[4];
}
/* fxv (RBDReference.py:149-164): y = crf(a) b */
static void fxv(const double* a, const double* b, double* y) {
  y[0] = -a[2] * b[1] + a[1] * b[2] - a[5] * b[4] + a[4] * b[5];
  y[1] = a[2] * b[0] - a[0] * b[2] + a[5] * b[3] - a[3] * b[5];
  y[2] = -a[1] * b[0] + a[0] * b[1] - a[4] * b[3] + a[3] * b[4];
  y[3] = -a[2] * b[4] + a[1] * b[5];
  y[4] = a[2] * b[3] - a[0] * b[5];
  y[5] = -a[1] * b[3] + a[0] * b[4];
}

/* rnea (RBDReference.py:559-628): v, a, f are [6][n] column-per-body like the reference's (6, NB);
 * f comes back ACCUMULATED (:619).  X (n x 36) is filled for reuse by the callers. */
static void rnea_one(const rbdo_model* m, const double* q, const double* qd, const double* qdd, double g,
                     double* c, double* v, double* a, double* f, double* X) {
  const int n = m->n;
  double gv[6] = {0, 0, 0, 0, 0, -g};
  for (int i = 0; i < n; ++i) {
    double* Xi = X + i * 36;
    xmat(m, i, q[i], Xi);
    const int p = m->parent[i];
    double vi[6] = {0}, ai[6], t[6], vp[6], ap[6];
    if (p == -1) {
      mv(Xi, gv, ai);                                                        /* :578 */
    } else {
      for (int r = 0; r < 6; ++r) { vp[r] = v[r * n + p]; ap[r] = a[r * n + p]; }
      mv(Xi, vp, vi);                                                        /* :580 */
      mv(Xi, ap, ai);                                                        /* :581 */
    }
    double vJ[6];
    for (int r = 0; r < 6; ++r) { vJ[r] = m->S[i * 6 + r] * qd[i]; vi[r] += vJ[r]; }   /* :586-587 */
    crm_mul(vi, vJ, t);                                                      /* :588 mxS(vJ, v_i) */
    for (int r = 0; r < 6; ++r) ai[r] += t[r] + (qdd ? m->S[i * 6 + r] * qdd[i] : 0.0);  /* :589-593 */
    double Iv[6], Ia[6], w[6];
    mv(m->I + i * 36, vi, Iv);
    mv(m->I + i * 36, ai, Ia);
    fxv(vi, Iv, w);                                                          /* :596 vxIv */
    for (int r = 0; r < 6; ++r) { v[r * n + i] = vi[r]; a[r * n + i] = ai[r]; f[r * n + i] = Ia[r] + w[r]; }
  }
  for (int i = n - 1; i >= 0; --i) {                                         /* :607-619 */
    const int p = m->parent[i];
    double fi[6], t[6], ci = 0;
    for (int r = 0; r < 6; ++r) { fi[r] = f[r * n + i]; ci += m->S[i * 6 + r] * fi[r]; }
    c[i] = ci;
    if (p != -1) {
      mtv(X + i * 36, fi, t);
      for (int r = 0; r < 6; ++r) f[r * n + p] += t[r];
    }
  }
}

/* rnea_grad (RBDReference.py:1127-1368).  Scratch: dv, da, df are [6][n][n] (component, column, body). */
static void rnea_grad_one(const rbdo_model* m, const double* q, const double* qd, const double* qdd, double g,
                          int use_damping, double* c, double* dc_du, double* scratch) {
  const int n = m->n;
  double* v = scratch;            /* 6n */
  double* a = v + 6 * n;
  double* f = a + 6 * n;
  double* X = f + 6 * n;          /* 36n */
  double* dv = X + 36 * n;        /* 6 n n */
  double* da = dv + 6 * n * n;
  double* df = da + 6 * n * n;
  rnea_one(m, q, qd, qdd, g, c, v, a, f, X);                                 /* :1353 */
  double gv[6] = {0, 0, 0, 0, 0, -g};
#define T3(P, r, col, body) (P)[((r) * n + (col)) * n + (body)]
  for (int variant = 0; variant < 2; ++variant) {        /* 0: d/dq (:1127-1187), 1: d/dqd (:1189-1255) */
    memset(dv, 0, sizeof(double) * 6 * n * n);
    memset(da, 0, sizeof(double) * 6 * n * n);
    memset(df, 0, sizeof(double) * 6 * n * n);
    for (int i = 0; i < n; ++i) {
      const int p = m->parent[i];
      const double* Xi = X + i * 36;
      const double* S = m->S + i * 6;
      double vi[6], vp[6], ap[6], x[6], y[6], t[6];
      for (int r = 0; r < 6; ++r) vi[r] = v[r * n + i];
      if (p != -1) {
        for (int col = 0; col < n; ++col) {                                  /* :1158,:1163 / :1230,:1234 */
          for (int r = 0; r < 6; ++r) x[r] = T3(dv, r, col, p);
          mv(Xi, x, y);
          for (int r = 0; r < 6; ++r) T3(dv, r, col, i) = y[r];
          for (int r = 0; r < 6; ++r) x[r] = T3(da, r, col, p);
          mv(Xi, x, y);
          for (int r = 0; r < 6; ++r) T3(da, r, col, i) = y[r];
        }
      }
      if (variant == 0) {
        if (p != -1) {                                                       /* :1159 */
          for (int r = 0; r < 6; ++r) vp[r] = v[r * n + p];
          mv(Xi, vp, x);
          crm_mul(x, S, t);
          for (int r = 0; r < 6; ++r) T3(dv, r, i, i) += t[r];
        }
      } else {
        for (int r = 0; r < 6; ++r) T3(dv, r, i, i) += S[r];                 /* :1231 */
      }
      for (int col = 0; col < n; ++col) {                                    /* :1164-1170 / :1235-1240 */
        for (int r = 0; r < 6; ++r) x[r] = T3(dv, r, col, i);
        crm_mul(x, S, t);
        for (int r = 0; r < 6; ++r) T3(da, r, col, i) += t[r] * qd[i];
      }
      if (variant == 0) {
        if (p != -1) { for (int r = 0; r < 6; ++r) ap[r] = a[r * n + p]; mv(Xi, ap, x); }   /* :1173 */
        else mv(Xi, gv, x);                                                                  /* :1175 */
        crm_mul(x, S, t);
      } else {
        crm_mul(vi, S, t);                                                   /* :1243 */
      }
      for (int r = 0; r < 6; ++r) T3(da, r, i, i) += t[r];
      double Iv[6];
      mv(m->I + i * 36, vi, Iv);
      for (int col = 0; col < n; ++col) {                                    /* :1179-1185 / :1247-1252 */
        double dvc[6], dac[6], Ida[6], Idv[6], t1[6], t2[6];
        for (int r = 0; r < 6; ++r) { dvc[r] = T3(dv, r, col, i); dac[r] = T3(da, r, col, i); }
        mv(m->I + i * 36, dac, Ida);
        mv(m->I + i * 36, dvc, Idv);
        fxv(dvc, Iv, t1);
        fxv(vi, Idv, t2);
        for (int r = 0; r < 6; ++r) T3(df, r, col, i) = Ida[r] + t1[r] + t2[r];
      }
    }
    /* backward pass (:1257-1297 / :1299-1343) */
    for (int i = n - 1; i >= 0; --i) {
      const int p = m->parent[i];
      const double* S = m->S + i * 6;
      for (int col = 0; col < n; ++col) {
        double s = 0;
        for (int r = 0; r < 6; ++r) s += S[r] * T3(df, r, col, i);
        dc_du[i * 2 * n + variant * n + col] = s;                            /* :1284 / :1325, hstack :1367 */
      }
      if (p != -1) {
        const double* Xi = X + i * 36;
        double x[6], y[6];
        for (int col = 0; col < n; ++col) {                                  /* :1291 / :1331 */
          for (int r = 0; r < 6; ++r) x[r] = T3(df, r, col, i);
          mtv(Xi, x, y);
          for (int r = 0; r < 6; ++r) T3(df, r, col, p) += y[r];
        }
        if (variant == 0) {                                                  /* :1292-1294: X^T fxS(S, f_i) */
          double fi[6], t[6];
          for (int r = 0; r < 6; ++r) fi[r] = f[r * n + i];
          crm_mul(fi, S, t);
          for (int r = 0; r < 6; ++r) t[r] = -t[r];
          mtv(Xi, t, y);
          for (int r = 0; r < 6; ++r) T3(df, r, i, p) += y[r];
        }
      }
    }
    if (variant == 1 && use_damping)
      for (int i = 0; i < n; ++i) dc_du[i * 2 * n + n + i] += m->damping[i];   /* :1336-1341 */
  }
#undef T3
}

/* minv (RBDReference.py:630-806), dense output.  Scratch: IA [n][36], F [n][6][n], U [n][6], D [n], X [n][36]. */
static void minv_one(const rbdo_model* m, const double* q, int dense, double* Minv, double* scratch) {
  const int n = m->n;
  double* IA = scratch;
  double* F = IA + 36 * n;
  double* U = F + 6 * n * n;
  double* D = U + 6 * n;
  double* X = D + n;
  memcpy(IA, m->I, sizeof(double) * 36 * n);                                 /* :662 */
  memset(F, 0, sizeof(double) * 6 * n * n);
  memset(Minv, 0, sizeof(double) * n * n);
  for (int i = 0; i < n; ++i) xmat(m, i, q[i], X + i * 36);
#define FF(b, r, col) F[((b) * 6 + (r)) * n + (col)]
  for (int i = n - 1; i >= 0; --i) {
    const int p = m->parent[i];
    const double* S = m->S + i * 6;
    mv(IA + i * 36, S, U + i * 6);                                           /* :697 */
    double d = 0;
    for (int r = 0; r < 6; ++r) d += S[r] * U[i * 6 + r];
    D[i] = d;                                                                /* :698 */
    Minv[i * n + i] = 1.0 / d;                                               /* :700 */
    for (int j = i; j < n; ++j) {                                            /* subtree(i) subset of j >= i */
      int k = j, insub = 0;
      while (k != -1) { if (k == i) { insub = 1; break; } k = m->parent[k]; }
      if (!insub) continue;
      double s = 0;
      for (int r = 0; r < 6; ++r) s += S[r] * FF(i, r, j);
      Minv[i * n + j] -= s / d;                                              /* :702-708 */
      if (p != -1) {
        double fi[6], y[6];
        for (int r = 0; r < 6; ++r) { FF(i, r, j) += U[i * 6 + r] * Minv[i * n + j]; fi[r] = FF(i, r, j); }   /* :721-723 */
        mtv(X + i * 36, fi, y);
        for (int r = 0; r < 6; ++r) FF(p, r, j) += y[r];                     /* :724-726 */
      }
    }
    if (p != -1) {                                                           /* :728-733 */
      double Ia[36], A[36];
      for (int r = 0; r < 6; ++r)
        for (int cc = 0; cc < 6; ++cc) Ia[r * 6 + cc] = IA[i * 36 + r * 6 + cc] - U[i * 6 + r] * U[i * 6 + cc] / d;
      const double* Xi = X + i * 36;
      for (int r = 0; r < 6; ++r)          /* A = Ia X */
        for (int cc = 0; cc < 6; ++cc) {
          double s = 0;
          for (int k = 0; k < 6; ++k) s += Ia[r * 6 + k] * Xi[k * 6 + cc];
          A[r * 6 + cc] = s;
        }
      for (int r = 0; r < 6; ++r)          /* IA_p += X^T A */
        for (int cc = 0; cc < 6; ++cc) {
          double s = 0;
          for (int k = 0; k < 6; ++k) s += Xi[k * 6 + r] * A[k * 6 + cc];
          IA[p * 36 + r * 6 + cc] += s;
        }
    }
  }
  for (int i = 0; i < n; ++i) {                                              /* :760-781 */
    const int p = m->parent[i];
    const double* S = m->S + i * 6;
    if (p != -1) {
      const double* Xi = X + i * 36;
      double UX[6];
      mtv(Xi, U + i * 6, UX);                                                /* U^T X */
      for (int col = 0; col < n; ++col) {
        double s = 0, fp[6], y[6];
        for (int r = 0; r < 6; ++r) { fp[r] = FF(p, r, col); s += UX[r] * fp[r]; }
        Minv[i * n + col] -= s / D[i];                                       /* :771-773 */
        mv(Xi, fp, y);
        for (int r = 0; r < 6; ++r) FF(i, r, col) = y[r] + S[r] * Minv[i * n + col];   /* :774-776 */
      }
    } else {
      for (int col = 0; col < n; ++col)
        for (int r = 0; r < 6; ++r) FF(i, r, col) = S[r] * Minv[i * n + col];          /* :781 */
    }
  }
#undef FF
  if (dense)                                                                 /* :799-804 */
    for (int r = 0; r < n; ++r)
      for (int cc = 0; cc < r; ++cc) Minv[r * n + cc] = Minv[cc * n + r];
}

/* ---- batched entry points (ctypes) ------------------------------------------------------------ */
int rbdo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void rbdo_rnea_grad(const rbdo_model* m, const double* q, const double* qd, const double* qdd, double g,
                    int use_damping, int64_t B, double* c, double* dc_du, int threads) {
  const int n = m->n;
  const size_t ns = (size_t)(18 * n + 36 * n + 18 * n * n);
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
  {
    double* scratch = (double*)malloc(sizeof(double) * ns);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t b = 0; b < B; ++b)
      rnea_grad_one(m, q + b * n, qd + b * n, qdd ? qdd + b * n : NULL, g, use_damping, c + b * n,
                    dc_du + b * 2 * n * n, scratch);
    free(scratch);
  }
}

void rbdo_rnea(const rbdo_model* m, const double* q, const double* qd, const double* qdd, double g, int64_t B,
               double* c, double* v, double* a, double* f, int threads) {
  const int n = m->n;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
  {
    double* X = (double*)malloc(sizeof(double) * 36 * n);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t b = 0; b < B; ++b)
      rnea_one(m, q + b * n, qd + b * n, qdd ? qdd + b * n : NULL, g, c + b * n, v + b * 6 * n, a + b * 6 * n,
               f + b * 6 * n, X);
    free(X);
  }
}

void rbdo_minv(const rbdo_model* m, const double* q, int dense, int64_t B, double* Minv, int threads) {
  const int n = m->n;
  const size_t ns = (size_t)(36 * n + 6 * n * n + 6 * n + n + 36 * n);
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
  {
    double* scratch = (double*)malloc(sizeof(double) * ns);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t b = 0; b < B; ++b) minv_one(m, q + b * n, dense, Minv + b * n * n, scratch);
    free(scratch);
  }
}
