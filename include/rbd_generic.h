/* rbd_generic.h -- C-ABI of the MODEL-HANDLE library (librbd_generic.so): the same batched rnea / rnea_grad /
 * minv / forward_dynamics(_grad) as include/rbd_hip.h, for ANY robot, with the model passed at run time.
 *
 * Why it exists.  The per-robot libraries of rbd_hip.h are compiled against the robot (tens of seconds for an arm,
 * minutes for a humanoid, and they need hipcc on the machine).  This library is compiled ONCE, needs no compiler on
 * the user's machine, and serves a new robot the moment rbd_model_create() returns: rbdreference_amd/api.py routes
 * calls here until the specialised library of the robot is ready, and stays here if it never will be.  It is the
 * interface SURVEY.md §8b sketches (`rbd_model_create(desc)` + entry points taking the handle).
 *
 * What it computes.  The reference's own recursions, literally, on dense 6x6 operands in body coordinates:
 *   rbd_g_rnea       /root/reference/RBDReference.py:559-628   (rnea_fpass, rnea_bpass, rnea)
 *   rbd_g_rnea_grad  :1127-1368  (rnea_grad_fpass_dq/dqd, rnea_grad_bpass_dq/dqd incl. the literal fxS term, rnea_grad)
 *   rbd_g_minv       :630-806    (minv_bpass, minv_fpass, minv; fixed- and floating-base branches)
 *   rbd_g_forward_dynamics(_grad)  :1371-1384
 * Joints: X_i(q) = X0_i + Xs_i f1(q) + Xc_i f2(q) with (f1, f2) = (sin q, cos q) for joint_type 0 (revolute, any
 * axis) and (q, 0) for joint_type 1 (prismatic) -- SURVEY.md Appendix A: three samples of the robot's Xmat closure
 * recover the three matrices exactly; S_i is any 6-vector.  1-DoF joints, n <= RBD_G_MAX_BODIES, fixed base or ONE
 * floating base at body 0 (rbd_model_desc.floating_base; rbd_g_rnea_grad then needs n >= 6, as the reference does).
 * One configuration per lane, per-lane state in private memory: correct and general, several times slower than the
 * specialised kernels (DESIGN.md §3.8 has the measured ratio) -- a first-use path, not the headline path.
 *
 * Conventions: as rbd_hip.h (device pointers, dense row-major, batch outermost, caller owns every buffer, `stream`
 * is a hipStream_t as void*, 0 = OK, <0 = RBD_G_ERR_*, >0 = hipError_t; rbd_g_last_error() is thread-local; no C++
 * exception crosses the boundary).  The handle is immutable after creation: concurrent calls are safe.  Calls run on
 * the device the model was created on (the calling thread's current device must be that device).
 */
#ifndef RBD_GENERIC_H
#define RBD_GENERIC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBD_G_MAX_BODIES 64
#define RBD_G_ABI_VERSION 1

#define RBD_G_ERR_ARG (-1)
#define RBD_G_ERR_UNSUPPORTED (-2)
#define RBD_G_ERR_WORKSPACE (-3)

typedef struct rbd_model rbd_model;

/* Host arrays, float64, read once by rbd_model_create (nothing is retained). */
typedef struct rbd_model_desc {
  int32_t abi_version;        /* RBD_G_ABI_VERSION                                                        */
  int32_t n;                  /* bodies == joints == velocities                                           */
  const int32_t* parent;      /* [n]   -1 = child of the fixed base; parent[i] < i (RBDReference.py:569)   */
  const int32_t* joint_type;  /* [n]   0: X = X0 + Xs sin q + Xc cos q;  1: X = X0 + Xs q                  */
  const double* S;            /* [n,6] motion subspace (get_S_by_id)                                      */
  const double* X0;           /* [n,36] row-major 6x6                                                     */
  const double* Xs;           /* [n,36]                                                                   */
  const double* Xc;           /* [n,36] (ignored for joint_type 1)                                        */
  const double* I;            /* [n,36] spatial inertia (get_Imat_by_id)                                  */
  const double* damping;      /* [n]   (get_damping_by_id)                                                */
  int32_t floating_base;      /* 1: body 0 is attached by the 6-DoF base joint (S = eye(6), q[0:6] = px py pz rx ry rz,
                                 X_0 = plux(Rz Ry Rx, p); RBDReference.py:585-593, :634-637, :652-691): its entries of
                                 joint_type / S / X0 / Xs / Xc are not read, every other body must descend from it, and
                                 q, qd, qdd, c, u have nv = n + 5 columns (body i >= 1 owns index i + 5), dc_du is
                                 [B, nv, 2 nv], Minv [B, nv, nv]; v, a, f stay [B, 6, n]                               */
} rbd_model_desc;

int rbd_g_abi_version(void);
const char* rbd_g_last_error(void);

/* Validates the description, uploads an fp32 and an fp64 copy to `device`, returns the handle in *out. */
int rbd_model_create(const rbd_model_desc* desc, int device, rbd_model** out);
void rbd_model_destroy(rbd_model* m);
int rbd_model_n(const rbd_model* m);   /* bodies */
int rbd_model_nv(const rbd_model* m);  /* velocities: n, or n + 5 with a floating base */

/* Replaces RBDReference.rnea (RBDReference.py:623): c [B,n]; v, a, f [B,6,n] (nullable, f accumulated);
 * qdd NULL = the reference's qdd=None. */
int rbd_g_rnea_f32(const rbd_model*, const float* q, const float* qd, const float* qdd, float gravity, int64_t B,
                   float* c, float* v, float* a, float* f, void* stream);
int rbd_g_rnea_f64(const rbd_model*, const double* q, const double* qd, const double* qdd, double gravity, int64_t B,
                   double* c, double* v, double* a, double* f, void* stream);

/* Replaces RBDReference.rnea_grad (:1345): dc_du [B,n,2n] = [dc_dq | dc_dqd]; c [B,n] nullable. */
int rbd_g_rnea_grad_f32(const rbd_model*, const float* q, const float* qd, const float* qdd, float gravity,
                        int use_damping, int64_t B, float* c, float* dc_du, void* stream);
int rbd_g_rnea_grad_f64(const rbd_model*, const double* q, const double* qd, const double* qdd, double gravity,
                        int use_damping, int64_t B, double* c, double* dc_du, void* stream);

/* rbd_g_rnea_grad has two kernels that compute the same result: COLUMNS, the reference's column recursions (any robot),
 * and WORLD, the world-frame identities the specialised libraries use (fixed base, revolute joints with S = (axis; 0),
 * rigid-body inertias: 3-4x faster).  AUTO picks WORLD where it applies.  Process-wide, thread-safe (tests run both). */
#define RBD_G_GRAD_KERNEL_AUTO 0
#define RBD_G_GRAD_KERNEL_COLUMNS 1
#define RBD_G_GRAD_KERNEL_WORLD 2
int rbd_g_set_grad_kernel(int which);
int rbd_g_grad_kernel_of(const rbd_model* m);   /* the kernel rbd_g_rnea_grad would run for this model now */
/* How results leave the kernels: 0 AUTO (staged through private memory and an LDS tile into 256-byte runs per
 * configuration when the launch has >= 1 024 waves -- one per SIMD --, direct 4-byte stores below that), 1 never staged, 2 always
 * (robots of <= 32 bodies).  The values are the same either way. */
int rbd_g_set_output_staging(int which);

/* Replaces RBDReference.minv (:785): Minv [B,n,n]; output_dense = 0 leaves a zero strict lower triangle. */
int rbd_g_minv_f32(const rbd_model*, const float* q, int64_t B, int output_dense, float* Minv, void* stream);
int rbd_g_minv_f64(const rbd_model*, const double* q, int64_t B, int output_dense, double* Minv, void* stream);

/* Replaces RBDReference.forward_dynamics / forward_dynamics_grad (:1371-1384): qdd [B,n] = Minv (u - c);
 * dqdd_du [B,n,2n] = -Minv [dc_dq | dc_dqd] at that qdd.  `workspace`: rbd_g_fd_workspace_bytes(m, B, elem, grad). */
size_t rbd_g_fd_workspace_bytes(const rbd_model*, int64_t B, int elem_size, int with_grad);
int rbd_g_forward_dynamics_f32(const rbd_model*, const float* q, const float* qd, const float* u, float gravity, int64_t B,
                               float* qdd, void* workspace, size_t workspace_bytes, void* stream);
int rbd_g_forward_dynamics_f64(const rbd_model*, const double* q, const double* qd, const double* u, double gravity, int64_t B,
                               double* qdd, void* workspace, size_t workspace_bytes, void* stream);
int rbd_g_forward_dynamics_grad_f32(const rbd_model*, const float* q, const float* qd, const float* u, float gravity, int64_t B,
                                    float* qdd, float* dqdd_du, void* workspace, size_t workspace_bytes, void* stream);
int rbd_g_forward_dynamics_grad_f64(const rbd_model*, const double* q, const double* qd, const double* u, double gravity, int64_t B,
                                    double* qdd, double* dqdd_du, void* workspace, size_t workspace_bytes, void* stream);

/* The per-pass surface the reference designates for accelerator testing (README.md:19) and crba, for FIXED-base models
 * (a floating-base model is refused with RBD_G_ERR_UNSUPPORTED: the reference's own crba raises for one, RBDReference.py:1063,
 * and its floating-base pass layouts are served by the robot's own library, include/rbd_hip.h).  Arguments, layouts and
 * in-place behaviour are those of the same-named entry points of include/rbd_hip.h -- v, a, f [B, 6, n]; dv, da, df
 * [B, 6, n, n] (element [b, r, c, i]); F [B, n, 6, n]; Dinv holds D -- with the model handle in front:
 *   rnea_fpass (RBDReference.py:559-598, local f)        rnea_bpass (:600-621, f accumulated IN PLACE)
 *   rnea_grad_fpass_dq (:1127-1187)                      rnea_grad_fpass_dqd (:1189-1255)
 *   rnea_grad_bpass_dq (:1257-1297, df IN PLACE, literal fxS term)   rnea_grad_bpass_dqd (:1299-1343, damping)
 *   minv_bpass (:630-735)                                minv_fpass (:737-783, whole rows of Minv IN PLACE, F rebuilt)
 *   crba (:1091-1124)
 * One configuration per lane, the literal recurrences with the output tensors as working storage: they exist so that an
 * accelerator port can be compared pass by pass against the reference on a machine WITHOUT a compiler. */
int rbd_g_rnea_fpass_f32(const rbd_model*, const float* q, const float* qd, const float* qdd, float gravity, int64_t B, float* v, float* a, float* f, void* stream);
int rbd_g_rnea_bpass_f32(const rbd_model*, const float* q, float* f, int64_t B, float* c, void* stream);
int rbd_g_rnea_grad_fpass_dq_f32(const rbd_model*, const float* q, const float* qd, const float* v, const float* a, float gravity, int64_t B,
                                  float* dv_dq, float* da_dq, float* df_dq, void* stream);
int rbd_g_rnea_grad_fpass_dqd_f32(const rbd_model*, const float* q, const float* qd, const float* v, int64_t B, float* dv_dqd, float* da_dqd,
                                   float* df_dqd, void* stream);
int rbd_g_rnea_grad_bpass_dq_f32(const rbd_model*, const float* q, const float* f, float* df_dq, int64_t B, float* dc_dq, void* stream);
int rbd_g_rnea_grad_bpass_dqd_f32(const rbd_model*, const float* q, float* df_dqd, int use_damping, int64_t B, float* dc_dqd, void* stream);
int rbd_g_minv_bpass_f32(const rbd_model*, const float* q, int64_t B, float* Minv, float* F, float* U, float* Dinv, void* stream);
int rbd_g_minv_fpass_f32(const rbd_model*, const float* q, int64_t B, float* Minv, float* F, const float* U, const float* Dinv, void* stream);
int rbd_g_crba_f32(const rbd_model*, const float* q, int64_t B, float* H, void* stream);
int rbd_g_rnea_fpass_f64(const rbd_model*, const double* q, const double* qd, const double* qdd, double gravity, int64_t B, double* v, double* a, double* f, void* stream);
int rbd_g_rnea_bpass_f64(const rbd_model*, const double* q, double* f, int64_t B, double* c, void* stream);
int rbd_g_rnea_grad_fpass_dq_f64(const rbd_model*, const double* q, const double* qd, const double* v, const double* a, double gravity, int64_t B,
                                  double* dv_dq, double* da_dq, double* df_dq, void* stream);
int rbd_g_rnea_grad_fpass_dqd_f64(const rbd_model*, const double* q, const double* qd, const double* v, int64_t B, double* dv_dqd, double* da_dqd,
                                   double* df_dqd, void* stream);
int rbd_g_rnea_grad_bpass_dq_f64(const rbd_model*, const double* q, const double* f, double* df_dq, int64_t B, double* dc_dq, void* stream);
int rbd_g_rnea_grad_bpass_dqd_f64(const rbd_model*, const double* q, double* df_dqd, int use_damping, int64_t B, double* dc_dqd, void* stream);
int rbd_g_minv_bpass_f64(const rbd_model*, const double* q, int64_t B, double* Minv, double* F, double* U, double* Dinv, void* stream);
int rbd_g_minv_fpass_f64(const rbd_model*, const double* q, int64_t B, double* Minv, double* F, const double* U, const double* Dinv, void* stream);
int rbd_g_crba_f64(const rbd_model*, const double* q, int64_t B, double* H, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RBD_GENERIC_H */
