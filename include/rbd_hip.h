/* rbd_hip.h -- C-ABI of the MI355X batched rigid-body-dynamics back-end.
 *
 * One shared library is built PER ROBOT (librbd_<name>_<hash>.so): topology, joint axes, tree
 * transforms and inertias are compile-time constants of its kernels, so the entry points take no
 * model handle.  The reference offers no FFI of its own; the boundary it does offer is the Python
 * class  RBDReference(robot).rnea / .rnea_grad / .minv  (/root/reference/RBDReference.py:623,
 * :1345, :785; README.md:15-17).  Each entry point below replaces one of those methods for a whole
 * batch of configurations; rbdreference_amd/api.py binds them with ctypes (INTEGRATION.md shows the
 * stub a maintainer of the reference would add).
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer to a dense row-major array, batch index outermost;
 *     the caller owns all buffers; nothing is allocated per call;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); launches are asynchronous;
 *   - return 0 on success, <0 for argument errors (RBD_ERR_*), >0 = hipError_t of a failed launch;
 *     rbd_last_error() returns a thread-local message for the last non-zero return;
 *   - no C++ exceptions cross this boundary; the model is immutable, so concurrent calls from
 *     several host threads / streams are safe.
 */
#ifndef RBD_HIP_H
#define RBD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBD_MAX_BODIES 64

#define RBD_ERR_ARG (-1)          /* null / inconsistent arguments                         */
#define RBD_ERR_UNSUPPORTED (-2)  /* robot too large for this kernel's on-chip working set  */
#define RBD_ERR_WORKSPACE (-3)    /* workspace missing or too small                        */

typedef struct rbd_model_info {
  int32_t abi_version;
  int32_t n;                          /* bodies == joints == velocities (fixed base, 1-DoF)  */
  int32_t max_depth;                  /* longest root path, in bodies                        */
  uint64_t hash;                      /* first 64 bits of sha256 over the packed model       */
  char name[64];
  int32_t parent[RBD_MAX_BODIES];     /* -1 = child of the fixed base                        */
  int32_t joint_type[RBD_MAX_BODIES]; /* 0 revolute, 1 prismatic                             */
  int32_t joint_axis[RBD_MAX_BODIES]; /* 0/1/2 = x/y/z of the body frame                     */
} rbd_model_info_t;

int rbd_abi_version(void);
const char* rbd_last_error(void);
/* The robot this library was compiled for (host-side, no GPU needed). */
int rbd_model_info(rbd_model_info_t* out);

/* RBDReference.rnea(q, qd, qdd=None, GRAVITY)            (RBDReference.py:623-628)
 *   q, qd, qdd : [B, n]   (qdd may be NULL == the reference's qdd=None, :589)
 *   c          : [B, n]
 *   v, a, f    : [B, 6, n] or all three NULL; f is the ACCUMULATED force the reference returns
 *                (its backward pass adds child forces in place, :619).                        */
int rbd_rnea_f32(const float* q, const float* qd, const float* qdd, float gravity, int64_t B,
                 float* c, float* v, float* a, float* f, void* stream);
int rbd_rnea_f64(const double* q, const double* qd, const double* qdd, double gravity, int64_t B,
                 double* c, double* v, double* a, double* f, void* stream);

/* Per-pass surface the reference designates for accelerator testing (README.md:19):
 * RBDReference.rnea_fpass(q, qd, qdd=None, GRAVITY) -> (v, a, f) with f LOCAL  (RBDReference.py:559-598)
 * RBDReference.rnea_bpass(q, f) -> (c, f): accumulates child forces into f IN PLACE (RBDReference.py:600-621) */
int rbd_rnea_fpass_f32(const float* q, const float* qd, const float* qdd, float gravity, int64_t B,
                       float* v, float* a, float* f, void* stream);
int rbd_rnea_fpass_f64(const double* q, const double* qd, const double* qdd, double gravity, int64_t B,
                       double* v, double* a, double* f, void* stream);
int rbd_rnea_bpass_f32(const float* q, float* f, int64_t B, float* c, void* stream);
int rbd_rnea_bpass_f64(const double* q, double* f, int64_t B, double* c, void* stream);

/* RBDReference.rnea_grad(q, qd, qdd=None, GRAVITY, USE_VELOCITY_DAMPING)   (RBDReference.py:1345-1368)
 *   dc_du : [B, n, 2n] = [dc_dq | dc_dqd]  (np.hstack, :1367)
 *   c     : [B, n] or NULL -- the bias force the reference computes on the way (:1353) and drops. */
int rbd_rnea_grad_f32(const float* q, const float* qd, const float* qdd, float gravity,
                      int use_damping, int64_t B, float* c, float* dc_du, void* stream);
int rbd_rnea_grad_f64(const double* q, const double* qd, const double* qdd, double gravity,
                      int use_damping, int64_t B, double* c, double* dc_du, void* stream);

/* RBDReference.minv(q, output_dense)                      (RBDReference.py:785-806)
 *   Minv : [B, n, n].  output_dense != 0: symmetric matrix (:799-804).  output_dense == 0: upper
 *   triangle as the reference defines it, strict lower triangle ZERO (the reference leaves
 *   by-products of its forward pass there, :771; documented deviation).
 *   workspace: device scratch of at least rbd_minv_workspace_bytes(B, sizeof(T)) bytes, 16-byte
 *   aligned (0 bytes -- and then ignored -- for robots that use the fused one-lane kernel).       */
size_t rbd_minv_workspace_bytes(int64_t B, int elem_size);
int rbd_minv_f32(const float* q, int64_t B, int output_dense, float* Minv, void* workspace,
                 size_t workspace_bytes, void* stream);
int rbd_minv_f64(const double* q, int64_t B, int output_dense, double* Minv, void* workspace,
                 size_t workspace_bytes, void* stream);

/* RBDReference.crba(q)  (fixed-base branch, RBDReference.py:1091-1124): joint-space inertia H [B, n, n]. */
int rbd_crba_f32(const float* q, int64_t B, float* H, void* stream);
int rbd_crba_f64(const double* q, int64_t B, double* H, void* stream);

/* RBDReference.forward_dynamics(q, qd, u)                 (RBDReference.py:1371-1374)
 *   qdd = minv(q) @ (u - rnea(q, qd)[0])        u, qdd : [B, n]
 * RBDReference.forward_dynamics_grad(q, qd, u)            (RBDReference.py:1376-1384)
 *   dqdd_du : [B, n, 2n] = [qdd_dq | qdd_dqd] = -minv(q) @ rnea_grad(q, qd, qdd)  (the reference
 *   returns the two halves as a tuple); qdd (nullable) also receives the forward dynamics itself.
 *   Three / four launches on `stream`: rnea (bias force), minv phases A and B with the
 *   Minv (u - c) product fused into phase B, rnea_grad with the -Minv product fused into its epilogue.
 *   workspace: device scratch of at least rbd_fd_workspace_bytes(B, sizeof(T)) bytes, 16-byte aligned. */
size_t rbd_fd_workspace_bytes(int64_t B, int elem_size);
int rbd_forward_dynamics_f32(const float* q, const float* qd, const float* u, float gravity, int64_t B,
                             float* qdd, void* workspace, size_t workspace_bytes, void* stream);
int rbd_forward_dynamics_f64(const double* q, const double* qd, const double* u, double gravity, int64_t B,
                             double* qdd, void* workspace, size_t workspace_bytes, void* stream);
int rbd_forward_dynamics_grad_f32(const float* q, const float* qd, const float* u, float gravity, int64_t B,
                                  float* qdd, float* dqdd_du, void* workspace, size_t workspace_bytes,
                                  void* stream);
int rbd_forward_dynamics_grad_f64(const double* q, const double* qd, const double* u, double gravity,
                                  int64_t B, double* qdd, double* dqdd_du, void* workspace,
                                  size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RBD_HIP_H */
