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
 *     the caller owns all buffers; nothing is allocated per call -- with ONE exception: rbd_rnea_grad_f64 /
 *     rbd_rnea_with_grad_f64 of a tree too big for registers and LDS (the 30-body humanoid; rbd_kernel_name says
 *     rnea_grad_tree_ws_kernel) keeps a LIBRARY-OWNED scratch buffer per (device, stream), sized by what is resident
 *     at once, not by B (126 MB for that robot): allocated (hipMalloc) by the first such call on a stream and reused by
 *     every later one.  A buffer that was handed out is never freed or moved by a later call (launches in flight,
 *     bound launches and captured graphs may hold its address); rbd_release_workspaces() frees them all.  GRAPH
 *     CAPTURE: make one such call on a stream before capturing calls on it into a hipGraph (hipMalloc is not
 *     capturable), and do not call rbd_release_workspaces() while such a graph may still be launched.  Calls on
 *     different streams use different buffers, calls on one stream are ordered and share one;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream of the calling thread's current device);
 *     launches are asynchronous; every entry point runs on the STREAM's device (it switches the calling thread's
 *     current device for the duration of the call if that is another one), and per-kernel launch attributes, grid
 *     sizes and the library-owned workspace are cached per that device;
 *   - return 0 on success, <0 for argument errors (RBD_ERR_*), >0 = hipError_t of a failed launch;
 *     rbd_last_error() returns a thread-local message for the last non-zero return;
 *   - no C++ exceptions cross this boundary; the model is immutable, so concurrent calls from
 *     several host threads / streams are safe.
 *
 * Output buffers (c, v, a, f, dc_du, Minv, qdd, ...) and workspaces must be 16-byte aligned -- any device
 * allocation is; a view into the middle of one may not be -- the kernels store 16-byte pieces.  A misaligned
 * output is refused with RBD_ERR_ARG.  Inputs may have any alignment of their element type.
 *
 * Floating-base robots (RBDReference.py:585-593, :652-691, :761-779; robot.floating_base): body 0
 * owns indices 0..5 of q, qd, qdd, c (q[0:6] = px, py, pz, rx, ry, rz of the world -> base transform,
 * qd[0:6] = the base twist in base coordinates), body i >= 1 owns index i + 5; "n" in the shapes
 * below then reads nv = n + 5 for q, qd, qdd, c, u, Minv and stays the body count for v, a, f.  Such a
 * library serves rbd_rnea, rbd_rnea_grad and rbd_rnea_with_grad -- dc_du [B, nv, 2 nv]; the base's six position
 * columns are derivatives along a base-frame twist, as in the reference, :1168-1175; robots with fewer
 * than six bodies are refused: the reference raises IndexError for them, :1168 -- rbd_minv and
 * rbd_forward_dynamics; every other entry point returns RBD_ERR_UNSUPPORTED (the reference's own
 * crba / aba raise for floating bases).
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
#define RBD_ERR_NOT_BUILT (-4)    /* a FAMILY library (first-use build, rbdreference_amd/build.py) was asked for an entry
                                     point of another family: the caller picked the wrong library, nothing ran      */

typedef struct rbd_model_info {
  int32_t abi_version;
  int32_t n;                          /* bodies == joints == velocities (fixed base, 1-DoF)  */
  int32_t max_depth;                  /* longest root path, in bodies                        */
  uint64_t hash;                      /* first 64 bits of sha256 over the packed model       */
  char name[64];
  int32_t parent[RBD_MAX_BODIES];     /* -1 = child of the fixed base                        */
  int32_t joint_type[RBD_MAX_BODIES]; /* 0 revolute, 1 prismatic, 2 = the 6-DoF floating base */
  int32_t joint_axis[RBD_MAX_BODIES]; /* 0/1/2 = x/y/z of the body frame                     */
  int32_t floating_base;              /* 1: body 0 is attached by a 6-DoF joint (S = eye(6)), */
  int32_t nv;                         /*    and q, qd, qdd, c have nv = n + 5 columns (else n) */
} rbd_model_info_t;

int rbd_abi_version(void);
const char* rbd_last_error(void);
/* Frees every library-owned workspace (see "Conventions").  Synchronises each device that holds one first.  The caller
 * guarantees that no call into this library is in flight on another thread and that no captured graph containing one
 * of its launches will be replayed.  Returns 0, or the hipError_t of a failed synchronisation. */
int rbd_release_workspaces(void);
/* The robot this library was compiled for (host-side, no GPU needed). */
int rbd_model_info(rbd_model_info_t* out);

/* Kernel selection.  Several kernels may serve one entry point; which one runs is
 * decided per robot at compile time and per launch from B.  rbd_set_option overrides the choice
 * where the override is built into the library (tests run every kernel of every robot this way;
 * a value the library cannot honour is ignored).  Options only ever select between kernels that
 * compute the same result.  Process-wide, thread-safe, no environment variables are read.
 *   RBD_OPT_GRAD_KERNEL    rbd_rnea_grad: AUTO | TREE (chain-by-chain world-frame kernel) | COLS (one
 *                          lane per derivative column: AUTO picks it for small batches) | BATCH (the
 *                          robot's batch-parallel kernel at every batch size)
 *   RBD_OPT_MINV_PHASE_A   rbd_minv, robots too big for the one-lane kernel: AUTO | LANE (phase A with one lane
 *                          per configuration, then the column kernel) | IA8 (eight lanes per configuration,
 *                          then the column kernel) | FUSED (one launch from q to Minv; robots whose big
 *                          root subtrees have limbs; what AUTO picks for them)
 *   RBD_OPT_RNEA_KERNEL    rbd_rnea with v, a, f: AUTO | BATCH (one lane per configuration) | GROUPS (one
 *                          wave per independent root subtree).  AUTO: robots whose root subtrees carry
 *                          several big branches (Atlas' arms) get one wave per branch / stem / root
 *                          subtree, other multi-root robots GROUPS, single chains BATCH
 *   RBD_OPT_SELECT_BATCH   rows of the GLOBAL batch a call is a shard of (0 = the call's own B, the default).
 *                          Wherever AUTO decides from the batch size (the small-batch column kernel of
 *                          rbd_rnea_grad, phase A of rbd_minv), it decides from this number instead: a shard of
 *                          a sharded batch then runs the kernel the unsharded call would run, so sharded and
 *                          unsharded results are bit-identical row by row (rbdreference_amd.dist.ShardedRBD sets it)
 * rbd_kernel_name writes the name of the kernel (the dominant one of a multi-launch entry point) that
 * `op` would launch for a batch of B rows of elem_size-byte scalars under the current options. */
#define RBD_OPT_GRAD_KERNEL 0
#define RBD_OPT_MINV_PHASE_A 1
#define RBD_OPT_RNEA_KERNEL 2
#define RBD_OPT_SELECT_BATCH 3
#define RBD_OPT_COUNT_ 4
#define RBD_GRAD_KERNEL_AUTO 0
#define RBD_GRAD_KERNEL_TREE 1
#define RBD_GRAD_KERNEL_COLS 2
#define RBD_GRAD_KERNEL_BATCH 3
#define RBD_RNEA_KERNEL_AUTO 0
#define RBD_RNEA_KERNEL_BATCH 1
#define RBD_RNEA_KERNEL_GROUPS 2
#define RBD_MINV_PHASE_A_AUTO 0
#define RBD_MINV_PHASE_A_LANE 1
#define RBD_MINV_PHASE_A_IA8 2
#define RBD_MINV_PHASE_A_FUSED 3
#define RBD_OP_RNEA 0
#define RBD_OP_RNEA_GRAD 1
#define RBD_OP_MINV 2
int rbd_set_option(int option, int value);
int rbd_get_option(int option);
int rbd_kernel_name(int op, int elem_size, int64_t B, char* buf, size_t len);

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

/* Gradient passes (README.md:19).  Layouts as the reference returns them, batch outermost:
 *   dv, da, df : [B, 6, n, NB]  -- element [b, r, c, i] = d(component r of body i) / d u_c
 * RBDReference.rnea_grad_fpass_dq(q, qd, v, a, GRAVITY) -> (dv_dq, da_dq, df_dq)     (RBDReference.py:1127-1187)
 *   v, a [B,6,NB] are the outputs of rnea (a includes the S qdd term, :1353-1358).
 * RBDReference.rnea_grad_fpass_dqd(q, qd, v) -> (dv_dqd, da_dqd, df_dqd)             (RBDReference.py:1189-1255)
 * RBDReference.rnea_grad_bpass_dq(q, f, df_dq) -> dc_dq [B,n,n]                      (RBDReference.py:1257-1297)
 *   f [B,6,NB] is the ACCUMULATED rnea force (:1353,:1362); df_dq is accumulated child -> parent
 *   IN PLACE as the reference does (:1291-1294).
 * RBDReference.rnea_grad_bpass_dqd(q, df_dqd, USE_VELOCITY_DAMPING) -> dc_dqd [B,n,n] (RBDReference.py:1299-1343)
 *   df_dqd accumulated in place (:1331). */
int rbd_rnea_grad_fpass_dq_f32(const float* q, const float* qd, const float* v, const float* a, float gravity, int64_t B,
                               float* dv_dq, float* da_dq, float* df_dq, void* stream);
int rbd_rnea_grad_fpass_dqd_f32(const float* q, const float* qd, const float* v, int64_t B, float* dv_dqd, float* da_dqd,
                                float* df_dqd, void* stream);
int rbd_rnea_grad_bpass_dq_f32(const float* q, const float* f, float* df_dq, int64_t B, float* dc_dq, void* stream);
int rbd_rnea_grad_bpass_dqd_f32(const float* q, float* df_dqd, int use_damping, int64_t B, float* dc_dqd, void* stream);
int rbd_rnea_grad_fpass_dq_f64(const double* q, const double* qd, const double* v, const double* a, double gravity, int64_t B,
                               double* dv_dq, double* da_dq, double* df_dq, void* stream);
int rbd_rnea_grad_fpass_dqd_f64(const double* q, const double* qd, const double* v, int64_t B, double* dv_dqd, double* da_dqd,
                                double* df_dqd, void* stream);
int rbd_rnea_grad_bpass_dq_f64(const double* q, const double* f, double* df_dq, int64_t B, double* dc_dq, void* stream);
int rbd_rnea_grad_bpass_dqd_f64(const double* q, double* df_dqd, int use_damping, int64_t B, double* dc_dqd, void* stream);

/* RBDReference.rnea_grad(q, qd, qdd=None, GRAVITY, USE_VELOCITY_DAMPING)   (RBDReference.py:1345-1368)
 *   dc_du : [B, n, 2n] = [dc_dq | dc_dqd]  (np.hstack, :1367)
 *   c     : [B, n] or NULL -- the bias force the reference computes on the way (:1353) and drops. */
int rbd_rnea_grad_f32(const float* q, const float* qd, const float* qdd, float gravity,
                      int use_damping, int64_t B, float* c, float* dc_du, void* stream);
int rbd_rnea_grad_f64(const double* q, const double* qd, const double* qdd, double gravity,
                      int use_damping, int64_t B, double* c, double* dc_du, void* stream);

/* rnea + rnea_grad in one call: everything RBDReference.rnea (:623-628) and RBDReference.rnea_grad
 * (:1345-1368) return for the same (q, qd, qdd) -- the reference's rnea_grad runs rnea internally
 * (:1353) and drops its outputs.  c [B,n]; v, a, f [B,6,n] (f accumulated); dc_du [B,n,2n]; all
 * non-null.  For small batches this is ONE launch (one lane per derivative column), otherwise the rnea
 * kernel followed by the gradient kernel on `stream`. */
int rbd_rnea_with_grad_f32(const float* q, const float* qd, const float* qdd, float gravity, int use_damping,
                           int64_t B, float* c, float* v, float* a, float* f, float* dc_du, void* stream);
int rbd_rnea_with_grad_f64(const double* q, const double* qd, const double* qdd, double gravity, int use_damping,
                           int64_t B, double* c, double* v, double* a, double* f, double* dc_du, void* stream);

/* RBDReference.minv(q, output_dense)                      (RBDReference.py:785-806)
 *   Minv : [B, n, n].  output_dense != 0: symmetric matrix (:799-804).  output_dense == 0: upper
 *   triangle as the reference defines it, strict lower triangle ZERO (the reference leaves
 *   by-products of its forward pass there, :771; documented deviation).
 *   workspace: device scratch of at least rbd_minv_workspace_bytes(B, sizeof(T)) bytes, 16-byte
 *   aligned.  0 bytes -- and then ignored, NULL is fine -- whenever the kernel selected by the current
 *   RBD_OPT_MINV_PHASE_A option does not go through HBM (the one-lane kernel of small robots, the
 *   one-launch kernel that AUTO picks for robots whose big groups have limbs): query it again after
 *   changing that option.                                                                          */
size_t rbd_minv_workspace_bytes(int64_t B, int elem_size);
int rbd_minv_f32(const float* q, int64_t B, int output_dense, float* Minv, void* workspace,
                 size_t workspace_bytes, void* stream);
int rbd_minv_f64(const double* q, int64_t B, int output_dense, double* Minv, void* workspace,
                 size_t workspace_bytes, void* stream);

/* Minv passes (README.md:19).
 * RBDReference.minv_bpass(q) -> (Minv, F, U, Dinv)                                   (RBDReference.py:630-735)
 *   Minv [B,n,n]: row i filled on the columns of subtree(i) only (:700-708), zero elsewhere;
 *   F [B,n,6,n]; U [B,n,6]; Dinv [B,n] holds D = S^T U, NOT its inverse, exactly as the reference's
 *   array of that name does (:698).
 * RBDReference.minv_fpass(q, Minv, F, U, Dinv) -> Minv                               (RBDReference.py:737-783)
 *   Minv is updated IN PLACE over whole rows (:771), so its strict lower triangle receives the
 *   same by-products as in the reference; F is rebuilt (:774-781), its incoming contents are not read. */
int rbd_minv_bpass_f32(const float* q, int64_t B, float* Minv, float* F, float* U, float* Dinv, void* stream);
int rbd_minv_fpass_f32(const float* q, int64_t B, float* Minv, float* F, const float* U, const float* Dinv, void* stream);
int rbd_minv_bpass_f64(const double* q, int64_t B, double* Minv, double* F, double* U, double* Dinv, void* stream);
int rbd_minv_fpass_f64(const double* q, int64_t B, double* Minv, double* F, const double* U, const double* Dinv, void* stream);

/* RBDReference.crba(q)  (fixed-base branch, RBDReference.py:1091-1124): joint-space inertia H [B, n, n]. */
int rbd_crba_f32(const float* q, int64_t B, float* H, void* stream);
int rbd_crba_f64(const double* q, int64_t B, double* H, void* stream);

/* RBDReference.aba(q, qd, tau, f_ext=[], GRAVITY)  (fixed-base branch, RBDReference.py:940-1024) -> qdd [B, n].
 * Articulated-body algorithm; f_ext is not part of the C-ABI (the fixed-base branch never reads it).
 * Same result as the forward-dynamics entry points below, i.e. Minv (tau - c), to rounding; no workspace. */
int rbd_aba_f32(const float* q, const float* qd, const float* tau, float gravity, int64_t B, float* qdd, void* stream);
int rbd_aba_f64(const double* q, const double* qd, const double* tau, double gravity, int64_t B, double* qdd, void* stream);

/* RBDReference.forward_dynamics(q, qd, u)                 (RBDReference.py:1371-1374)
 *   qdd = minv(q) @ (u - rnea(q, qd)[0])        u, qdd : [B, n]
 * RBDReference.forward_dynamics_grad(q, qd, u)            (RBDReference.py:1376-1384)
 *   dqdd_du : [B, n, 2n] = [qdd_dq | qdd_dqd] = -minv(q) @ rnea_grad(q, qd, qdd)  (the reference
 *   returns the two halves as a tuple); qdd (nullable) also receives the forward dynamics itself.
 *   forward_dynamics is one launch (the articulated-body sweeps of rbd_aba give Minv (u - c) without
 *   forming either factor; its workspace arguments are accepted and unused).  forward_dynamics_grad
 *   is three / four launches on `stream`: rnea (bias force), minv with the Minv (u - c) product
 *   fused in, rnea_grad with the -Minv product fused into its epilogue; it needs a device scratch of
 *   at least rbd_fd_workspace_bytes(B, sizeof(T)) bytes, 16-byte aligned. */
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
