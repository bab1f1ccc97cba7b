// rbd_fb_kernels.hip -- the translation units of a FLOATING-BASE robot's library (besides COMMON, which
// comes from rbd_kernels.hip): rbd_fb.h's kernels behind the same C-ABI (include/rbd_hip.h).
//
// A floating-base library exports every symbol of the header; the entry points the reference itself
// cannot serve for a floating base (its crba and aba raise, its rnea_grad raises IndexError for NB < 6:
// RBDReference.py:1063, :900, :1168) and the per-pass / forward_dynamics_grad entry points return
// RBD_ERR_UNSUPPORTED with a message.  Shapes: q, qd, qdd, c, u [B, NV]; v, a, f [B, 6, N]; Minv [B, NV, NV].
// Units: -DRBD_TU_FB_F32 / -DRBD_TU_FB_F64 (rbdreference_amd/build.py).
#include "rbd_fb.h"
#include "rbd_fb_world.h"
#include "rbd_fb_passes.h"
#include "rbd_fb_minv.h"
#include "../../include/rbd_hip.h"
#include "rbd_host.h"
#include <cstdio>
#include <cstring>
#include <atomic>
#include <cstdint>

static_assert(rbdm::FLOATING_BASE, "rbd_fb_kernels.hip is for floating-base robots");

extern "C" __attribute__((visibility("hidden"))) char* rbd_err_buf(void);
namespace {
constexpr size_t RBD_ERR_LEN = 512;
int fail(int code, const char* msg) { std::snprintf(rbd_err_buf(), RBD_ERR_LEN, "%s", msg); return code; }
int hip_fail(hipError_t e, const char* where) {
  std::snprintf(rbd_err_buf(), RBD_ERR_LEN, "%s: %s", where, hipGetErrorString(e));
  return (int)e > 0 ? (int)e : 1;
}
int unsupported(const char* who) {
  std::snprintf(rbd_err_buf(), RBD_ERR_LEN,
                "%s: not available for floating-base robots (the reference's own crba / aba raise for them, "
                "RBDReference.py:1063, :900)", who);
  return RBD_ERR_UNSUPPORTED;
}
constexpr size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

extern "C" __attribute__((visibility("hidden"))) std::atomic<int>* rbd_option_slot(int option);
int rbd_option(int option) { return rbd_option_slot(option)->load(std::memory_order_relaxed); }

// dynamic LDS above 64 KB needs the attribute once per kernel
template <class K>
int ensure_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? 0 : hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
}

template <class T>
int rnea_fb_launch(const T* q, const T* qd, const T* qdd, T gravity, int64_t B, T* c, T* v, T* a, T* f, void* stream) {
  using namespace rbdk;
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_rnea: B < 0");
  if (B == 0) return 0;
  if (!q || !qd || !c) return fail(RBD_ERR_ARG, "rbd_rnea: q, qd and c must be non-null");
  const bool vaf = v || a || f;
  if (vaf && !(v && a && f)) return fail(RBD_ERR_ARG, "rbd_rnea: v, a, f must be all null or all non-null");
  if (misaligned(c) || misaligned(v) || misaligned(a) || misaligned(f)) return fail(RBD_ERR_ARG, "rbd_rnea: output buffers must be 16-byte aligned");
  const int64_t blocks = (B + 63) / 64;
  if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_rnea: B too large");
  constexpr size_t lds = rnea_fbw_lds_bytes<T>();
  if constexpr (lds <= 160 * 1024) {
    // one configuration per lane, outputs through LDS images as flat 16-byte stores (rbd_fb_world.h)
    int rc;
    if (qdd) {
      auto k = rnea_fbw_kernel<T, true>;
      if ((rc = ensure_lds(k, lds)) != 0) return rc;
      hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, q, qd, qdd, gravity, (long long)B, c, v, a, f);
    } else {
      auto k = rnea_fbw_kernel<T, false>;
      if ((rc = ensure_lds(k, lds)) != 0) return rc;
      hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, q, qd, qdd, gravity, (long long)B, c, v, a, f);
    }
  } else {
    if (qdd) hipLaunchKernelGGL((rnea_fb_kernel<T, true>), dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, q, qd, qdd, gravity, (long long)B, c, v, a, f);
    else hipLaunchKernelGGL((rnea_fb_kernel<T, false>), dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, q, qd, qdd, gravity, (long long)B, c, v, a, f);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_rnea (floating base) launch");
}
// which gradient kernel serves this robot: the world-frame kernel (rbd_fb_world.h) where its identities apply and
// its LDS plan fits, else the column recursion (rbd_fb.h); RBD_OPT_GRAD_KERNEL = COLS forces the latter
template <class T>
bool grad_fb_use_world() {
  if constexpr (!rbdk::grad_fbw_ok<T>()) return false;
  if constexpr (!rbdk::grad_fb_ok<T>()) return true;
  return rbd_option(RBD_OPT_GRAD_KERNEL) != RBD_GRAD_KERNEL_COLS;
}
template <class T>
int grad_fb_launch(const char* who, const T* q, const T* qd, const T* qdd, T gravity, int use_damping, int64_t B, T* c, T* v,
                   T* a, T* f, T* dc_du, void* stream) {
  using namespace rbdk;
  if constexpr (!grad_fb_ok<T>() && !grad_fbw_ok<T>()) {
    std::snprintf(rbd_err_buf(), RBD_ERR_LEN,
                  "%s: floating-base rnea_grad needs 6 <= NB (the reference raises IndexError below, RBDReference.py:1168) "
                  "and an LDS working set that fits; this robot has NB = %d", who, N);
    return RBD_ERR_UNSUPPORTED;
  } else {
    if (B < 0) return fail(RBD_ERR_ARG, "rbd_rnea_grad: B < 0");
    if (B == 0) return 0;
    if (!q || !qd || !dc_du) return fail(RBD_ERR_ARG, "rbd_rnea_grad: q, qd and dc_du must be non-null");
    const bool vaf = v || a || f;
    if (vaf && !(v && a && f && c)) return fail(RBD_ERR_ARG, "rbd_rnea_with_grad: c, v, a, f must be all non-null");
    if (misaligned(c) || misaligned(dc_du)) return fail(RBD_ERR_ARG, "rbd_rnea_grad: output buffers must be 16-byte aligned");
    if (vaf) {   // RBDReference.rnea's outputs: the rnea kernel's own launch
      int rc = rnea_fb_launch<T>(q, qd, qdd, gravity, B, c, v, a, f, stream);
      if (rc != 0) return rc;
    }
    T* cg = vaf ? nullptr : c;
    if (grad_fb_use_world<T>()) {
      if constexpr (grad_fbw_ok<T>()) {
        const int64_t blocks = (B + 63) / 64;
        if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_rnea_grad: B too large");
        constexpr size_t lds = fbw_lds_bytes<T>();
        int rc;
        if (qdd) {
          auto k = rnea_grad_fbw_kernel<T, true>;
          if ((rc = ensure_lds(k, lds)) != 0) return rc;
          hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * FBW_W), lds, (hipStream_t)stream, q, qd, qdd, gravity, use_damping, (long long)B, cg, dc_du);
        } else {
          auto k = rnea_grad_fbw_kernel<T, false>;
          if ((rc = ensure_lds(k, lds)) != 0) return rc;
          hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * FBW_W), lds, (hipStream_t)stream, q, qd, qdd, gravity, use_damping, (long long)B, cg, dc_du);
        }
      }
    } else {
      if constexpr (grad_fb_ok<T>()) {
        const int64_t blocks = (B + FB_GRAD_C - 1) / FB_GRAD_C;
        if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_rnea_grad: B too large");
        if (qdd) hipLaunchKernelGGL((rnea_grad_fb_kernel<T, true>), dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, q, qd, qdd, gravity, use_damping, (long long)B, cg, dc_du);
        else hipLaunchKernelGGL((rnea_grad_fb_kernel<T, false>), dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, q, qd, qdd, gravity, use_damping, (long long)B, cg, dc_du);
      }
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "rbd_rnea_grad (floating base) launch");
  }
}
template <class T>
int minv_fb_launch(const T* q, int64_t B, int dense, T* Minv, void* stream, const T* u = nullptr, const T* cbias = nullptr, T* qdd = nullptr,
                   bool* fused_qdd = nullptr, const T* qd = nullptr, T gravity = T(0)) {
  using namespace rbdk;
  if (fused_qdd) *fused_qdd = false;
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_minv: B < 0");
  if (B == 0) return 0;
  if (!q || (!Minv && !qdd)) return fail(RBD_ERR_ARG, "rbd_minv: q and Minv must be non-null");
  if (misaligned(Minv)) return fail(RBD_ERR_ARG, "rbd_minv: Minv must be 16-byte aligned");
  if constexpr (minv_fbm_ok<T>()) {
    // one wave per subtree of the base (rbd_fb_minv.h); RBD_OPT_MINV_PHASE_A = LANE keeps the four-lanes kernel
    if (rbd_option(RBD_OPT_MINV_PHASE_A) != RBD_MINV_PHASE_A_LANE) {
      const int64_t nb = (B + 63) / 64;
      if (nb > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_minv: B too large");
      constexpr size_t ldsm = minv_fbm_lds_bytes<T>();
      auto km = minv_fbm_kernel<T>;
      int rcm;
      if ((rcm = ensure_lds(km, ldsm)) != 0) return rcm;
      // with u, c, qdd: qdd = Minv (u - c) leaves the same launch (rbd_fb_minv.h); Minv may then be null
      // (qd instead of c: the kernel computes the bias force itself)
      hipLaunchKernelGGL(km, dim3((unsigned)nb), dim3(64 * FBW_W), ldsm, (hipStream_t)stream, q, (long long)B, dense, Minv, u, cbias, qdd,
                         cbias ? (const T*)nullptr : qd, gravity);
      if (fused_qdd) *fused_qdd = qdd != nullptr;
      hipError_t em = hipGetLastError();
      return em == hipSuccess ? 0 : hip_fail(em, "rbd_minv (floating base, wave per subtree) launch");
    }
  }
  if (!Minv) return fail(RBD_ERR_ARG, "rbd_minv: Minv must be non-null");
  const int64_t blocks = (B + 64 / FB_MINV_L - 1) / (64 / FB_MINV_L);
  if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_minv: B too large");
  constexpr size_t lds = minv_fb_lds_bytes<T>();
  if (lds > 160 * 1024) return fail(RBD_ERR_UNSUPPORTED, "rbd_minv: the block's matrices do not fit LDS for this robot size");
  auto k = minv_fb_kernel<T>;
  int rc;
  if ((rc = ensure_lds(k, lds)) != 0) return rc;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, q, (long long)B, dense, Minv);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_minv (floating base) launch");
}
template <class T>
int fd_fb_launch(const T* q, const T* qd, const T* u, T gravity, int64_t B, T* qdd, void* workspace, size_t wsb, void* stream) {
  using namespace rbdk;
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_forward_dynamics: B < 0");
  if (B == 0) return 0;
  if (!q || !qd || !u || !qdd) return fail(RBD_ERR_ARG, "rbd_forward_dynamics: q, qd, u, qdd must be non-null");
  const size_t off_c = 0, off_m = align16((size_t)B * NV * sizeof(T)), total = off_m + align16((size_t)B * NV * NV * sizeof(T));
  if (!workspace || wsb < total) return fail(RBD_ERR_WORKSPACE, "rbd_forward_dynamics: workspace missing or smaller than rbd_fd_workspace_bytes()");
  if (misaligned(workspace)) return fail(RBD_ERR_WORKSPACE, "rbd_forward_dynamics: workspace must be 16-byte aligned");
  if (misaligned(qdd)) return fail(RBD_ERR_ARG, "rbd_forward_dynamics: qdd must be 16-byte aligned");
  char* w = reinterpret_cast<char*>(workspace);
  T* c = reinterpret_cast<T*>(w + off_c);
  T* Mi = reinterpret_cast<T*>(w + off_m);
  int rc;
  // :1372-1374 in ONE launch where the wave-per-subtree kernel serves the robot: the bias force from qd, qdd = Minv (u - c) from
  // the columns in registers, the matrix itself never written (Minv = nullptr); else rnea, minv into the workspace and
  // the product kernel
  bool fused = false;
#ifdef RBD_FB_EXP_NO_OWN_BIAS      // timing experiment: the c-only rnea launch as before
  const bool wave_kernel = false;
#else
  const bool wave_kernel = minv_fbm_ok<T>() && rbd_option(RBD_OPT_MINV_PHASE_A) != RBD_MINV_PHASE_A_LANE;
#endif
  if (wave_kernel) {
    if ((rc = minv_fb_launch<T>(q, B, 1, (T*)nullptr, stream, u, (const T*)nullptr, qdd, &fused, qd, gravity)) != 0) return rc;
    if (fused) return 0;
  }
  if ((rc = rnea_fb_launch<T>(q, qd, nullptr, gravity, B, c, nullptr, nullptr, nullptr, stream)) != 0) return rc;   // :1372
  if ((rc = minv_fb_launch<T>(q, B, 1, Mi, stream, u, (const T*)c, qdd, &fused)) != 0) return rc;
  if (fused) return 0;
  const int64_t ab = ((int64_t)B * NV + 255) / 256;
  if (ab > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_forward_dynamics: B too large");
  hipLaunchKernelGGL((fb_apply_kernel<T>), dim3((unsigned)ab), dim3(256), 0, (hipStream_t)stream, (const T*)Mi, u, (const T*)c, (long long)B, qdd);   // :1374
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_forward_dynamics (floating base) launch");
}
// ---- per-pass surface (README.md:19) ---------------------------------------------------------------------------
template <class T>
int rnea_pass_fb_launch(int mode, const T* q, const T* qd, const T* qdd, T gravity, int64_t B, T* c, T* v, T* a, T* f, void* stream) {
  using namespace rbdk;
  const char* who = mode == 1 ? "rbd_rnea_fpass" : "rbd_rnea_bpass";
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_rnea pass: B < 0");
  if (B == 0) return 0;
  if (mode == 1 && (!q || !qd || !v || !a || !f)) return fail(RBD_ERR_ARG, "rbd_rnea_fpass: q, qd, v, a, f must be non-null");
  if (mode == 2 && (!q || !f || !c)) return fail(RBD_ERR_ARG, "rbd_rnea_bpass: q, f, c must be non-null");
  if (misaligned(c) || misaligned(v) || misaligned(a) || misaligned(f)) return fail(RBD_ERR_ARG, "rbd_rnea pass: output buffers must be 16-byte aligned");
  const int64_t blocks = (B + 63) / 64;
  if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_rnea pass: B too large");
  constexpr size_t lds = rnea_fbw_lds_bytes<T>();
  if constexpr (lds > 160 * 1024) {
    std::snprintf(rbd_err_buf(), RBD_ERR_LEN, "%s: the [64][6 NB] image does not fit LDS for this robot size", who);
    return RBD_ERR_UNSUPPORTED;
  } else {
    int rc;
    if (mode == 2) {
      auto k = rnea_fbw_kernel<T, false, 2>;
      if ((rc = ensure_lds(k, lds)) != 0) return rc;
      hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, q, (const T*)nullptr, (const T*)nullptr, gravity, (long long)B, c, (T*)nullptr, (T*)nullptr, f);
    } else if (qdd) {
      auto k = rnea_fbw_kernel<T, true, 1>;
      if ((rc = ensure_lds(k, lds)) != 0) return rc;
      hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, q, qd, qdd, gravity, (long long)B, (T*)nullptr, v, a, f);
    } else {
      auto k = rnea_fbw_kernel<T, false, 1>;
      if ((rc = ensure_lds(k, lds)) != 0) return rc;
      hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, q, qd, qdd, gravity, (long long)B, (T*)nullptr, v, a, f);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, who);
  }
}
int grad_pass_needs_six(const char* who) {
  std::snprintf(rbd_err_buf(), RBD_ERR_LEN, "%s: floating-base gradient passes need NB >= 6 (the reference raises IndexError below, "
                "RBDReference.py:1168); this robot has NB = %d", who, rbdk::N);
  return RBD_ERR_UNSUPPORTED;
}
template <class T, bool ISQD>
int grad_fpass_fb_launch(const T* q, const T* qd, const T* v, const T* a, T gravity, int64_t B, T* dv, T* da, T* df, void* stream) {
  using namespace rbdk;
  const char* who = ISQD ? "rbd_rnea_grad_fpass_dqd" : "rbd_rnea_grad_fpass_dq";
  if constexpr (N < 6) return grad_pass_needs_six(who);
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_rnea_grad_fpass: B < 0");
  if (B == 0) return 0;
  if (!q || !qd || !v || (!ISQD && !a) || !dv || !da || !df) return fail(RBD_ERR_ARG, "rbd_rnea_grad_fpass: null argument");
  const int64_t blocks = (B + FBP_C - 1) / FBP_C;
  if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_rnea_grad_fpass: B too large");
  hipLaunchKernelGGL((fb_grad_fpass_kernel<T, ISQD>), dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, q, qd, v, a, gravity, (long long)B, dv, da, df);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, who);
}
template <class T, bool ISQD>
int grad_bpass_fb_launch(const T* q, const T* f, T* df, int use_damping, int64_t B, T* dc, void* stream) {
  using namespace rbdk;
  const char* who = ISQD ? "rbd_rnea_grad_bpass_dqd" : "rbd_rnea_grad_bpass_dq";
  if constexpr (N < 6) return grad_pass_needs_six(who);
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_rnea_grad_bpass: B < 0");
  if (B == 0) return 0;
  if (!q || (!ISQD && !f) || !df || !dc) return fail(RBD_ERR_ARG, "rbd_rnea_grad_bpass: null argument");
  const int64_t blocks = (B + FBP_C - 1) / FBP_C;
  if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_rnea_grad_bpass: B too large");
  hipLaunchKernelGGL((fb_grad_bpass_kernel<T, ISQD>), dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, q, f, df, use_damping, (long long)B, dc);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, who);
}
template <class T>
int minv_bpass_fb_launch(const T* q, int64_t B, T* Minv, T* F, T* U, T* Dinv, void* stream) {
  using namespace rbdk;
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_minv_bpass: B < 0");
  if (B == 0) return 0;
  if (!q || !Minv || !F || !U || !Dinv) return fail(RBD_ERR_ARG, "rbd_minv_bpass: null argument");
  const int64_t blocks = (B + 64 / FB_MINV_L - 1) / (64 / FB_MINV_L);
  if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_minv_bpass: B too large");
  hipError_t e = hipMemsetAsync(Minv, 0, (size_t)B * NV * NV * sizeof(T), (hipStream_t)stream);      // entries outside the subtrees stay zero (:700-708)
  if (e == hipSuccess) e = hipMemsetAsync(F, 0, (size_t)B * NV * 6 * NV * sizeof(T), (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "rbd_minv_bpass (clear)");
  hipLaunchKernelGGL((fb_minv_bpass_kernel<T>), dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, q, (long long)B, Minv, F, U, Dinv);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_minv_bpass (floating base) launch");
}
template <class T>
int minv_fpass_fb_launch(const T* q, int64_t B, T* Minv, T* F, const T* U, const T* Dinv, void* stream) {
  using namespace rbdk;
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_minv_fpass: B < 0");
  if (B == 0) return 0;
  if (!q || !Minv || !F || !U || !Dinv) return fail(RBD_ERR_ARG, "rbd_minv_fpass: null argument");
  const int64_t blocks = (B + FBP_C - 1) / FBP_C;
  if (blocks > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_minv_fpass: B too large");
  hipLaunchKernelGGL((fb_minv_fpass_kernel<T>), dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, q, (long long)B, Minv, F, U, Dinv);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_minv_fpass (floating base) launch");
}
// forward_dynamics_grad (:1376-1384): forward dynamics, rnea_grad at that qdd, one batched product.
// workspace: c [B, NV] | Minv [B, NV, NV] | qdd [B, NV] | dc_du [B, NV, 2 NV]
template <class T>
size_t fdg_fb_bytes(int64_t B) {
  using namespace rbdk;
  return align16((size_t)B * NV * sizeof(T)) + align16((size_t)B * NV * NV * sizeof(T)) + align16((size_t)B * NV * sizeof(T)) +
         align16((size_t)B * NV * 2 * NV * sizeof(T));
}
template <class T>
int fdg_fb_launch(const T* q, const T* qd, const T* u, T gravity, int64_t B, T* qdd_out, T* dqdd_du, void* workspace, size_t wsb, void* stream) {
  using namespace rbdk;
  if constexpr (N < 6) return grad_pass_needs_six("rbd_forward_dynamics_grad");
  if (B < 0) return fail(RBD_ERR_ARG, "rbd_forward_dynamics_grad: B < 0");
  if (B == 0) return 0;
  if (!q || !qd || !u || !dqdd_du) return fail(RBD_ERR_ARG, "rbd_forward_dynamics_grad: q, qd, u, dqdd_du must be non-null");
  if (misaligned(dqdd_du) || misaligned(qdd_out)) return fail(RBD_ERR_ARG, "rbd_forward_dynamics_grad: output buffers must be 16-byte aligned");
  if (!workspace || wsb < fdg_fb_bytes<T>(B)) return fail(RBD_ERR_WORKSPACE, "rbd_forward_dynamics_grad: workspace missing or smaller than rbd_fd_workspace_bytes()");
  if (misaligned(workspace)) return fail(RBD_ERR_WORKSPACE, "rbd_forward_dynamics_grad: workspace must be 16-byte aligned");
  char* w = reinterpret_cast<char*>(workspace);
  size_t o = 0;
  T* c = reinterpret_cast<T*>(w + o); o += align16((size_t)B * NV * sizeof(T));
  T* Mi = reinterpret_cast<T*>(w + o); o += align16((size_t)B * NV * NV * sizeof(T));
  T* qdd_ws = reinterpret_cast<T*>(w + o); o += align16((size_t)B * NV * sizeof(T));
  T* dc = reinterpret_cast<T*>(w + o);
  T* qdd = qdd_out ? qdd_out : qdd_ws;
  int rc;
  bool fused = false;
#ifdef RBD_FB_EXP_NO_OWN_BIAS
  const bool wave_kernel = false;
#else
  const bool wave_kernel = minv_fbm_ok<T>() && rbd_option(RBD_OPT_MINV_PHASE_A) != RBD_MINV_PHASE_A_LANE;
#endif
  if (wave_kernel) {   // :1372-1374 and :1381 in one launch (bias force from qd inside the minv kernel)
    if ((rc = minv_fb_launch<T>(q, B, 1, Mi, stream, u, (const T*)nullptr, qdd, &fused, qd, gravity)) != 0) return rc;
    if (!fused) return fail(RBD_ERR_UNSUPPORTED, "rbd_forward_dynamics_grad: the wave-per-subtree minv kernel did not run");
  } else {
    if ((rc = rnea_fb_launch<T>(q, qd, nullptr, gravity, B, c, nullptr, nullptr, nullptr, stream)) != 0) return rc;   // :1372
    if ((rc = minv_fb_launch<T>(q, B, 1, Mi, stream, u, (const T*)c, qdd, &fused)) != 0) return rc;                   // :1373, :1381 (+ :1374 where fused)
  }
  if (!fused) {
    const int64_t ab = ((int64_t)B * NV + 255) / 256;
    if (ab > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_forward_dynamics_grad: B too large");
    hipLaunchKernelGGL((fb_apply_kernel<T>), dim3((unsigned)ab), dim3(256), 0, (hipStream_t)stream, (const T*)Mi, u, (const T*)c, (long long)B, qdd);   // :1374
  }
  if ((rc = grad_fb_launch<T>("rbd_forward_dynamics_grad", q, qd, (const T*)qdd, gravity, 0, B, (T*)nullptr, (T*)nullptr, (T*)nullptr, (T*)nullptr, dc, stream)) != 0) return rc;   // :1378
  constexpr int MMC = negmm_cfgs<T, NV>();
  const int64_t mb = (B + MMC - 1) / MMC;
  if (mb > 0x7fffffffLL) return fail(RBD_ERR_ARG, "rbd_forward_dynamics_grad: B too large");
  hipLaunchKernelGGL((neg_mm_kernel<T, NV>), dim3((unsigned)mb), dim3(negmm_threads<T, NV>()), 0, (hipStream_t)stream, (const T*)Mi, (const T*)dc, (long long)B, dqdd_du);   // :1382-1383
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail(e, "rbd_forward_dynamics_grad (floating base) launch");
}
}  // namespace

extern "C" {
// kernel names for rbd_kernel_name (COMMON unit)
__attribute__((visibility("hidden"))) int rbd_grad_kernel_name_f32(int64_t B, char* buf, size_t len);
__attribute__((visibility("hidden"))) int rbd_grad_kernel_name_f64(int64_t B, char* buf, size_t len);
__attribute__((visibility("hidden"))) int rbd_minv_kernel_name_f32(int64_t B, char* buf, size_t len);
__attribute__((visibility("hidden"))) int rbd_minv_kernel_name_f64(int64_t B, char* buf, size_t len);
__attribute__((visibility("hidden"))) int rbd_rnea_kernel_name_f32(int64_t B, char* buf, size_t len);
__attribute__((visibility("hidden"))) int rbd_rnea_kernel_name_f64(int64_t B, char* buf, size_t len);

__attribute__((visibility("hidden"))) int rbd_minv_needs_ws_f32(void);
__attribute__((visibility("hidden"))) int rbd_minv_needs_ws_f64(void);

#define RBD_FB_DEFS(SFX, T)                                                                                                 \
  int rbd_minv_needs_ws_##SFX(void) { return 0; }                                                                           \
  int rbd_grad_kernel_name_##SFX(int64_t, char* buf, size_t len) {                                                          \
    std::snprintf(buf, len, "%s<%s,true>", grad_fb_use_world<T>() ? "rnea_grad_fbw_kernel" : "rnea_grad_fb_kernel", sizeof(T) == 4 ? "float" : "double"); \
    return 0;                                                                                                               \
  }                                                                                                                         \
  int rbd_rnea_kernel_name_##SFX(int64_t, char* buf, size_t len) {                                                          \
    std::snprintf(buf, len, "%s<%s,true>", rbdk::rnea_fbw_lds_bytes<T>() <= 160 * 1024 ? "rnea_fbw_kernel" : "rnea_fb_kernel", sizeof(T) == 4 ? "float" : "double"); \
    return 0;                                                                                                               \
  }                                                                                                                         \
  int rbd_minv_kernel_name_##SFX(int64_t, char* buf, size_t len) {                                                          \
    const bool m = rbdk::minv_fbm_ok<T>() && rbd_option(RBD_OPT_MINV_PHASE_A) != RBD_MINV_PHASE_A_LANE;                      \
    std::snprintf(buf, len, "%s<%s>", m ? "minv_fbm_kernel" : "minv_fb_kernel", sizeof(T) == 4 ? "float" : "double");      \
    return 0;                                                                                                               \
  }                                                                                                                         \
  int rbd_rnea_##SFX(const T* q, const T* qd, const T* qdd, T gravity, int64_t B, T* c, T* v, T* a, T* f, void* stream) {   \
    RbdStreamDevice sd_(stream); return rnea_fb_launch<T>(q, qd, qdd, gravity, B, c, v, a, f, stream);                                                   \
  }                                                                                                                         \
  int rbd_minv_##SFX(const T* q, int64_t B, int output_dense, T* Minv, void*, size_t, void* stream) {                       \
    RbdStreamDevice sd_(stream); return minv_fb_launch<T>(q, B, output_dense, Minv, stream);                                                             \
  }                                                                                                                         \
  int rbd_forward_dynamics_##SFX(const T* q, const T* qd, const T* u, T gravity, int64_t B, T* qdd, void* ws, size_t wsb,   \
                                 void* stream) {                                                                            \
    RbdStreamDevice sd_(stream); return fd_fb_launch<T>(q, qd, u, gravity, B, qdd, ws, wsb, stream);                                                     \
  }                                                                                                                         \
  int rbd_rnea_fpass_##SFX(const T* q, const T* qd, const T* qdd, T gravity, int64_t B, T* v, T* a, T* f, void* stream) {   \
    RbdStreamDevice sd_(stream); return rnea_pass_fb_launch<T>(1, q, qd, qdd, gravity, B, nullptr, v, a, f, stream);                                     \
  }                                                                                                                         \
  int rbd_rnea_bpass_##SFX(const T* q, T* f, int64_t B, T* c, void* stream) {                                               \
    RbdStreamDevice sd_(stream); return rnea_pass_fb_launch<T>(2, q, nullptr, nullptr, T(0), B, c, nullptr, nullptr, f, stream);                         \
  }                                                                                                                         \
  int rbd_rnea_grad_##SFX(const T* q, const T* qd, const T* qdd, T gravity, int use_damping, int64_t B, T* c, T* dc_du, void* stream) { \
    RbdStreamDevice sd_(stream); return grad_fb_launch<T>("rbd_rnea_grad", q, qd, qdd, gravity, use_damping, B, c, nullptr, nullptr, nullptr, dc_du, stream); \
  }                                                                                                                         \
  int rbd_rnea_with_grad_##SFX(const T* q, const T* qd, const T* qdd, T gravity, int use_damping, int64_t B, T* c, T* v, T* a, T* f, \
                               T* dc_du, void* stream) {                                                                    \
    RbdStreamDevice sd_(stream); return grad_fb_launch<T>("rbd_rnea_with_grad", q, qd, qdd, gravity, use_damping, B, c, v, a, f, dc_du, stream);          \
  }                                                                                                                         \
  int rbd_rnea_grad_fpass_dq_##SFX(const T* q, const T* qd, const T* v, const T* a, T gravity, int64_t B, T* dv, T* da, T* df, void* stream) { \
    RbdStreamDevice sd_(stream); return grad_fpass_fb_launch<T, false>(q, qd, v, a, gravity, B, dv, da, df, stream);                                     \
  }                                                                                                                         \
  int rbd_rnea_grad_fpass_dqd_##SFX(const T* q, const T* qd, const T* v, int64_t B, T* dv, T* da, T* df, void* stream) {    \
    RbdStreamDevice sd_(stream); return grad_fpass_fb_launch<T, true>(q, qd, v, nullptr, T(0), B, dv, da, df, stream);                                   \
  }                                                                                                                         \
  int rbd_rnea_grad_bpass_dq_##SFX(const T* q, const T* f, T* df, int64_t B, T* dc, void* stream) {                         \
    RbdStreamDevice sd_(stream); return grad_bpass_fb_launch<T, false>(q, f, df, 0, B, dc, stream);                                                      \
  }                                                                                                                         \
  int rbd_rnea_grad_bpass_dqd_##SFX(const T* q, T* df, int use_damping, int64_t B, T* dc, void* stream) {                   \
    RbdStreamDevice sd_(stream); return grad_bpass_fb_launch<T, true>(q, nullptr, df, use_damping, B, dc, stream);                                       \
  }                                                                                                                         \
  int rbd_minv_bpass_##SFX(const T* q, int64_t B, T* Minv, T* F, T* U, T* Dinv, void* stream) {                             \
    RbdStreamDevice sd_(stream); return minv_bpass_fb_launch<T>(q, B, Minv, F, U, Dinv, stream);                                                         \
  }                                                                                                                         \
  int rbd_minv_fpass_##SFX(const T* q, int64_t B, T* Minv, T* F, const T* U, const T* Dinv, void* stream) {                 \
    RbdStreamDevice sd_(stream); return minv_fpass_fb_launch<T>(q, B, Minv, F, U, Dinv, stream);                                                         \
  }                                                                                                                         \
  int rbd_crba_##SFX(const T*, int64_t, T*, void*) { return unsupported("rbd_crba"); }                                      \
  int rbd_aba_##SFX(const T*, const T*, const T*, T, int64_t, T*, void*) { return unsupported("rbd_aba"); }                 \
  int rbd_forward_dynamics_grad_##SFX(const T* q, const T* qd, const T* u, T gravity, int64_t B, T* qdd, T* dqdd_du, void* ws, \
                                      size_t wsb, void* stream) {                                                           \
    RbdStreamDevice sd_(stream); return fdg_fb_launch<T>(q, qd, u, gravity, B, qdd, dqdd_du, ws, wsb, stream);                                           \
  }

#ifdef RBD_TU_FB_F32
RBD_FB_DEFS(f32, float)
#endif
#ifdef RBD_TU_FB_F64
RBD_FB_DEFS(f64, double)
#endif
}  // extern "C"
