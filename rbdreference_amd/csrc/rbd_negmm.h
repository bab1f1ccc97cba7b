// rbd_negmm.h -- out [B, NV, 2 NV] = -Minv [B, NV, NV] . dc_du [B, NV, 2 NV]: the product that closes forward_dynamics_grad
// (/root/reference/RBDReference.py:1381-1383) wherever it cannot be folded into the gradient kernel's epilogue (trees whose
// rows leave the gradient kernel one at a time; the floating base).  It moves 5 NV^2 scalars per configuration and does
// 2 NV^3 FMAs: at NV = 18, fp32 that is 6.5 KB against 11.7 k FMAs -- an HBM-bound kernel.
//
// A thread owns VE = 16 / sizeof(T) CONSECUTIVE columns of one configuration: every global access is a 16-byte piece of a
// row (round 3: one 4-byte load per thread and row), the NV x VE accumulators stay in registers, and Minv comes from an LDS
// copy the block staged with flat 16-byte loads -- read ROW-wise: Minv is symmetric (:799-804), so the column of Minv that
// multiplies row k of dc_du is row k of Minv, contiguous, two entries per LDS read for 2 VE FMAs each.
// (Round 4 also tried dc_du and the output through LDS as flat copies, VERDICT r3's suggestion: 122-129 us against the
// 58 / 108 us of round 3's kernels for the 30-body robot / the floating quadruped -- 11-33 KB of LDS per block leave 1.6-3.7
// waves per SIMD, and a block's copy-in, multiply, copy-out are serial; profiles/r04_negmm.txt.)
#pragma once
#include <hip/hip_runtime.h>

namespace rbdk {

template <class T, int NV_>
constexpr int negmm_ve() { return (2 * NV_) % (16 / (int)sizeof(T)) == 0 ? 16 / (int)sizeof(T) : 1; }
template <class T, int NV_>
constexpr int negmm_tpc() { return 2 * NV_ / negmm_ve<T, NV_>(); }                  // threads per configuration
template <class T, int NV_>
constexpr int negmm_cfgs() { return 256 / negmm_tpc<T, NV_>() > 0 ? 256 / negmm_tpc<T, NV_>() : 1; }
template <class T, int NV_>
constexpr int negmm_threads() { return (negmm_cfgs<T, NV_>() * negmm_tpc<T, NV_>() + 63) / 64 * 64; }

template <class T, int NV_>
__global__ __launch_bounds__((negmm_threads<T, NV_>())) void neg_mm_kernel(const T* __restrict__ Minv, const T* __restrict__ dc, long long B,
                                                                          T* __restrict__ out) {
  constexpr int C = negmm_cfgs<T, NV_>(), TPB = negmm_threads<T, NV_>(), VE = negmm_ve<T, NV_>(), TPC = negmm_tpc<T, NV_>();
  constexpr int MM = NV_ * NV_;
  constexpr int CV = 16 / (int)sizeof(T);                       // scalars per 16-byte copy piece
  typedef T V __attribute__((ext_vector_type(VE)));
  typedef T VC __attribute__((ext_vector_type(CV)));
  typedef T V2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) T Ms[C * MM];
  const long long cfg0 = (long long)blockIdx.x * C;
  const long long rem = B - cfg0;
  const int nvalid = rem < C ? (int)rem : C;
  const int tid = threadIdx.x;
  const T* msrc = Minv + cfg0 * MM;
  if constexpr (MM % CV == 0) {      // every configuration's Minv starts on a 16-byte boundary: flat copies, all loads issued first
    constexpr int NM = (C * (MM / CV) + TPB - 1) / TPB;
    VC bm[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) { const int g = tid + i * TPB; bm[i] = reinterpret_cast<const VC*>(msrc)[g < nvalid * (MM / CV) ? g : 0]; }
#pragma unroll
    for (int i = 0; i < NM; ++i) { const int g = tid + i * TPB; if (g < C * (MM / CV)) reinterpret_cast<VC*>(Ms)[g] = bm[i]; }
  } else {
    for (int g = tid; g < nvalid * MM; g += TPB) Ms[g] = msrc[g];
  }
  __syncthreads();
  const int cl = tid / TPC, cv = tid - cl * TPC;
  if (cl >= nvalid) return;
  const V* D = reinterpret_cast<const V*>(dc + (cfg0 + cl) * (2LL * MM)) + cv;      // row k: D[k * TPC]
  V* O = reinterpret_cast<V*>(out + (cfg0 + cl) * (2LL * MM)) + cv;
  const T* M = Ms + cl * MM;
  V acc[NV_];
#pragma unroll
  for (int r = 0; r < NV_; ++r) acc[r] = V(T(0));
  constexpr int UNR = NV_ <= 20 ? 6 : 2;        // (big matrices: the unrolled rows' LDS reads are hoisted -- 286 VGPRs at NV = 30 with 6)
#pragma unroll UNR
  for (int k = 0; k < NV_; ++k) {
    const V d = D[k * TPC];
    if constexpr (NV_ % 2 == 0) {
#pragma unroll
      for (int r = 0; r < NV_; r += 2) {
        const V2 m = *reinterpret_cast<const V2*>(M + k * NV_ + r);          // M[k][r] == M[r][k]
        acc[r] -= m[0] * d;
        acc[r + 1] -= m[1] * d;
      }
    } else {
#pragma unroll
      for (int r = 0; r < NV_; ++r) acc[r] -= M[k * NV_ + r] * d;
    }
  }
#pragma unroll
  for (int r = 0; r < NV_; ++r) O[r * TPC] = acc[r];
}

}  // namespace rbdk
