// v_cndmask_b32 issue cost: e64 with an SGPR-pair mask, e32 with VCC set once, vs v_mov.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void k(float* out, int iters, float a, unsigned long long m) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  unsigned long long mask = m ^ (unsigned long long)blockIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"(mask));
    } else if (MODE == 1) {
      asm volatile("s_mov_b64 vcc, %9\n" REP8("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"(mask) : "vcc");
    } else if (MODE == 2) {   // DPP mov quad_perm
      asm volatile(REP8("v_mov_b32_dpp %0, %8 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %8 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %8 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %8 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %4, %8 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %8 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %8 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %8 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"(mask));
    } else if (MODE == 3) {   // v_fma with a literal constant operand (v_fmamk)
      asm volatile(REP8("v_fmamk_f32 %0, %0, 0x3f8ccccd, %8\n v_fmamk_f32 %1, %1, 0x3f8ccccd, %8\n v_fmamk_f32 %2, %2, 0x3f8ccccd, %8\n v_fmamk_f32 %3, %3, 0x3f8ccccd, %8\n v_fmamk_f32 %4, %4, 0x3f8ccccd, %8\n v_fmamk_f32 %5, %5, 0x3f8ccccd, %8\n v_fmamk_f32 %6, %6, 0x3f8ccccd, %8\n v_fmamk_f32 %7, %7, 0x3f8ccccd, %8\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"(mask));
    } else if (MODE == 4) {   // v_fmac with an SGPR operand
      asm volatile(REP8("v_fmac_f32_e32 %0, %9, %8\n v_fmac_f32_e32 %1, %9, %8\n v_fmac_f32_e32 %2, %9, %8\n v_fmac_f32_e32 %3, %9, %8\n v_fmac_f32_e32 %4, %9, %8\n v_fmac_f32_e32 %5, %9, %8\n v_fmac_f32_e32 %6, %9, %8\n v_fmac_f32_e32 %7, %9, %8\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"((float)(mask & 7)));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <class K>
float run(K kern, int blocks, int iters, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0x5555aaaa5555aaaaull);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0x5555aaaa5555aaaaull);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
  const int iters = 4000;
  const char* names[] = {"v_cndmask_b32_e64 (sgpr mask)", "v_cndmask_b32_e32 (vcc)", "v_mov_b32_dpp quad_perm", "v_fmamk_f32 (literal)", "v_fmac_f32_e32 (sgpr src)"};
  for (int wps : {1, 2, 4}) {
    int blocks = 256 * 4 * wps;
    double ninst = (double)iters * 64;
    float t[5] = {run(k<0>, blocks, iters, out), run(k<1>, blocks, iters, out), run(k<2>, blocks, iters, out), run(k<3>, blocks, iters, out), run(k<4>, blocks, iters, out)};
    printf("waves/SIMD %d (ns per wave-instruction per SIMD)\n", wps);
    for (int i = 0; i < 5; ++i) printf("   %-32s %.3f\n", names[i], t[i] * 1e6 / ninst / wps);
  }
  return 0;
}
