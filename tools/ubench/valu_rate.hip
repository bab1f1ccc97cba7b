// Microbenchmark: per-instruction issue cost of fp32 VALU forms on gfx950 at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {        // v_fmac_f32_e32: x += a*b  (VOP2)
      asm volatile(REP8("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 1) { // v_fma_f32 (VOP3): x = x*a + b
      asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 2) { // v_mul_f32_e32
      asm volatile(REP8("v_mul_f32_e32 %0, %0, %8\n v_mul_f32_e32 %1, %1, %8\n v_mul_f32_e32 %2, %2, %8\n v_mul_f32_e32 %3, %3, %8\n v_mul_f32_e32 %4, %4, %8\n v_mul_f32_e32 %5, %5, %8\n v_mul_f32_e32 %6, %6, %8\n v_mul_f32_e32 %7, %7, %8\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 3) { // v_add_f32_e32
      asm volatile(REP8("v_add_f32_e32 %0, %0, %8\n v_add_f32_e32 %1, %1, %8\n v_add_f32_e32 %2, %2, %8\n v_add_f32_e32 %3, %3, %8\n v_add_f32_e32 %4, %4, %8\n v_add_f32_e32 %5, %5, %8\n v_add_f32_e32 %6, %6, %8\n v_add_f32_e32 %7, %7, %8\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 4) { // v_mov_b32
      asm volatile(REP8("v_mov_b32_e32 %0, %8\n v_mov_b32_e32 %1, %8\n v_mov_b32_e32 %2, %8\n v_mov_b32_e32 %3, %8\n v_mov_b32_e32 %4, %8\n v_mov_b32_e32 %5, %8\n v_mov_b32_e32 %6, %8\n v_mov_b32_e32 %7, %8\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 5) { // v_cndmask_b32
      asm volatile(REP8("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE>
__global__ void kpk(float* out, int iters, float a, float b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 x0{(float)threadIdx.x, 1}, x1{2, 3}, x2{4, 5}, x3{6, 7}, x4{8, 9}, x5{1, 2}, x6{3, 4}, x7{5, 6};
  f2 A{a, a}, Bv{b, b};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(A), "v"(Bv));
    } else if (MODE == 1) {
      asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(A), "v"(Bv));
    } else {
      asm volatile(REP8("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(A), "v"(Bv));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0[0] + x1[1] + x2[0] + x3[1] + x4[0] + x5[1] + x6[0] + x7[1];
}
template <class K>
float run(K kern, int blocks, int iters, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
  const int iters = 4000;               // x 64 instructions per iteration
  const char* names[] = {"v_fmac_f32_e32", "v_fma_f32(VOP3)", "v_mul_f32_e32", "v_add_f32_e32", "v_mov_b32", "v_cndmask_b32"};
  const char* pkn[] = {"v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"};
  for (int wps : {1, 2, 4, 8}) {
    int blocks = 256 * 4 * wps;
    double ninst = (double)iters * 64;
    float t[6] = {run(k<0>, blocks, iters, out), run(k<1>, blocks, iters, out), run(k<2>, blocks, iters, out),
                  run(k<3>, blocks, iters, out), run(k<4>, blocks, iters, out), run(k<5>, blocks, iters, out)};
    float p[3] = {run(kpk<0>, blocks, iters, out), run(kpk<1>, blocks, iters, out), run(kpk<2>, blocks, iters, out)};
    printf("waves/SIMD %d  (ns per wave-instruction per SIMD; x2.4 = cycles at 2.4 GHz)\n", wps);
    for (int i = 0; i < 6; ++i) printf("   %-18s %.3f\n", names[i], t[i] * 1e6 / ninst / wps);
    for (int i = 0; i < 3; ++i) printf("   %-18s %.3f\n", pkn[i], p[i] * 1e6 / ninst / wps);
  }
  return 0;
}
