// Microbenchmark: SIMD cycles per wave64 VALU instruction by instruction form and by waves per SIMD (gfx950).
//   hipcc --offload-arch=gfx950 -O3 issue_mix.hip -o issue_mix && ./issue_mix
// Every form is forced with inline asm (32 independent accumulators, 32 instructions per loop iteration); cycles are
// core clocks from s_memtime around the loop (clock-independent), reported per instruction per SIMD =
// wave cycles / (instructions per wave * waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int ILP = 32;
#define FORM_MUL 0    // v_mul_f32   x = x * y          2 VGPR reads
#define FORM_ADD 1    // v_add_f32   x = x + y          2 VGPR reads
#define FORM_FMAC3 2  // v_fmac_f32  x += y * z         3 VGPR reads (VOP2)
#define FORM_FMACK 3  // v_fmac_f32  x += y * s         2 VGPR reads + SGPR
#define FORM_FMA3 4   // v_fma_f32   x = y * z + w      3 VGPR reads (VOP3), w != x
#define FORM_FMACLIT 5  // v_fmac_f32  x += LITERAL * z   32-bit literal constant
#define FORM_MULLIT 6   // v_mul_f32   x = LITERAL * x
#define FORM_MULS 7     // v_mul_f32   x = s * x          SGPR operand
#define FORM_FMAMK 8    // v_fmamk_f32 x = z * K + x
#define FORM_MULINL 9   // v_mul_f32   x = 0.5 * x        inline constant
template <int FORM>
__global__ void k(unsigned long long* cyc, float* out, int iters, float s) {
  float x[ILP], y[ILP], z[ILP];
  for (int i = 0; i < ILP; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = 1.0f + threadIdx.x * 1e-7f * i; z[i] = 1e-9f * (i + 1); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      if constexpr (FORM == FORM_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
      else if constexpr (FORM == FORM_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(z[i]));
      else if constexpr (FORM == FORM_FMAC3) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(z[i]));
      else if constexpr (FORM == FORM_FMACK) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "s"(s), "v"(z[i]));
      else if constexpr (FORM == FORM_FMA3) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(y[i]), "v"(z[i]), "v"(y[(i + 7) % ILP]));
      else if constexpr (FORM == FORM_FMACLIT) asm volatile("v_fmac_f32 %0, 0x3f7fbe77, %1" : "+v"(x[i]) : "v"(z[i]));
      else if constexpr (FORM == FORM_MULLIT) asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0" : "+v"(x[i]));
      else if constexpr (FORM == FORM_MULS) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "s"(s));
      else if constexpr (FORM == FORM_FMAMK) asm volatile("v_fmamk_f32 %0, %1, 0x3f7fbe77, %0" : "+v"(x[i]) : "v"(z[i]));
      else asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(x[i]));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0;
  for (int i = 0; i < ILP; ++i) acc += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int FORM>
double run(int wps, int iters, unsigned long long* dcyc, float* dout) {
  const int blocks = 256 * 4 * wps;
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(64), 0, 0, dcyc, dout, iters, 0.999f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), dcyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  return (double)h[blocks / 2] / ((double)iters * ILP * wps);   // median wave, cycles per instruction per SIMD
}
int main() {
  unsigned long long* dcyc; float* dout;
  hipMalloc(&dcyc, 256 * 4 * 8 * sizeof(unsigned long long)); hipMalloc(&dout, 256 * 4 * 8 * 64 * sizeof(float));
  const int iters = 4000;
  run<FORM_MUL>(4, 40000, dcyc, dout);   // clock ramp
  const char* names[10] = {"mul vv", "add vv", "fmac vvv", "fmac s,v,acc", "fma3 vvv", "fmac lit,v,acc", "mul lit,v", "mul s,v", "fmamk v,K,acc", "mul 0.5,v"};
  for (int wps : {1, 2, 3, 4}) {
    const double c[10] = {run<FORM_MUL>(wps, iters, dcyc, dout), run<FORM_ADD>(wps, iters, dcyc, dout), run<FORM_FMAC3>(wps, iters, dcyc, dout),
                          run<FORM_FMACK>(wps, iters, dcyc, dout), run<FORM_FMA3>(wps, iters, dcyc, dout), run<FORM_FMACLIT>(wps, iters, dcyc, dout),
                          run<FORM_MULLIT>(wps, iters, dcyc, dout), run<FORM_MULS>(wps, iters, dcyc, dout), run<FORM_FMAMK>(wps, iters, dcyc, dout),
                          run<FORM_MULINL>(wps, iters, dcyc, dout)};
    printf("waves/SIMD %d:", wps);
    for (int i = 0; i < 10; ++i) printf("  %s %.2f", names[i], c[i]);
    printf("\n");
  }
  return 0;
}
