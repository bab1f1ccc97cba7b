// Microbenchmark: cycles per v_fmac_f32 when only DEP independent accumulators rotate (DEP = 1: every instruction
// depends on its predecessor), at 1 and 2 waves per SIMD (gfx950).  hipcc --offload-arch=gfx950 -O3 dep_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int DEP>
__global__ void k(unsigned long long* cyc, float* out, int iters) {
  float x[DEP], y = 1.0f + threadIdx.x * 1e-7f, z = 1e-9f * (threadIdx.x + 1);
  for (int i = 0; i < DEP; ++i) x[i] = threadIdx.x * 1e-3f + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 48; ++u) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[u % DEP]) : "v"(y), "v"(z));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0;
  for (int i = 0; i < DEP; ++i) acc += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int DEP>
double run(int wps, int iters, unsigned long long* dcyc, float* dout) {
  const int blocks = 256 * 4 * wps;
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<DEP>, dim3(blocks), dim3(64), 0, 0, dcyc, dout, iters);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  (void)hipMemcpy(h.data(), dcyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  return (double)h[blocks / 2] / ((double)iters * 48);   // median wave: cycles per instruction of ONE wave
}
int main() {
  unsigned long long* dcyc; float* dout;
  (void)hipMalloc(&dcyc, 256 * 4 * 8 * sizeof(unsigned long long)); (void)hipMalloc(&dout, 256 * 4 * 8 * 64 * sizeof(float));
  run<8>(4, 40000, dcyc, dout);   // clock ramp
  for (int wps : {1, 2, 3}) {
    printf("waves/SIMD %d: wave cycles per v_fmac_f32 with 1 / 2 / 3 / 4 / 6 / 8 independent accumulators: %.2f %.2f %.2f %.2f %.2f %.2f\n", wps,
           run<1>(wps, 3000, dcyc, dout), run<2>(wps, 3000, dcyc, dout), run<3>(wps, 3000, dcyc, dout), run<4>(wps, 3000, dcyc, dout),
           run<6>(wps, 3000, dcyc, dout), run<8>(wps, 3000, dcyc, dout));
  }
  return 0;
}
