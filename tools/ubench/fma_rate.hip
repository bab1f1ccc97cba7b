// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (waves per SIMD = 1, 2, 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int ILP>
__global__ void k_fma(float* out, int iters, float a, float b) {
  float x[ILP];
  for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = __builtin_fmaf(x[i], a, b);
  }
  float s = 0;
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_pkfma(float* out, int iters, float a, float b) {
  float2_ x[ILP];
  for (int i = 0; i < ILP; ++i) x[i] = float2_{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  const float2_ A{a, a * 1.0001f}, Bv{b, b * 0.999f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = __builtin_elementwise_fma(x[i], A, Bv);
  }
  float s = 0;
  for (int i = 0; i < ILP; ++i) s += x[i][0] + x[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class K>
float run(K kern, int blocks, int threads, int iters, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
int main() {
  float* out; hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float) * 4);
  const int iters = 20000; constexpr int ILP = 8;
  for (int wps : {1, 2, 4}) {
    int blocks = 256 * 4 * wps;          // one wave per block, wps waves per SIMD
    float t1 = run(k_fma<ILP>, blocks, 64, iters, out);
    float t2 = run(k_pkfma<ILP>, blocks, 64, iters, out);
    double n_inst = (double)iters * ILP;  // instructions per wave
    // cycles per instruction per SIMD assuming 2.4 GHz
    printf("waves/SIMD %d: v_fma_f32 %.3f ms (%.2f TFLOP/s, %.2f ns/inst/wave)  v_pk_fma_f32 %.3f ms (%.2f TFLOP/s)\n", wps,
           t1, 2.0 * n_inst * 64 * blocks / (t1 * 1e-3) / 1e12, t1 * 1e6 / n_inst,
           t2, 4.0 * n_inst * 64 * blocks / (t2 * 1e-3) / 1e12);
  }
  return 0;
}
