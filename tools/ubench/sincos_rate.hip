// Cost of sincosf vs a Cody-Waite + polynomial sincos on gfx950 (ns per call per lane-wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__device__ __forceinline__ void fast_sincos(float q, float* s, float* c) {
  // k = nearest integer to q * 2/pi ; r = q - k*pi/2 in three steps (Cody-Waite), |r| <= pi/4
  const float k = __builtin_rintf(q * 0.63661977236758134f);
  float r = __builtin_fmaf(k, -1.5707962513e+00f, q);
  r = __builtin_fmaf(k, -7.5497894159e-08f, r);
  r = __builtin_fmaf(k, -5.3903029534e-15f, r);
  const float r2 = r * r;
  // minimax polynomials on [-pi/4, pi/4] (cephes sinf/cosf)
  float sp = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = __builtin_fmaf(sp, r2, -1.6666654611e-1f);
  sp = __builtin_fmaf(sp * r2, r, r);
  float cp = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = __builtin_fmaf(cp, r2, 4.166664568298827e-2f);
  cp = __builtin_fmaf(cp * r2, r2, __builtin_fmaf(r2, -0.5f, 1.0f));
  const int ki = (int)k;
  const bool swap = (ki & 1) != 0;
  float ss = swap ? cp : sp;
  float cc = swap ? sp : cp;
  // sign: sin negative for quadrants 2,3 ; cos negative for quadrants 1,2
  ss = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, ss) ^ ((unsigned)(ki & 2) << 30));
  cc = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, cc) ^ ((unsigned)((ki + 1) & 2) << 30));
  *s = ss; *c = cc;
}
template <int MODE>
__global__ void k(float* out, int iters, float a) {
  float q = (threadIdx.x + blockIdx.x * 64) * 0.001f - 3.0f, acc = 0;
  for (int it = 0; it < iters; ++it) {
    float s, c;
    if (MODE == 0) sincosf(q, &s, &c); else fast_sincos(q, &s, &c);
    acc += s * 0.5f + c;
    q += a;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void check(float* err, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float worst = 0;
  for (int j = i; j < n; j += gridDim.x * blockDim.x) {
    float q = -1000.0f + 2000.0f * (float)j / n;
    float s, c; fast_sincos(q, &s, &c);
    double sr = sin((double)q), cr = cos((double)q);
    worst = fmaxf(worst, fmaxf(fabsf((float)(s - sr)), fabsf((float)(c - cr))));
  }
  err[i] = worst;
}
template <class K>
float run(K kern, int blocks, int iters, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, 0.01f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, 0.01f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
  const int iters = 20000;
  for (int wps : {2, 4}) {
    int blocks = 256 * 4 * wps;
    float t0 = run(k<0>, blocks, iters, out), t1 = run(k<1>, blocks, iters, out);
    printf("waves/SIMD %d: sincosf %.2f ns per call per SIMD, fast_sincos %.2f ns\n", wps, t0 * 1e6 / iters / wps, t1 * 1e6 / iters / wps);
  }
  hipLaunchKernelGGL(check, dim3(256), dim3(256), 0, 0, out, 1 << 24);
  hipDeviceSynchronize();
  static float h[65536]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  float w = 0; for (float x : h) w = fmaxf(w, x);
  printf("fast_sincos max abs error on [-1000, 1000] (16.7M samples, vs fp64): %.3e\n", w);
  return 0;
}
