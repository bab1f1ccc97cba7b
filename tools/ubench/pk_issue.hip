// Microbenchmark: what a PACKED fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) costs a wave
// and the SIMD's vector pipe on gfx950, alone and mixed with scalar-per-lane fp32 instructions.
//   hipcc --offload-arch=gfx950 -O3 pk_issue.hip -o pk_issue && ./pk_issue
// Question behind it: a lone wave issues one VALU instruction per ~4 cycles while the pipe takes one per 2, so a
// kernel with two waves per SIMD whose waves stall part of the time leaves the pipe idle.  A packed instruction
// does two results per issue slot: if it costs the WAVE one slot (and the pipe two), pairing arithmetic by hand
// shortens a wave's issue time without adding pipe time.
// Reported: wave cycles per instruction (s_memtime around the loop; median wave) for 1, 2, 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ILP = 24;
enum { F_FMAC, F_PKFMA, F_PKFMA_BC, F_PKMUL, F_PKADD, F_MIX11, F_MIX13, F_PKDEP1, F_PKDEP2, F_PKDEP4, F_DEP1, F_DEP2, F_DEP4, F_PKFMA_SW, F_FMA3NEG, F_MULNEG, F_MIXREAL, F_CNDMASK, F_MIX71, F_MIX11V, F_MIXV2, F_SUB, F_CMP, F_BFI, F_XOR, F_CNDE64, F_FMA3POS, NFORMS };
template <int FORM>
__global__ __launch_bounds__(64) void k(unsigned long long* cyc, float* out, int iters) {
  f2 x[ILP], y[ILP], z[ILP];
  for (int i = 0; i < ILP; ++i) {
    x[i] = f2{threadIdx.x * 1e-3f + i, 1.0f + i};
    y[i] = f2{1.0f + threadIdx.x * 1e-7f * i, 0.999f};
    z[i] = f2{1e-9f * (i + 1), 2e-9f};
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      if constexpr (FORM == F_FMAC) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i].x) : "v"(y[i].x), "v"(z[i].x)); }
      else if constexpr (FORM == F_PKFMA) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y[i]), "v"(z[i])); }
      else if constexpr (FORM == F_PKFMA_BC) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(x[i]) : "v"(y[i]), "v"(z[i])); }
      else if constexpr (FORM == F_PKFMA_SW) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(x[i]) : "v"(y[i]), "v"(z[i])); }
      else if constexpr (FORM == F_PKMUL) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i])); }
      else if constexpr (FORM == F_PKADD) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(z[i])); }
      else if constexpr (FORM == F_MIX11) {
        if (i % 2 == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y[i]), "v"(z[i]));
        else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i].x) : "v"(y[i].x), "v"(z[i].x));
      } else if constexpr (FORM == F_MIX13) {
        if (i % 4 == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y[i]), "v"(z[i]));
        else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i].x) : "v"(y[i].x), "v"(z[i].x));
      } else if constexpr (FORM == F_PKDEP1) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[0]) : "v"(y[i]), "v"(z[i])); }
      else if constexpr (FORM == F_PKDEP2) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i % 2]) : "v"(y[i]), "v"(z[i])); }
      else if constexpr (FORM == F_PKDEP4) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i % 4]) : "v"(y[i]), "v"(z[i])); }
      else if constexpr (FORM == F_DEP1) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[0].x) : "v"(y[i].x), "v"(z[i].x)); }
      else if constexpr (FORM == F_DEP2) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i % 2].x) : "v"(y[i].x), "v"(z[i].x)); }
      else if constexpr (FORM == F_FMA3NEG) { asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(x[i].x) : "v"(y[i].x), "v"(z[i].x)); }
      else if constexpr (FORM == F_MULNEG) { asm volatile("v_mul_f32_e64 %0, %1, -%2" : "=v"(x[i].x) : "v"(y[i].x), "v"(z[i].x)); }
      else if constexpr (FORM == F_MIXREAL) {   // the headline kernel's mix: 2 fmac : 1 mul(neg, VOP3) : 1 fma(neg, VOP3) roughly 42 : 11 : 21 : plain mul 12 : add 10
        if (i % 8 < 3) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i].x) : "v"(y[i].x), "v"(z[i].x));
        else if (i % 8 == 3) asm volatile("v_mul_f32_e64 %0, %1, -%2" : "=v"(x[i].x) : "v"(y[i].x), "v"(z[i].x));
        else if (i % 8 < 6) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(x[i].x) : "v"(y[i].x), "v"(z[i].x));
        else if (i % 8 == 6) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[i].x) : "v"(y[i].x), "v"(z[i].x));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i].x) : "v"(z[i].x));
      }
      else if constexpr (FORM == F_CNDMASK) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[i].x) : "v"(y[i].x), "v"(z[i].x) : ); }
      else if constexpr (FORM == F_MIX71) {
        if (i % 8 == 7) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(x[i].x) : "v"(y[i].x), "v"(z[i].x));
        else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i].x) : "v"(y[i].x), "v"(z[i].x));
      } else if constexpr (FORM == F_MIX11V) {
        if (i % 2 == 1) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(x[i].x) : "v"(y[i].x), "v"(z[i].x));
        else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i].x) : "v"(y[i].x), "v"(z[i].x));
      } else if constexpr (FORM == F_MIXV2) {
        if (i % 3 == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i].x) : "v"(y[i].x), "v"(z[i].x));
        else if (i % 3 == 1) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[i].x) : "v"(y[i].x), "v"(z[i].x));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i].x) : "v"(z[i].x));
      }
      else if constexpr (FORM == F_SUB) { asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i].x) : "v"(z[i].x)); }
      else if constexpr (FORM == F_CMP) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i].x), "v"(z[i].x) : "vcc"); }
      else if constexpr (FORM == F_BFI) { asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(x[i].x) : "v"(y[i].x), "v"(z[i].x)); }
      else if constexpr (FORM == F_XOR) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i].x) : "v"(z[i].x)); }
      else if constexpr (FORM == F_CNDE64) { asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[10:11]" : "=v"(x[i].x) : "v"(y[i].x), "v"(z[i].x) : ); }
      else if constexpr (FORM == F_FMA3POS) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i].x) : "v"(y[i].x), "v"(z[i].x), "v"(y[(i + 5) % ILP].x)); }
      else if constexpr (FORM == F_DEP4) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i % 4].x) : "v"(y[i].x), "v"(z[i].x)); }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0;
  for (int i = 0; i < ILP; ++i) acc += x[i].x + x[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int FORM>
double run(int wps, int iters, unsigned long long* dcyc, float* dout) {
  const int blocks = 256 * 4 * wps;
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(64), 0, 0, dcyc, dout, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), dcyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  return (double)h[blocks / 2] / ((double)iters * ILP);   // median wave: cycles per instruction of ONE wave
}
template <int F>
void all(const char* const* names, int iters, unsigned long long* dcyc, float* dout, double (*res)[4]) {
  if constexpr (F < NFORMS) {
    for (int wps = 1; wps <= 3; ++wps) res[F][wps] = run<F>(wps, iters, dcyc, dout);
    all<F + 1>(names, iters, dcyc, dout, res);
  }
}
int main() {
  unsigned long long* dcyc; float* dout;
  hipMalloc(&dcyc, 256 * 4 * 8 * sizeof(unsigned long long)); hipMalloc(&dout, 256 * 4 * 8 * 64 * sizeof(float));
  run<F_FMAC>(4, 60000, dcyc, dout);   // clock ramp
  const char* names[NFORMS] = {"fmac (independent)", "pk_fma (independent)", "pk_fma op_sel_hi:[0,1,1] (broadcast lo)", "pk_mul", "pk_add",
                               "1 pk_fma : 1 fmac", "1 pk_fma : 3 fmac", "pk_fma chain on 1 acc", "pk_fma chain on 2 acc", "pk_fma chain on 4 acc",
                               "fmac chain on 1 acc", "fmac chain on 2 acc", "fmac chain on 4 acc", "pk_fma op_sel swap+broadcast",
                               "v_fma_f32 with neg (VOP3)", "v_mul_f32_e64 with neg (VOP3)", "mix 3 fmac:1 mul-neg:2 fma-neg:1 mul:1 add", "v_cndmask_b32 vcc (VOP2)",
                               "mix 7 fmac : 1 fma-neg", "mix 1 fmac : 1 fma-neg", "mix fmac:mul:add (all VOP2)", "v_sub_f32 (VOP2)", "v_cmp_lt_f32 vcc", "v_bfi_b32 (VOP3)", "v_xor_b32 (VOP2)",
                               "v_cndmask_b32_e64 sgpr pair", "v_fma_f32 d=a*b+c, 4 distinct regs"};
  double res[NFORMS][4];
  all<0>(names, 3000, dcyc, dout, res);
  printf("%-42s %10s %10s %10s   (wave cycles per instruction; per-SIMD pipe time = that / waves)\n", "form", "1 wave", "2 waves", "3 waves");
  for (int f = 0; f < NFORMS; ++f) printf("%-42s %10.2f %10.2f %10.2f\n", names[f], res[f][1], res[f][2], res[f][3]);
  return 0;
}
