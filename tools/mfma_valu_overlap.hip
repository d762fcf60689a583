// Do the matrix pipe and the VALU of one SIMD overlap when they are fed by DIFFERENT waves, by ONE wave with independent work
// interleaved, or not at all?  One workgroup of 8 waves per CU (two per SIMD: waves w and w + 4 share a SIMD).  Work items:
//   M = a run of v_mfma_f32_16x16x32_bf16 on four independent accumulators, V = a run of the GELU-like VALU mix
//   (1 v_exp_f32 + 3 v_fma_f32, four independent chains), sized to take about the same time alone.
// Modes: M alone (waves 0-3), V alone (waves 4-7), M beside V (different waves of a SIMD), both waves M then V in lockstep
// (the phase-synchronous form of the encoder kernels), both waves V then M / M then V (de-phased by one phase), one wave doing
// both interleaved 1 MFMA : 8 VALU (the software-pipelined form); M only / V only split over both waves (the rates two waves reach).
//   hipcc --offload-arch=gfx950 -O3 [-DPURE_FMA] [-DM32] -o tools/mfma_valu_overlap tools/mfma_valu_overlap.hip && tools/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s\n", hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct St { f32x4 c0, c1, c2, c3; f32x16 d0, d1; bf16x8 a, b; float x0, x1, x2, x3, y0, y1, y2, y3, k1, k2; };

#ifdef M32             // -DM32: the same flops as v_mfma_f32_32x32x16_bf16 (two per four 16x16x32)
#define MNAME "v_mfma_f32_32x32x16_bf16"
__device__ __forceinline__ void run_m(St& s, int n) {
  for (int i = 0; i < n; ++i) {
    s.d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.a, s.b, s.d0, 0, 0, 0);
    s.d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.a, s.b, s.d1, 0, 0, 0);
  }
}
#else
#define MNAME "v_mfma_f32_16x16x32_bf16"
__device__ __forceinline__ void run_m(St& s, int n) {          // n x 4 MFMAs
  for (int i = 0; i < n; ++i) {
    s.c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s.a, s.b, s.c0, 0, 0, 0);
    s.c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s.a, s.b, s.c1, 0, 0, 0);
    s.c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s.a, s.b, s.c2, 0, 0, 0);
    s.c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s.a, s.b, s.c3, 0, 0, 0);
  }
}
#endif
// eight INDEPENDENT chains (each instruction depends only on its own result eight instructions earlier): issue-bound, not
// latency-bound, already with one wave per SIMD
#ifdef PURE_FMA        // -DPURE_FMA: the same count of plain v_fma_f32, no transcendental
#define VMIX(s) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"   \
                             : "+v"(s.x0), "+v"(s.x1), "+v"(s.x2), "+v"(s.x3), "+v"(s.y0), "+v"(s.y1), "+v"(s.y2), "+v"(s.y3) : "v"(s.k1), "v"(s.k2))
#define VNAME "8 independent v_fma_f32"
#else
#define VMIX(s) asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
                             "v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"   \
                             : "+v"(s.x0), "+v"(s.x1), "+v"(s.x2), "+v"(s.x3), "+v"(s.y0), "+v"(s.y1), "+v"(s.y2), "+v"(s.y3) : "v"(s.k1), "v"(s.k2))
#define VNAME "2 v_exp_f32 + 6 v_fma_f32, independent"
#endif
__device__ __forceinline__ void run_v(St& s, int n) {          // n x 32 VALU instructions
  for (int i = 0; i < n; ++i) { VMIX(s); VMIX(s); VMIX(s); VMIX(s); }
}
__device__ __forceinline__ void run_mv(St& s, int n) {         // n x (the MFMAs of run_m with the VALU instructions of run_v spread behind them)
  for (int i = 0; i < n; ++i) {
#ifdef M32
    s.d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.a, s.b, s.d0, 0, 0, 0); VMIX(s); VMIX(s);
    s.d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.a, s.b, s.d1, 0, 0, 0); VMIX(s); VMIX(s);
#else
    s.c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s.a, s.b, s.c0, 0, 0, 0); VMIX(s);
    s.c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s.a, s.b, s.c1, 0, 0, 0); VMIX(s);
    s.c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s.a, s.b, s.c2, 0, 0, 0); VMIX(s);
    s.c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s.a, s.b, s.c3, 0, 0, 0); VMIX(s);
#endif
  }
}

// mode: 0 M alone, 1 V alone, 2 M beside V, 3 lockstep M;V x phases, 4 de-phased (first group M;V, second V;M), 5 one-wave interleave
__global__ __launch_bounds__(512) void k(float* out, int mode, int phases, int nm, int nv) {
  St s;
  const int w = threadIdx.x >> 6, grp = w >> 2;
  for (int e = 0; e < 8; ++e) { s.a[e] = (__bf16)(0.001f * (threadIdx.x & 7)); s.b[e] = (__bf16)0.002f; }
  for (int e = 0; e < 4; ++e) s.c0[e] = s.c1[e] = s.c2[e] = s.c3[e] = 0.f;
  for (int e = 0; e < 16; ++e) s.d0[e] = s.d1[e] = 0.f;
  s.x0 = threadIdx.x * 1e-3f + 0.5f; s.x1 = s.x0 + 0.25f; s.x2 = s.x0 + 0.5f; s.x3 = s.x0 + 0.75f;
  s.y0 = s.x0 * 0.5f; s.y1 = s.x1 * 0.5f; s.y2 = s.x2 * 0.5f; s.y3 = s.x3 * 0.5f; s.k1 = 0.999f; s.k2 = 1e-4f;
  for (int p = 0; p < phases; ++p) {
    if (mode == 0) { if (grp == 0) run_m(s, nm); }
    else if (mode == 1) { if (grp == 1) run_v(s, nv); }
    else if (mode == 2) { if (grp == 0) run_m(s, nm); else run_v(s, nv); }
    else if (mode == 3) { run_m(s, nm / 2); __syncthreads(); run_v(s, nv / 2); __syncthreads(); }
    else if (mode == 4) {
      if (grp == 0) { run_m(s, nm / 2); __syncthreads(); run_v(s, nv / 2); __syncthreads(); }
      else { run_v(s, nv / 2); __syncthreads(); run_m(s, nm / 2); __syncthreads(); }
    } else if (mode == 5) { run_mv(s, nm / 2); }
    else if (mode == 6) { run_m(s, nm / 2); }
    else if (mode == 7) { run_v(s, nv / 2); }
    else {        // 8 / 9 / 10: whole CUs -- even workgroups M, odd workgroups V (8: only the even ones work, 9: only the odd ones)
      const bool even = (blockIdx.x & 1) == 0;
      if (even && mode != 9) run_m(s, nm / 2);
      if (!even && mode != 8) run_v(s, nv / 2);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.c0[0] + s.c1[1] + s.c2[2] + s.c3[3] + s.d0[5] + s.d1[9] + s.x0 + s.x1 + s.x2 + s.x3 + s.y0 + s.y1 + s.y2 + s.y3;
}

int main() {
  float* out; CK(hipMalloc(&out, 1 << 22));
  const int phases = 200, nm = 32, nv = 32;             // per phase: 128 MFMAs / 1024 VALU instructions per wave
  const char* names[] = {"M alone (one wave per SIMD)", "V alone (one wave per SIMD)", "M beside V (two waves per SIMD, one each)",
                         "both waves: M/2 ; barrier ; V/2 ; barrier (lockstep)", "first wave M/2 ; V/2, second wave V/2 ; M/2 (de-phased)",
                         "both waves: 1 MFMA + 8 VALU interleaved (M/2 + V/2 each)", "M only, split over both waves (M/2 each)",
                         "V only, split over both waves (V/2 each)", "half of the CUs: M (the other half idle)",
                         "half of the CUs: V (the other half idle)", "half of the CUs M, the other half V, at the same time"};
  float t[11];
  for (int mode = 0; mode < 11; ++mode) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode, 5, nm, nv);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode, phases, nm, nv);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&t[mode], a, b));
    printf("mfma_valu_overlap %-62s %8.1f us\n", names[mode], t[mode] * 1e3);
  }
  printf("mfma_valu_overlap matrix instruction = %s, VALU block = %s\n", MNAME, VNAME);
  printf("mfma_valu_overlap the same total work per SIMD in modes 3-6: M (two waves) + V (two waves) = %.1f us, max of the two %.1f us\n",
         (t[6] + t[7]) * 1e3, (t[6] > t[7] ? t[6] : t[7]) * 1e3);
  return 0;
}
