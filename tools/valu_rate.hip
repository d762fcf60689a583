// Issue rate of the VALU instructions the GELU phases are made of (gfx950): cycles per wave64 instruction, one wave per SIMD
// and two waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate tools/valu_rate.hip && tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s\n", hipGetErrorString(e_)); exit(1); } } while (0)
#define REP8(X) X X X X X X X X
#define BODY(NAME, ASM)                                                                                   \
  __global__ void NAME(float* out, int iters) {                                                            \
    float a = threadIdx.x * 1e-3f + 0.5f, b = a + 0.25f, c = a + 0.5f, d = a + 0.75f;                     \
    for (int i = 0; i < iters; ++i) {                                                                      \
      REP8(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)                                        \
    }                                                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                                            \
  }
// four independent chains per asm block (no dependent-issue stalls): 32 instructions per loop iteration
BODY(k_fma32, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1")
BODY(k_exp32, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3")
BODY(k_rcp32, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3")
BODY(k_exp16, "v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3")
BODY(k_rcp16, "v_rcp_f16 %0, %0\n v_rcp_f16 %1, %1\n v_rcp_f16 %2, %2\n v_rcp_f16 %3, %3")
BODY(k_pkfma16, "v_pk_fma_f16 %0, %0, %1, %2\n v_pk_fma_f16 %1, %1, %2, %3\n v_pk_fma_f16 %2, %2, %3, %0\n v_pk_fma_f16 %3, %3, %0, %1")
typedef __attribute__((ext_vector_type(2))) float f2;
__global__ void k_pkmul32(float* out, int iters) {
  f2 a = {threadIdx.x * 1e-3f + 0.5f, 0.3f}, b = {0.9f, 1.1f}, c = {1.01f, 0.99f}, d = {0.7f, 1.3f};
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + b.y + c.x + d.y;
}
BODY(k_mix, "v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %1\n v_fma_f32 %3, %3, %1, %2")

int main() {
  float* out; CK(hipMalloc(&out, 1 << 22));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const double mhz = p.clockRate / 1e3;
  struct { const char* n; void (*k)(float*, int); int per_asm; } ks[] = {
      {"v_fma_f32", k_fma32, 4}, {"v_exp_f32", k_exp32, 4}, {"v_rcp_f32", k_rcp32, 4}, {"v_exp_f16", k_exp16, 4}, {"v_rcp_f16", k_rcp16, 4},
      {"v_pk_fma_f16", k_pkfma16, 4}, {"v_pk_mul_f32 (two fp32 multiplies per lane)", k_pkmul32, 4},
      {"1 v_exp_f32 + 3 v_fma_f32 interleaved", k_mix, 4}};
  const int iters = 4000;
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int threads = 64 * 4 * waves_per_simd;             // one workgroup per CU
    for (auto& e : ks) {
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, out, 100);
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, out, iters);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const double insts_per_wave = (double)iters * 8 * e.per_asm;
      const double cycles = ms * 1e-3 * mhz * 1e6;          // at the reported clock
      printf("valu_rate %d wave(s)/SIMD  %-70s %.2f cycles per instruction and wave (%.0f MHz nominal)\n", waves_per_simd, e.n,
             cycles / (insts_per_wave * waves_per_simd), mhz);
    }
  }
  return 0;
}
