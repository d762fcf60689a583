// Pure-MFMA ceiling probe for gfx950: what does v_mfma_f32_32x32x16_bf16 sustain on this box
// with NO operand movement at all, for the wave geometries the GEMM kernels use?
//   build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak     run: tools/mfma_peak
// Variants (template V): bit 0: an s_barrier pair around every 8-MFMA cluster (the 8-phase GEMM's phase structure),
// bit 1: the two wave rows one barrier apart (the GEMM's stagger).  Data: zeros or uniform random bf16 (DVFS: the
// clock the chip sustains depends on operand toggling, MI355X_MICROARCH.md "DVFS give-back").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int V>
__global__ __launch_bounds__(512) void mfma_loop(const bf16x8_t* __restrict__ src, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16x8_t a[2][4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[0][i] = src[(i * 64 + lane) & 1023];
    a[1][i] = src[(256 + i * 64 + lane) & 1023];
    b[i] = src[(512 + i * 64 + lane) & 1023];
  }
  f32x16_t acc[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[q][t][e] = 0.f;
  if ((V & 2) && (w >> 2) == 1) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (V & 1) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[q][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][ks], b[ks], acc[q][t], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if (V & 1) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if ((V & 2) && (w >> 2) == 0) __builtin_amdgcn_s_barrier();
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[q][t][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
static double run(const bf16x8_t* src, float* out, int threads, int wgs, int iters, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(mfma_loop<V>, dim3(wgs), dim3(threads), 0, 0, src, out, iters);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(mfma_loop<V>, dim3(wgs), dim3(threads), 0, 0, src, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)reps * wgs * (threads / 64) * (double)iters * 32.0 * (2.0 * 32 * 32 * 16);
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  bf16x8_t* src; float* out;
  hipMalloc(&src, 1024 * sizeof(bf16x8_t));
  hipMalloc(&out, 1024 * 512 * sizeof(float));
  std::vector<unsigned short> h(8192);
  for (int data = 0; data < 2; ++data) {
    for (auto& v : h) {
      if (!data) { v = 0; continue; }
      const float f = (float)rand() / RAND_MAX * 2.f - 1.f;
      unsigned u; memcpy(&u, &f, 4);
      v = (unsigned short)(u >> 16);
    }
    hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice);
    const char* dn = data ? "uniform[-1,1)" : "zeros";
    // iters = 1024 K-step-equivalents: ~ the 8192x2048x2048 tile's work x 8
    printf("%-14s 1 wave/SIMD  (256 thr, 256 WG)  no barriers : %7.1f TF\n", dn, run<0>(src, out, 256, 256, 1024, 20));
    printf("%-14s 2 waves/SIMD (512 thr, 256 WG)  no barriers : %7.1f TF\n", dn, run<0>(src, out, 512, 256, 1024, 20));
    printf("%-14s 2 waves/SIMD (512 thr, 256 WG)  phase barriers: %7.1f TF\n", dn, run<1>(src, out, 512, 256, 1024, 20));
    printf("%-14s 2 waves/SIMD (512 thr, 256 WG)  phase barriers + staggered rows: %7.1f TF\n", dn, run<3>(src, out, 512, 256, 1024, 20));
    printf("%-14s short launch (32 K-tiles, = one 8192x2048x2048 GEMM's MFMAs) staggered: %7.1f TF\n", dn, run<3>(src, out, 512, 256, 32, 50));
  }
  return 0;
}
