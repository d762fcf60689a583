// Which workgroups can share a CU with a long-running workgroup that holds a large LDS allocation?  (round 6)
// In the two-stream train step a main-stream LayerNorm backward (32 KiB of LDS) launched behind the side stream's four-problem
// weight-gradient GEMM (256 workgroups x 128 KiB, each resident for the kernel's ~290 us) does not start before that GEMM ends
// (profiles/r6g_train_stream_busy.txt), and tools/corun_probe.py shows the same for a kernel with 64 BYTES of LDS while an LDS-free
// kernel runs beside it.  This probe maps the rule with synthetic kernels: an occupier (256 workgroups x 512 threads, B bytes of LDS,
// spinning ~300 us) on a low-priority stream and a victim (1024 workgroups x 256 threads, V bytes of LDS, with or without a
// barrier) on a second stream, launched ~20 us later; printed: the victim's completion time after its launch.
//   hipcc --offload-arch=gfx950 -O2 -o tools/lds_corun_probe tools/lds_corun_probe.hip && tools/lds_corun_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

extern __shared__ unsigned char dyn_lds[];

__global__ __launch_bounds__(512) void occupier(unsigned long long ticks, int lds_bytes, int* sink) {
  if (lds_bytes > 0) dyn_lds[threadIdx.x % lds_bytes] = (unsigned char)threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (lds_bytes > 0 && dyn_lds[0] == 255 && threadIdx.x == 9999) *sink = 1;
}

__global__ __launch_bounds__(256) void victim(int lds_bytes, int use_barrier, float* out) {
  float v = (float)threadIdx.x;
  if (lds_bytes > 0) dyn_lds[threadIdx.x % lds_bytes] = (unsigned char)threadIdx.x;
  if (use_barrier) __syncthreads();
  if (lds_bytes > 0) v += (float)dyn_lds[(threadIdx.x * 7) % lds_bytes];
  for (int i = 0; i < 64; ++i) v = v * 1.0001f + 0.5f;
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t side, mainst;
  CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo));
  CK(hipStreamCreateWithPriority(&mainst, hipStreamNonBlocking, hi));
  float* out; int* sink;
  CK(hipMalloc(&out, 1024 * 256 * sizeof(float)));
  CK(hipMalloc(&sink, sizeof(int)));
  CK(hipFuncSetAttribute((const void*)occupier, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)victim, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipEvent_t e0, e1, s0, s1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&s0)); CK(hipEventCreate(&s1));
  const int occ_lds[] = {0, 32 * 1024, 64 * 1024, 65 * 1024, 96 * 1024, 128 * 1024};
  const int vic_lds[] = {0, 64, 16 * 1024, 32 * 1024};
  printf("victim completion time in us after its launch (alone: first column), occupier = 256 workgroups x 512 threads spinning 300 us\n");
  printf("%-34s %8s", "victim \\ occupier LDS", "alone");
  for (int ol : occ_lds) printf(" %7dK", ol / 1024);
  printf("\n");
  for (int bar = 0; bar < 2; ++bar)
    for (int vl : vic_lds) {
      char name[64];
      snprintf(name, sizeof(name), "LDS %6d B, %s", vl, bar ? "barrier" : "no barrier");
      printf("%-34s", name);
      for (int oi = -1; oi < (int)(sizeof(occ_lds) / sizeof(int)); ++oi) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipDeviceSynchronize());
          if (oi >= 0) {
            CK(hipEventRecord(s0, side));
            hipLaunchKernelGGL(occupier, dim3(256), dim3(512), occ_lds[oi], side, 30000ull, occ_lds[oi], sink);
            CK(hipEventRecord(s1, side));
            hipLaunchKernelGGL(occupier, dim3(1), dim3(64), 0, mainst, 2000ull, 0, sink);      // ~20 us spacer on the main stream
          }
          CK(hipEventRecord(e0, mainst));
          hipLaunchKernelGGL(victim, dim3(1024), dim3(256), vl, mainst, vl, bar, out);
          CK(hipEventRecord(e1, mainst));
          CK(hipDeviceSynchronize());
          float ms = 0.f;
          CK(hipEventElapsedTime(&ms, e0, e1));
          best = ms < best ? ms : best;
        }
        printf(" %8.1f", best * 1e3f);
      }
      printf("\n");
    }
  return 0;
}
