// How fast can 256 workgroups write a 8192 x 2048 bf16 matrix (33.5 MB) in the access patterns an output-tile epilogue can
// produce?  (gemm_nt256's exposed epilogue: 10.1 us = 3.3 TB/s.)  One 256 x 256 tile per workgroup of 8 waves (2 x 4):
//   A  per-wave 128 x 64 block, lane = 16 B of a row's 128-B segment, 8 rows per store instruction (the shipped epilogue)
//   B  workgroup-cooperative: a store instruction of a wave covers two whole 512-B tile rows (32 lanes x 16 B each)
//   C  like A, but the 8 waves walk their rows in a skewed order (wave w starts at row block w)
//   L  linear: the workgroup's 128 KiB as one contiguous stretch (not a tile layout: the HBM-side upper bound)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/store_pattern tools/store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) float f4;
constexpr int M = 8192, N = 2048;

template <int P>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* out, f4 v) {
  const int bid = blockIdx.x;
  const int q = gridDim.x >> 3, xcd = bid & 7;
  const int swz = xcd * q + (bid >> 3);
  const int tm = swz / 8, tn = swz % 8;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 2, wc = w & 3;
  char* base = reinterpret_cast<char*>(out);
  if (P == 0 || P == 2) {
    const int col = tn * 256 + wc * 64 + (lane & 7) * 8;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int ii = P == 2 ? ((i + w * 2) & 15) : i;
      const int row = tm * 256 + wr * 128 + ii * 8 + (lane >> 3);
      *reinterpret_cast<f4*>(base + ((size_t)row * N + col) * 2) = v;
    }
  } else if (P == 1) {
    const int col = tn * 256 + (lane & 31) * 8;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int row = tm * 256 + w * 32 + i * 2 + (lane >> 5);
      *reinterpret_cast<f4*>(base + ((size_t)row * N + col) * 2) = v;
    }
  } else {
    char* p = base + (size_t)swz * 131072 + tid * 16;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) *reinterpret_cast<f4*>(p + i * 8192) = v;
  }
}

template <int P>
float run(unsigned short* out, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f4 v = {1.f, 2.f, 3.f, 4.f};
  std::vector<float> t;
  for (int r = 0; r < 7; ++r) {
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_kernel<P>, dim3(256), dim3(512), 0, 0, out + (size_t)(i & 3) * M * N, v);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    t.push_back(ms / reps * 1e3f);
  }
  std::sort(t.begin(), t.end());
  return t[3];
}

int main() {
  unsigned short* out;
  hipMalloc(&out, (size_t)4 * M * N * 2);
  const char* names[4] = {"A per-wave 128-B segments (shipped)", "B workgroup rows of 512 B", "C per-wave, skewed row order", "L linear 128 KiB per workgroup"};
  float us[4] = {run<0>(out, 50), run<1>(out, 50), run<2>(out, 50), run<3>(out, 50)};
  for (int i = 0; i < 4; ++i) printf("%-40s %6.2f us = %5.2f TB/s (incl. ~2 us launch)\n", names[i], us[i], 33.554432e6 / us[i] * 1e-6);
  return 0;
}
