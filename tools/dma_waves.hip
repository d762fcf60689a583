// LDS-DMA rate of ONE workgroup per CU as a function of the number of waves issuing it and of the tiles each wave keeps
// in flight: no MFMA, no LDS reads; every workgroup streams its own contiguous 8 MiB of a 2 GiB buffer (HBM, single use)
// through a ring in LDS.  Measured (profiles/r2q_dma_waves.txt): 23.3-23.6 GB/s per CU = 6.0 TB/s over the chip for 4, 8
// and 16 waves and 2 or 4 tiles in flight alike -- the HBM roof of this path; wave count and depth are not the lever.
// Second part (round 3): every workgroup walks the SAME small region again and again (1 MiB: L2-resident like a weight panel;
// 16 MiB: Infinity-Cache-resident) -- the ceiling of the shared-operand path the GEMM / fused-encoder kernels use.
//   build: hipcc -O3 --offload-arch=gfx950 tools/dma_waves.hip -o tools/dma_waves     run: tools/dma_waves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// NW waves; each wave issues PIECES 1-KiB DMA instructions per tile (tile = NW * PIECES KiB) and keeps DEPTH tiles in flight
template <int NW, int PIECES, int DEPTH>
__global__ __launch_bounds__(64 * NW) void dma_stream(const unsigned char* __restrict__ src, size_t bytes_per_wg, int tiles,
                                                     unsigned* __restrict__ sink, int wrap_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int TILE = NW * PIECES * 1024;
  // wrap_tiles > 0: all workgroups share one region of wrap_tiles tiles (each starts at its own tile of it)
  const unsigned char* base = src + (wrap_tiles ? 0 : (size_t)blockIdx.x * bytes_per_wg) + (size_t)w * PIECES * 1024 + lane * 16;
  const int t0 = wrap_tiles ? (int)(blockIdx.x * 7u) : 0;
  auto issue = [&](int tt, int buf) {
    const int t = wrap_tiles ? (tt + t0) % wrap_tiles : tt;
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(base + (size_t)t * TILE + p * 1024),
                                       (lds_void_t*)(smem + buf * TILE + (w * PIECES + p) * 1024), 16, 0, 0);
  };
  for (int t = 0; t < DEPTH - 1 && t < tiles; ++t) issue(t, t);
  int buf = 0, wbuf = DEPTH - 1;
  for (int t = 0; t < tiles; ++t) {
    const int nxt = t + DEPTH - 1;
    issue(nxt < tiles ? nxt : tiles - 1, wbuf);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PIECES) : "memory");
    __builtin_amdgcn_s_barrier();          // the consumer phase of a GEMM would sit here
    __builtin_amdgcn_s_barrier();
    buf = buf + 1 == DEPTH ? 0 : buf + 1;
    wbuf = wbuf + 1 == DEPTH ? 0 : wbuf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(smem);      // keep the LDS writes observable
}

template <int NW, int PIECES, int DEPTH>
static void run(const unsigned char* d_src, size_t total, unsigned* d_sink, const char* what, size_t region = 0) {
  constexpr int TILE = NW * PIECES * 1024;
  const int nwg = 256;
  const size_t per = total / nwg / TILE * TILE;
  const int tiles = (int)(per / TILE);
  const int wrap_tiles = (int)(region / TILE);
  const size_t lds = (size_t)DEPTH * TILE;
  if (lds > 160 * 1024) { printf("%-52s skipped (%zu KiB LDS)\n", what, lds / 1024); return; }
  hipFuncSetAttribute((const void*)dma_stream<NW, PIECES, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((dma_stream<NW, PIECES, DEPTH>), dim3(nwg), dim3(64 * NW), lds, 0, d_src, per, tiles, d_sink, wrap_tiles);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((dma_stream<NW, PIECES, DEPTH>), dim3(nwg), dim3(64 * NW), lds, 0, d_src, per, tiles, d_sink, wrap_tiles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double gbs = (double)per * nwg * reps / (ms * 1e-3) / 1e9;
  printf("%-52s %7.1f GB/s per CU  %6.2f TB/s chip   (%d KiB tiles, %zu KiB LDS)\n", what, gbs / nwg, gbs / 1e3, TILE / 1024, lds / 1024);
}

int main() {
  const size_t total = (size_t)2 << 30;          // 2 GiB: every workgroup streams its own 8 MiB from HBM
  unsigned char* d_src;
  unsigned* d_sink;
  if (hipMalloc(&d_src, total) != hipSuccess || hipMalloc(&d_sink, 4096) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(d_src, 1, total);
  printf("one workgroup per CU (256 workgroups), contiguous 8 MiB per workgroup, global_load_lds 16 B per lane\n");
  run<4, 8, 2>(d_src, total, d_sink, " 4 waves x 8 KiB per tile, 2 tiles deep");
  run<4, 8, 4>(d_src, total, d_sink, " 4 waves x 8 KiB per tile, 4 tiles deep");
  run<4, 4, 4>(d_src, total, d_sink, " 4 waves x 4 KiB per tile, 4 tiles deep");
  run<8, 4, 2>(d_src, total, d_sink, " 8 waves x 4 KiB per tile, 2 tiles deep");
  run<8, 4, 4>(d_src, total, d_sink, " 8 waves x 4 KiB per tile, 4 tiles deep");
  run<8, 8, 2>(d_src, total, d_sink, " 8 waves x 8 KiB per tile, 2 tiles deep");
  run<16, 2, 4>(d_src, total, d_sink, "16 waves x 2 KiB per tile, 4 tiles deep");
  run<16, 4, 2>(d_src, total, d_sink, "16 waves x 4 KiB per tile, 2 tiles deep");
  run<16, 4, 2>(d_src, total, d_sink, "16 waves x 4 KiB per tile, 2 tiles deep (repeat)");
  printf("the same 8 MiB of DMA per workgroup, but every workgroup walks ONE shared region (operand panels)\n");
  run<8, 8, 2>(d_src, total, d_sink, " 8 waves x 8 KiB, 2 deep, shared  1 MiB (L2)", (size_t)1 << 20);
  run<8, 4, 4>(d_src, total, d_sink, " 8 waves x 4 KiB, 4 deep, shared  1 MiB (L2)", (size_t)1 << 20);
  run<4, 8, 4>(d_src, total, d_sink, " 4 waves x 8 KiB, 4 deep, shared  1 MiB (L2)", (size_t)1 << 20);
  run<16, 4, 2>(d_src, total, d_sink, "16 waves x 4 KiB, 2 deep, shared  1 MiB (L2)", (size_t)1 << 20);
  run<8, 8, 2>(d_src, total, d_sink, " 8 waves x 8 KiB, 2 deep, shared  8 MiB (L2 of all XCDs together / MALL)", (size_t)8 << 20);
  run<8, 8, 2>(d_src, total, d_sink, " 8 waves x 8 KiB, 2 deep, shared 64 MiB (MALL)", (size_t)64 << 20);
  run<8, 8, 2>(d_src, total, d_sink, " 8 waves x 8 KiB, 2 deep, shared  1 MiB (repeat)", (size_t)1 << 20);
  hipFree(d_src); hipFree(d_sink);
  return 0;
}
