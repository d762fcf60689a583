// Which XCDs / CUs does a stream made by hipExtStreamCreateWithCUMask run on?  (pin the two sampling
// chains to disjoint halves of the chip.)  Each workgroup records HW_REG_XCC_ID and HW_REG_HW_ID and spins ~20 us so that the
// grid spreads over every CU the queue may use; the same launch is repeated from a captured hipGraph on the masked stream
// (does a graph launch honour the launch stream's mask?).
//   build: hipcc -O3 --offload-arch=gfx950 tools/cumask_probe.hip -o tools/cumask_probe     run: tools/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64) void where_kernel(uint32_t* __restrict__ out, int spin_us) {
  // s_getreg_b32 simm16 = (size-1) << 11 | offset << 6 | id ; HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20
  const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();         // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)spin_us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
}

static void report(const char* tag, const std::vector<uint32_t>& h, int n) {
  std::map<int, int> per_xcc;
  std::set<uint32_t> cus;
  for (int i = 0; i < n; ++i) {
    const int xcc = h[2 * i] & 0xF;
    const uint32_t hw = h[2 * i + 1];
    const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcc[xcc]++;
    cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
  }
  std::string s;
  for (auto& kv : per_xcc) s += " x" + std::to_string(kv.first) + ":" + std::to_string(kv.second);
  printf("%-44s distinct CUs %3zu | workgroups per XCC:%s\n", tag, cus.size(), s.c_str());
}

int main() {
  const int n = 2048, spin = 20;
  uint32_t* d;
  CK(hipMalloc(&d, n * 2 * sizeof(uint32_t)));
  std::vector<uint32_t> h(2 * n);
  struct M { const char* name; uint32_t w[8]; };
  std::vector<M> masks;
  {
    M m{"no mask (hipStreamCreate)", {0}};
    masks.push_back(m);
    M a{"all 256 bits", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}};
    masks.push_back(a);
    M lo{"bits 0..127", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}};
    masks.push_back(lo);
    M hi{"bits 128..255", {0, 0, 0, 0, ~0u, ~0u, ~0u, ~0u}};
    masks.push_back(hi);
    M il{"bits with (i % 8) < 4", {0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu}};
    masks.push_back(il);
    M ih{"bits with (i % 8) >= 4", {0xF0F0F0F0u, 0xF0F0F0F0u, 0xF0F0F0F0u, 0xF0F0F0F0u, 0xF0F0F0F0u, 0xF0F0F0F0u, 0xF0F0F0F0u, 0xF0F0F0F0u}};
    masks.push_back(ih);
    M b0{"bit 0 only", {1, 0, 0, 0, 0, 0, 0, 0}};
    masks.push_back(b0);
    M b1{"bit 1 only", {2, 0, 0, 0, 0, 0, 0, 0}};
    masks.push_back(b1);
    M b8{"bit 8 only", {0x100, 0, 0, 0, 0, 0, 0, 0}};
    masks.push_back(b8);
    M e{"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}};
    masks.push_back(e);
  }
  for (size_t k = 0; k < masks.size(); ++k) {
    hipStream_t st;
    if (k == 0) CK(hipStreamCreate(&st));
    else CK(hipExtStreamCreateWithCUMask(&st, 8, masks[k].w));
    CK(hipMemsetAsync(d, 0xFF, n * 2 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(where_kernel, dim3(n), dim3(64), 0, st, d, spin);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d, n * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    report((std::string(masks[k].name) + " [plain]").c_str(), h, n);
    // the same launch replayed from a graph captured on (and launched into) the masked stream
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(where_kernel, dim3(n), dim3(64), 0, st, d, spin);
    hipLaunchKernelGGL(where_kernel, dim3(n), dim3(64), 0, st, d, spin);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipMemsetAsync(d, 0xFF, n * 2 * sizeof(uint32_t), st));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d, n * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    report((std::string(masks[k].name) + " [graph]").c_str(), h, n);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(st));
  }
  CK(hipFree(d));
  return 0;
}
