#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
// raw: lane l gets the 32 bytes Araw[l][0..31] / Braw[l][0..31]; scales sa[l], sb[l] (byte 0)
__global__ void probe(const unsigned char* A, const unsigned char* B, const unsigned* sa, const unsigned* sb, float* C) {
  const int lane = threadIdx.x;
  i32x8_t a, b;
  const int* pa = reinterpret_cast<const int*>(A + lane * 32);
  const int* pb = reinterpret_cast<const int*>(B + lane * 32);
  for (int i = 0; i < 8; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
  f32x16_t acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, (int)sa[lane], 0, (int)sb[lane]);
  for (int e = 0; e < 16; ++e) C[lane * 16 + e] = acc[e];
}
static float fp8_to_f(unsigned char v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}
int main() {
  std::vector<unsigned char> A(64 * 32), B(64 * 32);
  std::vector<unsigned> sa(64), sb(64);
  srand(1);
  for (auto& v : A) { v = rand() & 0xFF; if ((v & 0x7F) == 0x7F) v = 0x10; }
  for (auto& v : B) { v = rand() & 0xFF; if ((v & 0x7F) == 0x7F) v = 0x20; }
  unsigned char *dA, *dB; unsigned *dsa, *dsb; float* dC;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  std::vector<float> C(1024);
  auto run = [&]() {
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  };
  // C element (lane, e): col n = lane&31, row m = (e&3) + 8*(e>>2) + 4*(lane>>5)   (standard 32x32 C layout)
  // data-layout hypotheses: logical element (row r, k) of operand held by lane L(r,k), byte J(r,k)
  auto kmapH = [&](int hyp, int r, int k, int& L, int& J) {
    if (hyp == 0) { L = r + 32 * (k >> 5); J = k & 31; }                                  // contiguous 32 per lane half
    else if (hyp == 1) { L = r + 32 * ((k >> 4) & 1); J = (k & 15) + 16 * (k >> 5); }      // two K=32 sub-steps, 16 per half each
    else if (hyp == 2) { L = r + 32 * ((k >> 3) & 1); J = (k & 7) + 8 * (k >> 4); }         // four K=16 sub-steps, 8 per half each
    else { L = r + 32 * ((k >> 2) & 1); J = (k & 3) + 4 * (k >> 3); }
  };
  for (auto& v : sa) v = 127; for (auto& v : sb) v = 127;
  run();
  for (int hyp = 0; hyp < 4; ++hyp) {
    double num = 0, den = 0;
    for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 16; ++e) {
      const int n = lane & 31, m = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      double ref = 0;
      for (int k = 0; k < 64; ++k) { int La, Ja, Lb, Jb; kmapH(hyp, m, k, La, Ja); kmapH(hyp, n, k, Lb, Jb); ref += (double)fp8_to_f(A[La * 32 + Ja]) * fp8_to_f(B[Lb * 32 + Jb]); }
      num += (C[lane * 16 + e] - ref) * (C[lane * 16 + e] - ref); den += ref * ref;
    }
    printf("unit scales, data hypothesis %d: rel err %.3e\n", hyp, sqrt(num / den));
  }
  // scale semantics: A scale non-uniform. Hypotheses: S0: scale of lane L applies to the 32 bytes lane L holds;
  // S1: only lanes 0..31 supply scales: lane r -> whole row r (all 64 k); S2: lane r + 32*j supplies block j under data hyp 0 regardless
  for (int l = 0; l < 64; ++l) sa[l] = (unsigned)(120 + (l * 5) % 11) ;
  run();
  std::vector<float> Csa = C;
  for (int dh = 0; dh < 4; ++dh) for (int sh = 0; sh < 3; ++sh) {
    double num = 0, den = 0;
    for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 16; ++e) {
      const int n = lane & 31, m = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      double ref = 0;
      for (int k = 0; k < 64; ++k) {
        int La, Ja, Lb, Jb; kmapH(dh, m, k, La, Ja); kmapH(dh, n, k, Lb, Jb);
        const int sl = sh == 0 ? La : (sh == 1 ? m : m + 32 * (k >> 5));
        ref += ldexp(1.0, (int)(sa[sl] & 0xFF) - 127) * fp8_to_f(A[La * 32 + Ja]) * fp8_to_f(B[Lb * 32 + Jb]);
      }
      num += (Csa[lane * 16 + e] - ref) * (Csa[lane * 16 + e] - ref); den += ref * ref;
    }
    if (dh == 0 && sh == 0) for (int q = 0; q < 4; ++q) { int lane = q * 17, e = q * 3; const int n = lane & 31, m = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); double ref = 0; for (int k = 0; k < 64; ++k) { int La, Ja, Lb, Jb; kmapH(0, m, k, La, Ja); kmapH(0, n, k, Lb, Jb); ref += ldexp(1.0, (int)(sa[La] & 0xFF) - 127) * fp8_to_f(A[La * 32 + Ja]) * fp8_to_f(B[Lb * 32 + Jb]); } printf("  sample m=%d n=%d: got %g ref %g\n", m, n, Csa[lane * 16 + e], ref); }
    printf("A scales varying, data hyp %d, scale hyp %s: rel err %.3e\n", dh, sh == 0 ? "per-lane(own bytes)" : (sh == 1 ? "lanes0-31 per row" : "lane r+32b -> k-block b"), sqrt(num / den));
  }
  return 0;
}
