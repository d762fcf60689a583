// Shared device/host helpers for the smd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

#define SMD_WAVE 64

// ---- error convention (include/smd_hip.h): 0 ok, <0 argument error, >0 hipError_t ----
void smd_set_error(const char* fmt, ...);
#define SMD_ARG_CHECK(cond, ...)                         \
  do {                                                   \
    if (!(cond)) {                                       \
      smd_set_error(__VA_ARGS__);                        \
      return -1;                                         \
    }                                                    \
  } while (0)
#define SMD_LAUNCH_CHECK()                                                     \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      smd_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e__));  \
      return (int)e__;                                                         \
    }                                                                          \
  } while (0)

// ---- XCD-aware work distribution: hardware places block b on XCD b % 8; this bijection gives XCD x the contiguous
// band [x*n/8, (x+1)*n/8) of a 1-D index space (the mapping every row-tiled kernel here uses) ----
__device__ __forceinline__ int smd_xcd_band(int bid, int n) {
  const int q = n >> 3, r = n & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// table row of a device timestep: a sampler walked past t = 0 (or started at t >= T) reads row 0 / T - 1 instead of out of
// bounds (the reverse step itself is a no-op for such t)
__device__ __forceinline__ int smd_clamp_t(int t, int T) { return t < 0 ? 0 : (t >= T ? T - 1 : t); }

// ---- scalar math ----
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

// LayerNorm 1/sqrt(var + eps): v_rsq_f32 (the argument is >= 1e-6, never a denormal: same bits as rsqrtf) behind eight idle
// issue cycles.  History: round 3 attributed the intermittently wrong row statistics of the 128-wide
// LayerNorm backward next to a two-buffer weight-gradient workgroup to this instruction and put the idle cycles in front of
// it, which lowered the rate without closing it.  The stand-alone reproducer of round 4 (tools/rsq_repro.hip) shows the value
// this instruction WRITES is always right; what reads a stale register in lanes 48..63 is the v_pk_mul_f32 that hipcc's SLP
// vectoriser forms for the first use of the result -- with a plain FMA as the producer alike.  The cure is in the build
// (no packed-fp32 arithmetic in the small-LDS kernels, build.py) and in the library (only the four-buffer weight-gradient
// kernel ships); the idle cycles stay as they are part of the code every determinism soak has run on.
// -DSMD_LN_RSTD_BARE (tools/build_rsq_repro.sh): the bare instruction, for the reproducer's victim build.
__device__ __forceinline__ float smd_ln_rstd(float v) {
#ifdef SMD_LN_RSTD_BARE
  return __builtin_amdgcn_rsqf(v);
#else
  float r;
  asm volatile("s_nop 7\n\tv_rsq_f32 %0, %1\n\ts_nop 0" : "=v"(r) : "v"(v));
  return r;
#endif
}

// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: the FiLM LayerNorms evaluate this for 16.8 M elements each
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float swish_gradf_(float x) {
  float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}
// tanh-approximation GELU (flax.nn.gelu), reference models/ncsn.py:166
__device__ __forceinline__ float tanhf_(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); safe for large |x|
  float e = __expf(2.0f * x);
  return 1.0f - 2.0f / (e + 1.0f);
}
// gelu(x) = 0.5 x (1 + tanh(u)), u = k0 (x + k1 x^3)  ==  x / (1 + exp(-2u)); one exp2 + one rcp:
// exp(-2u) = exp2(x * (A + B x^2)),  A = -2 k0 log2(e),  B = A k1
__device__ __forceinline__ float geluf_(float x) {
  const float A = -2.0f * 0.7978845608028654f * 1.4426950408889634f, Bc = A * 0.044715f;
  const float e = __builtin_amdgcn_exp2f(x * fmaf(Bc, x * x, A));
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_gradf_(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  float t = tanhf_(u);
  float du = k0 * (1.0f + 3.0f * k1 * x * x);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}

// gelu(x) AND gelu'(x) from ONE exp2 + ONE rcp (the recompute backward evaluates both for every hidden activation:
// 16.8 M per encoder layer at B = 256, VALU-bound).  s = 1 / (1 + exp(-2u)) = 0.5 (1 + tanh u):
//   gelu = x s   (the same expression, bit for bit, as geluf_)      gelu' = s + 2 x s (1 - s) du/dx,  du/dx = k0 (1 + 3 k1 x^2)
// exp2 overflowing to +inf for very negative x gives s = 0, gelu = -0, gelu' = 0; underflow for large x gives s = 1, gelu' = 1.
__device__ __forceinline__ void gelu_fwd_grad_(float x, float& g, float& dg) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float A = -2.0f * k0 * 1.4426950408889634f, Bc = A * k1;
  const float x2 = x * x;
  const float e = __builtin_amdgcn_exp2f(x * fmaf(Bc, x2, A));
  const float s = __builtin_amdgcn_rcpf(1.0f + e);
  g = x * s;
  const float xd = x * fmaf(6.0f * k0 * k1, x2, 2.0f * k0);
  dg = fmaf(xd, s * (1.0f - s), s);
}

// Two hidden activations per instruction: the same expressions as geluf_ / gelu_fwd_grad_, element for element (every packed
// instruction rounds each half exactly like its scalar form: v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32 are two IEEE operations), so
// the results are bit-identical; the transcendentals stay scalar.  The GELU phase of the hidden-split MLP kernels is VALU-bound
// (encoder_fused.hip): 11 instead of ~14.5 instructions per pair forward, 17 instead of ~27 in the recompute backward.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ f32x2_t geluf2_(f32x2_t x) {
  const float A = -2.0f * 0.7978845608028654f * 1.4426950408889634f, Bc = A * 0.044715f;
  const f32x2_t A2 = {A, A}, B2 = {Bc, Bc}, one = {1.0f, 1.0f};
  const f32x2_t t = x * __builtin_elementwise_fma(B2, x * x, A2);
  f32x2_t e;
  e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
  const f32x2_t d = one + e;
  f32x2_t r;
  r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
  return x * r;
}
__device__ __forceinline__ void gelu_fwd_grad2_(f32x2_t x, f32x2_t& g, f32x2_t& dg) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float A = -2.0f * k0 * 1.4426950408889634f, Bc = A * k1;
  const f32x2_t A2 = {A, A}, B2 = {Bc, Bc}, one = {1.0f, 1.0f}, C2 = {6.0f * k0 * k1, 6.0f * k0 * k1}, D2 = {2.0f * k0, 2.0f * k0};
  const f32x2_t x2 = x * x;
  const f32x2_t t = x * __builtin_elementwise_fma(B2, x2, A2);
  f32x2_t e;
  e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
  const f32x2_t d = one + e;
  f32x2_t s;
  s.x = __builtin_amdgcn_rcpf(d.x); s.y = __builtin_amdgcn_rcpf(d.y);
  g = x * s;
  const f32x2_t xd = x * __builtin_elementwise_fma(C2, x2, D2);
  dg = __builtin_elementwise_fma(xd, s * (one - s), s);
}

// ---- OCP e4m3 with a per-row power-of-two scale: the smallest e with amax * 2^-e <= 448 (the format's maximum), i.e.
// floor(log2(amax)) - 8, plus one when the mantissa of amax exceeds 1.75 (until round 3 those rows -- about one in five -- had
// their largest elements saturated by up to 12.5 %, far above e4m3's rounding error).  The E8M0 byte the scaled MFMA takes is e + 127.
__device__ __forceinline__ int e4m3_row_exponent(float amax) {
  if (!(amax > 0.0f)) return 0;
  const unsigned bits = __builtin_bit_cast(unsigned, amax);
  int e = (int)((bits >> 23) & 0xFF) - 127 - 8;
  if ((bits & 0x7FFFFFu) > 0x600000u) e += 1;                 // mantissa > 1.75: 2^8 * mantissa would pass 448
  return e < -126 ? -126 : e;
}
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d, int e) {
  const float lim = 448.0f;
  a = fminf(fmaxf(ldexpf(a, -e), -lim), lim); b = fminf(fmaxf(ldexpf(b, -e), -lim), lim);
  c = fminf(fmaxf(ldexpf(c, -e), -lim), lim); d = fminf(fmaxf(ldexpf(d, -e), -lim), lim);
  int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (unsigned)v;
}

// ---- wave64 reductions (DPP/bpermute via __shfl_xor) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
