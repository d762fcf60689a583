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

// Load-settle fence.  Placed between a kernel's up-front global loads and their first use: every load
// has returned (vmcnt(0)) and SMD_LOAD_SETTLE x 16 further idle issue cycles have passed before any VALU instruction reads
// a loaded register.  -1 = no fence (the compiler's counted vmcnt waits directly in front of the first use).
#ifndef SMD_LOAD_SETTLE
#define SMD_LOAD_SETTLE -1
#endif
__device__ __forceinline__ void smd_load_settle() {
#if SMD_LOAD_SETTLE >= 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int k = 0; k < SMD_LOAD_SETTLE; ++k) asm volatile("s_nop 15" ::: "memory");
#endif
}

// LayerNorm 1/sqrt(var + eps) as the bare v_rsq_f32 (the argument is >= 1e-6, never a denormal: same bits as rsqrtf) with
// EIGHT IDLE ISSUE CYCLES IN FRONT of it (default form, 98).  Round 3 found (profiles/r3_det_root_cause.txt):
// with a two-buffer weight-gradient workgroup on the same CU, a v_rsq_f32 issued directly behind the VALU instruction that
// writes its source operand -- what hipcc schedules -- intermittently reads the STALE register in lanes 48..63 (whole rows
// of the 128-wide LayerNorm backward wrong by ~1e-2 with bit-identical inputs; 99 of 99 repeated steps differ).  Idle cycles
// behind the instruction do not help (24 of 39), in front of it they do (0 of 499 on the encoder backward).
// Other values are the experiment builds of that hunt: 0 rsqrtf(), -1 bare v_rsq_f32, N in 1..89 s_nop N-1 behind it, 90 + n
// s_nop n in front, 99 s_nop 7 on both sides.
#ifndef SMD_TRANS_SETTLE
#define SMD_TRANS_SETTLE 98
#endif
__device__ __forceinline__ float smd_ln_rstd(float v) {
#if SMD_TRANS_SETTLE == 99
  float r;                 // idle issue cycles on BOTH sides of the transcendental instruction
  asm volatile("s_nop 7\n\tv_rsq_f32 %0, %1\n\ts_nop 7" : "=v"(r) : "v"(v));
  return r;
#elif SMD_TRANS_SETTLE >= 90 && SMD_TRANS_SETTLE <= 97
  float r;                 // 90 + n: s_nop n in front, nothing behind
  asm volatile("s_nop %2\n\tv_rsq_f32 %0, %1" : "=v"(r) : "v"(v), "n"(SMD_TRANS_SETTLE - 90));
  return r;
#elif SMD_TRANS_SETTLE == 98
  float r;                 // idle issue cycles only in FRONT of it (its source operand was just written)
  asm volatile("s_nop 7\n\tv_rsq_f32 %0, %1\n\ts_nop 0" : "=v"(r) : "v"(v));
  return r;
#elif SMD_TRANS_SETTLE > 0
  float r = __builtin_amdgcn_rsqf(v);
  asm volatile("s_nop %1" : "+v"(r) : "n"(SMD_TRANS_SETTLE - 1));
  return r;
#elif SMD_TRANS_SETTLE < 0
  return __builtin_amdgcn_rsqf(v);
#else
  return rsqrtf(v);
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
