// jax.random-compatible counter RNG: Threefry-2x32 (20 rounds, Random123) and the jax 0.2.8 conventions on top of it.
//
// Reference call sites: utils/losses.py:271-294 (split / randint / uniform / normal of the training draws),
// utils/ebm_utils.py:329,342-345,360-362 (per-iteration splits and normals of the reverse sampler),
// train_ncsn.py:318-319,358,536-540 (key flow).  jax itself is not vendored: the algorithm is restated and pinned by
// the Random123 vectors and the keys / normals printed in JAX's documentation (tests/golden/jax_random_kat.json).
//
//   random_bits(key, n)[i] : the counter vector iota(n) (odd n: one zero appended) is cut into halves x0 | x1, every
//                            (x0[j], x1[j]) pair is one Threefry block, and the outputs are concatenated y0 | y1.
//                            With h = ceil(n/2): element i < h is word 0 of block (i, i+h < n ? i+h : 0), element
//                            i >= h is word 1 of block (i-h, i).
//   uniform(lo, hi)        : u01 = bitcast((bits >> 9) | 0x3F800000) - 1 ; max(lo, u01 * (hi - lo) + lo)
//   normal                 : sqrt(2) * erf_inv(uniform(nextafter(-1, 0), 1)), XLA's float32 erf_inv polynomial
//   randint(lo, hi)        : two draws from split(key); ((hi_bits % span) * (2^32 % span) + lo_bits % span) % span
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

struct TfKey { uint32_t k0, k1; };

__host__ __device__ __forceinline__ uint32_t tf_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__host__ __device__ __forceinline__ void threefry2x32(TfKey k, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks0 = k.k0, ks1 = k.k1, ks2 = k.k0 ^ k.k1 ^ 0x1BD11BDAu;
#define TF_R(r) x0 += x1; x1 = tf_rotl(x1, r) ^ x0;
  x0 += ks0; x1 += ks1;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)
  x0 += ks1; x1 += ks2 + 1u;
  TF_R(17) TF_R(29) TF_R(16) TF_R(24)
  x0 += ks2; x1 += ks0 + 2u;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)
  x0 += ks0; x1 += ks1 + 3u;
  TF_R(17) TF_R(29) TF_R(16) TF_R(24)
  x0 += ks1; x1 += ks2 + 4u;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)
  x0 += ks2; x1 += ks0 + 5u;
#undef TF_R
}

// element i of random_bits(key, n)
__host__ __device__ __forceinline__ uint32_t jax_bits_at(TfKey k, uint64_t i, uint64_t n) {
  const uint64_t h = (n + 1) >> 1;
  const uint64_t j = i < h ? i : i - h;
  uint32_t x0 = (uint32_t)j, x1 = (j + h < n) ? (uint32_t)(j + h) : 0u;
  threefry2x32(k, x0, x1);
  return i < h ? x0 : x1;
}

// child c of split(key, num)
__host__ __device__ __forceinline__ TfKey jax_split_at(TfKey k, uint32_t c, uint32_t num) {
  TfKey o;
  o.k0 = jax_bits_at(k, 2ull * c, 2ull * num);
  o.k1 = jax_bits_at(k, 2ull * c + 1, 2ull * num);
  return o;
}

__device__ __forceinline__ float jax_u01(uint32_t bits) { return __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f; }

__device__ __forceinline__ float jax_uniform_from_bits(uint32_t bits, float lo, float hi) {
#pragma clang fp contract(off)                          // mul then add, rounded separately (like the restatement)
  const float v = jax_u01(bits) * (hi - lo) + lo;
  return fmaxf(lo, v);
}

// XLA ErfInv (float32): w = -log((1-x)(1+x)); two degree-8 polynomials in w - 2.5 / sqrt(w) - 3 (Giles 2010)
__device__ __forceinline__ float xla_erfinv_f32(float x) {
#pragma clang fp contract(off)
  float w = -logf((1.0f - x) * (1.0f + x));
  const bool lt = w < 5.0f;
  w = lt ? w - 2.5f : sqrtf(w) - 3.0f;
  float p = lt ? 2.81022636e-08f : -0.000200214257f;
#define TF_P(a, b) p = (lt ? (a) : (b)) + p * w;
  TF_P(3.43273939e-07f, 0.000100950558f)
  TF_P(-3.5233877e-06f, 0.00134934322f)
  TF_P(-4.39150654e-06f, -0.00367342844f)
  TF_P(0.00021858087f, 0.00573950773f)
  TF_P(-0.00125372503f, -0.0076224613f)
  TF_P(-0.00417768164f, 0.00943887047f)
  TF_P(0.246640727f, 1.00167406f)
  TF_P(1.50140941f, 2.83297682f)
#undef TF_P
  return p * x;
}

__device__ __forceinline__ float jax_normal_from_bits(uint32_t bits) {
  const float lo = -0.99999994f;                       // nextafter(-1, 0); 1 - lo rounds to 2.0f
  return 1.41421356f * xla_erfinv_f32(jax_uniform_from_bits(bits, lo, 1.0f));
}
