// Counter-based on-device RNG: Philox4x32-10 (Random123 constants) + Box-Muller.
//
// This is the engine's own stream (throughput mode).  It is NOT jax.random's threefry2x32;
// bit-compatible JAX streams are a "next" row.  Parity tests pass the normal
// draws explicitly instead (reference draw sites: utils/losses.py:272-294, utils/ebm_utils.py:
// 329-362, train_ncsn.py:540).  oracle/ddpm_oracle.py::philox4x32 restates this bit-exactly.
//
// Counter convention: (x = element_index/4 within the sample, y = GLOBAL sample index,
// z = stream id, w = step) ; key = (seed_lo, seed_hi).  Keying by the global sample index makes
// every draw independent of how samples are sharded over GPUs.
#pragma once
#include "smd_common.h"

enum : uint32_t {
  SMD_STREAM_EPS = 0,       // q-sample noise          (utils/losses.py:294)
  SMD_STREAM_LABEL = 1,     // timestep labels         (utils/losses.py:272)
  SMD_STREAM_Z = 2,         // reverse-step noise      (utils/ebm_utils.py:360-362)
  SMD_STREAM_INIT = 3,      // initial state           (train_ncsn.py:540)
  SMD_STREAM_INFILL = 4     // infill template noise   (utils/ebm_utils.py:342-345)
};

__host__ __device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
    k0 += W0;
    k1 += W1;
  }
  return c;
}

// (0,1] : never 0 so log() is finite
__device__ __forceinline__ float u01(uint32_t bits) { return ((float)(bits >> 8) + 1.0f) * 5.9604644775390625e-8f; }

// four standard normals from one Philox block (two Box-Muller pairs)
__device__ __forceinline__ float4 philox_normal4(uint32_t idx4, uint32_t sample, uint32_t stream, uint32_t step,
                                                 uint32_t k0, uint32_t k1) {
  const uint4 b = philox4x32_10(make_uint4(idx4, sample, stream, step), k0, k1);
  const float r0 = sqrtf(-2.0f * logf(u01(b.x)));
  const float r1 = sqrtf(-2.0f * logf(u01(b.z)));
  // sin / cos of the angle 2 pi u on the transcendental unit: v_sin_f32 / v_cos_f32 take their argument in REVOLUTIONS, so the
  // angle is u itself -- 4 instructions instead of two software sincosf with range reduction (~300 of the ~400 instructions of
  // this function until round 6, which made q_sample and the reverse step VALU-bound at 2 TB/s: 1.05 M calls x 400 instructions
  // = 9 us of issue slots on 1024 SIMDs).  Absolute error of a normal <= 2e-6 (tests/test_gpu_kernels.py prints it).
  const float a0 = u01(b.y), a1 = u01(b.w);
  const float s0 = __builtin_amdgcn_sinf(a0), c0 = __builtin_amdgcn_cosf(a0);
  const float s1 = __builtin_amdgcn_sinf(a1), c1 = __builtin_amdgcn_cosf(a1);
  return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}
__device__ __forceinline__ float pick4(const float4& v, int i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}
