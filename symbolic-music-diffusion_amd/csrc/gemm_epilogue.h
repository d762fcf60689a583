// Fused GEMM epilogue shared by the NT kernels (gemm_nt.hip 128-wide tiles, gemm_nt256.hip 256x256 tile).
// Order of operations: see GemmEpilogue in smd_kernels.h.
#pragma once
#include "smd_kernels.h"

namespace smd_epi {

__device__ __forceinline__ float epilogue_scalar(float acc, int row, int col, const GemmEpilogue& ep) {
  float v = ep.alpha * acc;
  if (ep.bias) v += ep.bias[col];
  if (ep.pre_bf16) ep.pre_bf16[(size_t)row * ep.ld_pre + col] = f2bf(v);
  if (ep.act == SMD_ACT_GELU) v = geluf_(v);
  else if (ep.act == SMD_ACT_SWISH) v = swishf_(v);
  if (ep.aux_mode != SMD_AUX_NONE) {
    const float z = bf2f(ep.aux[(size_t)row * ep.ld_aux + col]);
    v *= (ep.aux_mode == SMD_AUX_GELU_GRAD) ? gelu_gradf_(z) : swish_gradf_(z);
  }
  if (ep.res_f32) {
    const int rr = ep.res_row_mod > 0 ? (row % ep.res_row_mod) : row;
    v += ep.res_f32[(size_t)rr * ep.ld_res + col];
  }
  if (ep.res_bf16) v += bf2f(ep.res_bf16[(size_t)row * ep.ld_resb + col]);
  return v;
}

// 4 consecutive columns of one row; `vec_ok`: every pointer/ld is 4-element aligned and col+3 < N
__device__ __forceinline__ void epilogue_quad(const float4 a, int row, int col, int N, bool vec_ok,
                                              const GemmEpilogue& ep) {
  if (vec_ok) {
    float v[4] = {ep.alpha * a.x, ep.alpha * a.y, ep.alpha * a.z, ep.alpha * a.w};
    if (ep.bias) {
      const float4 b = *reinterpret_cast<const float4*>(ep.bias + col);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (ep.pre_bf16) {
      bf16x4_t p;
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = f2bf(v[i]);
      *reinterpret_cast<bf16x4_t*>(ep.pre_bf16 + (size_t)row * ep.ld_pre + col) = p;
    }
    if (ep.act == SMD_ACT_GELU) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = geluf_(v[i]);
    } else if (ep.act == SMD_ACT_SWISH) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = swishf_(v[i]);
    }
    if (ep.aux_mode != SMD_AUX_NONE) {
      const bf16x4_t z = *reinterpret_cast<const bf16x4_t*>(ep.aux + (size_t)row * ep.ld_aux + col);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        v[i] *= (ep.aux_mode == SMD_AUX_GELU_GRAD) ? gelu_gradf_(bf2f(z[i])) : swish_gradf_(bf2f(z[i]));
    }
    if (ep.res_f32) {
      const int rr = ep.res_row_mod > 0 ? (row % ep.res_row_mod) : row;
      const float4 r = *reinterpret_cast<const float4*>(ep.res_f32 + (size_t)rr * ep.ld_res + col);
      v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    }
    if (ep.res_bf16) {
      const bf16x4_t r = *reinterpret_cast<const bf16x4_t*>(ep.res_bf16 + (size_t)row * ep.ld_resb + col);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += bf2f(r[i]);
    }
    if (ep.out_f32) {
      float4* o = reinterpret_cast<float4*>(ep.out_f32 + (size_t)row * ep.ld_out + col);
      if (ep.accumulate) {
        const float4 c = *o;
        v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w;
      }
      *o = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (ep.out_bf16) {
      bf16x4_t p;
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = f2bf(v[i]);
      *reinterpret_cast<bf16x4_t*>(ep.out_bf16 + (size_t)row * ep.ld_outb + col) = p;
    }
  } else {
    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (col + i < N) {
        const float v = epilogue_scalar(av[i], row, col + i, ep);
        if (ep.out_f32) {
          float* o = ep.out_f32 + (size_t)row * ep.ld_out + col + i;
          *o = ep.accumulate ? (*o + v) : v;
        }
        if (ep.out_bf16) ep.out_bf16[(size_t)row * ep.ld_outb + col + i] = f2bf(v);
      }
    }
  }
}

// ---- 8-column form used by the 256x256 kernel: every lane owns 8 consecutive columns of a row; the global
// operands of the epilogue (aux, fp32 residual) are fetched BEFORE the accumulators are staged so that a
// whole pass of loads is in flight at once.  Supports: bias, pre_bf16, act, aux, res_f32 (+row mod), out_f32,
// res_bf16, out_bf16 (alpha == 1, no accumulate: see oct_ok()).
struct EpiPre8 {
  bf16x8_t aux;
  bf16x8_t rb;
  float4 r0, r1;
};
__device__ __forceinline__ void epi8_prefetch(EpiPre8& p, int row, int col, const GemmEpilogue& ep) {
  if (ep.aux_mode != SMD_AUX_NONE) p.aux = *reinterpret_cast<const bf16x8_t*>(ep.aux + (size_t)row * ep.ld_aux + col);
  if (ep.res_f32) {
    const int rr = ep.res_row_mod > 0 ? (row % ep.res_row_mod) : row;
    const float4* r = reinterpret_cast<const float4*>(ep.res_f32 + (size_t)rr * ep.ld_res + col);
    p.r0 = r[0]; p.r1 = r[1];
  }
  if (ep.res_bf16) p.rb = *reinterpret_cast<const bf16x8_t*>(ep.res_bf16 + (size_t)row * ep.ld_resb + col);
}
__device__ __forceinline__ void epi8_apply(float (&v)[8], const float (&bias)[8], const EpiPre8& p, int row, int col,
                                           const GemmEpilogue& ep) {
  if (ep.bias) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += bias[i];
  }
  if (ep.pre_bf16) {
    bf16x8_t o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
    *reinterpret_cast<bf16x8_t*>(ep.pre_bf16 + (size_t)row * ep.ld_pre + col) = o;
  }
  if (ep.act == SMD_ACT_GELU) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = geluf_(v[i]);
  } else if (ep.act == SMD_ACT_SWISH) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = swishf_(v[i]);
  }
  if (ep.aux_mode == SMD_AUX_GELU_GRAD) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= gelu_gradf_(bf2f(p.aux[i]));
  } else if (ep.aux_mode == SMD_AUX_SWISH_GRAD) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= swish_gradf_(bf2f(p.aux[i]));
  }
  if (ep.res_f32) {
    v[0] += p.r0.x; v[1] += p.r0.y; v[2] += p.r0.z; v[3] += p.r0.w;
    v[4] += p.r1.x; v[5] += p.r1.y; v[6] += p.r1.z; v[7] += p.r1.w;
  }
  if (ep.res_bf16) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += bf2f(p.rb[i]);
  }
  if (ep.out_f32) {
    float4* o = reinterpret_cast<float4*>(ep.out_f32 + (size_t)row * ep.ld_out + col);
    o[0] = make_float4(v[0], v[1], v[2], v[3]);
    o[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
  if (ep.out_bf16) {
    bf16x8_t o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
    *reinterpret_cast<bf16x8_t*>(ep.out_bf16 + (size_t)row * ep.ld_outb + col) = o;
  }
}

inline bool al4(const void* p, int ld) { return p == nullptr || (((uintptr_t)p & 15) == 0 && ld % 4 == 0); }
inline bool al4h(const void* p, int ld) { return p == nullptr || (((uintptr_t)p & 7) == 0 && ld % 4 == 0); }
// 1 when every epilogue pointer / leading dimension allows the 4-wide vector path
inline int vec_ok(const GemmEpilogue& ep) {
  return (al4(ep.bias, 4) && al4(ep.res_f32, ep.ld_res) && al4(ep.out_f32, ep.ld_out) && al4h(ep.pre_bf16, ep.ld_pre) &&
          al4h(ep.aux, ep.ld_aux) && al4h(ep.res_bf16, ep.ld_resb) && al4h(ep.out_bf16, ep.ld_outb)) ? 1 : 0;
}

inline bool al8h(const void* p, int ld) { return p == nullptr || (((uintptr_t)p & 15) == 0 && ld % 8 == 0); }
// the 8-column epilogue applies (N is a multiple of 256 there)
inline bool oct_ok(const GemmEpilogue& ep) {
  return ep.alpha == 1.0f && !ep.accumulate && al4(ep.bias, 4) && al4(ep.res_f32, ep.ld_res) && al8h(ep.res_bf16, ep.ld_resb) &&
         al4(ep.out_f32, ep.ld_out) && al8h(ep.pre_bf16, ep.ld_pre) && al8h(ep.aux, ep.ld_aux) && al8h(ep.out_bf16, ep.ld_outb);
}

}  // namespace smd_epi
