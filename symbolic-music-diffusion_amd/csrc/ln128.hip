// The two stand-alone 128-wide LayerNorm kernels of the hidden-split encoder dataflow (encoder_fused.hip holds the MFMA-fused
// half-layers): ln128_parts (sum of the four partial tiles + the encoder's final norm, models/ncsn.py:170) and ln128_bwd_parts
// (LayerNorm 2 backward on the summed da2 tiles + the residual gradient, models/ncsn.py:164).
//
// Their own translation unit because it is compiled WITHOUT packed-fp32 VALU arithmetic (-fno-slp-vectorize, build.py): these
// small-LDS workgroups share CUs with the side stream's weight-gradient workgroups, and tools/rsq_repro.hip shows that the
// v_pk_mul_f32 hipcc forms on a register pair reads a stale source in lanes 48..63 there.
#include "smd_kernels.h"

namespace {

constexpr int E_DIM = 128;
constexpr float LN_EPS = 1e-6f;

template <int R, int DPP = 0>
__device__ __forceinline__ void wave_allreduce_sum(float (&v)[R]) {
  if constexpr (DPP == 1) {
#define SMD_DPP_ADD(CTRL)                                                                                          \
  _Pragma("unroll") for (int i = 0; i < R; ++i)                                                                    \
    v[i] += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v[i]), CTRL, 0xF, 0xF, true));
    SMD_DPP_ADD(0xB1)    // quad_perm [1,0,3,2]
    SMD_DPP_ADD(0x4E)    // quad_perm [2,3,0,1]
    SMD_DPP_ADD(0x141)   // row_half_mirror
    SMD_DPP_ADD(0x140)   // row_mirror
#undef SMD_DPP_ADD
  } else {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
      for (int i = 0; i < R; ++i) v[i] += __shfl_xor(v[i], o, 64);
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i) v[i] += __shfl_xor(v[i], 16, 64);
#pragma unroll
  for (int i = 0; i < R; ++i) v[i] += __shfl_xor(v[i], 32, 64);
}

// x = (p0 + p1) + (p2 + p3) of four partial tiles (one wave per 128-wide row), optionally written out, optionally
// followed by LayerNorm -> bf16: the consumer of the hidden-split MLP's output where no attention kernel follows
// (the encoder's final norm, models/ncsn.py:170) and the stand-alone / test entry.
__global__ __launch_bounds__(256) void ln128_parts_kernel(const float* __restrict__ parts, size_t part_stride, int rows,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ x_out, bf16_t* __restrict__ ln_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t o = (size_t)row * E_DIM + lane * 2;
  const float2 p0 = *reinterpret_cast<const float2*>(parts + o), p1 = *reinterpret_cast<const float2*>(parts + part_stride + o);
  const float2 p2 = *reinterpret_cast<const float2*>(parts + 2 * part_stride + o), p3 = *reinterpret_cast<const float2*>(parts + 3 * part_stride + o);
  float2 x;
  x.x = (p0.x + p1.x) + (p2.x + p3.x);
  x.y = (p0.y + p1.y) + (p2.y + p3.y);
  if (x_out) *reinterpret_cast<float2*>(x_out + o) = x;
  if (ln_out) {
    float st[2] = {x.x + x.y, x.x * x.x + x.y * x.y};
    wave_allreduce_sum<2>(st);
    const float mean = st[0] * (1.0f / E_DIM);
    const float rstd = smd_ln_rstd(st[1] * (1.0f / E_DIM) - mean * mean + LN_EPS);
    const float2 g2 = *reinterpret_cast<const float2*>(gamma + lane * 2), b2 = *reinterpret_cast<const float2*>(beta + lane * 2);
    bf16x2_t t;
    t[0] = f2bf((x.x - mean) * rstd * g2.x + b2.x);
    t[1] = f2bf((x.y - mean) * rstd * g2.y + b2.y);
    *reinterpret_cast<bf16x2_t*>(ln_out + o) = t;
  }
}

// LayerNorm backward on a gradient given as four partial tiles: dout = (p0 + p1) + (p2 + p3) (fp32), x fp32,
//   dx = LN-backward(dout) + dres  -> fp32 (may alias dres) and bf16;  dgamma / dbeta partial sums per 32-row group.
// One wave per row (two columns per lane), 8 rows per wave in flight, 32 rows per workgroup.
// DRES / F32 / B16: which optional operands exist -- compile-time, so that every load and store of the kernel is
// unconditional straight-line code (kernels with runtime-conditional loads in unrolled loops were the
// ones that lost bitwise repeatability next to the side stream).
template <bool DRES, bool F32, bool B16>
__global__ __launch_bounds__(256) void ln128_bwd_parts_kernel(const float* __restrict__ x, const float* __restrict__ parts,
                                                              size_t part_stride, const float* __restrict__ gamma,
                                                              const float* dres, float* dx_f32, bf16_t* __restrict__ dx_bf16,
                                                              float* __restrict__ partial) {
  __shared__ float red[4][2][E_DIM];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t r0 = (size_t)blockIdx.x * 32 + w * 8;
  const float2 g2 = *reinterpret_cast<const float2*>(gamma + lane * 2);
  float2 xv[8], dv[8], rv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const size_t o = (r0 + i) * E_DIM + lane * 2;
    xv[i] = *reinterpret_cast<const float2*>(x + o);
    const float2 p0 = *reinterpret_cast<const float2*>(parts + o), p1 = *reinterpret_cast<const float2*>(parts + part_stride + o);
    const float2 p2 = *reinterpret_cast<const float2*>(parts + 2 * part_stride + o), p3 = *reinterpret_cast<const float2*>(parts + 3 * part_stride + o);
    dv[i].x = (p0.x + p1.x) + (p2.x + p3.x);
    dv[i].y = (p0.y + p1.y) + (p2.y + p3.y);
    if constexpr (DRES) rv[i] = *reinterpret_cast<const float2*>(dres + o);
    else rv[i] = make_float2(0.f, 0.f);
  }
  float st[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) { st[2 * i] = xv[i].x + xv[i].y; st[2 * i + 1] = xv[i].x * xv[i].x + xv[i].y * xv[i].y; }
  wave_allreduce_sum<16>(st);
  float Px = 0.f, Py = 0.f, Qx = 0.f, Qy = 0.f;
  float tt[16], rs[8];
  float2 xh[8], dxh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float mean = st[2 * i] * (1.0f / E_DIM);
    rs[i] = smd_ln_rstd(st[2 * i + 1] * (1.0f / E_DIM) - mean * mean + LN_EPS);
    xh[i].x = (xv[i].x - mean) * rs[i];
    xh[i].y = (xv[i].y - mean) * rs[i];
    Qx += dv[i].x; Qy += dv[i].y;
    Px += dv[i].x * xh[i].x; Py += dv[i].y * xh[i].y;
    dxh[i].x = dv[i].x * g2.x; dxh[i].y = dv[i].y * g2.y;
    tt[2 * i] = dxh[i].x + dxh[i].y;
    tt[2 * i + 1] = dxh[i].x * xh[i].x + dxh[i].y * xh[i].y;
  }
  wave_allreduce_sum<16>(tt);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float t1 = tt[2 * i] * (1.0f / E_DIM), t2 = tt[2 * i + 1] * (1.0f / E_DIM);
    float2 o;
    o.x = rs[i] * (dxh[i].x - t1 - xh[i].x * t2) + rv[i].x;
    o.y = rs[i] * (dxh[i].y - t1 - xh[i].y * t2) + rv[i].y;
    const size_t off = (r0 + i) * E_DIM + lane * 2;
    if constexpr (F32) *reinterpret_cast<float2*>(dx_f32 + off) = o;
    if constexpr (B16) {
      bf16x2_t t;
      t[0] = f2bf(o.x); t[1] = f2bf(o.y);
      *reinterpret_cast<bf16x2_t*>(dx_bf16 + off) = t;
    }
  }
  red[w][0][lane * 2] = Px; red[w][0][lane * 2 + 1] = Py;
  red[w][1][lane * 2] = Qx; red[w][1][lane * 2 + 1] = Qy;
  __syncthreads();
  {
    const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
    partial[((size_t)blockIdx.x * 2 + which) * E_DIM + c] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
  }
}

}  // namespace

int launch_ln128_bwd_parts(const float* x, const float* parts, size_t part_stride, int rows, const float* gamma, const float* dres,
                           float* dx_f32, bf16_t* dx_bf16, float* partial, hipStream_t st) {
  SMD_ARG_CHECK(x && parts && gamma && partial && (dx_f32 || dx_bf16), "ln128_bwd_parts: null pointer");
  SMD_ARG_CHECK(rows > 0 && rows % 32 == 0, "ln128_bwd_parts: rows=%d must be a multiple of 32", rows);
  const dim3 grid(rows / 32), block(256);
  // "ln_excl" (experiment): pad the launch with dynamic LDS so that the workgroup cannot share a CU
  // with a 64-KiB workgroup of the side stream's weight-gradient GEMMs
  const int pad = smd_tuning_get("ln_excl") > 0 ? smd_tuning_get("ln_excl") * 1024 : 0;
#define SMD_LNB(D_, F_, B_)                                                                                           \
  do {                                                                                                                \
    if (pad > 65536) (void)hipFuncSetAttribute((const void*)ln128_bwd_parts_kernel<D_, F_, B_>, hipFuncAttributeMaxDynamicSharedMemorySize, pad); \
    hipLaunchKernelGGL((ln128_bwd_parts_kernel<D_, F_, B_>), grid, block, pad, st, x, parts, part_stride, gamma, dres, dx_f32, dx_bf16, partial); \
  } while (0)
  if (dres) {
    if (dx_f32 && dx_bf16) SMD_LNB(true, true, true); else if (dx_f32) SMD_LNB(true, true, false); else SMD_LNB(true, false, true);
  } else {
    if (dx_f32 && dx_bf16) SMD_LNB(false, true, true); else if (dx_f32) SMD_LNB(false, true, false); else SMD_LNB(false, false, true);
  }
#undef SMD_LNB
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_ln128_parts(const float* parts, size_t part_stride, int rows, const float* gamma, const float* beta, float* x_out,
                       bf16_t* ln_out, hipStream_t st) {
  SMD_ARG_CHECK(parts && rows > 0 && (x_out || ln_out) && (!ln_out || (gamma && beta)), "ln128_parts: bad arguments");
  hipLaunchKernelGGL(ln128_parts_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, parts, part_stride, rows, gamma, beta, x_out, ln_out);
  SMD_LAUNCH_CHECK();
  return 0;
}
