// LayerNorm (+ FiLM featurewise affine + swish) forward / backward, wave64 row reductions.
//
// Reference semantics: flax.nn.LayerNorm (eps 1e-6, biased variance E[x^2]-E[x]^2) as called at
// models/ncsn.py:160,164,170,177 and models/shared.py:62,66; FeaturewiseAffine scale*x+shift
// (models/shared.py:54-55) and nn.swish (models/shared.py:64,68).
//
// HBM-bound elementwise work: one wave owns one row (D/64 values per lane held in registers,
// 16-byte loads), statistics by xor-shuffle reductions, bf16 output written as 8-byte packs.
#include "smd_kernels.h"

namespace {

constexpr float LN_EPS = 1e-6f;

// lane owns elements  v*256 + lane*4 + {0..3}  for v < NV  (D = NV*256), or lane*2+{0,1} for D=128
template <int D> struct RowLayout {
  static constexpr int NV = D / 256;
  static constexpr int PER_LANE = NV * 4;
  __device__ static int col(int lane, int i) { return (i >> 2) * 256 + lane * 4 + (i & 3); }
};
template <> struct RowLayout<128> {
  static constexpr int NV = 0;
  static constexpr int PER_LANE = 2;
  __device__ static int col(int lane, int i) { return lane * 2 + i; }
};

template <int D>
__device__ __forceinline__ void load_row(const float* xf, const bf16_t* xb, size_t row, int lane,
                                         float (&v)[RowLayout<D>::PER_LANE]) {
  typedef RowLayout<D> L;
  if constexpr (D == 128) {
    if (xf) {
      const float2 t = *reinterpret_cast<const float2*>(xf + row * D + lane * 2);
      v[0] = t.x; v[1] = t.y;
    } else {
      const bf16x2_t t = *reinterpret_cast<const bf16x2_t*>(xb + row * D + lane * 2);
      v[0] = bf2f(t[0]); v[1] = bf2f(t[1]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < L::NV; ++k) {
      const size_t off = row * D + k * 256 + lane * 4;
      if (xf) {
        const float4 t = *reinterpret_cast<const float4*>(xf + off);
        v[k * 4 + 0] = t.x; v[k * 4 + 1] = t.y; v[k * 4 + 2] = t.z; v[k * 4 + 3] = t.w;
      } else {
        const bf16x4_t t = *reinterpret_cast<const bf16x4_t*>(xb + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[k * 4 + e] = bf2f(t[e]);
      }
    }
  }
}

template <int D>
__device__ __forceinline__ void load_vec(const float* p, int lane, float (&v)[RowLayout<D>::PER_LANE]) {
  load_row<D>(p, nullptr, 0, lane, v);
}

template <int D>
__device__ __forceinline__ void store_row_bf16(bf16_t* out, size_t row, int lane,
                                               const float (&v)[RowLayout<D>::PER_LANE]) {
  typedef RowLayout<D> L;
  if constexpr (D == 128) {
    bf16x2_t t; t[0] = f2bf(v[0]); t[1] = f2bf(v[1]);
    *reinterpret_cast<bf16x2_t*>(out + row * D + lane * 2) = t;
  } else {
#pragma unroll
    for (int k = 0; k < L::NV; ++k) {
      bf16x4_t t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = f2bf(v[k * 4 + e]);
      *reinterpret_cast<bf16x4_t*>(out + row * D + k * 256 + lane * 4) = t;
    }
  }
}
template <int D>
__device__ __forceinline__ void store_row_f32(float* out, size_t row, int lane,
                                              const float (&v)[RowLayout<D>::PER_LANE]) {
  typedef RowLayout<D> L;
  if constexpr (D == 128) {
    *reinterpret_cast<float2*>(out + row * D + lane * 2) = make_float2(v[0], v[1]);
  } else {
#pragma unroll
    for (int k = 0; k < L::NV; ++k)
      *reinterpret_cast<float4*>(out + row * D + k * 256 + lane * 4) =
          make_float4(v[k * 4 + 0], v[k * 4 + 1], v[k * 4 + 2], v[k * 4 + 3]);
  }
}

template <int D>
__device__ __forceinline__ void row_stats(const float (&x)[RowLayout<D>::PER_LANE], float& mean, float& rstd) {
  float s = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < RowLayout<D>::PER_LANE; ++i) { s += x[i]; s2 += x[i] * x[i]; }
  s = wave_sum(s);
  s2 = wave_sum(s2);
  mean = s * (1.0f / D);
  const float var = s2 * (1.0f / D) - mean * mean;
  rstd = smd_ln_rstd(var + LN_EPS);
}

// ------------------------------------------------------------------------------ forward
template <int D>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(LnArgs a) {
  typedef RowLayout<D> L;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  float x[L::PER_LANE], g[L::PER_LANE], b[L::PER_LANE];
  load_row<D>(a.x, a.x_bf16, row, lane, x);
  load_vec<D>(a.gamma, lane, g);
  load_vec<D>(a.beta, lane, b);
  float mean, rstd;
  row_stats<D>(x, mean, rstd);
  float y[L::PER_LANE];
#pragma unroll
  for (int i = 0; i < L::PER_LANE; ++i) y[i] = (x[i] - mean) * rstd * g[i] + b[i];
  if (a.film_scale) {
    const int frow = a.t_ptr ? smd_clamp_t(*a.t_ptr, a.film_rows) : row / a.rows_per_sample;
    float sc[L::PER_LANE], sh[L::PER_LANE];
    load_vec<D>(a.film_scale + (size_t)frow * a.ld_film, lane, sc);
    load_vec<D>(a.film_shift + (size_t)frow * a.ld_film, lane, sh);
#pragma unroll
    for (int i = 0; i < L::PER_LANE; ++i) y[i] = sc[i] * y[i] + sh[i];
  }
  if (a.swish) {
#pragma unroll
    for (int i = 0; i < L::PER_LANE; ++i) y[i] = swishf_(y[i]);
  }
  store_row_bf16<D>(a.out, row, lane, y);
}

// a row held raw in registers (converted at use)
template <int D, bool XBF> struct WideRow {
  typedef RowLayout<D> L;
  float4 xf[XBF ? 1 : L::NV];
  bf16x4_t xb[XBF ? L::NV : 1];
  bf16x4_t dy[L::NV];
  __device__ __forceinline__ float x(int k, int e) const {
    if constexpr (XBF) return bf2f(xb[k][e]);
    else return e == 0 ? xf[k].x : e == 1 ? xf[k].y : e == 2 ? xf[k].z : xf[k].w;
  }
};

// Wide rows (D = 1024 / 2048) in row groups: the one-wave-per-row kernel above re-reads gamma / beta (/ FiLM scale /
// shift) from the L2 for every row -- 4x the bytes of the row itself.  Here a workgroup of 8 waves owns one row group
// (a sample's rows with per-sample FiLM, else 32 rows), keeps the parameters in LDS and walks the group's rows
// (SGPR row bases, all of a row's loads issued up front).  FS: FiLM + swish (ResBlock norms) or neither.
// F8: also (or only) write the row as e4m3 with a per-row E8M0 scale (the A operand of the e4m3 GEMM); the row's outputs
// then wait in registers for the row maximum.
template <int D, bool XBF, bool FS, int NW, bool F8 = false, bool PF = false>
__global__ __launch_bounds__(64 * NW) void layernorm_fwd_wide_kernel(LnArgs a, int group_rows) {
  typedef RowLayout<D> L;
  constexpr int NV = L::NV;
  __shared__ __attribute__((aligned(16))) float prm[FS ? 4 : 2][D];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r_begin = blockIdx.x * group_rows;
  int r_end = r_begin + group_rows;
  r_end = r_end < a.rows ? r_end : a.rows;
  {
    const int frow = FS ? (a.t_ptr ? smd_clamp_t(*a.t_ptr, a.film_rows) : r_begin / a.rows_per_sample) : 0;
    const float* src[4] = {a.gamma, a.beta, FS ? a.film_scale + (size_t)frow * a.ld_film : nullptr,
                           FS ? a.film_shift + (size_t)frow * a.ld_film : nullptr};
#pragma unroll
    for (int q = 0; q < (FS ? 4 : 2); ++q)
      for (int c = threadIdx.x * 4; c < D; c += 256 * NW)
        *reinterpret_cast<float4*>(&prm[q][c]) = *reinterpret_cast<const float4*>(src[q] + c);
  }
  __syncthreads();
  const uint32_t l4 = lane * 4;
  auto load_row_raw = [&](int row, WideRow<D, XBF>& b) {
    const float* xr = a.x + (size_t)row * D;
    const bf16_t* xbr = a.x_bf16 + (size_t)row * D;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if constexpr (XBF) b.xb[k] = *reinterpret_cast<const bf16x4_t*>(xbr + (l4 + k * 256));
      else b.xf[k] = *reinterpret_cast<const float4*>(xr + (l4 + k * 256));
    }
  };
  // PF: the NEXT row of this wave is requested before the current one is reduced (a bf16 row is 16 registers): twice the
  // bytes in flight per wave, the row's arithmetic runs under the next row's HBM latency
  WideRow<D, XBF> nxt;
  if constexpr (PF) { if (r_begin + w < r_end) load_row_raw(r_begin + w, nxt); }
  for (int row = r_begin + w; row < r_end; row += NW) {
    WideRow<D, XBF> b;                       // dy unused here
    if constexpr (PF) {
      b = nxt;
      if (row + NW < r_end) load_row_raw(row + NW, nxt);
    } else {
      load_row_raw(row, b);
    }
    float s = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float v = b.x(k, e); s += v; sq += v * v; }
    s = wave_sum(s);
    sq = wave_sum(sq);
    const float mean = s * (1.0f / D);
    const float rstd = smd_ln_rstd(sq * (1.0f / D) - mean * mean + LN_EPS);
    bf16_t* orow = a.out + (size_t)row * D;
    float yk[F8 ? NV : 1][4];
    float amax = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = k * 256 + lane * 4;
      const float4 g4 = *reinterpret_cast<const float4*>(&prm[0][c]);
      const float4 b4 = *reinterpret_cast<const float4*>(&prm[1][c]);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = (b.x(k, e) - mean) * rstd * gg[e] + bb[e];
      if constexpr (FS) {
        const float4 s4 = *reinterpret_cast<const float4*>(&prm[2][c]);
        const float4 h4 = *reinterpret_cast<const float4*>(&prm[3][c]);
        const float ss[4] = {s4.x, s4.y, s4.z, s4.w}, hh[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = swishf_(ss[e] * y[e] + hh[e]);
      }
      if (!F8 || a.out) {
        bf16x4_t t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = f2bf(y[e]);
        *reinterpret_cast<bf16x4_t*>(orow + (l4 + k * 256)) = t;
      }
      if constexpr (F8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { yk[k][e] = y[e]; amax = fmaxf(amax, fabsf(y[e])); }
      }
      __builtin_amdgcn_sched_barrier(0);      // keep the LDS parameter reads of later chunks from being hoisted (VGPRs)
    }
    if constexpr (F8) {
      amax = wave_max(amax);
      const int ex = e4m3_row_exponent(amax);
      unsigned char* o8 = a.out_f8 + (size_t)row * D;
#pragma unroll
      for (int k = 0; k < NV; ++k)
        *reinterpret_cast<unsigned*>(o8 + (l4 + k * 256)) = pack4_e4m3(yk[k][0], yk[k][1], yk[k][2], yk[k][3], ex);
      if (lane == 0) a.out_scale[row] = (unsigned)(ex + 127);
    }
  }
}

// ------------------------------------------------------------------------------ backward
// One workgroup = one row group (a sample's rows_per_sample rows when FiLM is on, else 32 rows);
// wave w walks rows w, w+4, ... ; per-column sums live in registers and are combined through LDS.
struct LnBwdDev {
  LnArgs f;
  const bf16_t* dout;
  const float* dres;      // optional fp32 residual gradient added to dx (may alias dx_f32)
  const bf16_t* dres_bf16; // or the same in bf16 (wide8 kernel only; may alias dx_bf16)
  float* dx_f32;
  bf16_t* dx_bf16;
  float* dscale;
  float* dshift;
  int dfilm_accumulate;
  float* partial;         // [ngroups][2][D]
  int group_rows;
};

template <int D>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(LnBwdDev a) {
  typedef RowLayout<D> L;
  constexpr int PL = L::PER_LANE;
  __shared__ float red[4][D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int grp = blockIdx.x;
  const int r_begin = grp * a.group_rows;
  int r_end = r_begin + a.group_rows;
  r_end = r_end < a.f.rows ? r_end : a.f.rows;
  const bool film = a.f.film_scale != nullptr;

  float g[PL], b[PL], sc[PL], sh[PL];
  load_vec<D>(a.f.gamma, lane, g);
  load_vec<D>(a.f.beta, lane, b);
  if (film) {
    const int frow = a.f.t_ptr ? smd_clamp_t(*a.f.t_ptr, a.f.film_rows) : r_begin / a.f.rows_per_sample;
    load_vec<D>(a.f.film_scale + (size_t)frow * a.f.ld_film, lane, sc);
    load_vec<D>(a.f.film_shift + (size_t)frow * a.f.ld_film, lane, sh);
  }
  float acc_dg[PL], acc_db[PL], acc_dsc[PL], acc_dsh[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) acc_dg[i] = acc_db[i] = acc_dsc[i] = acc_dsh[i] = 0.f;

  for (int row = r_begin + w; row < r_end; row += 4) {
    float x[PL], dy[PL];
    load_row<D>(a.f.x, a.f.x_bf16, row, lane, x);
    load_row<D>(nullptr, a.dout, row, lane, dy);
    float mean, rstd;
    row_stats<D>(x, mean, rstd);
    float s1 = 0.f, s2 = 0.f;
    float xh[PL], dxh[PL];
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      xh[i] = (x[i] - mean) * rstd;
      const float n = xh[i] * g[i] + b[i];
      float d = dy[i];
      if (film) {
        const float pre = sc[i] * n + sh[i];
        if (a.f.swish) d *= swish_gradf_(pre);
        acc_dsc[i] += d * n;
        acc_dsh[i] += d;
        d *= sc[i];
      } else if (a.f.swish) {
        d *= swish_gradf_(n);
      }
      acc_dg[i] += d * xh[i];
      acc_db[i] += d;
      dxh[i] = d * g[i];
      s1 += dxh[i];
      s2 += dxh[i] * xh[i];
    }
    s1 = wave_sum(s1) * (1.0f / D);
    s2 = wave_sum(s2) * (1.0f / D);
    float dx[PL];
#pragma unroll
    for (int i = 0; i < PL; ++i) dx[i] = rstd * (dxh[i] - s1 - xh[i] * s2);
    if (a.dres) {
      float r[PL];
      load_row<D>(a.dres, nullptr, row, lane, r);
#pragma unroll
      for (int i = 0; i < PL; ++i) dx[i] += r[i];
    }
    if (a.dx_f32) store_row_f32<D>(a.dx_f32, row, lane, dx);
    if (a.dx_bf16) store_row_bf16<D>(a.dx_bf16, row, lane, dx);
  }

  // ---- combine the 4 waves' column sums through LDS, one quantity at a time
  auto combine = [&](float (&v)[PL], float* dst, int accumulate) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PL; ++i) red[w][L::col(lane, i)] = v[i];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
      const float s = red[0][c] + red[1][c] + red[2][c] + red[3][c];
      if (dst) dst[c] = accumulate ? dst[c] + s : s;
    }
  };
  combine(acc_dg, a.partial + ((size_t)grp * 2 + 0) * D, 0);
  combine(acc_db, a.partial + ((size_t)grp * 2 + 1) * D, 0);
  if (film && a.dscale) {
    const int srow = r_begin / a.f.rows_per_sample;
    combine(acc_dsc, a.dscale + (size_t)srow * a.f.ld_film, a.dfilm_accumulate);
    combine(acc_dsh, a.dshift + (size_t)srow * a.f.ld_film, a.dfilm_accumulate);
  }
}

// Wide rows (D >= 1024): the register-resident kernel above needs 8 x D/64 values per lane (256 VGPRs at
// D = 2048: one wave per SIMD, latency-bound at ~1.3 TB/s).  This variant
//   * keeps only TWO column accumulators per element.  With e = dy * act'(.) (the gradient wrt the FiLM output),
//     P_c = sum_r e_rc * xhat_rc and Q_c = sum_r e_rc give every column gradient of the row group:
//        dshift = Q, dscale = gamma*P + beta*Q, dgamma = scale*P, dbeta = scale*Q   (FiLM: scale is per group)
//        dgamma = P, dbeta = Q                                                        (no FiLM)
//   * holds gamma/beta/scale/shift in LDS (16 KiB..32 KiB per workgroup) instead of registers,
// so a wave needs < 256 VGPRs without spilling -> 2 waves per SIMD (twice the rows in flight per CU) and half the
// accumulate work per element.
template <int D>
__global__ __launch_bounds__(256, 2) void layernorm_bwd_wide_kernel(LnBwdDev a) {
  typedef RowLayout<D> L;
  constexpr int PL = L::PER_LANE, NV = L::NV;
  __shared__ __attribute__((aligned(16))) float prm[4][D];      // gamma, beta, scale, shift; later the combine buffer
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int grp = blockIdx.x;
  const int r_begin = grp * a.group_rows;
  int r_end = r_begin + a.group_rows;
  r_end = r_end < a.f.rows ? r_end : a.f.rows;
  const bool film = a.f.film_scale != nullptr;
  const bool swish = a.f.swish != 0;
  {
    const int frow = film ? (a.f.t_ptr ? smd_clamp_t(*a.f.t_ptr, a.f.film_rows) : r_begin / a.f.rows_per_sample) : 0;
    const float* src[4] = {a.f.gamma, a.f.beta, film ? a.f.film_scale + (size_t)frow * a.f.ld_film : nullptr,
                           film ? a.f.film_shift + (size_t)frow * a.f.ld_film : nullptr};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (src[q])
        for (int c = threadIdx.x * 4; c < D; c += 1024)
          *reinterpret_cast<float4*>(&prm[q][c]) = *reinterpret_cast<const float4*>(src[q] + c);
  }
  __syncthreads();

  float P[PL], Q[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) P[i] = Q[i] = 0.f;

  for (int row = r_begin + w; row < r_end; row += 4) {
    float x[PL], dxh[PL];
    bf16x4_t dyb[NV];
    load_row<D>(a.f.x, a.f.x_bf16, row, lane, x);
#pragma unroll
    for (int k = 0; k < NV; ++k) dyb[k] = *reinterpret_cast<const bf16x4_t*>(a.dout + (size_t)row * D + k * 256 + lane * 4);
    float mean, rstd;
    row_stats<D>(x, mean, rstd);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = k * 256 + lane * 4;
      const float4 g4 = *reinterpret_cast<const float4*>(&prm[0][c]);
      const float4 b4 = *reinterpret_cast<const float4*>(&prm[1][c]);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
      float ss[4] = {1.f, 1.f, 1.f, 1.f}, hh[4] = {0.f, 0.f, 0.f, 0.f};
      if (film) {
        const float4 s4 = *reinterpret_cast<const float4*>(&prm[2][c]);
        const float4 h4 = *reinterpret_cast<const float4*>(&prm[3][c]);
        ss[0] = s4.x; ss[1] = s4.y; ss[2] = s4.z; ss[3] = s4.w;
        hh[0] = h4.x; hh[1] = h4.y; hh[2] = h4.z; hh[3] = h4.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = k * 4 + e;
        const float xh = (x[i] - mean) * rstd;
        float d = bf2f(dyb[k][e]);
        if (swish) d *= swish_gradf_(ss[e] * (xh * gg[e] + bb[e]) + hh[e]);     // scale 1 / shift 0 without FiLM
        Q[i] += d;
        P[i] += d * xh;
        const float dh = d * ss[e] * gg[e];
        x[i] = xh;
        dxh[i] = dh;
        s1 += dh;
        s2 += dh * xh;
      }
      __builtin_amdgcn_sched_barrier(0);    // keep the LDS parameter reads of later chunks from being hoisted (VGPRs)
    }
    s1 = wave_sum(s1) * (1.0f / D);
    s2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
    for (int i = 0; i < PL; ++i) dxh[i] = rstd * (dxh[i] - s1 - x[i] * s2);
    if (a.dres) {
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const float4 r = *reinterpret_cast<const float4*>(a.dres + (size_t)row * D + k * 256 + lane * 4);
        dxh[k * 4 + 0] += r.x; dxh[k * 4 + 1] += r.y; dxh[k * 4 + 2] += r.z; dxh[k * 4 + 3] += r.w;
      }
    }
    if (a.dx_f32) store_row_f32<D>(a.dx_f32, row, lane, dxh);
    if (a.dx_bf16) store_row_bf16<D>(a.dx_bf16, row, lane, dxh);
  }

  // ---- combine the 4 waves' P and Q through LDS (fixed order), then expand to the column gradients
  constexpr int CPT = D / 256;                      // columns per thread: c = threadIdx.x + 256*j
  float gc[CPT], bc[CPT], sc[CPT], Pc[CPT], Qc[CPT];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int c = threadIdx.x + 256 * j;
    gc[j] = prm[0][c]; bc[j] = prm[1][c]; sc[j] = film ? prm[2][c] : 1.0f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PL; ++i) prm[w][L::col(lane, i)] = P[i];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int c = threadIdx.x + 256 * j;
    Pc[j] = (prm[0][c] + prm[1][c]) + (prm[2][c] + prm[3][c]);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PL; ++i) prm[w][L::col(lane, i)] = Q[i];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int c = threadIdx.x + 256 * j;
    Qc[j] = (prm[0][c] + prm[1][c]) + (prm[2][c] + prm[3][c]);
  }
  float* pg = a.partial + ((size_t)grp * 2 + 0) * D;
  float* pb = a.partial + ((size_t)grp * 2 + 1) * D;
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int c = threadIdx.x + 256 * j;
    pg[c] = sc[j] * Pc[j];
    pb[c] = sc[j] * Qc[j];
  }
  if (film && a.dscale) {
    const int srow = r_begin / a.f.rows_per_sample;
    float* ds = a.dscale + (size_t)srow * a.f.ld_film;
    float* dh = a.dshift + (size_t)srow * a.f.ld_film;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int c = threadIdx.x + 256 * j;
      const float vs = gc[j] * Pc[j] + bc[j] * Qc[j], vh = Qc[j];
      ds[c] = a.dfilm_accumulate ? ds[c] + vs : vs;
      dh[c] = a.dfilm_accumulate ? dh[c] + vh : vh;
    }
  }
}

// Same mathematics, laid out for memory-level parallelism.  The 4-wave kernel above runs one wave per SIMD
// (one row group per CU at B = 256) and walks load -> compute -> store per row, so a SIMD neither computes while its
// row loads nor loads while it computes (2 - 3.4 TB/s in the training step).  Here
//   * 8 waves share a row group (2 per SIMD), each walks rows w, w+8, ...;
//   * all of a row's loads (x, dy, residual gradient) are issued up front and stay raw in registers (converted at
//     use), xhat is recomputed from raw x in the dx pass, the row bases live in SGPRs;
//   * the residual gradient may be bf16 (RM = 2): the engine keeps the ResBlock residual-gradient chain in bf16.

template <int D, bool XBF, int RM, bool FS, int OM>   // FS: FiLM + swish (the ResBlock norms) or neither (plain
__global__ __launch_bounds__(512) void layernorm_bwd_wide8_kernel(LnBwdDev a) {   // LayerNorm); OM: 1 fp32 dx, 2 bf16, 3 both
  typedef RowLayout<D> L;
  constexpr int PL = L::PER_LANE, NV = L::NV, NW = 8;
  constexpr bool film = FS, swish = FS;
  __shared__ __attribute__((aligned(16))) float prm[4][D];      // gamma, beta, scale, shift; later the combine buffer
  // dh stays in registers: with 32 KiB of LDS a workgroup of this kernel still fits beside a 128-KiB wgrad workgroup
  // of the side stream (an LDS-parked dh row per wave, 96 KiB in all, made every launch queue behind the wgrads)
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform -> row bases live in SGPRs
  const int grp = blockIdx.x;
  const int r_begin = grp * a.group_rows;
  int r_end = r_begin + a.group_rows;
  r_end = r_end < a.f.rows ? r_end : a.f.rows;
  {
    const int frow = film ? (a.f.t_ptr ? smd_clamp_t(*a.f.t_ptr, a.f.film_rows) : r_begin / a.f.rows_per_sample) : 0;
    const float* src[4] = {a.f.gamma, a.f.beta, film ? a.f.film_scale + (size_t)frow * a.f.ld_film : nullptr,
                           film ? a.f.film_shift + (size_t)frow * a.f.ld_film : nullptr};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (src[q])
        for (int c = threadIdx.x * 4; c < D; c += 2048)
          *reinterpret_cast<float4*>(&prm[q][c]) = *reinterpret_cast<const float4*>(src[q] + c);
  }
  __syncthreads();

  float P[PL], Q[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) P[i] = Q[i] = 0.f;
  const uint32_t l4 = lane * 4;                     // element offset of this lane inside a 256-column chunk

  auto issue = [&](int row, WideRow<D, XBF>& b) {
    const float* xr = a.f.x + (size_t)row * D;           // uniform row bases
    const bf16_t* xbr = a.f.x_bf16 + (size_t)row * D;
    const bf16_t* dyr = a.dout + (size_t)row * D;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if constexpr (XBF) b.xb[k] = *reinterpret_cast<const bf16x4_t*>(xbr + (l4 + k * 256));
      else b.xf[k] = *reinterpret_cast<const float4*>(xr + (l4 + k * 256));
      b.dy[k] = *reinterpret_cast<const bf16x4_t*>(dyr + (l4 + k * 256));
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto compute = [&](int row, WideRow<D, XBF>& b) {
    auto fence_x = [&]() {             // bf16 rows: re-convert per pass instead of keeping 32 converted registers alive
      if constexpr (XBF) {
#pragma unroll
        for (int k = 0; k < NV; ++k) asm volatile("" : "+v"(b.xb[k]));
      }
    };
    // pass 1: row statistics straight from the raw registers
    float s = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float v = b.x(k, e); s += v; sq += v * v; }
    s = wave_sum(s);
    sq = wave_sum(sq);
    const float mean = s * (1.0f / D);
    const float rstd = smd_ln_rstd(sq * (1.0f / D) - mean * mean + LN_EPS);
    fence_x();
    // pass 2: column sums P, Q and the row sums
    float s1 = 0.f, s2 = 0.f;
    float dhr[PL];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = k * 256 + lane * 4;
      const float4 g4 = *reinterpret_cast<const float4*>(&prm[0][c]);
      const float4 b4 = *reinterpret_cast<const float4*>(&prm[1][c]);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
      float ss[4] = {1.f, 1.f, 1.f, 1.f}, hh[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (film) {
        const float4 s4 = *reinterpret_cast<const float4*>(&prm[2][c]);
        const float4 h4 = *reinterpret_cast<const float4*>(&prm[3][c]);
        ss[0] = s4.x; ss[1] = s4.y; ss[2] = s4.z; ss[3] = s4.w;
        hh[0] = h4.x; hh[1] = h4.y; hh[2] = h4.z; hh[3] = h4.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = k * 4 + e;
        const float xh = (b.x(k, e) - mean) * rstd;
        float d = bf2f(b.dy[k][e]);
        if constexpr (swish) d *= swish_gradf_(ss[e] * (xh * gg[e] + bb[e]) + hh[e]);
        Q[i] += d;
        P[i] += d * xh;
        const float dhe = d * ss[e] * gg[e];
        dhr[i] = dhe;
        s1 += dhe;
        s2 += dhe * xh;
      }
      __builtin_amdgcn_sched_barrier(0);    // keep the LDS parameter reads of later chunks from being hoisted (VGPRs)
    }
    s1 = wave_sum(s1) * (1.0f / D);
    s2 = wave_sum(s2) * (1.0f / D);
    fence_x();
    // pass 3: dx = rstd * (dh - mean(dh) - xhat * mean(dh * xhat)) (+ residual gradient)
    float4 rf[RM == 1 ? NV : 1];
    bf16x4_t rb[RM == 2 ? NV : 1];
    if constexpr (RM == 1) {           // residual rows: all loads first (dres may alias dx: the stores below pin them)
      const float* rr_f = a.dres + (size_t)row * D;
#pragma unroll
      for (int k = 0; k < NV; ++k) rf[k] = *reinterpret_cast<const float4*>(rr_f + (l4 + k * 256));
    }
    if constexpr (RM == 2) {
      const bf16_t* rr_b = a.dres_bf16 + (size_t)row * D;
#pragma unroll
      for (int k = 0; k < NV; ++k) rb[k] = *reinterpret_cast<const bf16x4_t*>(rr_b + (l4 + k * 256));
    }
    float* of = (OM & 1) ? a.dx_f32 + (size_t)row * D : nullptr;
    bf16_t* ob = (OM & 2) ? a.dx_bf16 + (size_t)row * D : nullptr;
    const float m2 = rstd * rstd * s2, c0 = rstd * (s1 - mean * rstd * s2);     // dx = rstd*dh - m2*x - c0
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      float rr[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (RM == 1) { rr[0] = rf[k].x; rr[1] = rf[k].y; rr[2] = rf[k].z; rr[3] = rf[k].w; }
      if constexpr (RM == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) rr[e] = bf2f(rb[k][e]);
      }
      const float dh[4] = {dhr[k * 4 + 0], dhr[k * 4 + 1], dhr[k * 4 + 2], dhr[k * 4 + 3]};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (rstd * dh[e] - m2 * b.x(k, e) - c0) + rr[e];
      if constexpr ((OM & 1) != 0) *reinterpret_cast<float4*>(of + (l4 + k * 256)) = make_float4(o[0], o[1], o[2], o[3]);
      if constexpr ((OM & 2) != 0) {
        bf16x4_t t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = f2bf(o[e]);
        *reinterpret_cast<bf16x4_t*>(ob + (l4 + k * 256)) = t;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // (A register double buffer of the next row was tried: the compiler spills it at D = 2048 and waits on the spill
  // stores, which serialises exactly what it was meant to overlap.  Two waves per SIMD already overlap one wave's
  // loads with the other's arithmetic.)
  for (int row = r_begin + w; row < r_end; row += NW) {
    WideRow<D, XBF> buf;
    issue(row, buf);
    compute(row, buf);
  }

  // ---- combine the 8 waves' P and Q through LDS (two rounds of four, fixed order), then expand
  constexpr int CPT = D / 512;                      // columns per thread: c = threadIdx.x + 512*j
  float gc[CPT], bc[CPT], sc[CPT], Pc[CPT], Qc[CPT];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int c = threadIdx.x + 512 * j;
    gc[j] = prm[0][c]; bc[j] = prm[1][c]; sc[j] = film ? prm[2][c] : 1.0f;
    Pc[j] = Qc[j] = 0.f;
  }
#pragma unroll
  for (int which = 0; which < 2; ++which)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
      if ((w >> 2) == half) {
#pragma unroll
        for (int i = 0; i < PL; ++i) prm[w & 3][L::col(lane, i)] = which ? Q[i] : P[i];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        const int c = threadIdx.x + 512 * j;
        const float v = (prm[0][c] + prm[1][c]) + (prm[2][c] + prm[3][c]);
        if (which) Qc[j] += v; else Pc[j] += v;
      }
    }
  float* pg = a.partial + ((size_t)grp * 2 + 0) * D;
  float* pb = a.partial + ((size_t)grp * 2 + 1) * D;
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int c = threadIdx.x + 512 * j;
    pg[c] = sc[j] * Pc[j];
    pb[c] = sc[j] * Qc[j];
  }
  if (film && a.dscale) {
    const int srow = r_begin / a.f.rows_per_sample;
    float* ds = a.dscale + (size_t)srow * a.f.ld_film;
    float* dh = a.dshift + (size_t)srow * a.f.ld_film;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int c = threadIdx.x + 512 * j;
      const float vs = gc[j] * Pc[j] + bc[j] * Qc[j], vh = Qc[j];
      ds[c] = a.dfilm_accumulate ? ds[c] + vs : vs;
      dh[c] = a.dfilm_accumulate ? dh[c] + vh : vh;
    }
  }
}

// dgamma[c] += sum_g partial[g][0][c] ; dbeta likewise (fixed order -> deterministic)
// 64 columns x 4 group-slices per block: many independent loads in flight instead of one long
// dependent chain per column (the one-thread-per-column version was latency-bound at ~65 us).
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partial, int ngroups, int D,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);      // column in [0, 2D)
  const int slice = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < 2 * D) {
    const int which = c / D, cc = c - which * D;
    const float* base = partial + (size_t)which * D + cc;
    int gidx = slice;
    for (; gidx + 12 < ngroups; gidx += 16) {
      s0 += base[(size_t)gidx * 2 * D];
      s1 += base[(size_t)(gidx + 4) * 2 * D];
      s2 += base[(size_t)(gidx + 8) * 2 * D];
      s3 += base[(size_t)(gidx + 12) * 2 * D];
    }
    for (; gidx < ngroups; gidx += 4) s0 += base[(size_t)gidx * 2 * D];
  }
  red[slice][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (slice == 0 && c < 2 * D) {
    const int which = c / D, cc = c - which * D;
    float* dst = which ? dbeta : dgamma;
    const int l = threadIdx.x;
    dst[cc] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);      // written, not accumulated: no gradient memset
  }
}

// ------------------------------------------------------------------ D = 128 (the encoder's residual-stream norms)
// A 512-byte row is too small for one wave: 16 lanes own a row (8 columns each), a wave walks 4 rows at once and
// both of its row quartets are loaded up front; row statistics are 4-step xor-shuffles inside the 16-lane group;
// the P/Q column accumulators (see the wide kernel) are folded over the row slots with two more shuffles.
// XF: x is fp32 (else bf16); DRES: fp32 residual gradient present; F32 / B16: which outputs exist -- compile-time, so
// that every load / store is unconditional straight-line code.
template <bool XF, bool DRES, bool F32, bool B16>
__global__ __launch_bounds__(256) void layernorm_bwd_narrow128_kernel(LnBwdDev a) {
  constexpr int D = 128;
  __shared__ float red[16][2][D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = lane >> 4, c0 = (lane & 15) * 8;
  // XCD-banded row groups (block b runs on XCD b % 8): XCD x owns the contiguous band of groups [x*n/8, (x+1)*n/8),
  // the same band -> XCD affinity the row-tiled GEMMs use, so a row block is produced and consumed through one L2
  const int grp = smd_xcd_band(blockIdx.x, gridDim.x);
  const int r_begin = grp * a.group_rows;             // group_rows == 32
  float g[8];
  {
    const float4 g0 = *reinterpret_cast<const float4*>(a.f.gamma + c0), g1 = *reinterpret_cast<const float4*>(a.f.gamma + c0 + 4);
    g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
  }
  float P[8], Q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) P[i] = Q[i] = 0.f;
  float x[2][8], r[2][8];
  bf16x8_t dyb[2];
  bool valid[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int row = r_begin + it * 16 + w * 4 + sub;
    valid[it] = row < a.f.rows;
    const size_t off = (size_t)(valid[it] ? row : 0) * D + c0;
    if constexpr (XF) {
      const float4 t0 = *reinterpret_cast<const float4*>(a.f.x + off), t1 = *reinterpret_cast<const float4*>(a.f.x + off + 4);
      x[it][0] = t0.x; x[it][1] = t0.y; x[it][2] = t0.z; x[it][3] = t0.w; x[it][4] = t1.x; x[it][5] = t1.y; x[it][6] = t1.z; x[it][7] = t1.w;
    } else {
      const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(a.f.x_bf16 + off);
#pragma unroll
      for (int i = 0; i < 8; ++i) x[it][i] = bf2f(t[i]);
    }
    dyb[it] = *reinterpret_cast<const bf16x8_t*>(a.dout + off);
    if constexpr (DRES) {
      const float4 t0 = *reinterpret_cast<const float4*>(a.dres + off), t1 = *reinterpret_cast<const float4*>(a.dres + off + 4);
      r[it][0] = t0.x; r[it][1] = t0.y; r[it][2] = t0.z; r[it][3] = t0.w; r[it][4] = t1.x; r[it][5] = t1.y; r[it][6] = t1.z; r[it][7] = t1.w;
    }
  }
  // all-reduce over the 16 lanes of a row: four xor exchanges (__shfl_xor = ds_bpermute); every step adds two commuting
  // operands, so all 16 lanes end with the bitwise identical sum.  Until round 3 these were DPP row operations
  // (quad_perm / row_half_mirror / row_mirror), suspected of the co-residency miscompare while it was being hunted; the
  // root cause turned out to be the v_rsq_f32 behind the instruction writing its source (smd_ln_rstd)
  // and both reduction forms behaved alike in those runs.  SMD_NARROW_DPP=1 rebuilds the DPP form for the A/B.
#ifndef SMD_NARROW_DPP
#define SMD_NARROW_DPP 0
#endif
  auto sum16 = [](float v) {
#if SMD_NARROW_DPP
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
#else
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
#endif
    return v;
  };
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int row = r_begin + it * 16 + w * 4 + sub;
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s += x[it][i]; s2 += x[it][i] * x[it][i]; }
    s = sum16(s); s2 = sum16(s2);
    const float mean = s * (1.0f / D);
    const float rstd = smd_ln_rstd(s2 * (1.0f / D) - mean * mean + LN_EPS);
    float dxh[8], xh[8], t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xh[i] = (x[it][i] - mean) * rstd;
      const float d = valid[it] ? bf2f(dyb[it][i]) : 0.f;
      Q[i] += d;
      P[i] += d * xh[i];
      dxh[i] = d * g[i];
      t1 += dxh[i];
      t2 += dxh[i] * xh[i];
    }
    t1 = sum16(t1) * (1.0f / D);
    t2 = sum16(t2) * (1.0f / D);
    float dx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dx[i] = rstd * (dxh[i] - t1 - xh[i] * t2) + (DRES ? r[it][i] : 0.f);
    if (valid[it]) {
      const size_t off = (size_t)row * D + c0;
      if constexpr (F32) {
        *reinterpret_cast<float4*>(a.dx_f32 + off) = make_float4(dx[0], dx[1], dx[2], dx[3]);
        *reinterpret_cast<float4*>(a.dx_f32 + off + 4) = make_float4(dx[4], dx[5], dx[6], dx[7]);
      }
      if constexpr (B16) {
        bf16x8_t o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = f2bf(dx[i]);
        *reinterpret_cast<bf16x8_t*>(a.dx_bf16 + off) = o;
      }
    }
  }
  // the 16 row slots of the workgroup (4 waves x 4) are folded through LDS in a fixed order
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[w * 4 + sub][0][c0 + i] = P[i]; red[w * 4 + sub][1][c0 + i] = Q[i]; }
  __syncthreads();
  {
    const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
    float acc = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) acc += red[sl][which][c];
    a.partial[((size_t)grp * 2 + which) * D + c] = acc;
  }
}

// dgamma / dbeta of several LayerNorms in one launch: entry e sums partial_e[g][0|1][c] over its groups in a fixed
// order (4 group slices per block, combined through LDS) and WRITES the gradient (no memset needed, no accumulation).
__global__ __launch_bounds__(256) void ln_bwd_reduce_batched_kernel(LnReduceTable t) {
  __shared__ float red[4][64];
  int e = 0;
  while (e + 1 < t.n && (int)blockIdx.x >= t.e[e + 1].block_start) ++e;
  const LnReduceEntry en = t.e[e];
  const int D = en.D;
  const int c = ((int)blockIdx.x - en.block_start) * 64 + (threadIdx.x & 63);      // column in [0, 2D)
  const int slice = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < 2 * D) {
    const int which = c / D, cc = c - which * D;
    const float* base = en.partial + (size_t)which * D + cc;
    int gidx = slice;
    for (; gidx + 12 < en.ngroups; gidx += 16) {
      s0 += base[(size_t)gidx * 2 * D];
      s1 += base[(size_t)(gidx + 4) * 2 * D];
      s2 += base[(size_t)(gidx + 8) * 2 * D];
      s3 += base[(size_t)(gidx + 12) * 2 * D];
    }
    for (; gidx < en.ngroups; gidx += 4) s0 += base[(size_t)gidx * 2 * D];
  }
  red[slice][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (slice == 0 && c < 2 * D) {
    const int which = c / D, cc = c - which * D;
    float* dst = which ? en.dbeta : en.dgamma;
    const int l = threadIdx.x;
    dst[cc] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);      // written, not accumulated: no gradient memset
  }
}

int ln_group_rows(const LnArgs& f) { return f.film_scale ? f.rows_per_sample : 32; }

}  // namespace

size_t ln_bwd_partial_elems(int rows, int D) {
  // worst case group size 1 (DenseDDPM: one FiLM row per sample)
  return (size_t)rows * 2 * D;
}

static int check_ln(const LnArgs& a, bool need_out) {
  SMD_ARG_CHECK((a.x != nullptr) != (a.x_bf16 != nullptr), "layernorm: exactly one of x / x_bf16");
  SMD_ARG_CHECK(a.gamma && a.beta && (a.out || a.out_f8 || !need_out), "layernorm: null gamma/beta/out");
  SMD_ARG_CHECK(!a.out_f8 || (a.out_scale && (a.D == 1024 || a.D == 2048) && ((a.film_scale != nullptr) == (a.swish != 0))),
                "layernorm: e4m3 output needs out_scale, D in {1024, 2048} and FiLM+swish or neither");
  SMD_ARG_CHECK(a.rows > 0, "layernorm: rows=%d", a.rows);
  SMD_ARG_CHECK((a.film_scale != nullptr) == (a.film_shift != nullptr), "layernorm: scale/shift must come together");
  if (a.film_scale) {
    SMD_ARG_CHECK(a.rows_per_sample > 0 && a.rows % a.rows_per_sample == 0 && a.ld_film >= a.D,
                  "layernorm: bad FiLM geometry rows=%d rows_per_sample=%d ld_film=%d", a.rows,
                  a.rows_per_sample, a.ld_film);
  }
  return 0;
}

#define SMD_LN_DISPATCH(D_, KERNEL, ...)                                                     \
  switch (D_) {                                                                              \
    case 128: KERNEL<128> __VA_ARGS__; break;                                                \
    case 256: KERNEL<256> __VA_ARGS__; break;                                                \
    case 512: KERNEL<512> __VA_ARGS__; break;                                                \
    case 1024: KERNEL<1024> __VA_ARGS__; break;                                              \
    case 2048: KERNEL<2048> __VA_ARGS__; break;                                              \
    case 4096: KERNEL<4096> __VA_ARGS__; break;                                              \
    default: smd_set_error("layernorm: unsupported width D=%d (128,256,512,1024,2048,4096)", D_); return -1; \
  }

template <int D> static void run_fwd(const LnArgs& a, hipStream_t st) {
  if constexpr (D == 1024 || D == 2048) {
    const bool fs = a.film_scale && a.swish, plain = !a.film_scale && !a.swish;
    const int gr = (a.film_scale && !a.t_ptr) ? a.rows_per_sample : 32;
    if ((smd_tuning_get("ln_fwd_wide") || a.out_f8) && (fs || plain) && gr >= 8) {
      const int ng = (a.rows + gr - 1) / gr;
#define SMD_FW(XB, FS_)                                                                                               \
  do {                                                                                                               \
    if (a.out_f8 && smd_tuning_get("ln_fwd_wide") == 3) hipLaunchKernelGGL((layernorm_fwd_wide_kernel<D, XB, FS_, 16, true, true>), dim3(ng), dim3(1024), 0, st, a, gr); \
    else if (a.out_f8 && smd_tuning_get("ln_fwd_wide") == 2) hipLaunchKernelGGL((layernorm_fwd_wide_kernel<D, XB, FS_, 16, true>), dim3(ng), dim3(1024), 0, st, a, gr); \
    else if (a.out_f8) hipLaunchKernelGGL((layernorm_fwd_wide_kernel<D, XB, FS_, 8, true>), dim3(ng), dim3(512), 0, st, a, gr);       \
    else if (smd_tuning_get("ln_fwd_wide") == 2) hipLaunchKernelGGL((layernorm_fwd_wide_kernel<D, XB, FS_, 16>), dim3(ng), dim3(1024), 0, st, a, gr); \
    else if (smd_tuning_get("ln_fwd_wide") == 3) hipLaunchKernelGGL((layernorm_fwd_wide_kernel<D, XB, FS_, 16, false, true>), dim3(ng), dim3(1024), 0, st, a, gr); \
    else hipLaunchKernelGGL((layernorm_fwd_wide_kernel<D, XB, FS_, 8>), dim3(ng), dim3(512), 0, st, a, gr);            \
  } while (0)
      if (a.x_bf16) { if (fs) SMD_FW(true, true); else SMD_FW(true, false); }
      else          { if (fs) SMD_FW(false, true); else SMD_FW(false, false); }
#undef SMD_FW
      return;
    }
  }
  hipLaunchKernelGGL(layernorm_fwd_kernel<D>, dim3((a.rows + 3) / 4), dim3(256), 0, st, a);
}
template <int D> static void run_bwd(const LnBwdDev& d, int ngroups, hipStream_t st) {
  if constexpr (D == 2048) {
    const int mode = smd_tuning_get("ln_bwd_wide");
    const bool fs = d.f.film_scale && d.f.swish, plain = !d.f.film_scale && !d.f.swish;
    if ((mode >= 2 || d.dres_bf16) && (fs || plain)) {
      const int rm = d.dres_bf16 ? 2 : (d.dres ? 1 : 0);
#define SMD_W8O(XB, RM_, FS_, OM_) \
  hipLaunchKernelGGL((layernorm_bwd_wide8_kernel<D, XB, RM_, FS_, OM_>), dim3(ngroups), dim3(512), 0, st, d)
#define SMD_W8F(XB, RM_, FS_) \
  do { if (om == 3) SMD_W8O(XB, RM_, FS_, 3); else if (om == 2) SMD_W8O(XB, RM_, FS_, 2); else SMD_W8O(XB, RM_, FS_, 1); } while (0)
#define SMD_W8(XB, RM_) do { if (fs) SMD_W8F(XB, RM_, true); else SMD_W8F(XB, RM_, false); } while (0)
      const int om = (d.dx_f32 ? 1 : 0) | (d.dx_bf16 ? 2 : 0);
      if (d.f.x_bf16) { if (rm == 2) SMD_W8(true, 2); else if (rm == 1) SMD_W8(true, 1); else SMD_W8(true, 0); }
      else            { if (rm == 2) SMD_W8(false, 2); else if (rm == 1) SMD_W8(false, 1); else SMD_W8(false, 0); }
#undef SMD_W8O
#undef SMD_W8F
#undef SMD_W8
      return;
    }
  }
  if constexpr (D >= 1024 && D <= 2048) {      // 4096: 64 KiB of LDS parameters, keep the register kernel
    if (smd_tuning_get("ln_bwd_wide")) {
      hipLaunchKernelGGL(layernorm_bwd_wide_kernel<D>, dim3(ngroups), dim3(256), 0, st, d);
      return;
    }
  }
  hipLaunchKernelGGL(layernorm_bwd_kernel<D>, dim3(ngroups), dim3(256), 0, st, d);
}

int launch_layernorm_fwd(const LnArgs& a, hipStream_t st) {
  int rc = check_ln(a, true);
  if (rc) return rc;
  SMD_LN_DISPATCH(a.D, run_fwd, (a, st));
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_layernorm_bwd(const LnBwdArgs& a, hipStream_t st) {
  int rc = check_ln(a.f, false);
  if (rc) return rc;
  SMD_ARG_CHECK(a.dout && a.partial && a.dgamma && a.dbeta, "layernorm_bwd: null dout/partial/dgamma/dbeta");
  SMD_ARG_CHECK(a.dx || a.dx_bf16, "layernorm_bwd: no dx output");
  SMD_ARG_CHECK(!(a.dres && a.dres_bf16), "layernorm_bwd: dres and dres_bf16 are exclusive");
  SMD_ARG_CHECK(!a.dres_bf16 || (a.f.D == 2048 && ((a.f.film_scale != nullptr) == (a.f.swish != 0))),
                "layernorm_bwd: dres_bf16 needs D = 2048 and FiLM+swish or neither");
  const int gr = ln_group_rows(a.f);
  const int ngroups = (a.f.rows + gr - 1) / gr;
  SMD_ARG_CHECK(a.partial_elems >= (size_t)ngroups * 2 * a.f.D, "layernorm_bwd: workspace too small");
  LnBwdDev d;
  d.f = a.f;
  d.dout = a.dout;
  d.dres = a.dres;
  d.dres_bf16 = a.dres_bf16;
  d.dx_f32 = a.dx;
  d.dx_bf16 = a.dx_bf16;
  d.dscale = a.dscale;
  d.dshift = a.dshift;
  d.dfilm_accumulate = a.dfilm_accumulate;
  d.partial = a.partial;
  d.group_rows = gr;
  if (a.f.D == 128 && !a.f.film_scale && !a.f.swish && gr == 32 && smd_tuning_get("ln_bwd_narrow")) {
    // the engine's three uses: fp32 x, residual or not, fp32 + bf16 outputs or bf16 only
#define SMD_NARROW(XF_, DR_, F_, B_) hipLaunchKernelGGL((layernorm_bwd_narrow128_kernel<XF_, DR_, F_, B_>), dim3(ngroups), dim3(256), 0, st, d)
    const int key = (d.f.x ? 8 : 0) | (d.dres ? 4 : 0) | (d.dx_f32 ? 2 : 0) | (d.dx_bf16 ? 1 : 0);
    switch (key) {
      case 15: SMD_NARROW(true, true, true, true); break;
      case 11: SMD_NARROW(true, false, true, true); break;
      case 9: SMD_NARROW(true, false, false, true); break;
      case 14: SMD_NARROW(true, true, true, false); break;
      case 10: SMD_NARROW(true, false, true, false); break;
      case 13: SMD_NARROW(true, true, false, true); break;
      case 7: SMD_NARROW(false, true, true, true); break;
      case 3: SMD_NARROW(false, false, true, true); break;
      case 1: SMD_NARROW(false, false, false, true); break;
      default: SMD_LN_DISPATCH(a.f.D, run_bwd, (d, ngroups, st)); break;
    }
#undef SMD_NARROW
  } else {
    SMD_LN_DISPATCH(a.f.D, run_bwd, (d, ngroups, st));
  }
  SMD_LAUNCH_CHECK();
  if (a.deferred) {          // the caller batches the dgamma/dbeta reductions (launch_ln_bwd_reduce_batched)
    a.deferred->partial = a.partial; a.deferred->ngroups = ngroups; a.deferred->D = a.f.D;
    a.deferred->dgamma = a.dgamma; a.deferred->dbeta = a.dbeta; a.deferred->block_start = 0;
    return 0;
  }
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * a.f.D + 63) / 64), dim3(256), 0, st, a.partial, ngroups,
                     a.f.D, a.dgamma, a.dbeta);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_ln_bwd_reduce_batched(const LnReduceEntry* entries, int n, hipStream_t st) {
  int i = 0;
  while (i < n) {
    LnReduceTable t;
    t.n = 0;
    int blocks = 0;
    for (; i < n && t.n < SMD_LN_REDUCE_MAX; ++i) {
      t.e[t.n] = entries[i];
      t.e[t.n].block_start = blocks;
      blocks += (2 * entries[i].D + 63) / 64;
      ++t.n;
    }
    hipLaunchKernelGGL(ln_bwd_reduce_batched_kernel, dim3(blocks), dim3(256), 0, st, t);
    SMD_LAUNCH_CHECK();
  }
  return 0;
}

// rows of a bf16 matrix -> e4m3 + per-row E8M0 scale: one wave per row, 8 elements per lane and pass
namespace {
__global__ __launch_bounds__(256) void quantize_rows_e4m3_kernel(const bf16_t* __restrict__ in, int ld, int rows, int K,
                                                                 unsigned char* __restrict__ out8, uint32_t* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* src = in + (size_t)row * ld;
  float amax = 0.f;
  for (int c = lane * 8; c < K; c += 512) {
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(src + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(bf2f(v[e])));
  }
  amax = wave_max(amax);
  const int ex = e4m3_row_exponent(amax);
  unsigned char* dst = out8 + (size_t)row * K;
  for (int c = lane * 8; c < K; c += 512) {
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(src + c);
    uint2 o;
    o.x = pack4_e4m3(bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3]), ex);
    o.y = pack4_e4m3(bf2f(v[4]), bf2f(v[5]), bf2f(v[6]), bf2f(v[7]), ex);
    *reinterpret_cast<uint2*>(dst + c) = o;
  }
  if (lane == 0) scale[row] = (unsigned)(ex + 127);
}
}  // namespace

int launch_quantize_rows_e4m3(const bf16_t* in, int ld, int rows, int K, unsigned char* out8, uint32_t* scale, hipStream_t st) {
  SMD_ARG_CHECK(in && out8 && scale && rows > 0 && K > 0 && K % 8 == 0 && ld >= K && ld % 8 == 0, "quantize_rows_e4m3: bad arguments");
  hipLaunchKernelGGL(quantize_rows_e4m3_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, in, ld, rows, K, out8, scale);
  SMD_LAUNCH_CHECK();
  return 0;
}
