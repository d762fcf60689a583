// Fused multi-tensor optimiser over the flat fp32 parameter buffer (HBM-bound, 28 B/param + 4 B/param of bf16 operands):
//   pass 1  grad_sumsq_slots  per-block partial sums of g^2 into FIXED slots (global L2 norm, deterministic order); the
//                             engine runs the output-stage slice's pass on its side stream under the encoder backward
//   pass 2  opt_prepare       one workgroup: norm, clip factor, stepped LR, bias corrections -> 4 device constants; step += 1
//   pass 3  adam_recast       clip_grads + Adam + EMA AND the bf16 re-cast of every Dense kernel in both operand layouts in
//                             ONE sweep: a workgroup owns a 64 x 64 tile of a (K, N) kernel, a thread a 4 x 4 micro-tile that it
//                             transposes in registers (no LDS, no second read of the 106 MB the update just wrote)
// The engine-less entry point smd_adam_clip_ema keeps the plain flat kernels (grad_sumsq / adam_clip_ema), and
// recast_all stays for loads / initialisation.
//
// Reference: jax.experimental.optimizers.clip_grads / l2_norm and flax.optim.Adam as used at
// train_ncsn.py:284-287; stepped LR schedule train_ncsn.py:340-342; EMAHelper.update
// utils/train_utils.py:73-78.  The step counter lives in device memory: the LR, the Adam bias
// correction and the RNG stream offset are all derived from it on device (graph-replayable).
#include "smd_kernels.h"

namespace {

constexpr int MAX_NORM_BLOCKS = 1024;

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ g, size_t n,
                                                         float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
    const float v = g[n4 * 4 + threadIdx.x];
    acc += v * v;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

struct AdamDev {
  float* p; const float* g; float* m; float* v; float* ema;
  size_t n;
  float lr0, lr_gamma; int lr_interval;
  float beta1, beta2, eps, grad_clip, mu, grad_scale;
  const uint32_t* step_ptr;
  const float* norm_partial; int norm_blocks;
  float* metrics_out;
};

__global__ __launch_bounds__(256) void adam_clip_ema_kernel(AdamDev a) {
  __shared__ float red[4];
  __shared__ float s_norm;
  // every block reduces the (<=1024) partials itself, in the same order: identical norm everywhere
  float acc = 0.f;
  for (int i = threadIdx.x; i < a.norm_blocks; i += 256) acc += a.norm_partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) s_norm = sqrtf(red[0] + red[1] + red[2] + red[3]) * a.grad_scale;
  __syncthreads();
  const float norm = s_norm;
  // clip_grads: where(norm < max_norm, g, g * max_norm / norm)
  const float gmul = a.grad_scale * (norm < a.grad_clip ? 1.0f : a.grad_clip / norm);
  const uint32_t step = *a.step_ptr;
  // stepped schedule: lr0 * gamma ** max(0, ceil(step/interval) - 1)
  const int idx = (int)((step + (uint32_t)a.lr_interval - 1u) / (uint32_t)a.lr_interval) - 1;
  const float lr = a.lr0 * powf(a.lr_gamma, (float)(idx > 0 ? idx : 0));
  const float tt = (float)(step + 1u);
  const float bc1 = 1.0f - powf(a.beta1, tt), bc2 = 1.0f - powf(a.beta2, tt);
  const float inv_bc1 = 1.0f / bc1, inv_bc2 = 1.0f / bc2;
  const float ob1 = 1.0f - a.beta1, ob2 = 1.0f - a.beta2, omu = 1.0f - a.mu;

  const size_t n4 = a.n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 p = reinterpret_cast<float4*>(a.p)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i];
    float4 m = reinterpret_cast<float4*>(a.m)[i];
    float4 v = reinterpret_cast<float4*>(a.v)[i];
    float* pp = &p.x; const float* gg = &g.x; float* mm = &m.x; float* vv = &v.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = gg[k] * gmul;
      mm[k] = a.beta1 * mm[k] + ob1 * gk;
      vv[k] = a.beta2 * vv[k] + ob2 * gk * gk;
      pp[k] -= lr * (mm[k] * inv_bc1) / (sqrtf(vv[k] * inv_bc2) + a.eps);
    }
    reinterpret_cast<float4*>(a.p)[i] = p;
    reinterpret_cast<float4*>(a.m)[i] = m;
    reinterpret_cast<float4*>(a.v)[i] = v;
    if (a.ema) {
      float4 e = reinterpret_cast<float4*>(a.ema)[i];
      e.x = e.x * a.mu + p.x * omu; e.y = e.y * a.mu + p.y * omu;
      e.z = e.z * a.mu + p.z * omu; e.w = e.w * a.mu + p.w * omu;
      reinterpret_cast<float4*>(a.ema)[i] = e;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(a.n - n4 * 4)) {
    const size_t i = n4 * 4 + threadIdx.x;
    const float gk = a.g[i] * gmul;
    const float m = a.beta1 * a.m[i] + ob1 * gk, v = a.beta2 * a.v[i] + ob2 * gk * gk;
    const float p = a.p[i] - lr * (m * inv_bc1) / (sqrtf(v * inv_bc2) + a.eps);
    a.m[i] = m; a.v[i] = v; a.p[i] = p;
    if (a.ema) a.ema[i] = a.ema[i] * a.mu + p * omu;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.metrics_out) {
    a.metrics_out[0] = norm;
    a.metrics_out[1] = norm < a.grad_clip ? norm : a.grad_clip;   // l2_norm of the clipped grads
    a.metrics_out[2] = lr;
    a.metrics_out[3] = (float)step;
  }
}

__global__ void bump_step_kernel(uint32_t* step_ptr) { *step_ptr += 1u; }

// one (K_in, N_out) fp32 kernel -> W [K_in][ldw] (same orientation) and Wt [N_out][ldwt]
__global__ __launch_bounds__(256) void recast_weight_kernel(const float* __restrict__ w, int K_in, int N_out,
                                                            bf16_t* __restrict__ W, int ldw,
                                                            bf16_t* __restrict__ Wt, int ldwt) {
  __shared__ float tile[64][65];
  const int k0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int k = k0 + i, n = n0 + tx;
    float v = 0.f;
    if (k < K_in && n < N_out) {
      v = w[(size_t)k * N_out + n];
      if (W) W[(size_t)k * ldw + n] = f2bf(v);
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  if (Wt) {
    for (int i = ty; i < 64; i += 4) {
      const int n = n0 + i, k = k0 + tx;
      if (n < N_out && k < K_in) Wt[(size_t)n * ldwt + k] = f2bf(tile[tx][i]);
    }
  }
}

// every Dense kernel of the model in ONE launch: block -> (weight, 64x64 tile) through a small table.
// Interior tiles of 16-byte-aligned weights move as float4 loads and 8-byte bf16 stores in both orientations (thread
// = 4 consecutive n for W, 4 consecutive k for Wt, through a padded LDS tile); edge or unaligned tiles go scalar.
__global__ __launch_bounds__(256) void recast_all_kernel(const float* __restrict__ params, bf16_t* __restrict__ wpack,
                                                         RecastTable t) {
  __shared__ float tile[64][65];
  int wi = 0;
#pragma unroll 1
  for (int i = 1; i < t.n; ++i)
    if ((int)blockIdx.x >= (int)t.e[i].tile_start) wi = i;
  const RecastEntry e = t.e[wi];
  const int local = blockIdx.x - e.tile_start;
  const int tiles_n = (e.N + 63) / 64;
  const int k0 = (local / tiles_n) * 64, n0 = (local % tiles_n) * 64;
  const float* w = params + e.w_off;
  bf16_t* W = wpack + e.W_off;
  bf16_t* Wt = wpack + e.Wt_off;
  const bool full = k0 + 64 <= (int)e.K && n0 + 64 <= (int)e.N;
  const bool aligned = ((e.w_off | e.N | e.W_off | e.ldw | e.Wt_off | e.ldwt) & 3u) == 0;
  if (full && aligned) {
    const int c4 = (threadIdx.x & 15) * 4, r = threadIdx.x >> 4;          // 16 threads per row, 16 rows per pass
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int k = r + 16 * p;
      const float4 v = *reinterpret_cast<const float4*>(w + (size_t)(k0 + k) * e.N + n0 + c4);
      bf16x4_t o;
      o[0] = f2bf(v.x); o[1] = f2bf(v.y); o[2] = f2bf(v.z); o[3] = f2bf(v.w);
      *reinterpret_cast<bf16x4_t*>(W + (size_t)(k0 + k) * e.ldw + n0 + c4) = o;
      tile[k][c4 + 0] = v.x; tile[k][c4 + 1] = v.y; tile[k][c4 + 2] = v.z; tile[k][c4 + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int n = r + 16 * p;                                            // row of Wt; this thread: k = c4 .. c4+3
      bf16x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = f2bf(tile[c4 + j][n]);
      *reinterpret_cast<bf16x4_t*>(Wt + (size_t)(n0 + n) * e.ldwt + k0 + c4) = o;
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int k = k0 + i, n = n0 + tx;
    float v = 0.f;
    if (k < (int)e.K && n < (int)e.N) {
      v = w[(size_t)k * e.N + n];
      W[(size_t)k * e.ldw + n] = f2bf(v);
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int n = n0 + i, k = k0 + tx;
    if (n < (int)e.N && k < (int)e.K) Wt[(size_t)n * e.ldwt + k] = f2bf(tile[tx][i]);
  }
}

// ---- engine path: fixed-slot norm partials, one-workgroup prepare, Adam + EMA + bf16 re-cast in one sweep ----------------
// slot b of `partial` <- sum of g[i]^2 over block b's grid-stride share of [0, n): the same slots hold the same sums every
// step whichever stream or order the slices were reduced in (bitwise repeatable norm)
__global__ __launch_bounds__(256) void grad_sumsq_slots_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  const size_t head = ((4 - (reinterpret_cast<uintptr_t>(g) >> 2 & 3)) & 3);        // scalars in front of the 16-byte boundary
  const size_t h = head < n ? head : n;
  const float* g4 = g + h;
  const size_t n4 = (n - h) / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g4)[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x < (int)h) { const float v = g[threadIdx.x]; acc += v * v; }
    const size_t tail = n - h - n4 * 4;
    if (threadIdx.x >= 64 && threadIdx.x - 64 < (int)tail) { const float v = g4[n4 * 4 + threadIdx.x - 64]; acc += v * v; }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

struct PrepDev {
  float lr0, lr_gamma; int lr_interval;
  float beta1, beta2, grad_clip, grad_scale;
  uint32_t* step_ptr; const float* partial; int nslots;
  float* consts; float* metrics_out;
};

// consts[0..3] = gradient multiplier (grad_scale * clip factor), lr, 1/(1-beta1^t), 1/(1-beta2^t); then *step_ptr += 1.
// One workgroup sums the slots in a fixed order, so the norm is the same whoever launched it.
__global__ __launch_bounds__(256) void opt_prepare_kernel(PrepDev a) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < a.nslots; i += 256) acc += a.partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]) * a.grad_scale;
  // clip_grads: where(norm < max_norm, g, g * max_norm / norm)
  const float gmul = a.grad_scale * (norm < a.grad_clip ? 1.0f : a.grad_clip / norm);
  const uint32_t step = *a.step_ptr;
  // stepped schedule: lr0 * gamma ** max(0, ceil(step/interval) - 1)
  const int idx = (int)((step + (uint32_t)a.lr_interval - 1u) / (uint32_t)a.lr_interval) - 1;
  const float lr = a.lr0 * powf(a.lr_gamma, (float)(idx > 0 ? idx : 0));
  const float tt = (float)(step + 1u);
  const float bc1 = 1.0f - powf(a.beta1, tt), bc2 = 1.0f - powf(a.beta2, tt);
  a.consts[0] = gmul; a.consts[1] = lr; a.consts[2] = 1.0f / bc1; a.consts[3] = 1.0f / bc2;
  if (a.metrics_out) {
    a.metrics_out[0] = norm;
    a.metrics_out[1] = norm < a.grad_clip ? norm : a.grad_clip;   // l2_norm of the clipped grads
    a.metrics_out[2] = lr;
    a.metrics_out[3] = (float)step;
  }
  *a.step_ptr = step + 1u;
}

struct AdamTileDev {
  float* p; const float* g; float* m; float* v; float* ema;
  bf16_t* wpack;
  const float* consts;
  float beta1, beta2, eps, mu;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float gmul, float lr, float inv_bc1, float inv_bc2,
                                          float beta1, float ob1, float beta2, float ob2, float eps) {
  const float gk = g * gmul;
  m = beta1 * m + ob1 * gk;
  v = beta2 * v + ob2 * gk * gk;
  p -= lr * (m * inv_bc1) / (sqrtf(v * inv_bc2) + eps);
}

__device__ __forceinline__ void adam_recast_block(const AdamTileDev& a, const OptTable& t, uint32_t b, float gmul, float lr, float inv_bc1,
                                                  float inv_bc2) {
  const float ob1 = 1.0f - a.beta1, ob2 = 1.0f - a.beta2, omu = 1.0f - a.mu;
  if (b >= t.flat_blk0) {
    // biases / LayerNorm parameters: 1024 elements per workgroup, any alignment
    int si = 0;
#pragma unroll 1
    for (int i = 1; i < t.n_flat; ++i)
      if (b >= t.f[i].blk_start) si = i;
    const OptFlat f = t.f[si];
    const uint32_t base = (b - f.blk_start) * 1024u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t e = base + j * 256u + threadIdx.x;
      if (e < f.len) {
        const size_t i = (size_t)f.off + e;
        float p = a.p[i], m = a.m[i], v = a.v[i];
        adam_elem(p, a.g[i], m, v, gmul, lr, inv_bc1, inv_bc2, a.beta1, ob1, a.beta2, ob2, a.eps);
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
        if (a.ema) a.ema[i] = a.ema[i] * a.mu + p * omu;
      }
    }
    return;
  }
  int wi = 0;
#pragma unroll 1
  for (int i = 1; i < t.n_dense; ++i)
    if (b >= t.d[i].blk_start) wi = i;
  const OptDense e = t.d[wi];
  const uint32_t local = b - e.blk_start;
  const uint32_t tiles_n = (e.N + 63u) / 64u;
  const int k0 = (int)(local / tiles_n) * 64, n0 = (int)(local % tiles_n) * 64;
  const int cq = threadIdx.x & 15, kq = threadIdx.x >> 4;                 // this thread: rows k0 + 4 kq .. +3, columns n0 + 4 cq .. +3
  const int kb = k0 + 4 * kq, nb = n0 + 4 * cq;
  const size_t wo = (size_t)e.w_off;
  float pv[4][4], gv[4][4], mv[4][4], vv[4][4], ev[4][4];
  const bool fast = k0 + 64 <= (int)e.K && n0 + 64 <= (int)e.N && ((e.w_off | e.N | e.W_off | e.ldw | e.Wt_off | e.ldwt) & 3u) == 0;
  if (fast) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t o = wo + (size_t)(kb + i) * e.N + nb;
      *reinterpret_cast<float4*>(pv[i]) = *reinterpret_cast<const float4*>(a.p + o);
      *reinterpret_cast<float4*>(gv[i]) = *reinterpret_cast<const float4*>(a.g + o);
      *reinterpret_cast<float4*>(mv[i]) = *reinterpret_cast<const float4*>(a.m + o);
      *reinterpret_cast<float4*>(vv[i]) = *reinterpret_cast<const float4*>(a.v + o);
      if (a.ema) *reinterpret_cast<float4*>(ev[i]) = *reinterpret_cast<const float4*>(a.ema + o);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) adam_elem(pv[i][j], gv[i][j], mv[i][j], vv[i][j], gmul, lr, inv_bc1, inv_bc2, a.beta1, ob1, a.beta2, ob2, a.eps);
    bf16_t* W = a.wpack + e.W_off;
    bf16_t* Wt = a.wpack + e.Wt_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t o = wo + (size_t)(kb + i) * e.N + nb;
      *reinterpret_cast<float4*>(a.p + o) = *reinterpret_cast<float4*>(pv[i]);
      *reinterpret_cast<float4*>(a.m + o) = *reinterpret_cast<float4*>(mv[i]);
      *reinterpret_cast<float4*>(a.v + o) = *reinterpret_cast<float4*>(vv[i]);
      if (a.ema) {
        float4 q;
        q.x = ev[i][0] * a.mu + pv[i][0] * omu; q.y = ev[i][1] * a.mu + pv[i][1] * omu;
        q.z = ev[i][2] * a.mu + pv[i][2] * omu; q.w = ev[i][3] * a.mu + pv[i][3] * omu;
        *reinterpret_cast<float4*>(a.ema + o) = q;
      }
      bf16x4_t w;
      w[0] = f2bf(pv[i][0]); w[1] = f2bf(pv[i][1]); w[2] = f2bf(pv[i][2]); w[3] = f2bf(pv[i][3]);
      *reinterpret_cast<bf16x4_t*>(W + (size_t)(kb + i) * e.ldw + nb) = w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // the transposed layout: row n, four consecutive k
      bf16x4_t w;
      w[0] = f2bf(pv[0][j]); w[1] = f2bf(pv[1][j]); w[2] = f2bf(pv[2][j]); w[3] = f2bf(pv[3][j]);
      *reinterpret_cast<bf16x4_t*>(Wt + (size_t)(nb + j) * e.ldwt + kb) = w;
    }
    return;
  }
  // edge tiles and kernels whose rows are not 16-byte aligned (C = 42, 146): element by element
  bf16_t* W = a.wpack + e.W_off;
  bf16_t* Wt = a.wpack + e.Wt_off;
#pragma unroll 1
  for (int i = 0; i < 4; ++i)
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
      const int k = kb + i, n = nb + j;
      if (k >= (int)e.K || n >= (int)e.N) continue;
      const size_t o = wo + (size_t)k * e.N + n;
      float p = a.p[o], m = a.m[o], v = a.v[o];
      adam_elem(p, a.g[o], m, v, gmul, lr, inv_bc1, inv_bc2, a.beta1, ob1, a.beta2, ob2, a.eps);
      a.p[o] = p; a.m[o] = m; a.v[o] = v;
      if (a.ema) a.ema[o] = a.ema[o] * a.mu + p * omu;
      W[(size_t)k * e.ldw + n] = f2bf(p);
      Wt[(size_t)n * e.ldwt + k] = f2bf(p);
    }
}

// one work item per workgroup (gridDim.x == t.total_blocks), or a grid-stride walk of fewer workgroups: the deferred
// output-stage update runs THROTTLED on the side stream (a fraction of the HBM rate for a longer time) so that the
// latency-bound encoder kernels of the next forward pass keep their memory latency
__global__ __launch_bounds__(256) void adam_recast_kernel(AdamTileDev a, OptTable t) {
  const float gmul = a.consts[0], lr = a.consts[1], inv_bc1 = a.consts[2], inv_bc2 = a.consts[3];
#pragma unroll 1
  for (uint32_t b = blockIdx.x; b < t.total_blocks; b += gridDim.x) adam_recast_block(a, t, b, gmul, lr, inv_bc1, inv_bc2);
}

}  // namespace

int launch_grad_sumsq_slots(const float* g, size_t n, float* partial, int nslots, hipStream_t st) {
  SMD_ARG_CHECK(g && partial && n > 0 && nslots > 0, "grad_sumsq_slots: bad arguments");
  hipLaunchKernelGGL(grad_sumsq_slots_kernel, dim3(nslots), dim3(256), 0, st, g, n, partial);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_opt_prepare(const AdamArgs& a, int nslots, float* consts, hipStream_t st) {
  SMD_ARG_CHECK(a.step_ptr && a.norm_partial && consts && nslots > 0, "opt_prepare: null pointer");
  SMD_ARG_CHECK(a.lr_interval > 0, "opt_prepare: lr_interval must be positive");
  PrepDev d;
  d.lr0 = a.lr0; d.lr_gamma = a.lr_gamma; d.lr_interval = a.lr_interval; d.beta1 = a.beta1; d.beta2 = a.beta2;
  d.grad_clip = a.grad_clip; d.grad_scale = a.grad_scale; d.step_ptr = a.step_ptr; d.partial = a.norm_partial; d.nslots = nslots;
  d.consts = consts; d.metrics_out = a.metrics_out;
  hipLaunchKernelGGL(opt_prepare_kernel, dim3(1), dim3(256), 0, st, d);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_adam_recast(const AdamArgs& a, const float* consts, bf16_t* wpack, const OptTable& t, hipStream_t st, int max_blocks) {
  SMD_ARG_CHECK(a.params && a.grads && a.m && a.v && consts && wpack, "adam_recast: null pointer");
  SMD_ARG_CHECK(t.n_dense >= 0 && t.n_dense <= SMD_OPT_DENSE_MAX && t.n_flat >= 0 && t.n_flat <= SMD_OPT_FLAT_MAX, "adam_recast: bad table");
  if (t.total_blocks == 0) return 0;
  AdamTileDev d;
  d.p = a.params; d.g = a.grads; d.m = a.m; d.v = a.v; d.ema = a.ema; d.wpack = wpack; d.consts = consts;
  d.beta1 = a.beta1; d.beta2 = a.beta2; d.eps = a.eps; d.mu = a.mu;
  const unsigned grid = max_blocks > 0 && (unsigned)max_blocks < t.total_blocks ? (unsigned)max_blocks : t.total_blocks;
  hipLaunchKernelGGL(adam_recast_kernel, dim3(grid), dim3(256), 0, st, d, t);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_recast_all(const float* params, bf16_t* wpack, const RecastTable& t, int total_tiles, hipStream_t st) {
  SMD_ARG_CHECK(params && wpack && t.n > 0 && t.n <= SMD_RECAST_MAX && total_tiles > 0, "recast_all: bad arguments");
  hipLaunchKernelGGL(recast_all_kernel, dim3(total_tiles), dim3(256), 0, st, params, wpack, t);
  SMD_LAUNCH_CHECK();
  return 0;
}

static int norm_blocks_for(size_t n) {
  size_t b = (n / 4 + 255) / 256;
  if (b < 1) b = 1;
  if (b > MAX_NORM_BLOCKS) b = MAX_NORM_BLOCKS;
  return (int)b;
}

int launch_grad_sumsq(const AdamArgs& a, hipStream_t st) {
  SMD_ARG_CHECK(a.grads && a.norm_partial && a.n > 0, "grad_sumsq: bad arguments");
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(norm_blocks_for(a.n)), dim3(256), 0, st, a.grads, a.n, a.norm_partial);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_adam_clip_ema(const AdamArgs& a, hipStream_t st) {
  SMD_ARG_CHECK(a.params && a.grads && a.m && a.v && a.step_ptr && a.norm_partial && a.n > 0,
                "adam_clip_ema: null pointer");
  SMD_ARG_CHECK(a.lr_interval > 0, "adam_clip_ema: lr_interval must be positive");
  AdamDev d;
  d.p = a.params; d.g = a.grads; d.m = a.m; d.v = a.v; d.ema = a.ema; d.n = a.n;
  d.lr0 = a.lr0; d.lr_gamma = a.lr_gamma; d.lr_interval = a.lr_interval;
  d.beta1 = a.beta1; d.beta2 = a.beta2; d.eps = a.eps; d.grad_clip = a.grad_clip; d.mu = a.mu;
  d.grad_scale = a.grad_scale;
  d.step_ptr = a.step_ptr; d.norm_partial = a.norm_partial; d.norm_blocks = norm_blocks_for(a.n);
  d.metrics_out = a.metrics_out;
  size_t blocks = (a.n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_clip_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d);
  SMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(1), 0, st, a.step_ptr);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_recast_weight(const float* w, int K_in, int N_out, bf16_t* W, int ldw, bf16_t* Wt, int ldwt,
                         hipStream_t st) {
  SMD_ARG_CHECK(w && (W || Wt) && K_in > 0 && N_out > 0, "recast_weight: bad arguments");
  SMD_ARG_CHECK((!W || ldw >= N_out) && (!Wt || ldwt >= K_in), "recast_weight: leading dimension too small");
  hipLaunchKernelGGL(recast_weight_kernel, dim3((N_out + 63) / 64, (K_in + 63) / 64), dim3(256), 0, st, w, K_in,
                     N_out, W, ldw, Wt, ldwt);
  SMD_LAUNCH_CHECK();
  return 0;
}
