// Multi-head self-attention core for S = 32 (softmax(q k^T / sqrt(d)) v), forward and backward.
//
// Reference: flax.nn.SelfAttention as called at models/ncsn.py:161 (no mask, no dropout,
// q scaled by 1/sqrt(d) before the logits, softmax over keys).  The QKV / output projections
// are MFMA GEMMs (gemm_nt.hip); this file is the per-(sample, head) 32x32 core, which is
// <0.5 % of the network's FLOPs and latency-bound: one workgroup per sample stages that
// sample's [32][3E] qkv rows in LDS once, each wave owns heads w, w+4, ...; lane = (query i,
// key half), softmax statistics by one cross-half xor-shuffle.
#include "smd_kernels.h"

namespace {

constexpr int S = 32;
constexpr int PAD = 8;   // bf16 elements of row padding: 16-B row slots rotate across LDS banks

template <int DH>
__device__ __forceinline__ void load_head_vec(const bf16_t* row, float (&v)[DH]) {
#pragma unroll
  for (int c = 0; c < DH; c += 8) {
    const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(row + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[c + e] = bf2f(t[e]);
  }
}
template <int DH>
__device__ __forceinline__ void store_head_vec(bf16_t* row, const float (&v)[DH]) {
#pragma unroll
  for (int c = 0; c < DH; c += 8) {
    bf16x8_t t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = f2bf(v[c + e]);
    *reinterpret_cast<bf16x8_t*>(row + c) = t;
  }
}

__device__ __forceinline__ void stage_rows(const bf16_t* g, int ld_g, int ncols, bf16_t* lds, int ld_l) {
  // 32 rows x ncols bf16, 16-byte pieces
  const int pieces_per_row = ncols / 8;
  for (int p = threadIdx.x; p < S * pieces_per_row; p += blockDim.x) {
    const int r = p / pieces_per_row, c = (p - r * pieces_per_row) * 8;
    *reinterpret_cast<bf16x8_t*>(lds + r * ld_l + c) = *reinterpret_cast<const bf16x8_t*>(g + (size_t)r * ld_g + c);
  }
}

// probabilities of query i against the 16 keys of this lane's half; returns them normalised
template <int DH>
__device__ __forceinline__ void softmax_row(const bf16_t* sq, const bf16_t* sk, int ld, int i, int half,
                                            float (&q)[DH], float (&p)[16]) {
  load_head_vec<DH>(sq + i * ld, q);
  const float inv_sqrt_d = rsqrtf((float)DH);
#pragma unroll
  for (int c = 0; c < DH; ++c) q[c] *= inv_sqrt_d;
  float mx = -INFINITY;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    float k[DH];
    load_head_vec<DH>(sk + (half * 16 + jj) * ld, k);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) s += q[c] * k[c];
    p[jj] = s;
    mx = fmaxf(mx, s);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) { p[jj] = __expf(p[jj] - mx); sum += p[jj]; }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) p[jj] *= inv;
}

template <int DH>
__global__ __launch_bounds__(256) void attention_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            int E, int H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* sm = reinterpret_cast<bf16_t*>(smem_raw);
  const int ld = 3 * E + PAD;
  const int b = blockIdx.x;
  stage_rows(qkv + (size_t)b * S * 3 * E, 3 * E, 3 * E, sm, ld);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 31, half = lane >> 5;
  for (int h = w; h < H; h += 4) {
    float q[DH], p[16];
    softmax_row<DH>(sm + h * DH, sm + E + h * DH, ld, i, half, q, p);
    float o[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) o[c] = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      float v[DH];
      load_head_vec<DH>(sm + 2 * E + h * DH + (half * 16 + jj) * ld, v);
#pragma unroll
      for (int c = 0; c < DH; ++c) o[c] += p[jj] * v[c];
    }
#pragma unroll
    for (int c = 0; c < DH; ++c) o[c] += __shfl_xor(o[c], 32, 64);
    if (half == 0) store_head_vec<DH>(out + ((size_t)b * S + i) * E + h * DH, o);
  }
}

template <int DH>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const bf16_t* __restrict__ qkv,
                                                            const bf16_t* __restrict__ dout,
                                                            bf16_t* __restrict__ dqkv, int E, int H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ld = 3 * E + PAD, ldo = E + PAD;
  bf16_t* sm = reinterpret_cast<bf16_t*>(smem_raw);          // [32][ld]   qkv
  bf16_t* sdo = sm + S * ld;                                  // [32][ldo]  dout
  float* sP = reinterpret_cast<float*>(sdo + S * ldo);        // [4 waves][2][32][33]
  const int b = blockIdx.x;
  stage_rows(qkv + (size_t)b * S * 3 * E, 3 * E, 3 * E, sm, ld);
  stage_rows(dout + (size_t)b * S * E, E, E, sdo, ldo);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 31, half = lane >> 5;
  float* P = sP + w * 2 * S * 33;
  float* dS = P + S * 33;
  const float inv_sqrt_d = rsqrtf((float)DH);
  for (int h = w; h < H; h += 4) {
    const bf16_t* sq = sm + h * DH;
    const bf16_t* sk = sm + E + h * DH;
    const bf16_t* sv = sm + 2 * E + h * DH;
    const bf16_t* sd = sdo + h * DH;
    // ---- phase 1: lane = (query i, key half)
    float q[DH], p[16];
    softmax_row<DH>(sq, sk, ld, i, half, q, p);          // q is already scaled by 1/sqrt(d)
    float go[DH];
    load_head_vec<DH>(sd + i * ldo, go);
    float dp[16], dot = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      float v[DH];
      load_head_vec<DH>(sv + (half * 16 + jj) * ld, v);
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < DH; ++c) s += go[c] * v[c];
      dp[jj] = s;
      dot += p[jj] * s;
    }
    dot += __shfl_xor(dot, 32, 64);
    float dq[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) dq[c] = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const float ds = p[jj] * (dp[jj] - dot);
      P[i * 33 + half * 16 + jj] = p[jj];
      dS[i * 33 + half * 16 + jj] = ds;
      float k[DH];
      load_head_vec<DH>(sk + (half * 16 + jj) * ld, k);
#pragma unroll
      for (int c = 0; c < DH; ++c) dq[c] += ds * k[c];
    }
#pragma unroll
    for (int c = 0; c < DH; ++c) dq[c] = (dq[c] + __shfl_xor(dq[c], 32, 64)) * inv_sqrt_d;
    if (half == 0) store_head_vec<DH>(dqkv + ((size_t)b * S + i) * 3 * E + h * DH, dq);
    __builtin_amdgcn_wave_barrier();   // DS ops of one wave execute in order: no s_barrier needed
    // ---- phase 2: lane = (key j, query half)
    const int j = i;
    float dk[DH], dv[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) dk[c] = dv[c] = 0.f;
#pragma unroll 4
    for (int ii = 0; ii < 16; ++ii) {
      const int qi = half * 16 + ii;
      const float pij = P[qi * 33 + j], dsij = dS[qi * 33 + j];
      float qv[DH], gv[DH];
      load_head_vec<DH>(sq + qi * ld, qv);
      load_head_vec<DH>(sd + qi * ldo, gv);
#pragma unroll
      for (int c = 0; c < DH; ++c) { dk[c] += dsij * qv[c]; dv[c] += pij * gv[c]; }
    }
#pragma unroll
    for (int c = 0; c < DH; ++c) {
      dk[c] = (dk[c] + __shfl_xor(dk[c], 32, 64)) * inv_sqrt_d;
      dv[c] += __shfl_xor(dv[c], 32, 64);
    }
    if (half == 0) {
      store_head_vec<DH>(dqkv + ((size_t)b * S + j) * 3 * E + E + h * DH, dk);
      store_head_vec<DH>(dqkv + ((size_t)b * S + j) * 3 * E + 2 * E + h * DH, dv);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

size_t fwd_lds_bytes(int E) { return (size_t)S * (3 * E + PAD) * 2; }
size_t bwd_lds_bytes(int E) {
  return (size_t)S * (3 * E + PAD) * 2 + (size_t)S * (E + PAD) * 2 + (size_t)4 * 2 * S * 33 * 4;
}

}  // namespace

static int check_attn(int B, int Sarg, int E, int H) {
  SMD_ARG_CHECK(Sarg == S, "attention: sequence length %d unsupported (kernel is specialised for S=32)", Sarg);
  SMD_ARG_CHECK(B > 0 && H > 0 && E % H == 0 && E % 8 == 0, "attention: bad geometry B=%d E=%d H=%d", B, E, H);
  const int d = E / H;
  SMD_ARG_CHECK(d == 8 || d == 16 || d == 32, "attention: head_dim %d unsupported (8, 16, 32)", d);
  return 0;
}

int launch_attention_fwd(const bf16_t* qkv, bf16_t* out, int B, int Sarg, int E, int H, hipStream_t st) {
  int rc = check_attn(B, Sarg, E, H);
  if (rc) return rc;
  SMD_ARG_CHECK(qkv && out, "attention_fwd: null pointer");
  const size_t lds = fwd_lds_bytes(E);
  switch (E / H) {
    case 8: hipLaunchKernelGGL(attention_fwd_kernel<8>, dim3(B), dim3(256), lds, st, qkv, out, E, H); break;
    case 16: hipLaunchKernelGGL(attention_fwd_kernel<16>, dim3(B), dim3(256), lds, st, qkv, out, E, H); break;
    default: hipLaunchKernelGGL(attention_fwd_kernel<32>, dim3(B), dim3(256), lds, st, qkv, out, E, H); break;
  }
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_attention_bwd(const bf16_t* qkv, const bf16_t* dout, bf16_t* dqkv, int B, int Sarg, int E, int H,
                         hipStream_t st) {
  int rc = check_attn(B, Sarg, E, H);
  if (rc) return rc;
  SMD_ARG_CHECK(qkv && dout && dqkv, "attention_bwd: null pointer");
  const size_t lds = bwd_lds_bytes(E);
  static bool attr_set = false;   // > 64 KiB of dynamic LDS needs the opt-in attribute (once, host-side)
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  switch (E / H) {
    case 8: hipLaunchKernelGGL(attention_bwd_kernel<8>, dim3(B), dim3(256), lds, st, qkv, dout, dqkv, E, H); break;
    case 16: hipLaunchKernelGGL(attention_bwd_kernel<16>, dim3(B), dim3(256), lds, st, qkv, dout, dqkv, E, H); break;
    default: hipLaunchKernelGGL(attention_bwd_kernel<32>, dim3(B), dim3(256), lds, st, qkv, dout, dqkv, E, H); break;
  }
  SMD_LAUNCH_CHECK();
  return 0;
}
