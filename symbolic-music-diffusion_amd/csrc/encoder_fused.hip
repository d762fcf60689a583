// Fused transformer-encoder half-layers for the 128-wide residual stream (reference models/ncsn.py:159-168):
//
//   mlp_block_fwd:   h_out = h_mid + Dense_{2048->128}( gelu( Dense_{128->2048}( LN(h_mid) ) ) )     (:163-168)
//
// Why fused: as separate launches the encoder is 16 % of the flops but ~45 % of a denoising step -- every
// kernel is bound by launch boundaries and by re-streaming 2048-wide activations through HBM.  Here one
// workgroup owns one sample (S = 32 token rows = exactly one 32-row MFMA tile) and walks the hidden dimension
// in chunks of 128 units; the 2048-wide hidden activation never leaves the registers:
//
//   * LN(h_mid) -> a2 (bf16) goes to LDS once; its MFMA B-fragments (8 k-steps) stay in registers.
//   * per chunk c, wave w:  z^T[32 hidden x 32 tokens] = W1[c*128 + w*32 .. +32][:] * a2^T   (8 MFMA 32x32x16),
//     + b1, GELU in the accumulator registers, which then ARE the A-fragments of the second GEMM: the 32x32 C
//     layout (lane = token column, 16 hidden rows per lane) matches the A layout (lane = token row, 8 k per
//     k-step) up to a permutation of the contraction index, and the W2 B-fragments are gathered with the same
//     permutation (two 8-byte LDS reads).  h_part[32 tokens x 128] += u * W2[chunk]  (8 MFMA).
//   * the four waves contract disjoint hidden slices; their partial h_out tiles are summed through LDS,
//     + b2 + residual, one fp32 store of the [32 x 128] result.
//   * weights stream HBM/L2 -> LDS with buffer_load ... lds (64 KiB per chunk: W1 slice + W2 slice), double
//     buffered, one chunk in flight behind the compute; 256-B LDS rows, 16-B chunk c of row r stored at
//     c ^ (r & 15) (source-side swizzle) so that the 16-lane ds_read_b128 groups are conflict-free.
//   * training: a2, z1 (pre-GELU) and u are written out for the backward pass straight from the C layout.
#include "smd_kernels.h"

namespace {

constexpr int S_TOK = 32, E_DIM = 128, CH = 128;        // tokens per sample, stream width, hidden units per chunk
constexpr int W_SLICE = CH * E_DIM * 2;                 // 32 KiB: 128 rows x 256 B
constexpr int BUF_BYTES = 2 * W_SLICE;                  // W1 slice + W2 slice
constexpr int OFF_A2 = 2 * BUF_BYTES;                   // a2 tile [32][128] bf16 = 8 KiB
constexpr int MAX_HIDDEN = 8192;
constexpr int SMEM_BYTES = OFF_A2 + S_TOK * E_DIM * 2;  // 136 KiB
constexpr float LN_EPS = 1e-6f;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(3))) bf16x8_t* lds_b128_ptr;
typedef const __attribute__((address_space(3))) bf16x4_t* lds_b64_ptr;
typedef const __attribute__((address_space(3))) unsigned char* lds_byte_ptr;
typedef const __attribute__((address_space(3))) float4* lds_f4_ptr;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, unsigned char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_wave_base, 16, voff, soff, 0, 0);
}

struct MlpArgs {
  const float* h_in;        // [R][128] fp32 residual stream (h_mid)
  float* h_out;             // [R][128] (may alias h_in)
  const float* gamma; const float* beta;
  const bf16_t* W1t;        // [M][128]  (nn.Dense kernel transposed: hidden-major, contraction contiguous)
  const float* b1;          // [M]
  const bf16_t* W2t;        // [128][M]
  const float* b2;          // [128]
  int M;                    // hidden width (multiple of 128)
  bf16_t* save_a2;          // [R][128] or null
  bf16_t* save_z1;          // [R][M] or null   (pre-activation)
  bf16_t* save_u;           // [R][M] or null   (gelu output)
};

union Frag8 {
  bf16x8_t v;
  bf16x4_t h[2];
};

// Built with -mllvm -amdgpu-mfma-vgpr-form (build.py): otherwise hipcc parks the accumulators in AGPRs and copies
// all 64 of acc_o out and back every chunk.
template <int V>
__global__ __launch_bounds__(256) void mlp_block_fwd_kernel(MlpArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t row0 = (size_t)blockIdx.x * S_TOK;
  const int kh = lane >> 5, l31 = lane & 31;

  // ---- weight DMA: 8 rounds per slice, wave w + round j covers LDS rows j*16 + w*4 .. +4 (16 lanes per 256-B row)
  const int rl = w * 4 + (lane >> 4);                                    // row within a 16-row round == row & 15
  const uint32_t cs = (uint32_t)(((lane & 15) ^ rl) * 16);               // swizzled source chunk, bytes
  const uint32_t w1_v = (uint32_t)rl * 256u + cs;                        // W1t rows are 256 B, slice rows contiguous
  const uint32_t w2_v = (uint32_t)rl * (uint32_t)(a.M * 2) + cs;         // W2t rows are M*2 B
  const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W1t), 0, a.M * E_DIM * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W2t), 0, a.M * E_DIM * 2, 0x00020000);
  const uint32_t w2_round = (uint32_t)(16 * a.M * 2);
  unsigned char* lds_w = smem + w * 1024;
  auto stage_chunk = [&](int c, int buf) {
    unsigned char* d1 = lds_w + buf * BUF_BYTES;
#pragma unroll
    for (int j = 0; j < 8; ++j) glds16(w1_rsrc, w1_v, (uint32_t)(c * W_SLICE + j * 4096), d1 + j * 4096);
#pragma unroll
    for (int j = 0; j < 8; ++j) glds16(w2_rsrc, w2_v, (uint32_t)(c * 256) + j * w2_round, d1 + W_SLICE + j * 4096);
  };
  const int nchunks = a.M / CH;
  // fc1 bias of the lane's 16 hidden rows, fetched one chunk ahead and issued BEFORE that chunk's DMA so that the
  // counted vmcnt wait of the next iteration also covers it (hidden of element e: hb + (e&3) + 8*(e>>2))
  float4 bnext[4];
  auto fetch_bias = [&](int c) {
#pragma unroll
    for (int g = 0; g < 4; ++g) bnext[g] = *reinterpret_cast<const float4*>(a.b1 + c * CH + w * 32 + 4 * kh + 8 * g);
  };
  fetch_bias(0);
  stage_chunk(0, 0);

  // ---- LayerNorm of the sample's 32 rows (wave w: rows w*8 .. +8, two elements per lane) -> a2 tile in LDS
  {
    const float2 g2 = *reinterpret_cast<const float2*>(a.gamma + lane * 2);
    const float2 b2v = *reinterpret_cast<const float2*>(a.beta + lane * 2);
    float2 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = *reinterpret_cast<const float2*>(a.h_in + (row0 + w * 8 + i) * E_DIM + lane * 2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = w * 8 + i;
      float s = x[i].x + x[i].y, s2 = x[i].x * x[i].x + x[i].y * x[i].y;
      s = wave_sum(s);
      s2 = wave_sum(s2);
      const float mean = s * (1.0f / E_DIM);
      const float rstd = rsqrtf(s2 * (1.0f / E_DIM) - mean * mean + LN_EPS);
      bf16x2_t o;
      o[0] = f2bf((x[i].x - mean) * rstd * g2.x + b2v.x);
      o[1] = f2bf((x[i].y - mean) * rstd * g2.y + b2v.y);
      // element column = 2*lane -> 16-B chunk lane>>2, byte (lane&3)*4 ; swizzle chunk ^ (r & 15)
      *reinterpret_cast<bf16x2_t*>(smem + OFF_A2 + r * 256 + (((lane >> 2) ^ (r & 15)) << 4) + (lane & 3) * 4) = o;
      if (a.save_a2) *reinterpret_cast<bf16x2_t*>(a.save_a2 + (row0 + r) * E_DIM + lane * 2) = o;
    }
  }
  __syncthreads();

  // ---- B fragments of a2^T for the 8 k-steps (lane = token l31, k-chunk 2*ks + kh), kept for the whole kernel
  bf16x8_t a2f[8];
  {
    lds_byte_ptr base = (lds_byte_ptr)smem + OFF_A2 + l31 * 256;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) a2f[ks] = *reinterpret_cast<lds_b128_ptr>(base + (((ks * 2 + kh) ^ (l31 & 15)) << 4));
  }

  f32x16_t acc_o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc_o[t][e] = 0.0f;

  // W1 A-fragment: row = w*32 + l31 of the slice; W2 B-fragment: row n = nt*32 + l31, hidden bytes w*64 + 32*ks2 (+16), +8*kh
  const int w1_row_off = (w * 32 + l31) * 256;
  const int sw = l31 & 15;

  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    float4 bcur[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bcur[g] = bnext[g];
    if (c + 1 < nchunks) {
      fetch_bias(c + 1);
      if constexpr (!(V & 1)) stage_chunk(c + 1, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    lds_byte_ptr s1 = (lds_byte_ptr)smem + buf * BUF_BYTES;
    lds_byte_ptr s2 = s1 + W_SLICE;

    if constexpr ((V & 2) != 0) { __builtin_amdgcn_s_barrier(); continue; }
    // z^T tile (rows = hidden, cols = tokens)
    f32x16_t z, zb;                     // two accumulators: halves the dependent-MFMA chain (one wave per SIMD)
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = zb[e] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 8; ks += 2) {
      const bf16x8_t wf0 = *reinterpret_cast<lds_b128_ptr>(s1 + w1_row_off + (((ks * 2 + kh) ^ sw) << 4));
      const bf16x8_t wf1 = *reinterpret_cast<lds_b128_ptr>(s1 + w1_row_off + ((((ks + 1) * 2 + kh) ^ sw) << 4));
      z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0, a2f[ks], z, 0, 0, 0);
      zb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1, a2f[ks + 1], zb, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] += zb[e];
    // bias (per hidden row), save, GELU, save; hidden of element e: hb + (e&3) + 8*(e>>2), hb = c*128 + w*32 + 4*kh
    const int hb = c * CH + w * 32 + 4 * kh;
    Frag8 uf[2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bb = bcur[g];
      float v[4] = {z[4 * g + 0] + bb.x, z[4 * g + 1] + bb.y, z[4 * g + 2] + bb.z, z[4 * g + 3] + bb.w};
      if (a.save_z1) {
        bf16x4_t o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = f2bf(v[i]);
        *reinterpret_cast<bf16x4_t*>(a.save_z1 + (row0 + l31) * a.M + hb + 8 * g) = o;
      }
      bf16x4_t o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = f2bf(geluf_(v[i]));
      if (a.save_u) *reinterpret_cast<bf16x4_t*>(a.save_u + (row0 + l31) * a.M + hb + 8 * g) = o;
      uf[g >> 1].h[g & 1] = o;           // k-step g>>1, elements (g&1)*4 .. +4  <->  accumulator elements 4g .. 4g+3
    }
    // h_part[token][n] += u[token][hidden slice] * W2[hidden slice][n]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = nt * 32 + l31;
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const int ch = w * 4 + 2 * ks2;                      // 16-B chunk of hidden bytes w*64 + 32*ks2
        Frag8 bf;
        bf.h[0] = *reinterpret_cast<lds_b64_ptr>(s2 + n * 256 + ((ch ^ sw) << 4) + 8 * kh);
        bf.h[1] = *reinterpret_cast<lds_b64_ptr>(s2 + n * 256 + (((ch + 1) ^ sw) << 4) + 8 * kh);
        acc_o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uf[ks2].v, bf.v, acc_o[nt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();      // every wave is done with `buf` before the next iteration refills it
  }

  // ---- sum the four waves' partial tiles through LDS (the weight buffers are free), + b2 + residual
  float* red = reinterpret_cast<float*>(smem);           // [4][32][128] fp32 = 64 KiB
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int tok = (e & 3) + 8 * (e >> 2) + 4 * kh;
      red[(w * S_TOK + tok) * E_DIM + nt * 32 + l31] = acc_o[nt][e];
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = (i * 256 + tid) * 4;                 // 4 consecutive columns of one token row
    const int tok = idx >> 7, col = idx & 127;
    const float4 p0 = *reinterpret_cast<const float4*>(red + (0 * S_TOK + tok) * E_DIM + col);
    const float4 p1 = *reinterpret_cast<const float4*>(red + (1 * S_TOK + tok) * E_DIM + col);
    const float4 p2 = *reinterpret_cast<const float4*>(red + (2 * S_TOK + tok) * E_DIM + col);
    const float4 p3 = *reinterpret_cast<const float4*>(red + (3 * S_TOK + tok) * E_DIM + col);
    const float4 bb = *reinterpret_cast<const float4*>(a.b2 + col);
    const float4 hr = *reinterpret_cast<const float4*>(a.h_in + (row0 + tok) * E_DIM + col);
    float4 o;
    o.x = ((p0.x + p1.x) + (p2.x + p3.x)) + bb.x + hr.x;
    o.y = ((p0.y + p1.y) + (p2.y + p3.y)) + bb.y + hr.y;
    o.z = ((p0.z + p1.z) + (p2.z + p3.z)) + bb.z + hr.z;
    o.w = ((p0.w + p1.w) + (p2.w + p3.w)) + bb.w + hr.w;
    *reinterpret_cast<float4*>(a.h_out + (row0 + tok) * E_DIM + col) = o;
  }
}



// ---- 8-wave variant of the MLP half-layer on 16x16x32 MFMAs: two waves per SIMD hide each other's MFMA / LDS /
// transcendental latencies (the 4-wave kernel above runs one wave per SIMD and is bound by exactly those).
// Wave w = (hg = w & 3: 32 hidden units of the chunk, th = w >> 2: token half).  z^T tiles [16 hidden x 16 tokens]
// (C layout: lane = token column, 4 hidden rows) of the two m-tiles form the 8-element A fragment of the second
// GEMM (k permutation: element s <-> hidden 4g + s, 16 + 4g + s - 4).  Training saves (z1, u) are staged through
// LDS per chunk so that every thread writes 16 contiguous bytes of a 256-B row segment.
constexpr int M8_STAGE = OFF_A2 + S_TOK * E_DIM * 2;           // z1 | u staging tiles: 2 x [32][256 B]
constexpr int M8_SMEM = M8_STAGE + 2 * 8192;

template <bool SAVE>
__global__ __launch_bounds__(512) void mlp_block_fwd8_kernel(MlpArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[M8_SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hg = w & 3, th = w >> 2;
  const size_t row0 = (size_t)blockIdx.x * S_TOK;
  const int g = lane >> 4, j = lane & 15;
  lds_byte_ptr L = (lds_byte_ptr)smem;

  // ---- weight DMA: 4 rounds per slice, round r covers LDS rows r*32 + w*4 + (lane>>4)
  const int rl = w * 4 + (lane >> 4);                                    // 0..31 ; row & 15 == rl & 15
  const uint32_t cs = (uint32_t)(((lane & 15) ^ (rl & 15)) * 16);
  const uint32_t w1_v = (uint32_t)rl * 256u + cs;
  const uint32_t w2_v = (uint32_t)rl * (uint32_t)(a.M * 2) + cs;
  const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W1t), 0, a.M * E_DIM * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W2t), 0, a.M * E_DIM * 2, 0x00020000);
  const uint32_t w2_round = (uint32_t)(32 * a.M * 2);
  unsigned char* lds_w = smem + w * 1024;
  auto stage_chunk = [&](int c, int buf) {
    unsigned char* d1 = lds_w + buf * BUF_BYTES;
#pragma unroll
    for (int r = 0; r < 4; ++r) glds16(w1_rsrc, w1_v, (uint32_t)(c * W_SLICE + r * 8192), d1 + r * 8192);
#pragma unroll
    for (int r = 0; r < 4; ++r) glds16(w2_rsrc, w2_v, (uint32_t)(c * 256) + r * w2_round, d1 + W_SLICE + r * 8192);
  };
  const int nchunks = a.M / CH;
  float4 bnext[2];
  auto fetch_bias = [&](int c) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) bnext[mt] = *reinterpret_cast<const float4*>(a.b1 + c * CH + hg * 32 + mt * 16 + 4 * g);
  };
  fetch_bias(0);
  stage_chunk(0, 0);

  // ---- LayerNorm: wave w normalises rows w*4 .. +4 -> a2 tile
  {
    const float2 g2 = *reinterpret_cast<const float2*>(a.gamma + lane * 2);
    const float2 b2v = *reinterpret_cast<const float2*>(a.beta + lane * 2);
    float2 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const float2*>(a.h_in + (row0 + w * 4 + i) * E_DIM + lane * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = w * 4 + i;
      float s = x[i].x + x[i].y, s2 = x[i].x * x[i].x + x[i].y * x[i].y;
      s = wave_sum(s);
      s2 = wave_sum(s2);
      const float mean = s * (1.0f / E_DIM);
      const float rstd = rsqrtf(s2 * (1.0f / E_DIM) - mean * mean + LN_EPS);
      bf16x2_t o;
      o[0] = f2bf((x[i].x - mean) * rstd * g2.x + b2v.x);
      o[1] = f2bf((x[i].y - mean) * rstd * g2.y + b2v.y);
      *reinterpret_cast<bf16x2_t*>(smem + OFF_A2 + r * 256 + (((lane >> 2) ^ (r & 15)) << 4) + (lane & 3) * 4) = o;
      if (a.save_a2) *reinterpret_cast<bf16x2_t*>(a.save_a2 + (row0 + r) * E_DIM + lane * 2) = o;
    }
  }
  __syncthreads();

  // B fragments of a2^T (lane = token th*16 + j, k-chunk 4*ks + g), 4 k-steps of 32
  bf16x8_t a2f[4];
  {
    const int tok = th * 16 + j;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a2f[ks] = *reinterpret_cast<lds_b128_ptr>(L + OFF_A2 + tok * 256 + (((ks * 4 + g) ^ (tok & 15)) << 4));
  }
  f32x4_t acc_o[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc_o[t][e] = 0.0f;

  const int stok = tid >> 4, spc = tid & 15;          // staged-save mapping: thread -> (token, 16-B piece of a 256-B row)
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    float4 bcur[2] = {bnext[0], bnext[1]};
    if (c + 1 < nchunks) {
      fetch_bias(c + 1);
      stage_chunk(c + 1, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // the 2 bias loads + 8 DMAs just issued may stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (SAVE && c > 0) {
      // z1 / u of the previous chunk: LDS staging -> registers -> 16-byte global stores that drain behind this
      // chunk's compute (the next vmcnt wait is a whole iteration away)
      const bf16x8_t vz = *reinterpret_cast<lds_b128_ptr>(L + M8_STAGE + stok * 256 + spc * 16);
      const bf16x8_t vu = *reinterpret_cast<lds_b128_ptr>(L + M8_STAGE + 8192 + stok * 256 + spc * 16);
      __builtin_amdgcn_s_barrier();      // everybody has read the staging tiles before this chunk rewrites them
      *reinterpret_cast<bf16x8_t*>(a.save_z1 + (row0 + stok) * a.M + (c - 1) * CH + spc * 8) = vz;
      *reinterpret_cast<bf16x8_t*>(a.save_u + (row0 + stok) * a.M + (c - 1) * CH + spc * 8) = vu;
    }
    lds_byte_ptr s1 = L + buf * BUF_BYTES;
    lds_byte_ptr s2 = s1 + W_SLICE;

    // z^T tiles: rows = hidden hg*32 + mt*16 + (0..15), cols = tokens th*16 + (0..15)
    f32x4_t z[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) z[mt][e] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const bf16x8_t wf = *reinterpret_cast<lds_b128_ptr>(s1 + (hg * 32 + mt * 16 + j) * 256 + (((ks * 4 + g) ^ j) << 4));
        z[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, a2f[ks], z[mt], 0, 0, 0);
      }
    // bias, GELU -> A fragment of the second GEMM; hidden of z[mt][e]: hg*32 + mt*16 + 4g + e
    Frag8 uf;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float bb[4] = {bcur[mt].x, bcur[mt].y, bcur[mt].z, bcur[mt].w};
      bf16x4_t zz, uu;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = z[mt][e] + bb[e];
        zz[e] = f2bf(v);
        uu[e] = f2bf(geluf_(v));
      }
      uf.h[mt] = uu;
      if (SAVE) {
        const int tok = th * 16 + j, hcol = hg * 32 + mt * 16 + 4 * g;          // 4 consecutive hidden of this chunk
        const int off = tok * 256 + hcol * 2;
        *reinterpret_cast<bf16x4_t*>(smem + M8_STAGE + off) = zz;
        *reinterpret_cast<bf16x4_t*>(smem + M8_STAGE + 8192 + off) = uu;
      }
    }
    // h_part[token][n] += u[token][this wave's 32 hidden] * W2[hidden][n]
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const int n = nt * 16 + j;
      const int ch = hg * 4 + (g >> 1);
      Frag8 bf;
      bf.h[0] = *reinterpret_cast<lds_b64_ptr>(s2 + n * 256 + ((ch ^ j) << 4) + (g & 1) * 8);
      bf.h[1] = *reinterpret_cast<lds_b64_ptr>(s2 + n * 256 + (((ch + 2) ^ j) << 4) + (g & 1) * 8);
      acc_o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uf.v, bf.v, acc_o[nt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();      // `buf` free for the next refill; the staging tiles are complete
  }
  if (SAVE) {                          // the last chunk's staging tiles
    const bf16x8_t vz = *reinterpret_cast<lds_b128_ptr>(L + M8_STAGE + stok * 256 + spc * 16);
    const bf16x8_t vu = *reinterpret_cast<lds_b128_ptr>(L + M8_STAGE + 8192 + stok * 256 + spc * 16);
    *reinterpret_cast<bf16x8_t*>(a.save_z1 + (row0 + stok) * a.M + (nchunks - 1) * CH + spc * 8) = vz;
    *reinterpret_cast<bf16x8_t*>(a.save_u + (row0 + stok) * a.M + (nchunks - 1) * CH + spc * 8) = vu;
  }

  // ---- sum the four hidden-group partials of each token half through LDS, + b2 + residual
  float* red = reinterpret_cast<float*>(smem);           // [4 hg][32 tokens][128] fp32 = 64 KiB
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(hg * S_TOK + th * 16 + 4 * g + e) * E_DIM + nt * 16 + j] = acc_o[nt][e];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = (i * 512 + tid) * 4;
    const int tok = idx >> 7, col = idx & 127;
    const float4 p0 = *reinterpret_cast<const float4*>(red + (0 * S_TOK + tok) * E_DIM + col);
    const float4 p1 = *reinterpret_cast<const float4*>(red + (1 * S_TOK + tok) * E_DIM + col);
    const float4 p2 = *reinterpret_cast<const float4*>(red + (2 * S_TOK + tok) * E_DIM + col);
    const float4 p3 = *reinterpret_cast<const float4*>(red + (3 * S_TOK + tok) * E_DIM + col);
    const float4 bb = *reinterpret_cast<const float4*>(a.b2 + col);
    const float4 hr = *reinterpret_cast<const float4*>(a.h_in + (row0 + tok) * E_DIM + col);
    float4 o;
    o.x = ((p0.x + p1.x) + (p2.x + p3.x)) + bb.x + hr.x;
    o.y = ((p0.y + p1.y) + (p2.y + p3.y)) + bb.y + hr.y;
    o.z = ((p0.z + p1.z) + (p2.z + p3.z)) + bb.z + hr.z;
    o.w = ((p0.w + p1.w) + (p2.w + p3.w)) + bb.w + hr.w;
    *reinterpret_cast<float4*>(a.h_out + (row0 + tok) * E_DIM + col) = o;
  }
}

// =====================================================================================================
//   attn_block_fwd:  h_mid = h_in + out_proj( softmax(q k^T / sqrt(d)) v ),  q,k,v = Dense(LN(h_in))     (:159-162)
//
// One workgroup per sample, wave w owns the 32 features [32w, 32w+32) of q, k and v, i.e. whole heads:
//   * Wqkv (96 KiB) is DMA'd once; LN(h_in) -> a1 tile; its fragments serve as A and as B operand.
//   * q^T, k^T tiles (features x tokens) = W * a1^T, v tile (tokens x features) = a1 * Wv^T: the C layouts give
//     8-byte LDS writes of row-major q/k [token][feature] and of v^T [feature][token].
//   * per head: s^T = k_h q_h^T (one MFMA, lane = query column, 16 keys per lane), softmax over the lane's keys
//     + one xor-32 shuffle, p (bf16) straight from the accumulator registers as the B fragments of
//     o_h^T = v_h^T p^T (A fragment gathered from v^T with the matching key permutation).
//   * Wo is DMA'd into the (now free) Wq region behind the attention; o tile -> out_proj^T -> + bias + residual.
// =====================================================================================================
constexpr int AT_W = 0;                                   // Wqkv [384][256 B] (later Wo in the first 32 KiB)
constexpr int AT_A1 = 384 * 256;                          // a1, later o: [32][256 B]
constexpr int AT_Q = AT_A1 + 8192;                        // q (scaled) [32][256 B]
constexpr int AT_K = AT_Q + 8192;                         // k [32][256 B]
constexpr int VT_LD = 72;                                 // v^T row: 32 keys bf16 + 8 B pad (conflict-free b64 gathers)
constexpr int AT_VT = AT_K + 8192;                        // v^T [128][72 B]
constexpr int AT_SMEM = AT_VT + 128 * VT_LD;

struct AttnArgs {
  const float* h_in; float* h_out;
  const float* gamma; const float* beta;
  const bf16_t* Wqkv_t;     // [384][128]
  const float* b_qkv;       // [384]
  const bf16_t* Wo_t;       // [128][128]
  const float* b_o;         // [128]
  bf16_t* save_a1;          // [R][128] or null
  bf16_t* save_qkv;         // [R][384] or null (q unscaled, as the unfused path stores it)
  bf16_t* save_o;           // [R][128] or null
};

template <int DH>
__global__ __launch_bounds__(256) void attn_block_fwd_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[AT_SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t row0 = (size_t)blockIdx.x * S_TOK;
  const int kh = lane >> 5, l31 = lane & 31, sw = l31 & 15;
  lds_byte_ptr L = (lds_byte_ptr)smem;

  // ---- Wqkv DMA: 24 rounds of 16 rows (contiguous 4 KiB each), source-side swizzle as in the MLP kernel
  const int rl = w * 4 + (lane >> 4);
  const uint32_t wv = (uint32_t)rl * 256u + (uint32_t)(((lane & 15) ^ rl) * 16);
  const __amdgpu_buffer_rsrc_t wq_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.Wqkv_t), 0, 384 * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t wo_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.Wo_t), 0, 128 * 256, 0x00020000);
  unsigned char* lds_w = smem + w * 1024;
#pragma unroll
  for (int j = 0; j < 24; ++j) glds16(wq_rsrc, wv, (uint32_t)(j * 4096), lds_w + AT_W + j * 4096);

  // ---- LayerNorm -> a1 tile
  {
    const float2 g2 = *reinterpret_cast<const float2*>(a.gamma + lane * 2);
    const float2 b2v = *reinterpret_cast<const float2*>(a.beta + lane * 2);
    float2 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = *reinterpret_cast<const float2*>(a.h_in + (row0 + w * 8 + i) * E_DIM + lane * 2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = w * 8 + i;
      float s = x[i].x + x[i].y, s2 = x[i].x * x[i].x + x[i].y * x[i].y;
      s = wave_sum(s);
      s2 = wave_sum(s2);
      const float mean = s * (1.0f / E_DIM);
      const float rstd = rsqrtf(s2 * (1.0f / E_DIM) - mean * mean + LN_EPS);
      bf16x2_t o;
      o[0] = f2bf((x[i].x - mean) * rstd * g2.x + b2v.x);
      o[1] = f2bf((x[i].y - mean) * rstd * g2.y + b2v.y);
      *reinterpret_cast<bf16x2_t*>(smem + AT_A1 + r * 256 + (((lane >> 2) ^ (r & 15)) << 4) + (lane & 3) * 4) = o;
      if (a.save_a1) *reinterpret_cast<bf16x2_t*>(a.save_a1 + (row0 + r) * E_DIM + lane * 2) = o;
    }
  }
  // biases of this wave's features: q/k rows (per accumulator row), v column (per lane)
  float4 bq[4], bk[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bq[g] = *reinterpret_cast<const float4*>(a.b_qkv + w * 32 + 4 * kh + 8 * g);
    bk[g] = *reinterpret_cast<const float4*>(a.b_qkv + 128 + w * 32 + 4 * kh + 8 * g);
  }
  const float bv = a.b_qkv[256 + w * 32 + l31];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // a1 fragments (lane = token, k-chunk 2*ks + kh): B operand of the transposed GEMMs, A operand of the v GEMM
  bf16x8_t a1f[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) a1f[ks] = *reinterpret_cast<lds_b128_ptr>(L + AT_A1 + l31 * 256 + (((ks * 2 + kh) ^ sw) << 4));

  // ---- q^T, k^T (rows = features, cols = tokens) and v (rows = tokens, cols = features) of this wave's features
  f32x16_t cq, ck, cv;
#pragma unroll
  for (int e = 0; e < 16; ++e) cq[e] = ck[e] = cv[e] = 0.0f;
  {
    lds_byte_ptr wrow = L + AT_W + (w * 32 + l31) * 256;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int co = ((ks * 2 + kh) ^ sw) << 4;
      const bf16x8_t fq = *reinterpret_cast<lds_b128_ptr>(wrow + co);
      const bf16x8_t fk = *reinterpret_cast<lds_b128_ptr>(wrow + 128 * 256 + co);
      const bf16x8_t fv = *reinterpret_cast<lds_b128_ptr>(wrow + 256 * 256 + co);
      cq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq, a1f[ks], cq, 0, 0, 0);
      ck = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk, a1f[ks], ck, 0, 0, 0);
      cv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1f[ks], fv, cv, 0, 0, 0);
    }
  }
  __syncthreads();          // every wave is done reading Wqkv and a1: the regions can be reused
  // Wo -> first 32 KiB of the weight region, behind the attention
#pragma unroll
  for (int j = 0; j < 8; ++j) glds16(wo_rsrc, wv, (uint32_t)(j * 4096), lds_w + AT_W + j * 4096);

  // q (scaled), k -> row-major [token = l31][feature]; feature of element e: w*32 + 4*kh + (e&3) + 8*(e>>2)
  const float qscale = rsqrtf((float)DH);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int f = w * 32 + 4 * kh + 8 * g;                    // 4 consecutive features
    const float q4[4] = {cq[4 * g + 0] + bq[g].x, cq[4 * g + 1] + bq[g].y, cq[4 * g + 2] + bq[g].z, cq[4 * g + 3] + bq[g].w};
    const float k4[4] = {ck[4 * g + 0] + bk[g].x, ck[4 * g + 1] + bk[g].y, ck[4 * g + 2] + bk[g].z, ck[4 * g + 3] + bk[g].w};
    bf16x4_t qs, qu, kk;
#pragma unroll
    for (int i = 0; i < 4; ++i) { qu[i] = f2bf(q4[i]); qs[i] = f2bf(bf2f(qu[i]) * qscale); kk[i] = f2bf(k4[i]); }
    const int off = l31 * 256 + (((f >> 3) ^ sw) << 4) + (f & 7) * 2;
    *reinterpret_cast<bf16x4_t*>(smem + AT_Q + off) = qs;
    *reinterpret_cast<bf16x4_t*>(smem + AT_K + off) = kk;
    if (a.save_qkv) {
      *reinterpret_cast<bf16x4_t*>(a.save_qkv + (row0 + l31) * 384 + f) = qu;
      *reinterpret_cast<bf16x4_t*>(a.save_qkv + (row0 + l31) * 384 + 128 + f) = kk;
    }
  }
  // v^T [feature = w*32 + l31][token]; tokens of element e: 4*kh + (e&3) + 8*(e>>2)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bf16x4_t vv;
#pragma unroll
    for (int i = 0; i < 4; ++i) vv[i] = f2bf(cv[4 * g + i] + bv);
    *reinterpret_cast<bf16x4_t*>(smem + AT_VT + (w * 32 + l31) * VT_LD + (4 * kh + 8 * g) * 2) = vv;
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // wave-local LDS round trip (own features only)

  // ---- attention for the heads inside this wave's 32 features
  constexpr int HPW = 32 / DH;                                 // heads per wave
#pragma unroll
  for (int hh = 0; hh < HPW; ++hh) {
    const int f0 = w * 32 + hh * DH;                           // first feature of the head
    f32x16_t st;                                               // s^T: rows = keys, cols = queries
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < (DH + 15) / 16; ++ks) {
      bf16x8_t fk, fq;
      const int f = f0 + ks * 16 + kh * 8;                     // 8 consecutive features of this lane half
      if (DH >= 16 || kh == 0) {
        const int off = l31 * 256 + (((f >> 3) ^ sw) << 4);
        fk = *reinterpret_cast<lds_b128_ptr>(L + AT_K + off);
        fq = *reinterpret_cast<lds_b128_ptr>(L + AT_Q + off);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { fk[i] = (bf16_t)0.0f; fq[i] = (bf16_t)0.0f; }
      }
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk, fq, st, 0, 0, 0);
    }
    // softmax over keys: 16 in the lane + the other lane half
    float mx = st[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) mx = fmaxf(mx, st[e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] = __expf(st[e] - mx); sum += st[e]; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    Frag8 pf[2];                                               // B fragments of p^T: k-step ks2, element s <-> accumulator element 8*ks2 + s
#pragma unroll
    for (int e = 0; e < 16; ++e) pf[e >> 3].v[e & 7] = f2bf(st[e] * inv);
    // o_h^T [d][query] = v_h^T [d][keys] p^T ; A row i = feature f0 + (l31 % DH), keys of element s at k-step ks2:
    // 16*ks2 + 4*kh + (s&3) + 8*(s>>2)
    f32x16_t ot;
#pragma unroll
    for (int e = 0; e < 16; ++e) ot[e] = 0.0f;
    lds_byte_ptr vrow = L + AT_VT + (f0 + (l31 % DH)) * VT_LD;
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      Frag8 fv;
      fv.h[0] = *reinterpret_cast<lds_b64_ptr>(vrow + (16 * ks2 + 4 * kh) * 2);
      fv.h[1] = *reinterpret_cast<lds_b64_ptr>(vrow + (16 * ks2 + 4 * kh + 8) * 2);
      ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv.v, pf[ks2].v, ot, 0, 0, 0);
    }
    // rows i = (e&3) + 8*(e>>2) + 4*kh < DH are features f0 + i of token l31
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i0 = 8 * g + 4 * kh;
      if (i0 < DH) {
        bf16x4_t ov;
#pragma unroll
        for (int i = 0; i < 4; ++i) ov[i] = f2bf(ot[4 * g + i]);
        const int f = f0 + i0;
        *reinterpret_cast<bf16x4_t*>(smem + AT_A1 + l31 * 256 + (((f >> 3) ^ sw) << 4) + (f & 7) * 2) = ov;
        if (a.save_o) *reinterpret_cast<bf16x4_t*>(a.save_o + (row0 + l31) * E_DIM + f) = ov;
      }
    }
  }
  if (a.save_qkv) {      // v, row-major, from the wave's own v^T rows: thread -> (token l31, 4 features) x 4
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = w * 32 + 4 * kh + 8 * g;
      bf16x4_t vv;
#pragma unroll
      for (int i = 0; i < 4; ++i) vv[i] = *reinterpret_cast<const bf16_t*>(smem + AT_VT + (f + i) * VT_LD + l31 * 2);
      *reinterpret_cast<bf16x4_t*>(a.save_qkv + (row0 + l31) * 384 + 256 + f) = vv;
    }
  }
  // residual + bias for the output rows of this lane (features w*32 + 4*kh + 8*g .. +3 of token l31)
  float4 res[4], bo[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    res[g] = *reinterpret_cast<const float4*>(a.h_in + (row0 + l31) * E_DIM + w * 32 + 4 * kh + 8 * g);
    bo[g] = *reinterpret_cast<const float4*>(a.b_o + w * 32 + 4 * kh + 8 * g);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // Wo landed (and the loads above)
  __syncthreads();                                             // o tile complete, Wo visible to every wave

  // ---- out_proj^T: rows = output features (wave w: 32 of them), cols = tokens
  f32x16_t co;
#pragma unroll
  for (int e = 0; e < 16; ++e) co[e] = 0.0f;
  {
    lds_byte_ptr wrow = L + AT_W + (w * 32 + l31) * 256;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int cof = ((ks * 2 + kh) ^ sw) << 4;
      const bf16x8_t fw = *reinterpret_cast<lds_b128_ptr>(wrow + cof);
      const bf16x8_t fo = *reinterpret_cast<lds_b128_ptr>(L + AT_A1 + l31 * 256 + cof);
      co = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fo, co, 0, 0, 0);
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 o;
    o.x = co[4 * g + 0] + bo[g].x + res[g].x;
    o.y = co[4 * g + 1] + bo[g].y + res[g].y;
    o.z = co[4 * g + 2] + bo[g].z + res[g].z;
    o.w = co[4 * g + 3] + bo[g].w + res[g].w;
    *reinterpret_cast<float4*>(a.h_out + (row0 + l31) * E_DIM + w * 32 + 4 * kh + 8 * g) = o;
  }
}


// =====================================================================================================
//   attn_block_bwd:  backward of the attention half-layer between the two LayerNorms, one launch per layer:
//     dO   = dh_mid Wo^T                        (out_proj dgrad)
//     dqkv = attention backward (softmax recomputed from the saved q, k, v)
//     da1  = dqkv Wqkv^T                        (qkv dgrad; input of the ln1 backward)
//   The two weight gradients (X = o, dY = dh_mid and X = a1, dY = dqkv) stay separate GEMMs; dqkv is written out
//   for the second one.  One workgroup per sample, wave w owns features [32w, 32w+32) of q, k, v and dO.
//   Per head the score tile is needed in both orientations: s^T (lane = query; softmax statistics and the row
//   dot product are in-lane + one shuffle) feeds dq^T = k^T ds^T, and s (lane = key; statistics come back through
//   a 3x32-float LDS table) feeds dk^T = q^T ds and dv^T = do^T p.  All "transposed" A fragments are
//   ds_read_b64_tr_b16 gathers from the row-major q / k / dO tiles; p, ds go from the accumulators straight into
//   B fragments (C layout == B layout up to the contraction permutation the tr-reads are issued with).
// =====================================================================================================
constexpr int AB_W = 0;                        // Wo [128][256 B] + dh tile first; later Wqkv as 3 x [128][256 B]
constexpr int AB_DH = 128 * 256;               // dh_mid tile [32][256 B] (inside the future Wqkv region)
constexpr int AB_QKV = 3 * 128 * 256;          // saved q | k | v tiles, 3 x [32][256 B]
constexpr int AB_DO = AB_QKV + 3 * 8192;       // dO tile [32][256 B]
constexpr int AB_DQKV = AB_DO + 8192;          // dq | dk | dv tiles, 3 x [32][256 B]
constexpr int AB_ST = AB_DQKV + 3 * 8192;      // per wave: max, 1/sum, row-dot of 32 queries (3 x 32 floats)
constexpr int AB_SMEM = AB_ST + 4 * 3 * 32 * 4;

struct AttnBwdArgs {
  const bf16_t* dh_mid;     // [R][128] gradient wrt the half-layer output (bf16)
  const bf16_t* qkv;        // [R][384] saved (q unscaled)
  const bf16_t* Wo;         // out_proj kernel as W [in 128][out 128] bf16 (dgrad operand pack)
  const bf16_t* Wqkv;       // qkv kernel as W [in 128][out 384]
  bf16_t* dqkv;             // [R][384]
  bf16_t* da1;              // [R][128]
};

__device__ __forceinline__ bf16x4_t tr_read(unsigned addr) {
  bf16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

template <int DH>
__global__ __launch_bounds__(256) void attn_block_bwd_kernel(AttnBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[AB_SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t row0 = (size_t)blockIdx.x * S_TOK;
  const int kh = lane >> 5, l31 = lane & 31, sw = l31 & 15;
  lds_byte_ptr L = (lds_byte_ptr)smem;
  const unsigned L0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);

  // ---- DMA: 256-B-row tiles, 16 rows per round over the 4 waves, source chunk = position ^ (row & 15)
  const int rl = w * 4 + (lane >> 4);
  const uint32_t csw = (uint32_t)(((lane & 15) ^ rl) * 16);
  unsigned char* lds_w = smem + w * 1024;
  const __amdgpu_buffer_rsrc_t wo_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.Wo), 0, 128 * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t wq_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.Wqkv), 0, 128 * 768, 0x00020000);
  const __amdgpu_buffer_rsrc_t dh_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dh_mid + row0 * E_DIM), 0, 32 * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t qkv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.qkv + row0 * 384), 0, 32 * 768, 0x00020000);
#pragma unroll
  for (int j = 0; j < 8; ++j) glds16(wo_rsrc, (uint32_t)rl * 256u + csw, (uint32_t)(j * 4096), lds_w + AB_W + j * 4096);
#pragma unroll
  for (int j = 0; j < 2; ++j) glds16(dh_rsrc, (uint32_t)rl * 256u + csw, (uint32_t)(j * 4096), lds_w + AB_DH + j * 4096);
#pragma unroll
  for (int pp = 0; pp < 3; ++pp)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      glds16(qkv_rsrc, (uint32_t)rl * 768u + csw, (uint32_t)(pp * 256 + j * 16 * 768), lds_w + AB_QKV + pp * 8192 + j * 4096);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- dO^T (rows = in-features of out_proj = this wave's 32, cols = tokens) = Wo[i][:] . dh[token][:]
  {
    f32x16_t c0, c1;
#pragma unroll
    for (int e = 0; e < 16; ++e) c0[e] = c1[e] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 8; ks += 2) {
      const int o0 = ((ks * 2 + kh) ^ sw) << 4, o1 = (((ks + 1) * 2 + kh) ^ sw) << 4;
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<lds_b128_ptr>(L + AB_W + (w * 32 + l31) * 256 + o0),
                                                   *reinterpret_cast<lds_b128_ptr>(L + AB_DH + l31 * 256 + o0), c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<lds_b128_ptr>(L + AB_W + (w * 32 + l31) * 256 + o1),
                                                   *reinterpret_cast<lds_b128_ptr>(L + AB_DH + l31 * 256 + o1), c1, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = w * 32 + 4 * kh + 8 * g;
      bf16x4_t v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = f2bf(c0[4 * g + i] + c1[4 * g + i]);
      *reinterpret_cast<bf16x4_t*>(smem + AB_DO + l31 * 256 + (((f >> 3) ^ sw) << 4) + (f & 7) * 2) = v;
    }
  }
  __syncthreads();          // Wo and dh are dead: their space becomes the Wqkv tiles, DMA'd behind the attention
#pragma unroll
  for (int pp = 0; pp < 3; ++pp)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      glds16(wq_rsrc, (uint32_t)rl * 768u + csw, (uint32_t)(pp * 256 + j * 16 * 768), lds_w + AB_W + pp * 32768 + j * 4096);

  // ---- attention backward, heads inside this wave's 32 features
  const float scale = rsqrtf((float)DH);
  float* stats = reinterpret_cast<float*>(smem + AB_ST) + w * 96;
  const int gg = lane >> 4, ig = lane & 15;
  constexpr int HPW = 32 / DH;
  constexpr int KS = (DH + 15) / 16;
#pragma unroll
  for (int hh = 0; hh < HPW; ++hh) {
    const int f0 = w * 32 + hh * DH;
    // row-major fragments (lane = token l31, 8 consecutive features of this lane half), zero past the head
    bf16x8_t fq[KS], fk[KS], fv[KS], fd[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int f = f0 + ks * 16 + kh * 8;
      if (DH >= 16 || kh == 0) {
        const int off = l31 * 256 + (((f >> 3) ^ sw) << 4);
        fq[ks] = *reinterpret_cast<lds_b128_ptr>(L + AB_QKV + off);
        fk[ks] = *reinterpret_cast<lds_b128_ptr>(L + AB_QKV + 8192 + off);
        fv[ks] = *reinterpret_cast<lds_b128_ptr>(L + AB_QKV + 16384 + off);
        fd[ks] = *reinterpret_cast<lds_b128_ptr>(L + AB_DO + off);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { fq[ks][i] = (bf16_t)0.0f; fk[ks][i] = (bf16_t)0.0f; fv[ks][i] = (bf16_t)0.0f; fd[ks][i] = (bf16_t)0.0f; }
      }
    }
    // ---- orientation T: rows = keys, cols = queries
    f32x16_t st, dpt;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = dpt[e] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[ks], fq[ks], st, 0, 0, 0);
      dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[ks], fd[ks], dpt, 0, 0, 0);
    }
    float mx = st[0] * scale;
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] *= scale; mx = fmaxf(mx, st[e]); }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] = __expf(st[e] - mx); sum += st[e]; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] *= inv; dot += st[e] * dpt[e]; }
    dot += __shfl_xor(dot, 32, 64);
    if (kh == 0) { stats[l31] = mx; stats[32 + l31] = inv; stats[64 + l31] = dot; }
    Frag8 dst_[2];                                 // ds^T as B fragments (k-step ks2, element s <-> accumulator element 8*ks2 + s)
#pragma unroll
    for (int e = 0; e < 16; ++e) dst_[e >> 3].v[e & 7] = f2bf(st[e] * (dpt[e] - dot));
    // dq^T [d][query] = k^T [d][keys] ds^T : A gathered from the row-major k tile with transposing reads
    {
      f32x16_t dq;
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[e] = 0.0f;
      const int fcol = f0 + 16 * (gg & 1) + 4 * (ig & 3);      // the 4 features whose address this lane supplies
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const int r0 = 16 * ks2 + 4 * kh + (ig >> 2), r1 = r0 + 8;
        Frag8 fa;
        fa.h[0] = tr_read(L0 + AB_QKV + 8192 + r0 * 256 + (((fcol >> 3) ^ (r0 & 15)) << 4) + (fcol & 7) * 2);
        fa.h[1] = tr_read(L0 + AB_QKV + 8192 + r1 * 256 + (((fcol >> 3) ^ (r1 & 15)) << 4) + (fcol & 7) * 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, dst_[ks2].v, dq, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i0 = 8 * g + 4 * kh;
        if (i0 < DH) {
          const int f = f0 + i0;
          bf16x4_t v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = f2bf(dq[4 * g + i] * scale);
          *reinterpret_cast<bf16x4_t*>(smem + AB_DQKV + l31 * 256 + (((f >> 3) ^ sw) << 4) + (f & 7) * 2) = v;
          *reinterpret_cast<bf16x4_t*>(a.dqkv + (row0 + l31) * 384 + f) = v;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- orientation N: rows = queries, cols = keys
    f32x16_t sn, dpn;
#pragma unroll
    for (int e = 0; e < 16; ++e) sn[e] = dpn[e] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[ks], fk[ks], sn, 0, 0, 0);
      dpn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd[ks], fv[ks], dpn, 0, 0, 0);
    }
    Frag8 pn[2], dsn[2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int q0 = 8 * g + 4 * kh;                            // queries of elements 4g .. 4g+3
      const float4 m4 = *reinterpret_cast<const float4*>(stats + q0);
      const float4 i4 = *reinterpret_cast<const float4*>(stats + 32 + q0);
      const float4 d4 = *reinterpret_cast<const float4*>(stats + 64 + q0);
      const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = 4 * g + i;
        const float pv = __expf(sn[e] * scale - mm[i]) * ii[i];
        pn[e >> 3].v[e & 7] = f2bf(pv);
        dsn[e >> 3].v[e & 7] = f2bf(pv * (dpn[e] - dd[i]));
      }
    }
    {
      f32x16_t dk, dv;
#pragma unroll
      for (int e = 0; e < 16; ++e) dk[e] = dv[e] = 0.0f;
      const int fcol = f0 + 16 * (gg & 1) + 4 * (ig & 3);
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const int r0 = 16 * ks2 + 4 * kh + (ig >> 2), r1 = r0 + 8;
        const int o0 = r0 * 256 + (((fcol >> 3) ^ (r0 & 15)) << 4) + (fcol & 7) * 2;
        const int o1 = r1 * 256 + (((fcol >> 3) ^ (r1 & 15)) << 4) + (fcol & 7) * 2;
        Frag8 fqT, fdT;
        fqT.h[0] = tr_read(L0 + AB_QKV + o0);
        fqT.h[1] = tr_read(L0 + AB_QKV + o1);
        fdT.h[0] = tr_read(L0 + AB_DO + o0);
        fdT.h[1] = tr_read(L0 + AB_DO + o1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fqT.v, dsn[ks2].v, dk, 0, 0, 0);
        dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fdT.v, pn[ks2].v, dv, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i0 = 8 * g + 4 * kh;
        if (i0 < DH) {
          const int f = f0 + i0;
          bf16x4_t vk, vv;
#pragma unroll
          for (int i = 0; i < 4; ++i) { vk[i] = f2bf(dk[4 * g + i] * scale); vv[i] = f2bf(dv[4 * g + i]); }
          const int off = l31 * 256 + (((f >> 3) ^ sw) << 4) + (f & 7) * 2;
          *reinterpret_cast<bf16x4_t*>(smem + AB_DQKV + 8192 + off) = vk;
          *reinterpret_cast<bf16x4_t*>(smem + AB_DQKV + 16384 + off) = vv;
          *reinterpret_cast<bf16x4_t*>(a.dqkv + (row0 + l31) * 384 + 128 + f) = vk;
          *reinterpret_cast<bf16x4_t*>(a.dqkv + (row0 + l31) * 384 + 256 + f) = vv;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // Wqkv landed
  __syncthreads();                                       // dqkv tiles complete

  // ---- da1^T (rows = in-features k of the qkv projection: this wave's 32, cols = tokens)
  {
    f32x16_t c[3];
#pragma unroll
    for (int pp = 0; pp < 3; ++pp)
#pragma unroll
      for (int e = 0; e < 16; ++e) c[pp][e] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int o = ((ks * 2 + kh) ^ sw) << 4;
#pragma unroll
      for (int pp = 0; pp < 3; ++pp)
        c[pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<lds_b128_ptr>(L + AB_W + pp * 32768 + (w * 32 + l31) * 256 + o),
                                                        *reinterpret_cast<lds_b128_ptr>(L + AB_DQKV + pp * 8192 + l31 * 256 + o), c[pp], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4_t v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = f2bf((c[0][4 * g + i] + c[1][4 * g + i]) + c[2][4 * g + i]);
      *reinterpret_cast<bf16x4_t*>(a.da1 + (row0 + l31) * E_DIM + w * 32 + 4 * kh + 8 * g) = v;
    }
  }
}

}  // namespace

int launch_mlp_block_fwd(const float* h_in, float* h_out, int rows, const float* gamma, const float* beta,
                         const bf16_t* W1t, const float* b1, const bf16_t* W2t, const float* b2, int M, bf16_t* save_a2,
                         bf16_t* save_z1, bf16_t* save_u, hipStream_t st) {
  SMD_ARG_CHECK(h_in && h_out && gamma && beta && W1t && b1 && W2t && b2, "mlp_block_fwd: null pointer");
  SMD_ARG_CHECK(rows > 0 && rows % S_TOK == 0, "mlp_block_fwd: rows=%d must be a multiple of 32", rows);
  SMD_ARG_CHECK(M >= CH && M % CH == 0 && M <= MAX_HIDDEN, "mlp_block_fwd: hidden width %d must be a multiple of 128 and <= 8192", M);
  MlpArgs a;
  a.h_in = h_in; a.h_out = h_out; a.gamma = gamma; a.beta = beta; a.W1t = W1t; a.b1 = b1; a.W2t = W2t; a.b2 = b2; a.M = M;
  a.save_a2 = save_a2; a.save_z1 = save_z1; a.save_u = save_u;
  const int variant = smd_tuning_get("mlp_variant");
  if (variant == 0) {        // 8-wave kernel (default); saves need all three pointers
    const bool save = save_z1 && save_u;
    SMD_ARG_CHECK(save || (!save_z1 && !save_u), "mlp_block_fwd: save_z1 and save_u come together");
    if (save) hipLaunchKernelGGL(mlp_block_fwd8_kernel<true>, dim3(rows / S_TOK), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(mlp_block_fwd8_kernel<false>, dim3(rows / S_TOK), dim3(512), 0, st, a);
  } else {
    switch (variant) {
      case 1: hipLaunchKernelGGL(mlp_block_fwd_kernel<1>, dim3(rows / S_TOK), dim3(256), 0, st, a); break;
      case 2: hipLaunchKernelGGL(mlp_block_fwd_kernel<2>, dim3(rows / S_TOK), dim3(256), 0, st, a); break;
      default: hipLaunchKernelGGL(mlp_block_fwd_kernel<0>, dim3(rows / S_TOK), dim3(256), 0, st, a); break;
    }
  }
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_attn_block_fwd(const float* h_in, float* h_out, int rows, const float* gamma, const float* beta,
                          const bf16_t* Wqkv_t, const float* b_qkv, const bf16_t* Wo_t, const float* b_o, int num_heads,
                          bf16_t* save_a1, bf16_t* save_qkv, bf16_t* save_o, hipStream_t st) {
  SMD_ARG_CHECK(h_in && h_out && gamma && beta && Wqkv_t && b_qkv && Wo_t && b_o, "attn_block_fwd: null pointer");
  SMD_ARG_CHECK(rows > 0 && rows % S_TOK == 0, "attn_block_fwd: rows=%d must be a multiple of 32", rows);
  AttnArgs a;
  a.h_in = h_in; a.h_out = h_out; a.gamma = gamma; a.beta = beta; a.Wqkv_t = Wqkv_t; a.b_qkv = b_qkv; a.Wo_t = Wo_t; a.b_o = b_o;
  a.save_a1 = save_a1; a.save_qkv = save_qkv; a.save_o = save_o;
  const dim3 grid(rows / S_TOK), block(256);
  switch (num_heads) {
    case 4: hipLaunchKernelGGL(attn_block_fwd_kernel<32>, grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL(attn_block_fwd_kernel<16>, grid, block, 0, st, a); break;
    case 16: hipLaunchKernelGGL(attn_block_fwd_kernel<8>, grid, block, 0, st, a); break;
    default: smd_set_error("attn_block_fwd: num_heads=%d unsupported (4, 8, 16)", num_heads); return -1;
  }
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_attn_block_bwd(const bf16_t* dh_mid, const bf16_t* qkv, const bf16_t* Wo, const bf16_t* Wqkv, bf16_t* dqkv,
                          bf16_t* da1, int rows, int num_heads, hipStream_t st) {
  SMD_ARG_CHECK(dh_mid && qkv && Wo && Wqkv && dqkv && da1, "attn_block_bwd: null pointer");
  SMD_ARG_CHECK(rows > 0 && rows % S_TOK == 0, "attn_block_bwd: rows=%d must be a multiple of 32", rows);
  AttnBwdArgs a;
  a.dh_mid = dh_mid; a.qkv = qkv; a.Wo = Wo; a.Wqkv = Wqkv; a.dqkv = dqkv; a.da1 = da1;
  const dim3 grid(rows / S_TOK), block(256);
  switch (num_heads) {
    case 4: hipLaunchKernelGGL(attn_block_bwd_kernel<32>, grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL(attn_block_bwd_kernel<16>, grid, block, 0, st, a); break;
    case 16: hipLaunchKernelGGL(attn_block_bwd_kernel<8>, grid, block, 0, st, a); break;
    default: smd_set_error("attn_block_bwd: num_heads=%d unsupported (4, 8, 16)", num_heads); return -1;
  }
  SMD_LAUNCH_CHECK();
  return 0;
}
