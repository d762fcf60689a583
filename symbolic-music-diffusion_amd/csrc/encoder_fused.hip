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
      const float rstd = smd_ln_rstd(s2 * (1.0f / E_DIM) - mean * mean + LN_EPS);
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
      const float rstd = smd_ln_rstd(s2 * (1.0f / E_DIM) - mean * mean + LN_EPS);
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

// All-reduce of R independent values per lane over the 64 lanes of a wave; the R chains are interleaved step by
// step so that the exchange latencies overlap.  Every step adds two commuting operands, so all lanes end with the bitwise
// identical sum, and the DPP and the ds_bpermute form of a step give the same bits.
//   DPP = 0: six xor exchanges through the LDS crossbar (ds_bpermute, __shfl_xor): 6 R LDS-pipe operations per wave.
//   DPP = 1: the four intra-row steps as DPP moves on the VALU (quad_perm, row_half_mirror, row_mirror), the two cross-row
//            steps through the crossbar: 2 R LDS-pipe operations.  (The round-2 suspicion against the DPP form was cleared in
//            round 3: the reduction primitive is innocent of the co-residency miscompare)
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

// =====================================================================================================
//   mlp_hs_fwd: the MLP half-layer with the HIDDEN dimension split over workgroups ("hs").
//
// The one-sample-per-workgroup kernels above make every CU stream all of W1 + W2 (1 MiB) through its LDS (~36 GB/s per
// CU in every schedule tried, 29 us per layer).  Here a workgroup owns NS
// samples x one QUARTER of the hidden units: 256 KiB of weights per CU instead of 1 MiB, every weight fragment read
// from LDS once per NS samples, the same number of workgroups (B/NS x 4).  Per 128-unit chunk:
//   GEMM1  z^T[hidden][token] = W1[chunk] a2^T  (+ b1, GELU)  -> u tile in LDS (row-major [token][hidden], bf16)
//   GEMM2  out^T[n][token]   += W2t[:, chunk] u^T            (wave = 32 output features x 16 tokens: no cross-wave sum)
// Both are 16x16x32 MFMAs with the weight slice as the A operand (rows DMA'd as they lie in the operand pack).
// The LayerNorm is NOT in here: the attention kernel of the same layer emits a2 = ln2(h_mid) (bf16), DMA'd into LDS.
// The four hidden quarters leave four fp32 partial tiles (quarter 0 also carries b2 + the residual); the CONSUMER of
// the residual stream -- the next layer's attention kernel, or ln128_parts_kernel for the last layer -- adds them in
// the fixed order (p0 + p1) + (p2 + p3).  (An in-kernel last-arriver combine was measured first: 15.6 us of publish +
// combine on a 19 us kernel, profiles/README.md.)
// =====================================================================================================
constexpr int HS_NQ = 4;                       // hidden quarters
constexpr int HS_WBUF = 65536;                 // one chunk: W1 slice [128 hidden][256 B] + W2t slice [128 n][256 B]
constexpr int HS_TILE = 2 * HS_WBUF;           // token tiles: a2 (prologue), then u, NS x [32][256 B]

struct MlpHsArgs {
  const bf16_t* a2;         // [R][128] bf16 = ln2(h_mid)
  const float* h_res;       // [R][128] fp32 residual (h_mid)
  const bf16_t* W1t; const float* b1; const bf16_t* W2t; const float* b2;
  int M;
  float* part;              // [4][R][128] fp32 partial tiles of h_out
  int rows;
  int dbg;
};

// TS instantiation (tuning knob mlp_hs_dbg = 64, tools/mlp_hs_phases.py): every wave stamps s_memtime at its phase boundaries
// -- behind an explicit wait for what the phase issued -- and leaves the stamps in its partial tile instead of the result.
#define HS_TS(i)                                                             \
  if constexpr (TS) {                                                        \
    __builtin_amdgcn_sched_barrier(0);                                       \
    ts[i] = (uint32_t)__builtin_amdgcn_s_memtime();                          \
    __builtin_amdgcn_sched_barrier(0);                                       \
  }
// NWV = 16 (the default for groups of four samples; mlp_variant = 5 selects the eight-wave form): sixteen waves, four per SIMD;
// the second group of eight takes the second half of the workgroup's samples (NSW = NS / 2 per wave: 128 registers instead of
// 200), every DMA round covers 64 tile rows instead of 32.  Bit-identical; 18.9 -> 18.2 us isolated, sample step +1-2 %
// (profiles/r4w_mlp_fwd16_ab.txt) -- the phases stay barrier-synchronous, so the gain is only the latency four waves hide.
template <int NS, bool TS = false, int NWV = 8>
__global__ __launch_bounds__(64 * NWV) void mlp_hs_fwd_kernel(MlpHsArgs a) {
  constexpr int SGS = NWV / 8;                   // sample groups of waves
  constexpr int NSW = NS / SGS;                  // samples per wave
  constexpr int RR = 32 * SGS;                   // tile rows per DMA round
  constexpr int WR = 4 / SGS;                    // DMA rounds per 128-row weight slice
  static_assert(NS % SGS == 0 && (NWV == 8 || NWV == 16), "mlp_hs_fwd: wave count");
  uint32_t ts[36];
  HS_TS(0)
  __shared__ __attribute__((aligned(16))) unsigned char smem[HS_TILE + NS * 8192];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hg = w & 3, th = (w >> 2) & 1, s_lo = (w >> 3) * NSW;       // this wave's samples: s_lo .. s_lo + NSW - 1
  const int q = blockIdx.x & (HS_NQ - 1), grp = blockIdx.x >> 2;
  const size_t row0 = (size_t)grp * (S_TOK * NS);
  const int g = lane >> 4, j = lane & 15;
  const int tok = th * 16 + j;
  const int HQ = a.M / HS_NQ, hbase = q * HQ, nchunks = HQ / CH;
  lds_byte_ptr L = (lds_byte_ptr)smem;

  // ---- DMA (as mlp_block_fwd8): round r covers tile rows r*32 + w*4 + (lane>>4), 16 lanes per 256-B row, source
  // chunk = LDS chunk ^ (row & 15)
  const int rl = w * 4 + (lane >> 4);
  const uint32_t cs = (uint32_t)(((lane & 15) ^ (rl & 15)) * 16);
  const uint32_t w1_v = (uint32_t)rl * 256u + cs;
  const uint32_t w2_v = (uint32_t)rl * (uint32_t)(a.M * 2) + cs;
  const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W1t), 0, a.M * E_DIM * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W2t), 0, a.M * E_DIM * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t a2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.a2 + row0 * E_DIM), 0, NS * 8192, 0x00020000);
  const uint32_t w2_round = (uint32_t)(RR * a.M * 2);
  unsigned char* lds_w = smem + w * 1024;
  auto stage_chunk = [&](int c, int buf) {
    unsigned char* d1 = lds_w + buf * HS_WBUF;
    const uint32_t h0 = (uint32_t)(hbase + c * CH);
#pragma unroll
    for (int r = 0; r < WR; ++r) glds16(w1_rsrc, w1_v, h0 * 256u + (uint32_t)(r * RR * 256), d1 + r * RR * 256);
#pragma unroll
    for (int r = 0; r < WR; ++r) glds16(w2_rsrc, w2_v, h0 * 2u + (uint32_t)r * w2_round, d1 + W_SLICE + r * RR * 256);
  };
  float4 bnext[2];
  auto fetch_bias = [&](int c) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) bnext[mt] = *reinterpret_cast<const float4*>(a.b1 + hbase + c * CH + hg * 32 + mt * 16 + 4 * g);
  };
  fetch_bias(0);
#pragma unroll
  for (int s = 0; s < NS / SGS; ++s) glds16(a2_rsrc, w1_v, (uint32_t)(s * RR * 256), lds_w + HS_TILE + s * RR * 256);      // a2 tiles
  stage_chunk(0, 0);
  if constexpr (NWV == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // the a2 tiles (and the bias) have landed; chunk 0 may fly on
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // B fragments of a2^T per sample (lane = token th*16 + j, k-chunk 4 ks + g), kept in registers for the whole kernel
  bf16x8_t a2f[NSW][4];
#pragma unroll
  for (int s = 0; s < NSW; ++s)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      a2f[s][ks] = *reinterpret_cast<lds_b128_ptr>(L + HS_TILE + ((s_lo + s) * 32 + tok) * 256 + (((ks * 4 + g) ^ (tok & 15)) << 4));

  f32x4_t acc[NSW][2];
#pragma unroll
  for (int s = 0; s < NSW; ++s)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[s][mt][e] = 0.0f;

  const int arow = (hg * 32 + j) * 256;                    // A-fragment row of this lane inside a slice (m-tile 0)
  HS_TS(1)
  const int nloop = TS ? 4 : ((a.dbg & 1) ? 0 : nchunks);  // TS: constant stamp indices (the launcher checks hidden = 2048)
#pragma unroll
  for (int c = 0; c < nloop; ++c) {
    const int buf = c & 1;
    HS_TS(2 + c * 8)
    // The W1 slice of chunk c has landed (vmcnt(4): its W2 slice, issued right behind it, may still be in flight: it
    // is only needed after the second barrier); every wave is done with iteration c-1, so the other weight buffer and
    // the u tiles are free (the first pass also orders the a2 fragment reads before the u writes).
    if constexpr (NWV == 8) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    HS_TS(3 + c * 8)
    const float4 bcur[2] = {bnext[0], bnext[1]};
    const bool more = c + 1 < nchunks;
    if (more) {
      fetch_bias(c + 1);
      stage_chunk(c + 1, buf ^ 1);
    }
    lds_byte_ptr s1 = L + buf * HS_WBUF;
    lds_byte_ptr s2 = s1 + W_SLICE;
    {
      // ---- GEMM1: z^T tiles [16 hidden][16 tokens] x 2 m-tiles x NS samples
      bf16x8_t wf[2][4];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[mt][ks] = *reinterpret_cast<lds_b128_ptr>(s1 + arow + mt * 16 * 256 + (((ks * 4 + g) ^ j) << 4));
      if constexpr (TS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      HS_TS(4 + c * 8)
      f32x4_t z[NSW][2];
#pragma unroll
      for (int s = 0; s < NSW; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) z[s][mt][e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int s = 0; s < NSW; ++s)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) z[s][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[mt][ks], a2f[s][ks], z[s][mt], 0, 0, 0);
      if constexpr (TS) {                  // the last MFMA of every accumulator has retired
#pragma unroll
        for (int s = 0; s < NSW; ++s)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) asm volatile("v_mov_b32 %0, %0" : "+v"(z[s][mt][0]));
      }
      HS_TS(5 + c * 8)
      // + b1, GELU -> u[sample][token][hidden] (hidden of z[s][mt][e]: hg*32 + mt*16 + 4g + e)
#pragma unroll
      for (int s = 0; s < NSW; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const float bb[4] = {bcur[mt].x, bcur[mt].y, bcur[mt].z, bcur[mt].w};
          bf16x4_t uu;
          {   // two activations per VALU instruction (smd_common.h geluf2_: bit-identical to geluf_)
            const f32x2_t g01 = geluf2_(f32x2_t{z[s][mt][0], z[s][mt][1]} + f32x2_t{bb[0], bb[1]});
            const f32x2_t g23 = geluf2_(f32x2_t{z[s][mt][2], z[s][mt][3]} + f32x2_t{bb[2], bb[3]});
            uu[0] = f2bf(g01.x); uu[1] = f2bf(g01.y); uu[2] = f2bf(g23.x); uu[3] = f2bf(g23.y);
          }
          const int hid = hg * 32 + mt * 16 + 4 * g;
          *reinterpret_cast<bf16x4_t*>(smem + HS_TILE + ((s_lo + s) * 32 + tok) * 256 + (((hid >> 3) ^ (tok & 15)) << 4) + (hid & 7) * 2) = uu;
        }
    }
    if constexpr (TS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    HS_TS(6 + c * 8)
    // u tiles written; the W2 slice of this chunk has landed: behind it only the next chunk's 2 bias loads + 8 DMAs
    if (more) {
      if constexpr (NWV == 8) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    HS_TS(7 + c * 8)
    {
      // ---- GEMM2: out^T tiles [16 n][16 tokens] += W2t[n][chunk] u^T
      bf16x8_t vf[2][4];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) vf[mt][ks] = *reinterpret_cast<lds_b128_ptr>(s2 + arow + mt * 16 * 256 + (((ks * 4 + g) ^ j) << 4));
#pragma unroll
      for (int s = 0; s < NSW; ++s) {
        bf16x8_t uf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          uf[ks] = *reinterpret_cast<lds_b128_ptr>(L + HS_TILE + ((s_lo + s) * 32 + tok) * 256 + (((ks * 4 + g) ^ (tok & 15)) << 4));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[s][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[mt][ks], uf[ks], acc[s][mt], 0, 0, 0);
      }
    }
    if constexpr (TS) {
#pragma unroll
      for (int s = 0; s < NSW; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) asm volatile("v_mov_b32 %0, %0" : "+v"(acc[s][mt][0]));
    }
    HS_TS(8 + c * 8)
    __builtin_amdgcn_sched_barrier(0);
  }
  HS_TS(34)

  // ---- this quarter's partial tile: out[token][n], n = hg*32 + mt*16 + 4g + e (4 consecutive per lane); quarter 0
  // carries the bias and the residual
  {
    float* dst = a.part + ((size_t)q * a.rows + row0) * E_DIM;
#pragma unroll
    for (int s = 0; s < NSW; ++s)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int n0 = hg * 32 + mt * 16 + 4 * g;
        float4 v = make_float4(acc[s][mt][0], acc[s][mt][1], acc[s][mt][2], acc[s][mt][3]);
        if (q == 0) {
          const float4 r = *reinterpret_cast<const float4*>(a.h_res + (row0 + (s_lo + s) * 32 + tok) * E_DIM + n0);
          const float4 bb = *reinterpret_cast<const float4*>(a.b2 + n0);
          v.x += bb.x + r.x; v.y += bb.y + r.y; v.z += bb.z + r.z; v.w += bb.w + r.w;
        }
        *reinterpret_cast<float4*>(dst + (size_t)((s_lo + s) * 32 + tok) * E_DIM + n0) = v;
      }
    if constexpr (TS) {                    // 36 stamps per wave at the head of the workgroup's partial tile (40 dwords apart)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      HS_TS(35)
      __syncthreads();
      if (lane == 0) {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst) + w * 40;
#pragma unroll
        for (int i = 0; i < 36; ++i) d[i] = ts[i];
      }
    }
  }
}
#undef HS_TS

// =====================================================================================================
//   mlp_hs_bwd: backward of the MLP half-layer between ln2 and the residual add, hidden split as in mlp_hs_fwd, with
//   the hidden activations RECOMPUTED from a2 instead of saved by the forward pass (which then saves nothing):
//     z  = a2 W1 + b1,  u = gelu(z)            (written out: X operand of the fc2 weight gradient)
//     du = dh W2^T,     dz = du * gelu'(z)     (written out: dY operand of the fc1 weight gradient)
//     da2 += dz W1^T                           (four fp32 partial tiles, summed by ln128_bwd_parts_kernel)
//   A workgroup owns 4 samples x a quarter of the hidden units and walks it in chunks of 32 units; per chunk three 8-KiB
//   weight slices are DMA'd (ring of three buffers, two chunks in flight):
//     S1 = W1t[chunk][128 k]   A operand of z^T[h][token]  = S1 a2^T        (forward pack of fc1, rows as they lie)
//     S2 = W2[chunk][128 n]    A operand of du^T[h][token] = S2 dh^T        (dgrad pack of fc2)
//     S3 = W1[128 k][chunk]    A operand of da2^T[k][token] += S3 dz^T      (dgrad pack of fc1, 64-B row pieces)
//   Waves: ht = hidden 16-tile of the chunk (and k half of da2), th = token half, sp = sample pair.  u and dz meet the
//   third GEMM and the global stores through two small LDS tiles ([token][32 hidden], 64-B rows, chunk-swizzled).
// =====================================================================================================
constexpr int HB_CH = 32;
constexpr int HB_SL = 8192;                              // one weight slice
constexpr int HB_WBUF = 3 * HB_SL;                       // S1 | S2 | S3
constexpr int HB_NS = 4;                                 // samples per workgroup
constexpr int HB_A2 = 3 * HB_WBUF;                       // a2 tiles  NS x [32][256 B]
constexpr int HB_DH = HB_A2 + HB_NS * 8192;              // dh tiles
constexpr int HB_U = HB_DH + HB_NS * 8192;               // u stage  [NS*32 tokens][64 B]
constexpr int HB_DZ = HB_U + HB_NS * 2048;               // dz stage
constexpr int HB_B1 = HB_DZ + HB_NS * 2048;              // fc1 bias of this quarter (<= 2048 floats)
constexpr int HB_SMEM = HB_B1 + (MAX_HIDDEN / HS_NQ) * 4;

struct MlpHsBwdArgs {
  const bf16_t* a2;         // [R][128] ln2 output (saved by the attention kernel)
  const bf16_t* dh;         // [R][128] gradient wrt the half-layer output
  const bf16_t* W1t;        // [M][128]  fc1 forward pack
  const bf16_t* W2;         // [M][128]  fc2 dgrad pack  (kernel (in = M, out = 128))
  const bf16_t* W1;         // [128][M]  fc1 dgrad pack  (kernel (in = 128, out = M))
  const float* b1;
  int M;
  bf16_t* u;                // [R][M]
  bf16_t* dz;               // [R][M]
  float* part;              // [4][R][128] partial tiles of da2
  int rows;
};

// REGF: the wave's a2 / dh B fragments (its two samples x its token half: 16 x 16 bytes per lane) are read from LDS ONCE and
// kept in registers for all 16 chunks (the kernel needs 88 registers of the 256 two waves per SIMD may take) instead of being
// re-read every chunk: 16 of a chunk's ~32 ds_read_b128 and their latency in front of the first MFMA chain are gone.
// Piece swizzle of the 64-byte-row tiles (S3, u, dz): 16-byte piece p of row r lives at position p ^ hb_swz((r >> 2) & 3).  The
// permutation 0 2 3 1 instead of the identity because ds_read_b128 serves lanes {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} / ...
// together, not {0-15} / {16-31}: with the identity the rows j and j +- 4 of neighbouring k-groups met on one slot (2-way
// conflict on every S3 and dz fragment read: 24 of a chunk's 40 conflict cycles, tools/lds_conflicts.py).
__device__ __forceinline__ int hb_swz(int x) { return (0x78 >> (2 * x)) & 3; }

template <bool REGF>
__global__ __launch_bounds__(512) void mlp_hs_bwd_kernel(MlpHsBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[HB_SMEM];
  constexpr int NS = HB_NS, SPW = NS / 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ht = w & 1, th = (w >> 1) & 1, sp = w >> 2;
  const int q = blockIdx.x & (HS_NQ - 1), grp = blockIdx.x >> 2;
  const size_t row0 = (size_t)grp * (S_TOK * NS);
  const int g = lane >> 4, j = lane & 15;
  const int tok = th * 16 + j;
  const int HQ = a.M / HS_NQ, hbase = q * HQ, nchunks = HQ / HB_CH;
  lds_byte_ptr L = (lds_byte_ptr)smem;

  {
    // ---- DMA descriptors
    const int rl = w * 4 + (lane >> 4);                                   // row of a 32-row round of 256-B rows
    const uint32_t v256 = (uint32_t)rl * 256u + (uint32_t)(((lane & 15) ^ (rl & 15)) * 16);
    const int k3 = w * 16 + (lane >> 2);                                  // S3: 16 rows of 64 B per wave, 4 lanes per row
    const uint32_t v3 = (uint32_t)k3 * (uint32_t)(a.M * 2) + (uint32_t)((((lane & 3) ^ hb_swz((k3 >> 2) & 3))) * 16);
    const __amdgpu_buffer_rsrc_t w1t_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W1t), 0, a.M * E_DIM * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W2), 0, a.M * E_DIM * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W1), 0, a.M * E_DIM * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t a2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.a2 + row0 * E_DIM), 0, NS * 8192, 0x00020000);
    const __amdgpu_buffer_rsrc_t dh_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dh + row0 * E_DIM), 0, NS * 8192, 0x00020000);
    // fc1 bias of this quarter: 8 KiB window of b1 by DMA as well (reads past the end of b1 return 0: bounded descriptor)
    const __amdgpu_buffer_rsrc_t b1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.b1), 0, a.M * 4, 0x00020000);
    unsigned char* lds_w = smem + w * 1024;
    glds16(b1_rsrc, (uint32_t)(lane * 16), (uint32_t)(hbase * 4 + w * 1024), lds_w + HB_B1);
    auto stage_chunk = [&](int c, int buf) {
      unsigned char* d = lds_w + buf * HB_WBUF;
      const uint32_t h0 = (uint32_t)(hbase + c * HB_CH);
      glds16(w1t_rsrc, v256, h0 * 256u, d);
      glds16(w2_rsrc, v256, h0 * 256u, d + HB_SL);
      glds16(w1_rsrc, v3, h0 * 2u, d + 2 * HB_SL);
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      glds16(a2_rsrc, v256, (uint32_t)(s * 8192), lds_w + HB_A2 + s * 8192);
      glds16(dh_rsrc, v256, (uint32_t)(s * 8192), lds_w + HB_DH + s * 8192);
    }
    stage_chunk(0, 0);
    if (nchunks > 1) stage_chunk(1, 1);

    f32x4_t acc[SPW][4];
#pragma unroll
    for (int sl = 0; sl < SPW; ++sl)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[sl][mt][e] = 0.0f;

    const int arow = (ht * 16 + j) * 256;                               // A row of S1 / S2
    const int tswz = hb_swz((tok >> 2) & 3);
    const int srow = tid >> 2, spc = tid & 3;                           // store mapping: (token row of the group, 16-B piece)
    bf16x8_t a2f[REGF ? SPW : 1][4], dhf[REGF ? SPW : 1][4];
    if constexpr (REGF) {
      // the bias window + the a2 / dh tiles have landed (what may still fly: the 3 + 3 DMAs of chunks 0 and 1)
      if (nchunks > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sl = 0; sl < SPW; ++sl) {
        const int s = sp * SPW + sl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int co = ((ks * 4 + g) ^ (tok & 15)) << 4;
          a2f[sl][ks] = *reinterpret_cast<lds_b128_ptr>(L + HB_A2 + (s * 32 + tok) * 256 + co);
          dhf[sl][ks] = *reinterpret_cast<lds_b128_ptr>(L + HB_DH + (s * 32 + tok) * 256 + co);
        }
      }
    }
    for (int c = 0; c < nchunks; ++c) {
      const int buf = c % 3;
      // chunk c has landed; what may still be in flight behind it: chunk c+1 (3 DMAs) and the previous iteration's two
      // 16-B stores of this thread (vmcnt counts stores, in issue order)
      if (c + 1 < nchunks) {
        if (c >= 1) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
      } else {
        if (c >= 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (c + 2 < nchunks) stage_chunk(c + 2, (c + 2) % 3);            // its buffer was read in iteration c-1: free
      lds_byte_ptr s1 = L + buf * HB_WBUF, s2 = s1 + HB_SL, s3 = s1 + 2 * HB_SL;
      {
        bf16x8_t wf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          wf[ks] = *reinterpret_cast<lds_b128_ptr>(s1 + arow + (((ks * 4 + g) ^ j) << 4));
          vf[ks] = *reinterpret_cast<lds_b128_ptr>(s2 + arow + (((ks * 4 + g) ^ j) << 4));
        }
        const f32x4_t bb4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4_t*>(L + HB_B1 + (c * HB_CH + ht * 16 + 4 * g) * 4);
        const float bb[4] = {bb4[0], bb4[1], bb4[2], bb4[3]};
#pragma unroll
        for (int sl = 0; sl < SPW; ++sl) {
          const int s = sp * SPW + sl;
          f32x4_t z, du;
#pragma unroll
          for (int e = 0; e < 4; ++e) z[e] = du[e] = 0.0f;
          lds_byte_ptr ta = L + HB_A2 + (s * 32 + tok) * 256, td = L + HB_DH + (s * 32 + tok) * 256;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t fa, fd;
            if constexpr (REGF) { fa = a2f[sl][ks]; fd = dhf[sl][ks]; }
            else {
              const int co = ((ks * 4 + g) ^ (tok & 15)) << 4;
              fa = *reinterpret_cast<lds_b128_ptr>(ta + co);
              fd = *reinterpret_cast<lds_b128_ptr>(td + co);
            }
            z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], fa, z, 0, 0, 0);
            du = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ks], fd, du, 0, 0, 0);
          }
          // hidden of element e: c*32 + ht*16 + 4g + e, token = tok
          bf16x4_t u4, d4;
#pragma unroll
          for (int e = 0; e < 4; e += 2) {   // two activations per VALU instruction (gelu_fwd_grad2_: bit-identical to the scalar form)
            f32x2_t gz, dgz;
            gelu_fwd_grad2_(f32x2_t{z[e], z[e + 1]} + f32x2_t{bb[e], bb[e + 1]}, gz, dgz);
            const f32x2_t dd = f32x2_t{du[e], du[e + 1]} * dgz;
            u4[e] = f2bf(gz.x); u4[e + 1] = f2bf(gz.y);
            d4[e] = f2bf(dd.x); d4[e + 1] = f2bf(dd.y);
          }
          const int piece = ht * 2 + (g >> 1);
          const int off = (s * 32 + tok) * 64 + ((piece ^ tswz) << 4) + (g & 1) * 8;
          *reinterpret_cast<bf16x4_t*>(smem + HB_U + off) = u4;
          *reinterpret_cast<bf16x4_t*>(smem + HB_DZ + off) = d4;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      {
        // ---- da2^T[k][token] += S3[k][chunk] dz^T : wave = k half ht (4 m-tiles) x token half x sample pair
        bf16x8_t kf[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int kr = (ht * 4 + mt) * 16 + j;
          kf[mt] = *reinterpret_cast<lds_b128_ptr>(s3 + kr * 64 + ((g ^ hb_swz((kr >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int sl = 0; sl < SPW; ++sl) {
          const int s = sp * SPW + sl;
          const bf16x8_t fz = *reinterpret_cast<lds_b128_ptr>(L + HB_DZ + (s * 32 + tok) * 64 + ((g ^ tswz) << 4));
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[sl][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[mt], fz, acc[sl][mt], 0, 0, 0);
        }
        // u / dz of this chunk -> global, one 16-B piece of a 64-B row segment per thread and tensor
        const int po = srow * 64 + ((spc ^ hb_swz((srow >> 2) & 3)) << 4);
        const bf16x8_t vu = *reinterpret_cast<lds_b128_ptr>(L + HB_U + po);
        const bf16x8_t vz = *reinterpret_cast<lds_b128_ptr>(L + HB_DZ + po);
        const size_t go = (row0 + srow) * (size_t)a.M + hbase + c * HB_CH + spc * 8;
        *reinterpret_cast<bf16x8_t*>(a.u + go) = vu;
        *reinterpret_cast<bf16x8_t*>(a.dz + go) = vz;
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- partial tile of da2: k = (ht*4 + mt)*16 + 4g + e (4 consecutive per lane), token = tok
    float* dst = a.part + ((size_t)q * a.rows + row0) * E_DIM;
#pragma unroll
    for (int sl = 0; sl < SPW; ++sl)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        *reinterpret_cast<float4*>(dst + (size_t)((sp * SPW + sl) * 32 + tok) * E_DIM + (ht * 4 + mt) * 16 + 4 * g) =
            make_float4(acc[sl][mt][0], acc[sl][mt][1], acc[sl][mt][2], acc[sl][mt][3]);
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
constexpr int AT_XS = AT_VT + 128 * VT_LD;                // combined input rows [32][128] fp32 (partial-sum input only)
constexpr int AT_ST = AT_XS + S_TOK * E_DIM * 4;          // LN2 row statistics of the output: [4 waves][32 tokens][2]
constexpr int AT_SMEM = AT_ST + 4 * S_TOK * 2 * 4;

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
  // the input as a sum of partial tiles (hidden-split MLP of the layer below): x = ((p0 + p1) + (p2 + p3)), parts
  // h_parts + k * part_stride; h_in is then unused.  h_comb (nullable): x written out (training keeps it)
  const float* h_parts; size_t part_stride; float* h_comb;
  // LayerNorm of the OUTPUT rows (the ln2 of the same encoder layer) -> a2_out bf16 [R][128], or null
  const float* gamma2; const float* beta2; bf16_t* a2_out;
};

// TS instantiation (tuning knob mlp_hs_dbg = 128, tools/attn_phases.py): s_memtime stamps per wave at the phase boundaries, left in
// the first rows of the workgroup's h_out tile instead of the result.
#define AT_TS(i)                                                             \
  if constexpr (TS) {                                                        \
    __builtin_amdgcn_sched_barrier(0);                                       \
    ts[i] = (uint32_t)__builtin_amdgcn_s_memtime();                          \
    __builtin_amdgcn_sched_barrier(0);                                       \
  }
template <int DH, bool TS = false>
__global__ __launch_bounds__(256) void attn_block_fwd_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[AT_SMEM];
  uint32_t ts[16];
  AT_TS(0)
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
  // ---- LayerNorm -> a1 tile.  The input rows are requested FIRST: the phase stamps (tools/attn_phases.py) showed the 24 DMA
  // pieces of Wqkv (~90 cycles of issue apiece, needed only behind barrier 1) in front of them delaying the whole LayerNorm.
  {
    const float2 g2 = *reinterpret_cast<const float2*>(a.gamma + lane * 2);
    const float2 b2v = *reinterpret_cast<const float2*>(a.beta + lane * 2);
    float2 x[8];
    float2 pp[4][8];
    if (a.h_parts) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          pp[k][i] = *reinterpret_cast<const float2*>(a.h_parts + k * a.part_stride + (row0 + w * 8 + i) * E_DIM + lane * 2);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = *reinterpret_cast<const float2*>(a.h_in + (row0 + w * 8 + i) * E_DIM + lane * 2);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 24; ++j) glds16(wq_rsrc, wv, (uint32_t)(j * 4096), lds_w + AT_W + j * 4096);
    __builtin_amdgcn_sched_barrier(0);
    AT_TS(1)
    if (a.h_parts) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        x[i].x = (pp[0][i].x + pp[1][i].x) + (pp[2][i].x + pp[3][i].x);
        x[i].y = (pp[0][i].y + pp[1][i].y) + (pp[2][i].y + pp[3][i].y);
        // 16-byte slot s of row r lives at s ^ (r & 15): the epilogue reads one slot per ROW and lane (512-byte rows: all 16
        // lanes of a ds_read_b128 group would meet on one slot, a 16-way conflict = 240 of the kernel's ~290 conflict cycles per
        // wave, tools/lds_conflicts.py)
        *reinterpret_cast<float2*>(smem + AT_XS + (w * 8 + i) * (E_DIM * 4) + (((lane >> 1) ^ ((w * 8 + i) & 15)) << 4) + (lane & 1) * 8) = x[i];
        if (a.h_comb) *reinterpret_cast<float2*>(a.h_comb + (row0 + w * 8 + i) * E_DIM + lane * 2) = x[i];
      }
    }
    float st[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { st[2 * i] = x[i].x + x[i].y; st[2 * i + 1] = x[i].x * x[i].x + x[i].y * x[i].y; }
    if constexpr (TS) asm volatile("v_mov_b32 %0, %0" : "+v"(st[15]));       // the input rows have arrived
    AT_TS(2)
    // DPP form: 96 ds_bpermute fewer per wave, LN1 phase 4400 -> 3260 cycles (phase stamps); every value is re-read 15 DPP adds
    // behind the instruction that wrote it
    wave_allreduce_sum<16, 1>(st);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = w * 8 + i;
      const float mean = st[2 * i] * (1.0f / E_DIM);
      const float rstd = smd_ln_rstd(st[2 * i + 1] * (1.0f / E_DIM) - mean * mean + LN_EPS);
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
  if constexpr (TS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  AT_TS(3)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  AT_TS(4)
  __syncthreads();
  AT_TS(5)

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
  if constexpr (TS) asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2" : "+v"(cq[0]), "+v"(ck[0]), "+v"(cv[0]));
  AT_TS(6)
  __syncthreads();          // every wave is done reading Wqkv and a1: the regions can be reused
  AT_TS(7)
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
  AT_TS(8)

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
  if constexpr (TS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  AT_TS(9)
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
    if (a.h_parts) res[g] = *reinterpret_cast<const float4*>(smem + AT_XS + l31 * (E_DIM * 4) + (((w * 8 + kh + 2 * g) ^ sw) << 4));
    else res[g] = *reinterpret_cast<const float4*>(a.h_in + (row0 + l31) * E_DIM + w * 32 + 4 * kh + 8 * g);
    bo[g] = *reinterpret_cast<const float4*>(a.b_o + w * 32 + 4 * kh + 8 * g);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // Wo landed (and the loads above)
  AT_TS(10)
  __syncthreads();                                             // o tile complete, Wo visible to every wave
  AT_TS(11)

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
  if constexpr (TS) asm volatile("v_mov_b32 %0, %0" : "+v"(co[0]));
  AT_TS(12)
  float4 ov[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 o;
    o.x = co[4 * g + 0] + bo[g].x + res[g].x;
    o.y = co[4 * g + 1] + bo[g].y + res[g].y;
    o.z = co[4 * g + 2] + bo[g].z + res[g].z;
    o.w = co[4 * g + 3] + bo[g].w + res[g].w;
    *reinterpret_cast<float4*>(a.h_out + (row0 + l31) * E_DIM + w * 32 + 4 * kh + 8 * g) = o;
    ov[g] = o;
  }
  AT_TS(13)
  if (a.a2_out) {
    // ---- ln2 of the output rows: this lane holds 16 of token l31's 128 features; the other 112 sit in the other lane
    // half (xor 32) and in the other three waves (through LDS), summed in a fixed order
    float ps = 0.f, ps2 = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      ps += (ov[g].x + ov[g].y) + (ov[g].z + ov[g].w);
      ps2 += (ov[g].x * ov[g].x + ov[g].y * ov[g].y) + (ov[g].z * ov[g].z + ov[g].w * ov[g].w);
    }
    ps += __shfl_xor(ps, 32, 64);
    ps2 += __shfl_xor(ps2, 32, 64);
    float* stt = reinterpret_cast<float*>(smem + AT_ST);
    if (kh == 0) { stt[(w * S_TOK + l31) * 2] = ps; stt[(w * S_TOK + l31) * 2 + 1] = ps2; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the statistics only: __syncthreads() would also wait for the h_out stores
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);                       // s_barrier is IntrNoMem: keep the stt[] reads below it
    asm volatile("" ::: "memory");
    AT_TS(14)
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { s += stt[(ww * S_TOK + l31) * 2]; s2 += stt[(ww * S_TOK + l31) * 2 + 1]; }
    const float mean = s * (1.0f / E_DIM);
    const float rstd = smd_ln_rstd(s2 * (1.0f / E_DIM) - mean * mean + LN_EPS);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = w * 32 + 4 * kh + 8 * g;
      const float4 gg = *reinterpret_cast<const float4*>(a.gamma2 + f), bb = *reinterpret_cast<const float4*>(a.beta2 + f);
      bf16x4_t t;
      t[0] = f2bf((ov[g].x - mean) * rstd * gg.x + bb.x);
      t[1] = f2bf((ov[g].y - mean) * rstd * gg.y + bb.y);
      t[2] = f2bf((ov[g].z - mean) * rstd * gg.z + bb.z);
      t[3] = f2bf((ov[g].w - mean) * rstd * gg.w + bb.w);
      *reinterpret_cast<bf16x4_t*>(a.a2_out + (row0 + l31) * E_DIM + f) = t;
    }
  }
  if constexpr (TS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AT_TS(15)
    __syncthreads();
    if (lane == 0) {
      uint32_t* d = reinterpret_cast<uint32_t*>(a.h_out + row0 * E_DIM) + w * 16;
#pragma unroll
      for (int i = 0; i < 16; ++i) d[i] = ts[i];
    }
  }
}
#undef AT_TS


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
  const bf16_t* dh_mid;     // [R][128] gradient wrt the half-layer output (bf16); LNF: unused (produced in the kernel)
  const bf16_t* qkv;        // [R][384] saved (q unscaled)
  const bf16_t* Wo;         // out_proj kernel as W [in 128][out 128] bf16 (dgrad operand pack)
  const bf16_t* Wqkv;       // qkv kernel as W [in 128][out 384]
  bf16_t* dqkv;             // [R][384]
  bf16_t* da1;              // [R][128] (LNF: optional)
  // ---- LNF (attn_block_bwd_kernel<DH, true>): the two LayerNorm backwards either side of the attention, in the same launch
  const float* x2;          // [R][128] h_mid, the input of LayerNorm 2 (models/ncsn.py:164)
  const float* parts;       // da2 as the four partial tiles the hidden-split MLP backward left, part_stride floats apart
  size_t part_stride;
  const float* gamma2;
  float* dh;                // [R][128] fp32 residual-stream gradient: READ as the residual term of the LayerNorm-2 backward,
                            // WRITTEN with the gradient that leaves the layer (output of the LayerNorm-1 backward); in place
  bf16_t* dh_mid_out;       // [R][128] bf16 gradient between the two half-layers (operand of out_proj's weight gradient)
  float* partial2;          // [R/32][2][128] dgamma | dbeta partial sums of LayerNorm 2, one slot per sample
  const float* x1;          // [R][128] h, the input of LayerNorm 1 (models/ncsn.py:159)
  const float* gamma1;
  bf16_t* dh_out;           // [R][128] bf16 copy of the gradient that leaves the layer
  float* partial1;          // [R/32][2][128] of LayerNorm 1
};

constexpr int AB_T1_LD = 264;                  // LNF: row stride of the da1 tile handed to the LayerNorm-1 backward (256 B + 8: conflict-free
                                               // 8-byte column writes from the MFMA C layout, 4-byte row reads)

__device__ __forceinline__ bf16x4_t tr_read(unsigned addr) {
  bf16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// Slot swizzle of this kernel's 256-byte-row tiles: 16-byte slot s of row r lives at s ^ ab_swz(r & 15), the two bit pairs of the
// row index SWAPPED.  With the plain s ^ (r & 15) the transposing reads (32 lanes = 4 consecutive rows x 64 bytes) found all four
// rows on the same four slots -- the rows differ in r & 3, which only permuted the low slot bits: a 4-way conflict on each of the
// 24 ds_read_b64_tr_b16 per wave (144 of ~210 conflict cycles, tools/lds_conflicts.py); swapped, r & 3 selects the 64-byte
// quarter.  The ds_read_b128 fragment reads (16 distinct r & 15 per lane group) are conflict-free under any bijection.
__device__ __forceinline__ int ab_swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// LNF ("LayerNorm fused", engine option fused_attn_bwd = 2): the launch also runs the two 128-wide LayerNorm backwards that
// bracket the attention in the backward pass -- until round 5 two launches of their own per layer (ln128_bwd_parts before,
// layernorm_bwd_narrow128 after: 10 + 7 us of a 65 us layer, almost all of it launch / load latency):
//   prologue  dh_mid = LN2-backward((p0 + p1) + (p2 + p3); h_mid, gamma2) + dh   -- wave w owns rows 8w .. 8w+7, a lane two columns,
//             exactly the arithmetic of ln128_bwd_parts_kernel; the result goes to the LDS dh tile (instead of a DMA), to global
//             memory as bf16 (out_proj's weight gradient reads it) and stays in registers as the residual term of the epilogue
//   epilogue  dh    = LN1-backward(bf16(da1); h, gamma1) + dh_mid                -- da1 leaves the accumulators through an LDS tile
//             (MFMA C layout -> row layout), same arithmetic again; fp32 in place + the bf16 copy the next layer reads
// dgamma / dbeta partials: one [2][128] slot per sample and LayerNorm, folded over the four waves through LDS in a fixed order.
template <int DH, bool LNF = false>
__global__ __launch_bounds__(256) void attn_block_bwd_kernel(AttnBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[AB_SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t row0 = (size_t)blockIdx.x * S_TOK;
  const int kh = lane >> 5, l31 = lane & 31, sw = ab_swz(l31 & 15);
  lds_byte_ptr L = (lds_byte_ptr)smem;
  const unsigned L0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);

  // ---- DMA: 256-B-row tiles, 16 rows per round over the 4 waves, source chunk = position ^ ab_swz(row & 15)
  const int rl = w * 4 + (lane >> 4);
  const uint32_t csw = (uint32_t)(((lane & 15) ^ ab_swz(rl)) * 16);
  unsigned char* lds_w = smem + w * 1024;
  const __amdgpu_buffer_rsrc_t wo_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.Wo), 0, 128 * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t wq_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.Wqkv), 0, 128 * 768, 0x00020000);
  const __amdgpu_buffer_rsrc_t dh_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dh_mid + row0 * E_DIM), 0, 32 * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t qkv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.qkv + row0 * 384), 0, 32 * 768, 0x00020000);
#pragma unroll
  for (int j = 0; j < 8; ++j) glds16(wo_rsrc, (uint32_t)rl * 256u + csw, (uint32_t)(j * 4096), lds_w + AB_W + j * 4096);
  if constexpr (!LNF) {
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(dh_rsrc, (uint32_t)rl * 256u + csw, (uint32_t)(j * 4096), lds_w + AB_DH + j * 4096);
  }
#pragma unroll
  for (int pp = 0; pp < 3; ++pp)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      glds16(qkv_rsrc, (uint32_t)rl * 768u + csw, (uint32_t)(pp * 256 + j * 16 * 768), lds_w + AB_QKV + pp * 8192 + j * 4096);
  // ---- LNF prologue: LayerNorm-2 backward on this sample's 32 rows (row layout: wave w rows 8w .. 8w+7, lane = two columns)
  float2 rv2[8], xv1[8], g1v;                       // dh_mid (fp32) and the LayerNorm-1 input rows / scale: live until the epilogue
  if constexpr (LNF) {
    const size_t r0 = row0 + (size_t)w * 8;
    const float2 g2 = *reinterpret_cast<const float2*>(a.gamma2 + lane * 2);
    float2 xv[8], dv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t o = (r0 + i) * E_DIM + lane * 2;
      xv[i] = *reinterpret_cast<const float2*>(a.x2 + o);
      const float2 p0 = *reinterpret_cast<const float2*>(a.parts + o), p1 = *reinterpret_cast<const float2*>(a.parts + a.part_stride + o);
      const float2 p2 = *reinterpret_cast<const float2*>(a.parts + 2 * a.part_stride + o), p3 = *reinterpret_cast<const float2*>(a.parts + 3 * a.part_stride + o);
      dv[i].x = (p0.x + p1.x) + (p2.x + p3.x);
      dv[i].y = (p0.y + p1.y) + (p2.y + p3.y);
      rv2[i] = *reinterpret_cast<const float2*>(a.dh + o);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) xv1[i] = *reinterpret_cast<const float2*>(a.x1 + (r0 + i) * E_DIM + lane * 2);
    g1v = *reinterpret_cast<const float2*>(a.gamma1 + lane * 2);
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
      rv2[i].x = rs[i] * (dxh[i].x - t1 - xh[i].x * t2) + rv2[i].x;
      rv2[i].y = rs[i] * (dxh[i].y - t1 - xh[i].y * t2) + rv2[i].y;
      bf16x2_t t;
      t[0] = f2bf(rv2[i].x); t[1] = f2bf(rv2[i].y);
      *reinterpret_cast<bf16x2_t*>(a.dh_mid_out + (r0 + i) * E_DIM + lane * 2) = t;
      const int r = w * 8 + i;                      // the dh tile as the DMA would have laid it out: 16-byte slot s at s ^ ab_swz(r & 15)
      *reinterpret_cast<bf16x2_t*>(smem + AB_DH + r * 256 + (((lane >> 2) ^ ab_swz(r & 15)) << 4) + (lane & 3) * 4) = t;
    }
    float* red = reinterpret_cast<float*>(smem + AB_DQKV);          // [4 waves][2][128]: the dq | dk | dv tiles are not written before the next barrier but one
    red[(w * 2 + 0) * E_DIM + lane * 2] = Px; red[(w * 2 + 0) * E_DIM + lane * 2 + 1] = Py;
    red[(w * 2 + 1) * E_DIM + lane * 2] = Qx; red[(w * 2 + 1) * E_DIM + lane * 2 + 1] = Qy;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (LNF) {
    const float* red = reinterpret_cast<const float*>(smem + AB_DQKV);
    const int which = tid >> 7, c = tid & 127;
    a.partial2[((size_t)blockIdx.x * 2 + which) * E_DIM + c] =
        (red[(0 * 2 + which) * E_DIM + c] + red[(1 * 2 + which) * E_DIM + c]) + (red[(2 * 2 + which) * E_DIM + c] + red[(3 * 2 + which) * E_DIM + c]);
  }

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
        fa.h[0] = tr_read(L0 + AB_QKV + 8192 + r0 * 256 + (((fcol >> 3) ^ ab_swz(r0 & 15)) << 4) + (fcol & 7) * 2);
        fa.h[1] = tr_read(L0 + AB_QKV + 8192 + r1 * 256 + (((fcol >> 3) ^ ab_swz(r1 & 15)) << 4) + (fcol & 7) * 2);
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
        const int o0 = r0 * 256 + (((fcol >> 3) ^ ab_swz(r0 & 15)) << 4) + (fcol & 7) * 2;
        const int o1 = r1 * 256 + (((fcol >> 3) ^ ab_swz(r1 & 15)) << 4) + (fcol & 7) * 2;
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
      if (!LNF || a.da1) *reinterpret_cast<bf16x4_t*>(a.da1 + (row0 + l31) * E_DIM + w * 32 + 4 * kh + 8 * g) = v;
      // LNF: the q | k | v tiles are dead (every wave has left the attention loop: the barrier above)
      if constexpr (LNF) *reinterpret_cast<bf16x4_t*>(smem + AB_QKV + l31 * AB_T1_LD + (w * 32 + 4 * kh + 8 * g) * 2) = v;
    }
  }
  // ---- LNF epilogue: LayerNorm-1 backward in the row layout of the prologue, residual term = dh_mid still in registers
  if constexpr (LNF) {
    __syncthreads();
    const size_t r0 = row0 + (size_t)w * 8;
    float2 dv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bf16x2_t t = *reinterpret_cast<const bf16x2_t*>(smem + AB_QKV + (w * 8 + i) * AB_T1_LD + lane * 4);
      dv[i].x = bf2f(t[0]); dv[i].y = bf2f(t[1]);
    }
    float st[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { st[2 * i] = xv1[i].x + xv1[i].y; st[2 * i + 1] = xv1[i].x * xv1[i].x + xv1[i].y * xv1[i].y; }
    wave_allreduce_sum<16>(st);
    float Px = 0.f, Py = 0.f, Qx = 0.f, Qy = 0.f;
    float tt[16], rs[8];
    float2 xh[8], dxh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float mean = st[2 * i] * (1.0f / E_DIM);
      rs[i] = smd_ln_rstd(st[2 * i + 1] * (1.0f / E_DIM) - mean * mean + LN_EPS);
      xh[i].x = (xv1[i].x - mean) * rs[i];
      xh[i].y = (xv1[i].y - mean) * rs[i];
      Qx += dv[i].x; Qy += dv[i].y;
      Px += dv[i].x * xh[i].x; Py += dv[i].y * xh[i].y;
      dxh[i].x = dv[i].x * g1v.x; dxh[i].y = dv[i].y * g1v.y;
      tt[2 * i] = dxh[i].x + dxh[i].y;
      tt[2 * i + 1] = dxh[i].x * xh[i].x + dxh[i].y * xh[i].y;
    }
    wave_allreduce_sum<16>(tt);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float t1 = tt[2 * i] * (1.0f / E_DIM), t2 = tt[2 * i + 1] * (1.0f / E_DIM);
      float2 o;
      o.x = rs[i] * (dxh[i].x - t1 - xh[i].x * t2) + rv2[i].x;
      o.y = rs[i] * (dxh[i].y - t1 - xh[i].y * t2) + rv2[i].y;
      const size_t off = (r0 + i) * E_DIM + lane * 2;
      *reinterpret_cast<float2*>(a.dh + off) = o;
      bf16x2_t t;
      t[0] = f2bf(o.x); t[1] = f2bf(o.y);
      *reinterpret_cast<bf16x2_t*>(a.dh_out + off) = t;
    }
    float* red = reinterpret_cast<float*>(smem + AB_DO);            // dO is dead since the attention loop
    red[(w * 2 + 0) * E_DIM + lane * 2] = Px; red[(w * 2 + 0) * E_DIM + lane * 2 + 1] = Py;
    red[(w * 2 + 1) * E_DIM + lane * 2] = Qx; red[(w * 2 + 1) * E_DIM + lane * 2 + 1] = Qy;
    __syncthreads();
    const int which = tid >> 7, c = tid & 127;
    a.partial1[((size_t)blockIdx.x * 2 + which) * E_DIM + c] =
        (red[(0 * 2 + which) * E_DIM + c] + red[(1 * 2 + which) * E_DIM + c]) + (red[(2 * 2 + which) * E_DIM + c] + red[(3 * 2 + which) * E_DIM + c]);
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

int mlp_hs_samples_per_group(int rows, int M) {
  if (rows <= 0 || rows % S_TOK || M % (HS_NQ * CH) || M > MAX_HIDDEN) return 0;
  const int B = rows / S_TOK;
  return B % 4 == 0 ? 4 : (B % 2 == 0 ? 2 : 1);
}

int launch_mlp_block_fwd_hs(const bf16_t* a2, const float* h_res, int rows, const bf16_t* W1t, const float* b1,
                            const bf16_t* W2t, const float* b2, int M, float* part, hipStream_t st) {
  SMD_ARG_CHECK(a2 && h_res && W1t && b1 && W2t && b2 && part, "mlp_block_fwd_hs: null pointer");
  const int ns = mlp_hs_samples_per_group(rows, M);
  SMD_ARG_CHECK(ns > 0, "mlp_block_fwd_hs: rows=%d must be a multiple of 32 and the hidden width %d a multiple of 512 (<= 8192)", rows, M);
  MlpHsArgs a;
  a.a2 = a2; a.h_res = h_res; a.W1t = W1t; a.b1 = b1; a.W2t = W2t; a.b2 = b2; a.M = M; a.part = part; a.rows = rows;
  a.dbg = smd_tuning_get("mlp_hs_dbg");
  const dim3 grid((rows / (S_TOK * ns)) * HS_NQ), block(512);
  if (a.dbg & 64) {                // phase stamps instead of the result (tools/mlp_hs_phases.py)
    SMD_ARG_CHECK(ns == 4 && M == 2048, "mlp_block_fwd_hs: the instrumented instantiation is built for groups of 4 samples, hidden 2048");
    hipLaunchKernelGGL((mlp_hs_fwd_kernel<4, true>), grid, block, 0, st, a);
  } else if (ns == 4 && smd_tuning_get("mlp_variant") != 5) hipLaunchKernelGGL((mlp_hs_fwd_kernel<4, false, 16>), grid, dim3(1024), 0, st, a);
  else if (ns == 4) hipLaunchKernelGGL(mlp_hs_fwd_kernel<4>, grid, block, 0, st, a);     // mlp_variant = 5: the eight-wave form (A/B, tools/mlp_fwd16_ab.py)
  else if (ns == 2) hipLaunchKernelGGL(mlp_hs_fwd_kernel<2>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(mlp_hs_fwd_kernel<1>, grid, block, 0, st, a);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_mlp_block_bwd_hs(const bf16_t* a2, const bf16_t* dh, int rows, const bf16_t* W1t, const bf16_t* W2, const bf16_t* W1,
                            const float* b1, int M, bf16_t* u, bf16_t* dz, float* part, hipStream_t st) {
  SMD_ARG_CHECK(a2 && dh && W1t && W2 && W1 && b1 && u && dz && part, "mlp_block_bwd_hs: null pointer");
  SMD_ARG_CHECK(rows > 0 && rows % (S_TOK * HB_NS) == 0 && M % (HS_NQ * CH) == 0 && M <= MAX_HIDDEN,
                "mlp_block_bwd_hs: rows=%d must be a multiple of 128 and the hidden width %d a multiple of 512 (<= 8192)", rows, M);
  MlpHsBwdArgs a;
  a.a2 = a2; a.dh = dh; a.W1t = W1t; a.W2 = W2; a.W1 = W1; a.b1 = b1; a.M = M; a.u = u; a.dz = dz; a.part = part; a.rows = rows;
  if (smd_tuning_get("mlp_variant") == 9) hipLaunchKernelGGL(mlp_hs_bwd_kernel<false>, dim3((rows / (S_TOK * HB_NS)) * HS_NQ), dim3(512), 0, st, a);   // A/B: fragments re-read per chunk
  else hipLaunchKernelGGL(mlp_hs_bwd_kernel<true>, dim3((rows / (S_TOK * HB_NS)) * HS_NQ), dim3(512), 0, st, a);
  SMD_LAUNCH_CHECK();
  return 0;
}


int launch_attn_block_fwd(const float* h_in, float* h_out, int rows, const float* gamma, const float* beta,
                          const bf16_t* Wqkv_t, const float* b_qkv, const bf16_t* Wo_t, const float* b_o, int num_heads,
                          bf16_t* save_a1, bf16_t* save_qkv, bf16_t* save_o, hipStream_t st, const AttnBlockExtra* ex) {
  SMD_ARG_CHECK((h_in || (ex && ex->h_parts)) && h_out && gamma && beta && Wqkv_t && b_qkv && Wo_t && b_o, "attn_block_fwd: null pointer");
  SMD_ARG_CHECK(rows > 0 && rows % S_TOK == 0, "attn_block_fwd: rows=%d must be a multiple of 32", rows);
  SMD_ARG_CHECK(!ex || !ex->a2_out || (ex->gamma2 && ex->beta2), "attn_block_fwd: a2_out needs the ln2 parameters");
  AttnArgs a;
  a.h_in = h_in; a.h_out = h_out; a.gamma = gamma; a.beta = beta; a.Wqkv_t = Wqkv_t; a.b_qkv = b_qkv; a.Wo_t = Wo_t; a.b_o = b_o;
  a.save_a1 = save_a1; a.save_qkv = save_qkv; a.save_o = save_o;
  a.h_parts = ex ? ex->h_parts : nullptr; a.part_stride = ex ? ex->part_stride : 0; a.h_comb = ex ? ex->h_comb : nullptr;
  a.gamma2 = ex ? ex->gamma2 : nullptr; a.beta2 = ex ? ex->beta2 : nullptr; a.a2_out = ex ? ex->a2_out : nullptr;
  const dim3 grid(rows / S_TOK), block(256);
  const bool stamps = (smd_tuning_get("mlp_hs_dbg") & 128) != 0;       // phase stamps instead of h_out (tools/attn_phases.py)
  switch (num_heads) {
    case 4: hipLaunchKernelGGL(attn_block_fwd_kernel<32>, grid, block, 0, st, a); break;
    case 8:
      if (stamps) hipLaunchKernelGGL((attn_block_fwd_kernel<16, true>), grid, block, 0, st, a);
      else hipLaunchKernelGGL(attn_block_fwd_kernel<16>, grid, block, 0, st, a);
      break;
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
  AttnBwdArgs a = {};
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

int launch_attn_block_bwd_ln(const AttnBwdLnArgs& x, int rows, int num_heads, hipStream_t st) {
  SMD_ARG_CHECK(x.qkv && x.Wo && x.Wqkv && x.dqkv && x.h_mid && x.da2_parts && x.gamma2 && x.dh && x.dh_mid_out && x.partial2 && x.h && x.gamma1 &&
                x.dh_out && x.partial1, "attn_block_bwd_ln: null pointer");
  SMD_ARG_CHECK(rows > 0 && rows % S_TOK == 0, "attn_block_bwd_ln: rows=%d must be a multiple of 32", rows);
  AttnBwdArgs a = {};
  a.dh_mid = nullptr; a.qkv = x.qkv; a.Wo = x.Wo; a.Wqkv = x.Wqkv; a.dqkv = x.dqkv; a.da1 = x.da1;
  a.x2 = x.h_mid; a.parts = x.da2_parts; a.part_stride = x.part_stride; a.gamma2 = x.gamma2; a.dh = x.dh; a.dh_mid_out = x.dh_mid_out;
  a.partial2 = x.partial2; a.x1 = x.h; a.gamma1 = x.gamma1; a.dh_out = x.dh_out; a.partial1 = x.partial1;
  const dim3 grid(rows / S_TOK), block(256);
  switch (num_heads) {
    case 4: hipLaunchKernelGGL((attn_block_bwd_kernel<32, true>), grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL((attn_block_bwd_kernel<16, true>), grid, block, 0, st, a); break;
    case 16: hipLaunchKernelGGL((attn_block_bwd_kernel<8, true>), grid, block, 0, st, a); break;
    default: smd_set_error("attn_block_bwd_ln: num_heads=%d unsupported (4, 8, 16)", num_heads); return -1;
  }
  SMD_LAUNCH_CHECK();
  return 0;
}
