// bf16 MFMA GEMM, "NT" form:  C[M,N] = A[M,K] * Bt[N,K]^T  with a fused epilogue.
//
// Replaces every nn.Dense on the eps-net forward path and every dgrad GEMM of the backward
// (reference models/ncsn.py:53-61,155,161,165-171,178; models/shared.py:65,69).
//
// gfx950 design (cdna_hip_programming.md section 5):
//   * 128x128x64 workgroup tile, 4 waves (2x2), each wave a 64x64 sub-tile = 2x2 MFMA
//     v_mfma_f32_32x32x16_bf16 tiles, fp32 accumulation in 64 accumulator registers.
//   * operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two LDS
//     buffers (2 x 32 KiB), the next K-tile's DMA stays in flight across the barrier behind a
//     counted s_waitcnt vmcnt(8) + raw s_barrier.
//   * LDS rows are 128 B (64 bf16); 16-byte chunk c of row r is stored at chunk c ^ ((r>>1)&7)
//     (swizzle applied on the per-lane *source* address, matching XOR on the ds_read_b128
//     fragment reads) so the 16-lane ds_read_b128 groups hit 16 distinct 16-B bank slots.
//   * XCD-aware bijective workgroup remap so one XCD's L2 sees a contiguous band of M-tiles.
#include "smd_kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;      // A + B

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ void glds16(const bf16_t* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float epilogue_value(float acc, int row, int col, const GemmEpilogue& ep) {
  float v = ep.alpha * acc;
  if (ep.bias) v += ep.bias[col];
  if (ep.pre_bf16) ep.pre_bf16[(size_t)row * ep.ld_pre + col] = f2bf(v);
  if (ep.act == SMD_ACT_GELU) v = geluf_(v);
  else if (ep.act == SMD_ACT_SWISH) v = swishf_(v);
  if (ep.aux_mode != SMD_AUX_NONE) {
    float z = bf2f(ep.aux[(size_t)row * ep.ld_aux + col]);
    v *= (ep.aux_mode == SMD_AUX_GELU_GRAD) ? gelu_gradf_(z) : swish_gradf_(z);
  }
  if (ep.res_f32) {
    int rr = ep.res_row_mod > 0 ? (row % ep.res_row_mod) : row;
    v += ep.res_f32[(size_t)rr * ep.ld_res + col];
  }
  if (ep.res_bf16) v += bf2f(ep.res_bf16[(size_t)row * ep.ld_resb + col]);
  return v;
}

// 32x32 MFMA C layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
__device__ __forceinline__ void store_one(float a, int row, int col, int M, int N, const GemmEpilogue& ep) {
  if (col < N && row < M) {
    const float v = epilogue_value(a, row, col, ep);
    if (ep.out_f32) {
      float* o = ep.out_f32 + (size_t)row * ep.ld_out + col;
      *o = ep.accumulate ? (*o + v) : v;
    }
    if (ep.out_bf16) ep.out_bf16[(size_t)row * ep.ld_outb + col] = f2bf(v);
  }
}
template <int... Es> struct IntSeq {};
typedef IntSeq<0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15> Seq16;
template <int... Es>
__device__ __forceinline__ void store_tile(const f32x16_t& acc, int row0, int col, int M, int N,
                                           const GemmEpilogue& ep, IntSeq<Es...>) {
  (store_one(acc[Es], row0 + (Es & 3) + 8 * (Es >> 2), col, M, N, ep), ...);
}

__global__ __launch_bounds__(256) void gemm_nt_128x128_kernel(
    const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ Bt, int ldb, int M, int N,
    int K, int tiles_n, int nwg, GemmEpilogue ep) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF_BYTES];

  // ---- XCD-aware bijective remap (block b runs on XCD b % 8; give each XCD a contiguous band)
  const int bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = swz / tiles_n, tn = swz - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;

  // ---- per-lane DMA source pointers: wave w, piece j covers LDS rows (w*4+j)*8 .. +8
  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (w * 4 + j) * 8 + (lane >> 3);
    const int gk = ((lane & 7) ^ ((row >> 1) & 7)) * 8;       // source chunk for LDS chunk lane&7
    int ga = m0 + row; ga = ga < M ? ga : M - 1;
    int gb = n0 + row; gb = gb < N ? gb : N - 1;
    a_src[j] = A + (size_t)ga * lda + gk;
    b_src[j] = Bt + (size_t)gb * ldb + gk;
  }

  auto issue_tile = [&](int kt, int buf) {
    unsigned char* base = smem + buf * BUF_BYTES + w * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(a_src[j] + kt * BK, base + j * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(b_src[j] + kt * BK, base + TILE_BYTES + j * 1024);
  };

  // ---- fragment read offsets (bytes) within an operand tile
  const int fsw = (lane >> 1) & 7;        // == (row>>1)&7 for row = 32*k + (lane&31)
  const int kh = lane >> 5;               // which 8-wide k half of a 16-wide MFMA k-step
  int a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_off[i] = (wr * 64 + i * 32 + (lane & 31)) * 128;
    b_off[i] = TILE_BYTES + (wc * 64 + i * 32 + (lane & 31)) * 128;
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int nk = K / BK;
  issue_tile(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {
      issue_tile(kt + 1, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile kt landed, tile kt+1 in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* tb = smem + buf * BUF_BYTES;
    // fragment reads for k-step ks+1 are issued ahead of the MFMAs of k-step ks
    bf16x8_t af[2][2], bfr[2][2];
    {
      const int coff = (kh ^ fsw) << 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[0][i] = *reinterpret_cast<const bf16x8_t*>(tb + a_off[i] + coff);
        bfr[0][i] = *reinterpret_cast<const bf16x8_t*>(tb + b_off[i] + coff);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) {
        const int coff = (((ks + 1) * 2 + kh) ^ fsw) << 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8_t*>(tb + a_off[i] + coff);
          bfr[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8_t*>(tb + b_off[i] + coff);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bfr[ks & 1][j],
                                                              acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                          // all waves done with buf before re-fill
  }

  // ---- epilogue (store_tile is expanded with compile-time accumulator indices)
  store_tile(acc[0][0], m0 + wr * 64 + 4 * kh, n0 + wc * 64 + (lane & 31), M, N, ep, Seq16{});
  store_tile(acc[0][1], m0 + wr * 64 + 4 * kh, n0 + wc * 64 + 32 + (lane & 31), M, N, ep, Seq16{});
  store_tile(acc[1][0], m0 + wr * 64 + 32 + 4 * kh, n0 + wc * 64 + (lane & 31), M, N, ep, Seq16{});
  store_tile(acc[1][1], m0 + wr * 64 + 32 + 4 * kh, n0 + wc * 64 + 32 + (lane & 31), M, N, ep, Seq16{});
}

}  // namespace

int launch_gemm_nt(const bf16_t* A, int lda, const bf16_t* Bt, int ldb, int M, int N, int K,
                   const GemmEpilogue& ep, hipStream_t st) {
  SMD_ARG_CHECK(A && Bt, "gemm_nt: null operand");
  SMD_ARG_CHECK(M > 0 && N > 0 && K > 0, "gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
  SMD_ARG_CHECK(K % BK == 0, "gemm_nt: K=%d must be a multiple of %d (pad the operands)", K, BK);
  SMD_ARG_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K,
                "gemm_nt: lda=%d ldb=%d must be >=K and multiples of 8", lda, ldb);
  SMD_ARG_CHECK(ep.out_f32 || ep.out_bf16 || ep.pre_bf16, "gemm_nt: no output");
  SMD_ARG_CHECK(ep.aux_mode == SMD_AUX_NONE || ep.aux, "gemm_nt: aux_mode without aux");
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  hipLaunchKernelGGL(gemm_nt_128x128_kernel, dim3(nwg), dim3(256), 0, st, A, lda, Bt, ldb, M, N, K,
                     tiles_n, nwg, ep);
  SMD_LAUNCH_CHECK();
  return 0;
}
