// bf16 MFMA GEMM, "NT" form:  C[M,N] = A[M,K] * Bt[N,K]^T  with a fused epilogue.
//
// Replaces every nn.Dense on the eps-net forward path and every dgrad GEMM of the backward
// (reference models/ncsn.py:53-61,155,161,165-171,178; models/shared.py:65,69).
//
// gfx950 design (cdna_hip_programming.md section 5):
//   * BM x 128 x 64 workgroup tile, BM in {128, 64, 32} chosen per launch so that skinny outputs
//     (N = 128 / 384, the transformer encoder) still put >= 256 workgroups on the 256 CUs;
//     4 waves, v_mfma_f32_32x32x16_bf16, fp32 accumulation.
//   * operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two LDS
//     buffers, the next K-tile's DMA stays in flight across the barrier behind a counted
//     s_waitcnt vmcnt(N) + raw s_barrier.
//   * LDS rows are 128 B (64 bf16); 16-byte chunk c of row r is stored at chunk c ^ ((r>>1)&7)
//     (swizzle applied on the per-lane *source* address, matching XOR on the ds_read_b128
//     fragment reads) so the 16-lane ds_read_b128 groups hit 16 distinct 16-B bank slots.
//   * epilogue: the fp32 accumulator tile is staged through the (now free) LDS so that every lane
//     handles 4 consecutive columns of one row: 16-byte bias/residual loads and 16-/8-byte
//     stores, 512 B contiguous per 32 lanes (the MFMA C layout alone gives 2-byte scattered
//     stores, which made the K=128 GEMMs store-bound).
//   * XCD-aware bijective workgroup remap so one XCD's L2 sees a contiguous band of M-tiles.
#include "smd_kernels.h"
#include "gemm_epilogue.h"
#include <string.h>

namespace {

constexpr int BN = 128, BK = 64;
constexpr int STAGE_LD = 132;   // floats per staged row (128 + 4 pad: 528 B, 16-B aligned)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ void glds16(const bf16_t* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

template <int N_> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <int... Es> struct IntSeq {};
typedef IntSeq<0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15> Seq16;
// 32x32 MFMA C layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
template <int... Es>
__device__ __forceinline__ void stage_tile(const f32x16_t& acc, float* stage, int row0, int col, IntSeq<Es...>) {
  ((stage[(row0 + (Es & 3) + 8 * (Es >> 2)) * STAGE_LD + col] = acc[Es]), ...);
}
template <int... Es>
__device__ __forceinline__ void stage_tile_add(const f32x16_t& acc, float* stage, int row0, int col, IntSeq<Es...>) {
  ((stage[(row0 + (Es & 3) + 8 * (Es >> 2)) * STAGE_LD + col] += acc[Es]), ...);
}

// NS = number of LDS K-tile buffers.  2: one tile prefetched ahead (MFMA-bound shapes, 64 KiB at BM = 128).
// 4 (BM <= 64, K >= 512): three tiles in flight -- the skinny-output GEMMs of the 128-wide encoder run one
// workgroup per CU with 4 MFMAs per wave per K-tile, so a 2-deep pipeline is bound by the DMA latency.
// KG = K-groups inside the workgroup (1 or 2): with KG = 2 the workgroup has 8 waves, wave group kg stages and
// multiplies the K-tiles kt*2 + kg in its own LDS ring, both groups share the barriers, and the two partial tiles
// are added in the epilogue's LDS stage (fixed order).  For the skinny-output GEMMs (one 32-row tile per CU) that is
// twice the DMA in flight and two waves per SIMD for the same output tile.
template <int BM, int NS, int KG = 1>
__global__ __launch_bounds__(256 * KG) void gemm_nt_kernel(const bf16_t* __restrict__ A, int lda,
                                                      const bf16_t* __restrict__ Bt, int ldb, int M, int N, int K,
                                                      int tiles_n, int nwg, int vec_epilogue, GemmEpilogue ep) {
  constexpr int WM = BM >= 64 ? 2 : 1;            // waves along M
  constexpr int WN = 4 / WM;                      // waves along N
  constexpr int WTM = BM / WM, WTN = BN / WN;     // wave tile
  constexpr int MT = WTM / 32, NT = WTN / 32;     // MFMA tiles per wave
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int BUF_BYTES = A_BYTES + B_BYTES;
  constexpr int A_PIECES = BM / 32;               // 1-KiB DMA pieces per wave per K-tile
  constexpr int SROWS = BM < 64 ? BM : 64;        // rows staged per epilogue pass
  constexpr int STAGE_BYTES = SROWS * STAGE_LD * 4;
  constexpr int SMEM_BYTES = NS * KG * BUF_BYTES > STAGE_BYTES ? NS * KG * BUF_BYTES : STAGE_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

  // ---- XCD-aware bijective remap (block b runs on XCD b % 8; give each XCD a contiguous band)
  const int bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = swz / tiles_n, tn = swz - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = KG == 1 ? 0 : w8 >> 2;           // K-group of this wave
  const int w = KG == 1 ? w8 : (w8 & 3);          // wave index inside its group
  const int wr = w / WN, wc = w % WN;

  // ---- per-lane DMA source pointers. A: wave w piece j covers LDS rows (w*A_PIECES+j)*8 .. +8,
  //      B: rows (w*4+j)*8 .. +8 ; 8 lanes per 128-B row, source chunk = lane&7 ^ ((row>>1)&7)
  const bf16_t* a_src[A_PIECES];
  const bf16_t* b_src[4];
#pragma unroll
  for (int j = 0; j < A_PIECES; ++j) {
    const int row = (w * A_PIECES + j) * 8 + (lane >> 3);
    const int gk = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    int ga = m0 + row; ga = ga < M ? ga : M - 1;
    a_src[j] = A + (size_t)ga * lda + gk;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (w * 4 + j) * 8 + (lane >> 3);
    const int gk = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    int gb = n0 + row; gb = gb < N ? gb : N - 1;
    b_src[j] = Bt + (size_t)gb * ldb + gk;
  }

  auto issue_tile = [&](int kt, int buf) {        // step kt of this K-group = K-tile kt*KG + kg
    unsigned char* base = smem + (buf * KG + kg) * BUF_BYTES;
    const int ko = (kt * KG + kg) * BK;
#pragma unroll
    for (int j = 0; j < A_PIECES; ++j) glds16(a_src[j] + ko, base + (w * A_PIECES + j) * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(b_src[j] + ko, base + A_BYTES + (w * 4 + j) * 1024);
  };

  // ---- fragment read offsets (bytes) within a buffer
  const int fsw = (lane >> 1) & 7;        // == (row>>1)&7 for row = 32*k + (lane&31)
  const int kh = lane >> 5;               // which 8-wide k half of a 16-wide MFMA k-step
  int a_off[MT], b_off[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) a_off[i] = (wr * WTM + i * 32 + (lane & 31)) * 128;
#pragma unroll
  for (int i = 0; i < NT; ++i) b_off[i] = A_BYTES + (wc * WTN + i * 32 + (lane & 31)) * 128;

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int nk = K / (BK * KG);
  auto compute_tile = [&](int buf) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* tb = smem + (buf * KG + kg) * BUF_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int coff = ((ks * 2 + kh) ^ fsw) << 4;
      bf16x8_t af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(tb + a_off[i] + coff);
#pragma unroll
      for (int i = 0; i < NT; ++i) bfr[i] = *reinterpret_cast<const bf16x8_t*>(tb + b_off[i] + coff);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();        // all waves done with buf before it is re-filled / re-used
  };
  if constexpr (NS == 2) {
    issue_tile(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) {
        issue_tile(kt + 1, buf ^ 1);
        wait_vmcnt<A_PIECES + 4>();        // tile kt landed, tile kt+1 stays in flight
      } else {
        wait_vmcnt<0>();
      }
      compute_tile(buf);
    }
  } else {
    // NS - 1 tiles in flight.  Past the end the last tile is re-requested into a buffer nobody reads any more,
    // which keeps the vmcnt arithmetic uniform; the stragglers are drained before the epilogue reuses the LDS.
#pragma unroll
    for (int s2 = 0; s2 < NS - 1; ++s2) issue_tile(s2 < nk ? s2 : nk - 1, s2);
    int rd = 0, wr = NS - 1;
    for (int kt = 0; kt < nk; ++kt) {
      const int nxt = kt + NS - 1;
      issue_tile(nxt < nk ? nxt : nk - 1, wr);
      wait_vmcnt<(NS - 1) * (A_PIECES + 4)>();
      compute_tile(rd);
      rd = rd + 1 == NS ? 0 : rd + 1;
      wr = wr + 1 == NS ? 0 : wr + 1;
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue through LDS: SROWS rows per pass, every lane owns 4 consecutive columns
  float* stage = reinterpret_cast<float*>(smem);
  constexpr int PASSES = BM / SROWS;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    if (p > 0) __syncthreads();
    // rows [p*SROWS, (p+1)*SROWS) of the block tile: wave rows wr*WTM .. +WTM
    if (kg == 0 && wr * WTM >= p * SROWS && wr * WTM < (p + 1) * SROWS) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          stage_tile(acc[i][j], stage, wr * WTM - p * SROWS + i * 32 + 4 * kh, wc * WTN + j * 32 + (lane & 31),
                     Seq16{});
    }
    __syncthreads();
    if constexpr (KG == 2) {              // the second K-group adds its partial tile (same lane -> element mapping)
      if (kg == 1 && wr * WTM >= p * SROWS && wr * WTM < (p + 1) * SROWS) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            stage_tile_add(acc[i][j], stage, wr * WTM - p * SROWS + i * 32 + 4 * kh, wc * WTN + j * 32 + (lane & 31),
                           Seq16{});
      }
      __syncthreads();
    }
    const int c4 = (tid & 31) * 4;
    const int col = n0 + c4;
#pragma unroll
    for (int i = 0; i < SROWS / (8 * KG); ++i) {
      const int rs = (tid >> 5) + 8 * KG * i;
      const int row = m0 + p * SROWS + rs;
      if (row < M && col < N) {
        const float4 a4 = *reinterpret_cast<const float4*>(stage + rs * STAGE_LD + c4);
        smd_epi::epilogue_quad(a4, row, col, N, vec_epilogue && (col + 3 < N), ep);
      }
    }
  }
}

template <int BM, int NS, int KG = 1>
void launch_bm(const bf16_t* A, int lda, const bf16_t* Bt, int ldb, int M, int N, int K, int vec, const GemmEpilogue& ep,
               hipStream_t st) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  hipLaunchKernelGGL((gemm_nt_kernel<BM, NS, KG>), dim3(nwg), dim3(256 * KG), 0, st, A, lda, Bt, ldb, M, N, K, tiles_n, nwg, vec, ep);
}

}  // namespace

// process-wide kernel-selection knobs (smd_set_tuning in the C-ABI)
namespace {
struct Knob { const char* key; int value; };
Knob g_knobs[] = {{"gemm_nt256", 1}, {"gemm_nt256_variant", 0}, {"gemm_nt256_pk", 1}, {"gemm_tn256", 1}, {"ln_bwd_wide", 2}, {"ln_bwd_narrow", 1}, {"gemm_nt_deep", 1}, {"mlp_variant", 0}, {"tn128_target_wgs", 512}, {"gemm_tn_deep", 0}, {"ln_fwd_wide", 3}, {"gemm_nt_kg", 1}, {"mlp_hs_dbg", 0}, {"ln_excl", 0}, {"tn_exclusive_cu", 2}, {"tn_split_model", 1}, {"tn128_loader_waves", 1}, {"tn_mode", 0}, {"gemm_nt_form", 0}, {"gemm_nt_form_wk", 0}};
}
int smd_tuning_set(const char* key, int value) {
  for (Knob& k : g_knobs)
    if (key && !strcmp(key, k.key)) { k.value = value; return 0; }
  smd_set_error("smd_set_tuning: unknown key '%s'", key ? key : "(null)");
  return -1;
}
int smd_tuning_get(const char* key) {
  for (const Knob& k : g_knobs)
    if (key && !strcmp(key, k.key)) return k.value;
  return -1;
}

int launch_gemm_nt(const bf16_t* A, int lda, const bf16_t* Bt, int ldb, int M, int N, int K,
                   const GemmEpilogue& ep, hipStream_t st) {
  SMD_ARG_CHECK(A && Bt, "gemm_nt: null operand");
  SMD_ARG_CHECK(M > 0 && N > 0 && K > 0, "gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
  SMD_ARG_CHECK(K % BK == 0, "gemm_nt: K=%d must be a multiple of %d (pad the operands)", K, BK);
  SMD_ARG_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K,
                "gemm_nt: lda=%d ldb=%d must be >=K and multiples of 8", lda, ldb);
  SMD_ARG_CHECK(ep.out_f32 || ep.out_bf16 || ep.pre_bf16, "gemm_nt: no output");
  SMD_ARG_CHECK(ep.aux_mode == SMD_AUX_NONE || ep.aux, "gemm_nt: aux_mode without aux");
  if (gemm_nt256_eligible(M, N, K, ep)) return launch_gemm_nt256(A, lda, Bt, ldb, M, N, K, ep, st);
  const int vec = smd_epi::vec_ok(ep);
  // tile height: keep >= ~256 workgroups on the chip when the output is skinny
  const int tiles_n = (N + BN - 1) / BN;
  const long wg128 = (long)((M + 127) / 128) * tiles_n;
  const long wg64 = (long)((M + 63) / 64) * tiles_n;
  const bool deep = K >= 8 * BK && smd_tuning_get("gemm_nt_deep");
  // measurement knob (tools/gemm_nt_forms_ab.py): force one tile form for the shapes that have a choice
  // ("gemm_nt_form_wk": the same for the wide-K, few-column shapes only -- out_proj: N <= 512, K >= 2048 -- so that an in-step A/B
  // of that one GEMM leaves every other launch on its default form)
  const int form_wk = (M > 64 && N <= 512 && K >= 2048) ? smd_tuning_get("gemm_nt_form_wk") : 0;
  const int form = form_wk ? form_wk : (M > 64 ? smd_tuning_get("gemm_nt_form") : 0);
  if (form == 1) launch_bm<64, 2>(A, lda, Bt, ldb, M, N, K, vec, ep, st);
  else if (form == 2) launch_bm<128, 2>(A, lda, Bt, ldb, M, N, K, vec, ep, st);
  else if (form == 3 && K % (2 * BK) == 0) launch_bm<128, 2, 2>(A, lda, Bt, ldb, M, N, K, vec, ep, st);
  else if (form == 4 && K % (2 * BK) == 0) launch_bm<64, 2, 2>(A, lda, Bt, ldb, M, N, K, vec, ep, st);
  else if (form == 5) launch_bm<128, 3>(A, lda, Bt, ldb, M, N, K, vec, ep, st);
  else if (form == 6 && K % (2 * BK) == 0) launch_bm<64, 3, 2>(A, lda, Bt, ldb, M, N, K, vec, ep, st);
  else if (N <= 512 && K >= 2048 && K % (2 * BK) == 0 && M >= 2048 && smd_tuning_get("gemm_nt_kg") && smd_tuning_get("gemm_nt_form_wk") == 0) {
    // out_proj (models/ncsn.py:177-178: rows x 2048 -> 512): 128-row tiles with two K-groups of four waves -- half the operand
    // traffic per flop of the 64-row form and two waves per SIMD on one tile.  In-step A/B (profiles/r6d_schedule_and_out_proj_form_ab.txt):
    // sample step +2 ... +3 % (1849 / 1855 -> 1914 / 1885 steps/s), train step unchanged
    launch_bm<128, 2, 2>(A, lda, Bt, ldb, M, N, K, vec, ep, st);
  } else if (M <= 32 || !(wg128 >= 512 || wg64 >= 256 || M <= 64)) {
    // K >= 1024: two K-groups of four waves per workgroup (A/B: 8-18 % over one group with a 4-deep ring; deeper rings
    // -- 7 stages, or 2 groups x 4 stages -- gain nothing: the step time follows the LDS-DMA landing cadence)
    if (deep && K >= 16 * BK && K % (2 * BK) == 0 && smd_tuning_get("gemm_nt_kg")) launch_bm<32, 3, 2>(A, lda, Bt, ldb, M, N, K, vec, ep, st);
    else if (deep) launch_bm<32, 4>(A, lda, Bt, ldb, M, N, K, vec, ep, st);
    else launch_bm<32, 2>(A, lda, Bt, ldb, M, N, K, vec, ep, st);
  } else if (wg128 >= 512) {
    launch_bm<128, 2>(A, lda, Bt, ldb, M, N, K, vec, ep, st);
  } else if (wg64 <= 256 && K >= 16 * BK && K % (2 * BK) == 0 && smd_tuning_get("gemm_nt_kg")) {
    // exactly one 64-row workgroup per CU and a long K (out_proj of one sampler chain, 4096 x 2048 -> 512; out_proj of the
    // C = 146 network): two K-groups of four waves = two waves per SIMD on the same tile, 19.6 -> 16.5 us and 18.9 -> 15.8 us
    // (profiles/r4w_gemm_nt_forms.txt); with two workgroups per CU (8192 x 2048 -> 512) the one-group form is the faster one
    launch_bm<64, 2, 2>(A, lda, Bt, ldb, M, N, K, vec, ep, st);
  } else {
    launch_bm<64, 2>(A, lda, Bt, ldb, M, N, K, vec, ep, st);   // 4 stages = 96 KiB: 1 workgroup per CU instead of 3, slower (A/B)
  }
  SMD_LAUNCH_CHECK();
  return 0;
}
