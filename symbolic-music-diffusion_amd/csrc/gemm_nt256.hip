// bf16 MFMA GEMM, "NT" form, 256 x 256 x 64 workgroup tile with an 8-phase software pipeline:
//   C[M,N] = A[M,K] * Bt[N,K]^T  + the fused epilogue of gemm_epilogue.h.
//
// This is the kernel for the MFMA-bound GEMMs of the eps-net: the two 2048x2048 Dense layers of every
// DenseResBlock (reference models/shared.py:65,69 -> 77 % of the forward flops) and their dgrads.
// Skinny / ragged shapes stay on gemm_nt.hip.
//
// gfx950 design (cdna_hip_programming.md section 5, "256^2 8-phase"):
//   * 512 threads = 8 waves as 2 (M) x 4 (N); every wave owns a contiguous 128 x 64 block of C, held as
//     2 x 2 quadrants of 64 x 32 (two 32x32 MFMA tiles each) = 128 accumulator registers.
//   * A K-tile (64 deep) is four 16-KiB half-tiles  B0 A0 B1 A1  (half h of A = the rows every wave needs
//     for its quadrant row h; same for B), DMA'd HBM/L2 -> LDS with global_load_lds_dwordx4.  Two K-tile
//     buffers = 128 KiB LDS, one workgroup per CU, two waves per SIMD.
//   * One K-tile = 4 phases, one per C quadrant: {ds_read the fragments this quadrant still needs, issue ONE
//     half-tile DMA for a later K-tile, s_barrier, 8 MFMA 32x32x16 under s_setprio(1), s_barrier}.  The two
//     wave rows run one barrier apart, so on every SIMD one wave is in its MFMA half-phase while the other
//     reads LDS / issues DMA.  DMAs are retired only by a counted s_waitcnt vmcnt(6) once per K-tile: three
//     half-tiles stay in flight across every barrier; vmcnt never reaches 0 inside the loop.
//   * Hazards are closed by construction, not by timing: a half-tile is read one phase after the
//     (vmcnt, barrier) that retires it, and re-staged only after a barrier that follows the lgkmcnt wait
//     that retired its reads (order B0 A0 B1 A1 = consumption order).
//   * LDS rows are 128 B; 16-byte chunk c of row r sits at chunk c ^ ((r>>1)&7): the swizzle is applied on
//     the per-lane DMA *source* address and again on the ds_read_b128 address (same involution), which makes
//     the 16-lane read groups of the 32x32x16 A/B fragments conflict-free.
//   * Epilogue: each wave stages its accumulators through its own slice of the (now free) LDS so that lanes
//     own 4 consecutive columns: 16-byte loads of bias/residual, 128..256-byte contiguous row stores.
//   * Bijective XCD-aware workgroup remap (block b runs on XCD b % 8): each XCD's L2 sees a band of M-tiles.
#include "smd_kernels.h"
#include "gemm_epilogue.h"

#include <type_traits>

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int HALF_BYTES = 128 * TK * 2;          // 16 KiB: 128 rows x 64 bf16
constexpr int KT_BYTES = 4 * HALF_BYTES;          // B0 A0 B1 A1
constexpr int SMEM_BYTES = 2 * KT_BYTES;          // 128 KiB
constexpr int OFF_B0 = 0, OFF_A0 = HALF_BYTES, OFF_B1 = 2 * HALF_BYTES, OFF_A1 = 3 * HALF_BYTES;
constexpr int SLD = 68;                           // staged epilogue row: 64 floats + 4 pad (272 B)
constexpr int WAVE_STAGE_BYTES = 32 * SLD * 4;    // 8704 B per wave

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// LDS-DMA of 16 B per lane: source = buffer descriptor base + per-lane voffset + wave-uniform soffset,
// destination = wave-uniform LDS base + lane*16.
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff,
                                       unsigned char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_wave_base, 16, voff, soff, 0, 0);
}

#define SMD_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define SMD_LGKMCNT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define SMD_PIN() __builtin_amdgcn_sched_barrier(0)
#define SMD_BAR() __builtin_amdgcn_s_barrier()

typedef const __attribute__((address_space(3))) bf16x8_t* lds_frag_ptr;
typedef const __attribute__((address_space(3))) unsigned char* lds_byte_ptr;

struct Frags {
  bf16x8_t a[2][4];    // [m-tile][k-step] of the current A half
  bf16x8_t b0[4], b1[4];
};
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
struct Frags8 {        // e4m3 operands: one 8-register tuple per 64-wide step = the two 16-byte chunks of its two K-blocks
  i32x8_t a[2][2];     // [m-tile][64-wide step]
  i32x8_t b0[2], b1[2];
};

// `ad[ks]` = LDS address of the lane's chunk for k-step ks in the wave's first row block of a K-tile buffer;
// `off` (half-tile offset) and the m-tile stride fold into the ds_read immediate (< 64 KiB).
__device__ __forceinline__ void read_a(Frags& f, const lds_byte_ptr (&ad)[4], int off) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) f.a[mt][ks] = *reinterpret_cast<lds_frag_ptr>(ad[ks] + off + mt * 32 * 128);
}
__device__ __forceinline__ void read_b(bf16x8_t (&b)[4], const lds_byte_ptr (&ad)[4], int off) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) b[ks] = *reinterpret_cast<lds_frag_ptr>(ad[ks] + off);
}
// e4m3: ad[ks] / ad[ks + 2] (ks = 0, 1) = addresses of the two 16-byte chunks (logical chunk L and L + 1) of the lane's 32
// bytes of 64-wide step ks -- swizzled independently over all eight chunk positions of the row like the bf16 kernel's
// (a pair-preserving 4-position swizzle made every ds_read_b128 a 2-way bank conflict: 46 % of the LDS cycles, PMC).
// Read as two bf16x8-typed 16-byte loads like the bf16 kernel's (with an int-typed 32-byte load hipcc orders every ds_read behind ALL outstanding
// LDS-DMA -- s_waitcnt vmcnt(0) in every phase, which serialises the pipeline) and joined in registers.
__device__ __forceinline__ i32x8_t join8(const bf16x8_t lo, const bf16x8_t hi) {
  return __builtin_shufflevector(__builtin_bit_cast(i32x4_t, lo), __builtin_bit_cast(i32x4_t, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void read_a(Frags8& f, const lds_byte_ptr (&ad)[4], int off) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      f.a[mt][ks] = join8(*reinterpret_cast<lds_frag_ptr>(ad[ks] + off + mt * 32 * 128), *reinterpret_cast<lds_frag_ptr>(ad[ks + 2] + off + mt * 32 * 128));
}
__device__ __forceinline__ void read_b(i32x8_t (&b)[2], const lds_byte_ptr (&ad)[4], int off) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) b[ks] = join8(*reinterpret_cast<lds_frag_ptr>(ad[ks] + off), *reinterpret_cast<lds_frag_ptr>(ad[ks + 2] + off));
}
template <int V>
__device__ __forceinline__ void mma_quadrant(f32x16_t (&acc)[2], const bf16x8_t (&a)[2][4], const bf16x8_t (&b)[4],
                                             const uint32_t (&)[2], uint32_t) {
  if constexpr (!(V & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][ks], b[ks], acc[mt], 0, 0, 0);
  if constexpr (!(V & 2)) __builtin_amdgcn_s_setprio(0);
}
template <int V>
__device__ __forceinline__ void mma_quadrant(f32x16_t (&acc)[2], const i32x8_t (&a)[2][2], const i32x8_t (&b)[2],
                                             const uint32_t (&sa)[2], uint32_t sb) {
  if constexpr (!(V & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      acc[mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[mt][ks], b[ks], acc[mt], 0, 0, 0, (int)sa[mt], 0, (int)sb);
  // the MFMAs are side-effect free for the optimiser, which otherwise sinks the last K-tiles' chains below their phase
  // barriers (and spills the fragments to get there): make the accumulators opaque right here
  asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
  if constexpr (!(V & 2)) __builtin_amdgcn_s_setprio(0);
}

template <int... Es> struct IntSeq {};
typedef IntSeq<0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15> Seq16;
// 32x32 MFMA C layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
template <int... Es>
__device__ __forceinline__ void stage_tile(const f32x16_t& acc, float* stage, int row0, int col, IntSeq<Es...>) {
  ((stage[(row0 + (Es & 3) + 8 * (Es >> 2)) * SLD + col] = acc[Es]), ...);
}

// V: schedule variants for A/B runs (bit 0: no explicit lgkmcnt(0) after the phase barrier -- the compiler's own
// counted waits sit between the MFMAs; bit 1: no s_setprio around the MFMA clusters).
// F8: the operands are OCP e4m3 bytes with one power-of-two (E8M0) scale per ROW of A and of Bt (sa[M], sb[N], byte 0 of
// a dword): the same 128-byte LDS rows then hold 128 K-elements, a K-tile is two v_mfma_scale_f32_32x32x64_f8f6f4 steps
// per 32x32 tile instead of four 32x32x16 bf16 steps (2x the MFMA rate, half the operand bytes per flop), and the
// hardware applies 2^(sa + sb - 254) to every product.  Operand layout of the scaled MFMA as measured on gfx950
// (tools/fp8_probe.hip): VGPRs 0-3 of a lane belong to K-block 0 of the 64-wide step, VGPRs 4-7 to K-block 1, the lane
// halves split each block's 32 elements 16 / 16; the scale of block b of row r is taken from lane r + 32 b.  With one
// scale per ROW both blocks carry the same exponent, so only the byte-for-byte pairing of the A and B fragments matters:
// lane half kh simply takes the 32 contiguous bytes [64 ks + 32 kh, +32) of its row (one 8-register LDS load).
template <int V, bool F8>
__global__ __launch_bounds__(512) void gemm_nt256_kernel(const void* __restrict__ Av, int lda,
                                                         const void* __restrict__ Btv, int ldb, int M, int N, int K,
                                                         int tiles_n, int nwg, GemmEpilogue ep,
                                                         const uint32_t* __restrict__ scale_a, const uint32_t* __restrict__ scale_b, int pk_epi) {
  constexpr int ESZ = F8 ? 1 : 2;                 // bytes per operand element
  const unsigned char* A = reinterpret_cast<const unsigned char*>(Av);
  const unsigned char* Bt = reinterpret_cast<const unsigned char*>(Btv);
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

  const int bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = swz / tiles_n, tn = swz - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;

  // ---- DMA sources.  Round j (0/1) of half h: wave w fills LDS rows j*64 + w*8 .. +8 of the half-tile (1 KiB),
  // 8 lanes per 128-B row; the lane's source chunk is its LDS chunk ^ ((row>>1)&7).
  //   A half h, LDS row r  <->  global row m0 + (r/64)*128 + h*64 + r%64     (wave row wr reads r in [wr*64, +64))
  //   B half h, LDS row r  <->  global row n0 + (r/32)*64  + h*32 + r%32     (wave col wc reads r in [wc*32, +32))
  // Source address = buffer descriptor over the tile's 256-row band + one 32-bit per-lane byte offset (VGPR) + a
  // wave-uniform byte offset (SGPR): no 64-bit per-lane pointers in the loop.
  const int lrow = lane >> 3;
  // (e4m3 uses the same 16-byte swizzle: a lane's fragment is two logical chunks L, L + 1, fetched from wherever each lies)
  const int sw_src = ((w & 1) * 4 + (lrow >> 1)) & 7;
  const int gkb = ((lane & 7) ^ sw_src) * 16;                       // source chunk, bytes
  const uint32_t a_lane = (uint32_t)((w * 8 + lrow) * lda * ESZ + gkb);
  const uint32_t b_lane = (uint32_t)(((w >> 2) * 64 + (w & 3) * 8 + lrow) * ldb * ESZ + gkb);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(A + (size_t)m0 * lda * ESZ), 0, TM * lda * ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(Bt + (size_t)n0 * ldb * ESZ), 0, TN * ldb * ESZ, 0x00020000);
  const uint32_t a_round = (uint32_t)(128 * lda * ESZ), a_half = (uint32_t)(64 * lda * ESZ);
  const uint32_t b_round = (uint32_t)(128 * ldb * ESZ), b_half = (uint32_t)(32 * ldb * ESZ);
  // row scales of this lane's fragment rows (F8): A rows m0 + wr*128 + qi*64 + mt*32 + (lane&31), B rows n0 + wc*64 +
  // qj*32 + (lane&31).  Loaded first (the oldest outstanding loads); their wait is placed behind the DMA prologue.
  uint32_t sa_[2][2] = {{0u, 0u}, {0u, 0u}}, sb_[2] = {0u, 0u};
  if constexpr (F8) {
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) sa_[qi][mt] = scale_a[m0 + (tid >> 8) * 128 + qi * 64 + mt * 32 + (lane & 31)];
#pragma unroll
    for (int qj = 0; qj < 2; ++qj) sb_[qj] = scale_b[n0 + ((tid >> 6) & 3) * 64 + qj * 32 + (lane & 31)];
  }
  unsigned char* lds_w = smem + w * 1024;

#define STAGE_A(buf, h, kt)                                                                       \
  do {                                                                                            \
    const uint32_t s_ = (h) * a_half + (uint32_t)(kt) * (TK * 2);                                 \
    glds16(a_rsrc, a_lane, s_, lds_w + (buf) * KT_BYTES + ((h) ? OFF_A1 : OFF_A0));               \
    glds16(a_rsrc, a_lane, s_ + a_round, lds_w + (buf) * KT_BYTES + ((h) ? OFF_A1 : OFF_A0) + 8192); \
  } while (0)
#define STAGE_B(buf, h, kt)                                                                       \
  do {                                                                                            \
    const uint32_t s_ = (h) * b_half + (uint32_t)(kt) * (TK * 2);                                 \
    glds16(b_rsrc, b_lane, s_, lds_w + (buf) * KT_BYTES + ((h) ? OFF_B1 : OFF_B0));               \
    glds16(b_rsrc, b_lane, s_ + b_round, lds_w + (buf) * KT_BYTES + ((h) ? OFF_B1 : OFF_B0) + 8192); \
  } while (0)

  // ---- fragment read addresses: row = (wave base) + (lane&31); chunk (2*ks + kh) ^ ((row>>1)&7).
  // One address register per (operand, k-step, K-tile buffer): everything else is a ds_read immediate.
  const int fsw = (lane >> 1) & 7, kh = lane >> 5;
  lds_byte_ptr lds0 = (lds_byte_ptr)smem;
  lds_byte_ptr a_ad0[4], a_ad1[4], b_ad0[4], b_ad1[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    // e4m3: ks = 0, 1 -> first chunk of 64-wide step ks (logical chunk (ks&1)*4 + 2 kh), ks = 2, 3 -> its second chunk (+1)
    const int ko = F8 ? (lane & 31) * 128 + (((((ks & 1) * 4 + kh * 2) + (ks >> 1)) ^ fsw) << 4)
                      : (lane & 31) * 128 + (((ks * 2 + kh) ^ fsw) << 4);
    a_ad0[ks] = lds0 + wr * 64 * 128 + ko;
    b_ad0[ks] = lds0 + wc * 32 * 128 + ko;
    a_ad1[ks] = a_ad0[ks] + KT_BYTES;
    b_ad1[ks] = b_ad0[ks] + KT_BYTES;
    asm volatile("" : "+v"(a_ad1[ks]), "+v"(b_ad1[ks]));   // keep them as registers: +64 KiB does not fit an immediate
  }

  f32x16_t acc[2][2][2];   // [quadrant row][quadrant col][m-tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][t][e] = 0.0f;
  typename std::conditional<F8, Frags8, Frags>::type f;

  const int nk = (V & 64) ? 0 : K * ESZ / 128;   // K-tiles of 128 operand bytes per row: even, >= 2 (checked by the launcher)
  if constexpr ((V & 64) == 0) {

  // ---- prologue: K-tile 0 complete in buffer 0, B0 A0 B1 of K-tile 1 in flight; wave row 1 one barrier behind
  STAGE_B(0, 0, 0); STAGE_A(0, 0, 0); STAGE_B(0, 1, 0); STAGE_A(0, 1, 0);
  if (wr == 1) SMD_BAR();
  SMD_VMCNT(4);
  SMD_BAR();
  STAGE_B(1, 0, 1); STAGE_A(1, 0, 1); STAGE_B(1, 1, 1);
  SMD_VMCNT(6);
  SMD_BAR();
  SMD_PIN();
  if constexpr (F8) {      // the compiler's wait for the scale loads lands here (they are older than every DMA above)
    asm volatile("" ::"v"(sa_[0][0]), "v"(sa_[0][1]), "v"(sa_[1][0]), "v"(sa_[1][1]), "v"(sb_[0]), "v"(sb_[1]));
    SMD_PIN();
  }

  // One K-tile from buffer `cur`; the four DMA slots of its phases are given by the caller.
#define KTILE(cur, S1, S2, S3, S4, WAIT4)                                                         \
  do {                                                                                            \
    /* phase 1: quadrant (0,0) */                                                                 \
    if constexpr (!(V & 4)) read_b(f.b0, (cur) ? b_ad1 : b_ad0, OFF_B0);                                           \
    SMD_PIN();                                                                                    \
    if constexpr (!(V & 4)) read_a(f, (cur) ? a_ad1 : a_ad0, OFF_A0);                                              \
    if constexpr (!(V & 8)) { S1; }                                                                                           \
    SMD_LGKMCNT(8);                                                                               \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    if constexpr (!(V & 1)) SMD_LGKMCNT(0);                                                       \
    SMD_PIN();                                                                                    \
    mma_quadrant<V>(acc[0][0], f.a, f.b0, sa_[0], sb_[0]);                                                           \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    SMD_PIN();                                                                                    \
    /* phase 2: quadrant (0,1) */                                                                 \
    if constexpr (!(V & 4)) read_b(f.b1, (cur) ? b_ad1 : b_ad0, OFF_B1);                                           \
    if constexpr (!(V & 8)) { S2; }                                                                                           \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    if constexpr (!(V & 1)) SMD_LGKMCNT(0);                                                       \
    SMD_PIN();                                                                                    \
    mma_quadrant<V>(acc[0][1], f.a, f.b1, sa_[0], sb_[1]);                                                           \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    SMD_PIN();                                                                                    \
    /* phase 3: quadrant (1,0) */                                                                 \
    if constexpr (!(V & 4)) read_a(f, (cur) ? a_ad1 : a_ad0, OFF_A1);                                              \
    if constexpr (!(V & 8)) { S3; }                                                                                           \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    if constexpr (!(V & 1)) SMD_LGKMCNT(0);                                                       \
    SMD_PIN();                                                                                    \
    mma_quadrant<V>(acc[1][0], f.a, f.b0, sa_[1], sb_[0]);                                                           \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    SMD_PIN();                                                                                    \
    /* phase 4: quadrant (1,1) */                                                                 \
    if constexpr (!(V & 8)) { S4; }                                                                                           \
    WAIT4;                                                                                        \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    SMD_PIN();                                                                                    \
    mma_quadrant<V>(acc[1][1], f.a, f.b1, sa_[1], sb_[1]);                                                           \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    SMD_PIN();                                                                                    \
  } while (0)

  if constexpr ((V & 4) != 0 && !F8) {   // ablation (wrong results): fragments read once
    read_b(f.b0, b_ad0, OFF_B0); read_b(f.b1, b_ad0, OFF_B1); read_a(f, a_ad0, OFF_A0);
  }
  int kt = 0;
  for (; kt < nk - 2; kt += 2) {
    KTILE(0, STAGE_A(1, 1, kt + 1), STAGE_B(0, 0, kt + 2), STAGE_A(0, 0, kt + 2), STAGE_B(0, 1, kt + 2), SMD_VMCNT(6));
    KTILE(1, STAGE_A(0, 1, kt + 2), STAGE_B(1, 0, kt + 3), STAGE_A(1, 0, kt + 3), STAGE_B(1, 1, kt + 3), SMD_VMCNT(6));
  }
  // last two K-tiles: only the outstanding A1 of the final tile is still to be issued
  KTILE(0, STAGE_A(1, 1, kt + 1), (void)0, (void)0, (void)0, SMD_VMCNT(0));
  KTILE(1, (void)0, (void)0, (void)0, (void)0, (void)0);
  if (wr == 0) SMD_BAR();
  SMD_PIN();
  }
#undef KTILE
#undef STAGE_A
#undef STAGE_B

  // ---- epilogue: per-wave staging, 4 passes of 32 rows x 64 columns; a lane owns 8 consecutive columns of
  // rows (lane>>3) + 8*i.  The pass's global operands (aux / residual) are requested before the staging.
  float* stage = reinterpret_cast<float*>(smem + w * WAVE_STAGE_BYTES);
  const int c8 = (lane & 7) * 8;
  const int col = n0 + wc * 64 + c8;
  const int rbase = m0 + wr * 128 + (lane >> 3);
  float bias8[8];
  if (ep.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(ep.bias + col), b1 = *reinterpret_cast<const float4*>(ep.bias + col + 4);
    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
    bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
  }
#define EPI_PASS(mi, mt)                                                                          \
  do {                                                                                            \
    smd_epi::EpiPre8 pre[4];                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
      smd_epi::epi8_prefetch(pre[i], rbase + (mi) * 64 + (mt) * 32 + i * 8, col, ep);             \
    stage_tile(acc[mi][0][mt], stage, 4 * kh, (lane & 31), Seq16{});                              \
    stage_tile(acc[mi][1][mt], stage, 4 * kh, 32 + (lane & 31), Seq16{});                         \
    __builtin_amdgcn_wave_barrier();                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
      const float* sp = stage + (i * 8 + (lane >> 3)) * SLD + c8;                                 \
      const float4 lo = *reinterpret_cast<const float4*>(sp), hi = *reinterpret_cast<const float4*>(sp + 4); \
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};                              \
      smd_epi::epi8_apply(v, bias8, pre[i], rbase + (mi) * 64 + (mt) * 32 + i * 8, col, ep);      \
    }                                                                                             \
    __builtin_amdgcn_wave_barrier();                                                              \
  } while (0)
  if (pk_epi) {
    // ---- (bias ->) bf16 outputs (fc1 of a DenseResBlock, `up`, every dgrad): the accumulators are rounded BEFORE the staging.
    // A lane holds rows r, r+1 (e, e+1) of its column: the pair goes into LDS as ONE dword (v_cvt_pk_bf16_f32), the reader
    // takes 8 columns x 2 rows with two ds_read_b128, splits the halves with v_perm_b32 and stores two 16-byte row segments:
    // 16 ds_write_b32 + 4 ds_read_b128 per pass instead of 32 + 8 (same values bit for bit: bias add in fp32, one RNE rounding).
    constexpr int SLD2 = 68;
    uint32_t* stage2 = reinterpret_cast<uint32_t*>(smem + w * WAVE_STAGE_BYTES);
    float bcol[2] = {0.f, 0.f};
    if (ep.bias) { bcol[0] = ep.bias[n0 + wc * 64 + (lane & 31)]; bcol[1] = ep.bias[n0 + wc * 64 + 32 + (lane & 31)]; }
    bf16_t* out = ep.out_bf16;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int jq = 0; jq < 2; ++jq)
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            bf16x2_t pr;
            pr[0] = f2bf(acc[mi][jq][mt][e] + bcol[jq]);
            pr[1] = f2bf(acc[mi][jq][mt][e + 1] + bcol[jq]);
            const int p = 2 * kh + ((e & 3) >> 1) + 4 * (e >> 2);                        // row pair inside the 32-row pass
            stage2[p * SLD2 + jq * 32 + (lane & 31)] = __builtin_bit_cast(uint32_t, pr);
          }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int p = (lane >> 3) + 8 * i;
          const uint32_t* sp = stage2 + p * SLD2 + c8;
          const uint4 lo = *reinterpret_cast<const uint4*>(sp), hi = *reinterpret_cast<const uint4*>(sp + 4);
          const uint32_t d[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          uint4 r0, r1;
          r0.x = __builtin_amdgcn_perm(d[1], d[0], 0x05040100u); r1.x = __builtin_amdgcn_perm(d[1], d[0], 0x07060302u);
          r0.y = __builtin_amdgcn_perm(d[3], d[2], 0x05040100u); r1.y = __builtin_amdgcn_perm(d[3], d[2], 0x07060302u);
          r0.z = __builtin_amdgcn_perm(d[5], d[4], 0x05040100u); r1.z = __builtin_amdgcn_perm(d[5], d[4], 0x07060302u);
          r0.w = __builtin_amdgcn_perm(d[7], d[6], 0x05040100u); r1.w = __builtin_amdgcn_perm(d[7], d[6], 0x07060302u);
          const size_t row = (size_t)(m0 + wr * 128 + mi * 64 + mt * 32 + 2 * p);
          *reinterpret_cast<uint4*>(out + row * ep.ld_outb + col) = r0;
          *reinterpret_cast<uint4*>(out + (row + 1) * ep.ld_outb + col) = r1;
        }
        __builtin_amdgcn_wave_barrier();
      }
    return;
  }
  if constexpr ((V & 32) != 0) {        // ablation: no epilogue (keep the accumulators alive)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) asm volatile("" ::"v"(acc[i][j][t]));
    return;
  }
  EPI_PASS(0, 0); EPI_PASS(0, 1); EPI_PASS(1, 0); EPI_PASS(1, 1);
#undef EPI_PASS
}

}  // namespace

// the packed-bf16 epilogue applies: (bias ->) bf16 output and nothing else (knob "gemm_nt256_pk" 0 switches it off: A/B, tests)
static int nt256_pk_epilogue(const GemmEpilogue& ep) {
  return (smd_tuning_get("gemm_nt256_pk") != 0 && ep.out_bf16 && !ep.out_f32 && !ep.pre_bf16 && ep.act == SMD_ACT_NONE &&
          ep.aux_mode == SMD_AUX_NONE && !ep.res_f32 && !ep.res_bf16) ? 1 : 0;
}

bool gemm_nt256_eligible(int M, int N, int K, const GemmEpilogue& ep, int min_tiles) {
  if (!smd_epi::oct_ok(ep)) return false;
  const int mode = smd_tuning_get("gemm_nt256");
  if (mode == 0 || M % TM || N % TN || K % (2 * TK) || K < 2 * TK) return false;
  if (mode == 2) return true;                  // forced (tests)
  const long tiles = (long)(M / TM) * (N / TN);
  // at least ~3/4 of the 256 CUs busy (192): smaller grids are better served by 128-wide tiles -- unless the caller runs
  // two such streams side by side (concurrent sampling chains ask for 128)
  return tiles >= min_tiles;
}

int launch_gemm_nt256(const bf16_t* A, int lda, const bf16_t* Bt, int ldb, int M, int N, int K,
                      const GemmEpilogue& ep, hipStream_t st) {
  SMD_ARG_CHECK(M % TM == 0 && N % TN == 0 && K % (2 * TK) == 0 && K >= 2 * TK,
                "gemm_nt256: M=%d N=%d must be multiples of 256 and K=%d a multiple of 128", M, N, K);
  SMD_ARG_CHECK((size_t)TM * lda * 2 < (1ull << 32) && (size_t)TN * ldb * 2 < (1ull << 32), "gemm_nt256: row band exceeds 4 GiB");
  const int tiles_m = M / TM, tiles_n = N / TN;
  const int nwg = tiles_m * tiles_n;
  SMD_ARG_CHECK(smd_epi::oct_ok(ep), "gemm_nt256: epilogue not supported (alignment / alpha / res_bf16 / accumulate)");
  const int pk = nt256_pk_epilogue(ep);
#define SMD_NT256_LAUNCH(V_) hipLaunchKernelGGL((gemm_nt256_kernel<V_, false>), dim3(nwg), dim3(512), 0, st, A, lda, Bt, ldb, M, N, K, tiles_n, nwg, ep, nullptr, nullptr, pk)
  // schedule variants 1..3 compute the same result (A/B runs); the ABLATION variants (bits 4, 8, 32, 64: parts of the
  // kernel removed, wrong results by construction) exist only in a -DSMD_ABLATIONS build (tools/kbench.py --gemm-ab)
  switch (smd_tuning_get("gemm_nt256_variant")) {
    case 0: SMD_NT256_LAUNCH(0); break;
    case 1: SMD_NT256_LAUNCH(1); break;
    case 2: SMD_NT256_LAUNCH(2); break;
    case 3: SMD_NT256_LAUNCH(3); break;
#ifdef SMD_ABLATIONS
    case 4: SMD_NT256_LAUNCH(4); break;
    case 8: SMD_NT256_LAUNCH(8); break;
    case 12: SMD_NT256_LAUNCH(12); break;
    case 32: SMD_NT256_LAUNCH(32); break;
    case 64: SMD_NT256_LAUNCH(64); break;
    case 96: SMD_NT256_LAUNCH(96); break;
#endif
    default:
      smd_set_error("gemm_nt256: variant %d is not in this build (ablation variants need -DSMD_ABLATIONS)", smd_tuning_get("gemm_nt256_variant"));
      return -1;
  }
#undef SMD_NT256_LAUNCH
  SMD_LAUNCH_CHECK();
  return 0;
}

// e4m3 operands with per-row E8M0 scales (see the kernel comment): C = (2^sa[m] A8[m,:]) . (2^sb[n] Bt8[n,:]) + epilogue
int launch_gemm_nt256_fp8(const unsigned char* A8, int lda, const uint32_t* scale_a, const unsigned char* Bt8, int ldb,
                          const uint32_t* scale_b, int M, int N, int K, const GemmEpilogue& ep, hipStream_t st) {
  SMD_ARG_CHECK(A8 && Bt8 && scale_a && scale_b, "gemm_nt256_fp8: null operand");
  SMD_ARG_CHECK(M % TM == 0 && N % TN == 0 && K % 256 == 0 && K >= 256,
                "gemm_nt256_fp8: M=%d N=%d must be multiples of 256 and K=%d a multiple of 256", M, N, K);
  SMD_ARG_CHECK(lda % 16 == 0 && ldb % 16 == 0 && lda >= K && ldb >= K, "gemm_nt256_fp8: lda=%d ldb=%d must be >= K and multiples of 16", lda, ldb);
  SMD_ARG_CHECK(smd_epi::oct_ok(ep), "gemm_nt256_fp8: epilogue not supported (alignment / alpha / res_bf16 / accumulate)");
  SMD_ARG_CHECK(ep.out_f32 || ep.out_bf16 || ep.pre_bf16, "gemm_nt256_fp8: no output");
  const int tiles_n = N / TN, nwg = (M / TM) * tiles_n;
  hipLaunchKernelGGL((gemm_nt256_kernel<0, true>), dim3(nwg), dim3(512), 0, st, A8, lda, Bt8, ldb, M, N, K, tiles_n, nwg, ep, scale_a,
                     scale_b, nt256_pk_epilogue(ep));
  SMD_LAUNCH_CHECK();
  return 0;
}
