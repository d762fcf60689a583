// bf16 MFMA weight-gradient GEMM, "TN" form, 256 x 256 output tile, 8-phase pipeline (the wgrad twin of
// gemm_nt256.hip):
//     dW[Kd,N] = sum_m X[m,Kd] * dY[m,N]      and      db[N] = sum_m dY[m,N]
// (value_and_grad of train_ncsn.py:282-283 through every 2048-wide nn.Dense: models/shared.py:65,69, ncsn.py:178).
//
// Both operands have the contraction index m as the ROW index, so the 8-consecutive-m MFMA fragments are read
// with the gfx950 transposing LDS read ds_read_b64_tr_b16 (lane semantics pinned by smd_probe_tr_read):
//   * K-tile = 64 rows of m; it is four 16-KiB half-tiles  B0 A0 B1 A1, each [64 m][128 cols] with 256-B rows
//     (half h of A = the Kd columns of every wave row's quadrant h, half h of B = the N columns of every wave
//     column's quadrant h).  DMA: buffer_load_dwordx4 ... lds, 2 per lane per half-tile; rows >= Mrows are
//     out of the descriptor's range and read as zeros (ragged M needs no zero page).
//   * 16-byte chunk c of LDS row r is stored at chunk c ^ ((r&3)<<2) (source-side swizzle + the same XOR on
//     the read): the 4 rows of one transpose block sit in 4 different bank quarters.
//   * schedule, staggering, hazard rules and register budget are those of gemm_nt256.hip: 4 phases per K-tile,
//     8 MFMA 32x32x16 per phase per wave, one counted vmcnt(6) per K-tile, two K-tile buffers (128 KiB LDS).
//   * split-K over m: grid = tiles x nsplit (flattened, XCD-remapped so an XCD works on one m-range); every
//     block writes an fp32 partial tile into its split's slab; reduce_slabs256_kernel adds the slabs in a fixed
//     order (deterministic, no atomics).
//   * bias gradient on the matrix cores, load-balanced: the block of Kd-tile row tk adds the all-ones MFMA only
//     on K-tiles with kt % tiles_k == tk (wave row 0 for the B0 columns, wave row 1 for B1), so every block
//     carries 1/tiles_k of the column-sum work; the partial rows are summed by the same reduce kernel.
#include "smd_kernels.h"

// dynamic-LDS pad that makes a weight-gradient workgroup fill a CU's LDS (smd_kernels.h)
namespace {

constexpr int TM = 256, TN = 256, TKM = 64;
constexpr int HALF_BYTES = TKM * 128 * 2;         // 16 KiB: 64 rows x 128 bf16
constexpr int KT_BYTES = 4 * HALF_BYTES;          // B0 A0 B1 A1
constexpr int SMEM_BYTES = 2 * KT_BYTES;          // 128 KiB
constexpr int OFF_B0 = 0, OFF_A0 = HALF_BYTES, OFF_B1 = 2 * HALF_BYTES, OFF_A1 = 3 * HALF_BYTES;
constexpr int SLD = 68;
constexpr int WAVE_STAGE_BYTES = 32 * SLD * 4;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff,
                                       unsigned char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_wave_base, 16, voff, soff, 0, 0);
}

#define SMD_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define SMD_LGKMCNT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define SMD_PIN() __builtin_amdgcn_sched_barrier(0)
#define SMD_BAR() __builtin_amdgcn_s_barrier()

union Frag8 {
  bf16x8_t v;
  s16x4_t h[2];
};

// Four k-steps of one 32-column fragment column: 8 transpose reads from one address register.
// OFF = half-tile offset within the K-tile buffer; k-step ks adds ks*4096, the second 4-row block 1024.
template <int OFF>
__device__ __forceinline__ void tr_read4(unsigned ad, Frag8 (&f)[4]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%9\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:%10\n\t"
      "ds_read_b64_tr_b16 %2, %8 offset:%11\n\t"
      "ds_read_b64_tr_b16 %3, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %4, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %5, %8 offset:%14\n\t"
      "ds_read_b64_tr_b16 %6, %8 offset:%15\n\t"
      "ds_read_b64_tr_b16 %7, %8 offset:%16"
      : "=&v"(f[0].h[0]), "=&v"(f[0].h[1]), "=&v"(f[1].h[0]), "=&v"(f[1].h[1]), "=&v"(f[2].h[0]), "=&v"(f[2].h[1]),
        "=&v"(f[3].h[0]), "=&v"(f[3].h[1])
      : "v"(ad), "i"(OFF), "i"(OFF + 1024), "i"(OFF + 4096), "i"(OFF + 5120), "i"(OFF + 8192), "i"(OFF + 9216),
        "i"(OFF + 12288), "i"(OFF + 13312)
      : "memory");
}

__device__ __forceinline__ void mma_quadrant(f32x16_t (&acc)[2], const Frag8 (&a)[2][4], const Frag8 (&b)[4]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][ks].v, b[ks].v, acc[mt], 0, 0, 0);
}
__device__ __forceinline__ void mma_ones(f32x16_t& acc, const bf16x8_t ones, const Frag8 (&b)[4]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, b[ks].v, acc, 0, 0, 0);
}

template <int... Es> struct IntSeq {};
typedef IntSeq<0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15> Seq16;
template <int... Es>
__device__ __forceinline__ void stage_tile(const f32x16_t& acc, float* stage, int row0, int col, IntSeq<Es...>) {
  ((stage[(row0 + (Es & 3) + 8 * (Es >> 2)) * SLD + col] = acc[Es]), ...);
}

struct Tn256Args {
  const bf16_t* X; int ldx;
  const bf16_t* dY; int ldy;
  int Mrows, Kd, N;
  float* dst; int ld; size_t split_stride;   // partial tile destination: dst + split*split_stride, row stride ld
  float* bias_dst;                           // [(split*tiles_k + tk)][N] partial column sums, or null
  int tiles_n, tiles_k, ktiles_per_split, nwg;
};

// Up to four problems per launch.  Four 2048 x 2048 weight gradients (the Dense layers of two DenseResBlocks) are 4 x 64 =
// 256 tiles: one per CU with NO split over m -- every block walks the whole contraction (128 K-tiles instead of 32: the
// fixed cost of a block is paid once), writes its tile straight into the gradient buffer, and the slab round trip and the
// reduce launches of the split form disappear (round 3: 4 x 69.6 us + 4 x 22.1 us -> one launch).  Two problems: two
// m-splits fill the chip (half the slab traffic of a lone problem's four splits).
#define SMD_TN256_GROUP_MAX 4
struct Tn256Group { int ngroups; int nwg_total; Tn256Args p[SMD_TN256_GROUP_MAX]; };

__global__ __launch_bounds__(512) void gemm_tn256_kernel(Tn256Group ga) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

  // flattened (problem, split, tile) id, XCD-remapped: one XCD's blocks share an m-range and neighbouring tiles
  const int bid = blockIdx.x;
  const int q = ga.nwg_total >> 3, r = ga.nwg_total & 7, xcd = bid & 7;
  int vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  int gsel = 0;
  while (gsel + 1 < ga.ngroups && vid >= ga.p[gsel].nwg) { vid -= ga.p[gsel].nwg; ++gsel; }
  const Tn256Args a = ga.p[gsel];
  const int tiles = a.tiles_k * a.tiles_n;
  const int split = vid / tiles, tile = vid - split * tiles;
  const int tk = tile / a.tiles_n, tn = tile - tk * a.tiles_n;
  const int kd0 = tk * TM, n0 = tn * TN;
  const int kt_begin = split * a.ktiles_per_split;
  const int nk = a.ktiles_per_split;            // even, >= 2

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;

  // ---- DMA sources.  Round j of half h: wave w fills LDS rows j*32 + w*4 .. +4 (1 KiB), 16 lanes per 256-B row;
  // LDS chunk (lane&15) of row r holds source chunk c' = (lane&15) ^ ((r&3)<<2), r&3 == lane>>4.
  //   A half h, LDS col c <-> global col kd0 + (c/64)*128 + h*64 + c%64  (wave row wr reads c in [wr*64, +64))
  //   B half h, LDS col c <-> global col n0  + (c/32)*64  + h*32 + c%32  (wave col wc reads c in [wc*32, +32))
  const int cs = (lane & 15) ^ ((lane >> 4) << 2);
  // Running per-lane byte offsets (one per DMA round), advanced by one K-tile after each half-1 stage: the ROW
  // part must live in the VGPR offset because the descriptor's range check ignores the SGPR offset, and rows
  // >= Mrows are zero-filled by that check (ragged M needs no zero page).  soffset only selects the half.
  const uint32_t a_row = (uint32_t)(a.ldx * 2), b_row = (uint32_t)(a.ldy * 2);
  uint32_t a_v[2], b_v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = kt_begin * TKM + j * 32 + w * 4 + (lane >> 4);
    a_v[j] = (uint32_t)row * a_row + (uint32_t)(((cs >> 3) * 128 + (cs & 7) * 8) * 2);
    b_v[j] = (uint32_t)row * b_row + (uint32_t)(((cs >> 2) * 64 + (cs & 3) * 8) * 2);
  }
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(a.X + kd0), 0, (int)(((size_t)a.Mrows * a.ldx - kd0) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(a.dY + n0), 0, (int)(((size_t)a.Mrows * a.ldy - n0) * 2), 0x00020000);
  unsigned char* lds_w = smem + w * 1024;

  // stage order per operand is half 0 then half 1 of the same K-tile, so one running offset serves both
#define STAGE_A(buf, h)                                                                           \
  do {                                                                                            \
    glds16(a_rsrc, a_v[0], (h) * 128, lds_w + (buf) * KT_BYTES + ((h) ? OFF_A1 : OFF_A0));        \
    glds16(a_rsrc, a_v[1], (h) * 128, lds_w + (buf) * KT_BYTES + ((h) ? OFF_A1 : OFF_A0) + 8192); \
    if (h) { a_v[0] += TKM * a_row; a_v[1] += TKM * a_row; }                                      \
  } while (0)
#define STAGE_B(buf, h)                                                                           \
  do {                                                                                            \
    glds16(b_rsrc, b_v[0], (h) * 64, lds_w + (buf) * KT_BYTES + ((h) ? OFF_B1 : OFF_B0));         \
    glds16(b_rsrc, b_v[1], (h) * 64, lds_w + (buf) * KT_BYTES + ((h) ? OFF_B1 : OFF_B0) + 8192);  \
    if (h) { b_v[0] += TKM * b_row; b_v[1] += TKM * b_row; }                                      \
  } while (0)

  // ---- transpose-read addresses (bytes within a half-tile): row = ks*16 + 8*(g>>1) + rd*4 + rsub,
  // column = (wave base) + mt*32 + 16*(g&1) + 4*(lane&3); chunk ^= rsub<<2.
  const int g = lane >> 4, rsub = (lane >> 2) & 3;
  const unsigned lds0 = (unsigned)(size_t)(lds_byte_t*)smem;
  const unsigned row_off = (unsigned)((8 * (g >> 1) + rsub) * 256);
  unsigned a_ad0[2], a_ad1[2], b_ad0, b_ad1;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int cb = (wr * 64 + mt * 32 + 16 * (g & 1) + 4 * (lane & 3)) * 2;
    a_ad0[mt] = lds0 + row_off + ((((cb >> 4) ^ (rsub << 2)) << 4) | (cb & 15));
    a_ad1[mt] = a_ad0[mt] + KT_BYTES;
  }
  {
    const int cb = (wc * 32 + 16 * (g & 1) + 4 * (lane & 3)) * 2;
    b_ad0 = lds0 + row_off + ((((cb >> 4) ^ (rsub << 2)) << 4) | (cb & 15));
    b_ad1 = b_ad0 + KT_BYTES;
  }

  f32x16_t acc[2][2][2];   // [quadrant row][quadrant col][m-tile]
  f32x16_t acc_b;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc_b[e] = 0.0f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][t][e] = 0.0f;
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
  Frag8 fa[2][4], fb0[4], fb1[4];
  const bool want_bias = a.bias_dst != nullptr;

  // ---- prologue (see gemm_nt256.hip)
  STAGE_B(0, 0); STAGE_A(0, 0); STAGE_B(0, 1); STAGE_A(0, 1);
  if (wr == 1) SMD_BAR();
  SMD_VMCNT(4);
  SMD_BAR();
  STAGE_B(1, 0); STAGE_A(1, 0); STAGE_B(1, 1);
  SMD_VMCNT(6);
  SMD_BAR();
  SMD_PIN();

  // One K-tile from buffer `cur`.  BIAS (wave-uniform): this K-tile contributes to the block's column sums.
#define KTILE(cur, S1, S2, S3, S4, WAIT4, BIAS)                                                   \
  do {                                                                                            \
    /* phase 1: quadrant (0,0); B0 reads first so that lgkmcnt(15) retires them before the barrier */ \
    tr_read4<OFF_B0>((cur) ? b_ad1 : b_ad0, fb0);                                                 \
    tr_read4<OFF_A0>((cur) ? a_ad1[0] : a_ad0[0], fa[0]);                                         \
    tr_read4<OFF_A0>((cur) ? a_ad1[1] : a_ad0[1], fa[1]);                                         \
    S1;                                                                                           \
    SMD_LGKMCNT(15);                                                                              \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    SMD_LGKMCNT(0);                                                                               \
    SMD_PIN();                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                \
    mma_quadrant(acc[0][0], fa, fb0);                                                             \
    if ((BIAS) && wr == 0) mma_ones(acc_b, ones, fb0);                                            \
    __builtin_amdgcn_s_setprio(0);                                                                \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    SMD_PIN();                                                                                    \
    /* phase 2: quadrant (0,1) */                                                                 \
    tr_read4<OFF_B1>((cur) ? b_ad1 : b_ad0, fb1);                                                 \
    S2;                                                                                           \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    SMD_LGKMCNT(0);                                                                               \
    SMD_PIN();                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                \
    mma_quadrant(acc[0][1], fa, fb1);                                                             \
    if ((BIAS) && wr == 1) mma_ones(acc_b, ones, fb1);                                            \
    __builtin_amdgcn_s_setprio(0);                                                                \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    SMD_PIN();                                                                                    \
    /* phase 3: quadrant (1,0) */                                                                 \
    tr_read4<OFF_A1>((cur) ? a_ad1[0] : a_ad0[0], fa[0]);                                         \
    tr_read4<OFF_A1>((cur) ? a_ad1[1] : a_ad0[1], fa[1]);                                         \
    S3;                                                                                           \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    SMD_LGKMCNT(0);                                                                               \
    SMD_PIN();                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                \
    mma_quadrant(acc[1][0], fa, fb0);                                                             \
    __builtin_amdgcn_s_setprio(0);                                                                \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    SMD_PIN();                                                                                    \
    /* phase 4: quadrant (1,1) */                                                                 \
    S4;                                                                                           \
    WAIT4;                                                                                        \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    SMD_PIN();                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                \
    mma_quadrant(acc[1][1], fa, fb1);                                                             \
    __builtin_amdgcn_s_setprio(0);                                                                \
    SMD_PIN();                                                                                    \
    SMD_BAR();                                                                                    \
    SMD_PIN();                                                                                    \
  } while (0)

  // bias K-tiles of this block: (kt_begin + kt) % tiles_k == tk
  int bc = (kt_begin - tk) % a.tiles_k;
  if (bc < 0) bc += a.tiles_k;
  for (int kt = 0; kt < nk - 2; kt += 2) {
    const bool bz0 = want_bias && bc == 0, bz1 = want_bias && (bc + 1 == a.tiles_k || (a.tiles_k == 1));
    KTILE(0, STAGE_A(1, 1), STAGE_B(0, 0), STAGE_A(0, 0), STAGE_B(0, 1), SMD_VMCNT(6), bz0);
    KTILE(1, STAGE_A(0, 1), STAGE_B(1, 0), STAGE_A(1, 0), STAGE_B(1, 1), SMD_VMCNT(6), bz1);
    bc += 2;
    while (bc >= a.tiles_k) bc -= a.tiles_k;
  }
  {
    const bool bz0 = want_bias && bc == 0, bz1 = want_bias && (bc + 1 == a.tiles_k || (a.tiles_k == 1));
    KTILE(0, STAGE_A(1, 1), (void)0, (void)0, (void)0, SMD_VMCNT(0), bz0);
    KTILE(1, (void)0, (void)0, (void)0, (void)0, (void)0, bz1);
  }
  if (wr == 0) SMD_BAR();
  SMD_PIN();
#undef KTILE
#undef STAGE_A
#undef STAGE_B

  // ---- partial column sums: row 0 of the ones-MFMA (lanes 0..31, element 0)
  const int kh = lane >> 5;
  if (want_bias && kh == 0) {
    float* bd = a.bias_dst + (size_t)(split * a.tiles_k + tk) * a.N;
    bd[n0 + wc * 64 + wr * 32 + (lane & 31)] = acc_b[0];
  }

  // ---- partial tile through per-wave LDS staging: 4 passes of 32 rows x 64 columns, 32-byte stores per lane
  float* stage = reinterpret_cast<float*>(smem + w * WAVE_STAGE_BYTES);
  const int c8 = (lane & 7) * 8;
  float* dst = a.dst + (size_t)split * a.split_stride + (size_t)(kd0 + wr * 128 + (lane >> 3)) * a.ld + n0 + wc * 64 + c8;
#define EPI_PASS(mi, mt)                                                                          \
  do {                                                                                            \
    stage_tile(acc[mi][0][mt], stage, 4 * kh, (lane & 31), Seq16{});                              \
    stage_tile(acc[mi][1][mt], stage, 4 * kh, 32 + (lane & 31), Seq16{});                         \
    __builtin_amdgcn_wave_barrier();                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
      const float* sp = stage + (i * 8 + (lane >> 3)) * SLD + c8;                                 \
      const float4 lo = *reinterpret_cast<const float4*>(sp), hi = *reinterpret_cast<const float4*>(sp + 4); \
      float4* o = reinterpret_cast<float4*>(dst + (size_t)((mi) * 64 + (mt) * 32 + i * 8) * a.ld); \
      o[0] = lo; o[1] = hi;                                                                       \
    }                                                                                             \
    __builtin_amdgcn_wave_barrier();                                                              \
  } while (0)
  EPI_PASS(0, 0); EPI_PASS(0, 1); EPI_PASS(1, 0); EPI_PASS(1, 1);
#undef EPI_PASS
}

// out_w[i] = sum_s part[s][i] for the dW elements, out_b[n] = sum_p bias_part[p][n]; fixed order, float4 wide.
__global__ __launch_bounds__(256) void reduce_slabs256_kernel(const float* __restrict__ part, size_t stride, int nsplit,
                                                              size_t n_w4, float* __restrict__ out_w, int ldo, int N,
                                                              const float* __restrict__ bias_part, int nparts,
                                                              float* __restrict__ out_b) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_w4) {
    const float4* p = reinterpret_cast<const float4*>(part) + i;
    float4 s = p[0];
    for (int k = 1; k < nsplit; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * stride + i * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const size_t e = i * 4;
    const size_t row = e / N, col = e - row * N;
    *reinterpret_cast<float4*>(out_w + row * ldo + col) = s;
  } else if (out_b) {
    const size_t j = i - n_w4;
    if (j * 4 < (size_t)N) {
      float4 s = make_float4(0, 0, 0, 0);
      for (int k = 0; k < nparts; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(bias_part + (size_t)k * N + j * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4*>(out_b + j * 4) = s;
    }
  }
}

}  // namespace

// Eligibility + split choice for the 256x256 wgrad kernel.  Returns nsplit (0 = not eligible).
int gemm_tn256_plan(const TnLaunch& t, int* ktiles_per_split) {
  if (smd_tuning_get("gemm_tn256") == 0) return 0;
  if (t.Kd % TM || t.N % TN || t.ldx % 8 || t.ldy % 8 || t.ldo % 4 || !t.slab) return 0;
  if ((size_t)t.Mrows * t.ldx * 2 >= (1ull << 31) || (size_t)t.Mrows * t.ldy * 2 >= (1ull << 31)) return 0;
  if ((((uintptr_t)t.out) & 15) || (t.bias_out && (((uintptr_t)t.bias_out) & 15))) return 0;
  const int tiles = (t.Kd / TM) * (t.N / TN);
  const int total_kt = (t.Mrows + TKM - 1) / TKM;
  if (total_kt < 8) return 0;
  int nsplit = (256 + tiles - 1) / tiles;               // one workgroup per CU (128 KiB LDS each)
  if (nsplit > 4 && smd_tuning_get("gemm_tn256") != 2) return 0;   // slab traffic would dominate: 128-wide kernel
  if (nsplit * 4 > total_kt) nsplit = total_kt / 4;     // at least 4 K-tiles per block
  if (nsplit < 1) nsplit = 1;
  int per = (total_kt + nsplit - 1) / nsplit;
  per = (per + 1) & ~1;                                 // the pipeline consumes K-tiles in pairs
  nsplit = (total_kt + per - 1) / per;
  const size_t need = (size_t)nsplit * t.Kd * t.N + (size_t)nsplit * (t.Kd / TM) * t.N;
  if (need > t.slab_elems) return 0;
  if ((long)tiles * nsplit < 96 && smd_tuning_get("gemm_tn256") != 2) return 0;   // tiny grids: 128-wide kernel
  *ktiles_per_split = per;
  return nsplit;
}

static void fill_tn256(Tn256Args& a, const TnLaunch& t, int nsplit, int per, float* slab) {
  a.X = t.X; a.ldx = t.ldx; a.dY = t.dY; a.ldy = t.ldy; a.Mrows = t.Mrows; a.Kd = t.Kd; a.N = t.N;
  a.tiles_k = t.Kd / TM; a.tiles_n = t.N / TN; a.ktiles_per_split = per;
  a.nwg = a.tiles_k * a.tiles_n * nsplit;
  const size_t n_w = (size_t)t.Kd * t.N;
  a.dst = slab; a.ld = t.N; a.split_stride = n_w;
  a.bias_dst = t.bias_out ? slab + (size_t)nsplit * n_w : nullptr;
}
static int reduce_tn256(const TnLaunch& t, const Tn256Args& a, int nsplit, hipStream_t st) {
  const size_t n_w = (size_t)t.Kd * t.N, n_w4 = n_w / 4;
  const size_t total = n_w4 + (t.bias_out ? (size_t)t.N / 4 : 0);
  hipLaunchKernelGGL(reduce_slabs256_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.dst, n_w, nsplit,
                     n_w4, t.out, t.ldo, t.N, a.dst + (size_t)nsplit * n_w, nsplit * a.tiles_k, t.bias_out);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_gemm_tn256(const TnLaunch& t, int nsplit, int ktiles_per_split, hipStream_t st) {
  Tn256Group ga;
  ga.ngroups = 1;
  fill_tn256(ga.p[0], t, nsplit, ktiles_per_split, t.slab);
  ga.nwg_total = ga.p[0].nwg;
  hipLaunchKernelGGL(gemm_tn256_kernel, dim3(ga.nwg_total), dim3(512), smd_tn_pad_bytes(SMEM_BYTES), st, ga);
  SMD_LAUNCH_CHECK();
  return reduce_tn256(t, ga.p[0], nsplit, st);
}

// n <= 4 tn256-eligible problems with the same contraction length in one launch.  The split over m is chosen for the whole
// group (one workgroup per CU): 4 x 64 tiles -> no split, 2 x 64 -> two, 1 x 64 -> four.  Without a split the tiles are
// written straight to the gradient (ldo) and only the bias partial rows (tiles_k per problem) go through the reduce kernel.
int launch_gemm_tn256_multi(const TnLaunch* ts, int n, hipStream_t st) {
  SMD_ARG_CHECK(ts && n >= 1 && n <= SMD_TN256_GROUP_MAX, "gemm_tn256_multi: 1..%d problems", SMD_TN256_GROUP_MAX);
  const int total_kt = (ts[0].Mrows + TKM - 1) / TKM;
  int tiles_all = 0;
  for (int i = 0; i < n; ++i) {
    int per_i = 0;
    SMD_ARG_CHECK(gemm_tn256_plan(ts[i], &per_i) > 0, "gemm_tn256_multi: problem %d not eligible for the 256x256 kernel", i);
    SMD_ARG_CHECK(ts[i].Mrows == ts[0].Mrows, "gemm_tn256_multi: different contraction lengths");
    tiles_all += (ts[i].Kd / TM) * (ts[i].N / TN);
  }
  int nsplit = 256 / tiles_all;
  if (nsplit * 4 > total_kt) nsplit = total_kt / 4;
  if (nsplit < 1) nsplit = 1;
  int per = (total_kt + nsplit - 1) / nsplit;
  per = (per + 1) & ~1;
  nsplit = (total_kt + per - 1) / per;
  Tn256Group ga;
  ga.ngroups = n;
  ga.nwg_total = 0;
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    const TnLaunch& t = ts[i];
    const size_t n_w = (size_t)t.Kd * t.N, bias_rows = (size_t)nsplit * (t.Kd / TM) * t.N;
    const size_t need = (nsplit > 1 ? (size_t)nsplit * n_w : 0) + bias_rows;
    SMD_ARG_CHECK(off + need <= ts[0].slab_elems, "gemm_tn256_multi: slab workspace too small");
    fill_tn256(ga.p[i], t, nsplit, per, ts[0].slab + off);
    if (nsplit == 1) {                    // tiles go straight to the gradient; the slab only holds the bias partial rows
      SMD_ARG_CHECK(t.ldo % 4 == 0, "gemm_tn256_multi: ldo must be a multiple of 4");
      ga.p[i].dst = t.out; ga.p[i].ld = t.ldo; ga.p[i].split_stride = 0;
      ga.p[i].bias_dst = t.bias_out ? ts[0].slab + off : nullptr;
    }
    ga.nwg_total += ga.p[i].nwg;
    off += (need + 3) / 4 * 4;
  }
  hipLaunchKernelGGL(gemm_tn256_kernel, dim3(ga.nwg_total), dim3(512), smd_tn_pad_bytes(SMEM_BYTES), st, ga);
  SMD_LAUNCH_CHECK();
  for (int i = 0; i < n; ++i) {
    const TnLaunch& t = ts[i];
    if (nsplit > 1) {
      const int rc = reduce_tn256(t, ga.p[i], nsplit, st);
      if (rc) return rc;
    } else if (t.bias_out) {              // column sums: tiles_k partial rows -> db
      hipLaunchKernelGGL(reduce_slabs256_kernel, dim3((unsigned)((t.N / 4 + 255) / 256)), dim3(256), 0, st, ga.p[i].bias_dst, (size_t)0, 1,
                         (size_t)0, t.out, t.ldo, t.N, ga.p[i].bias_dst, ga.p[i].tiles_k, t.bias_out);
      SMD_LAUNCH_CHECK();
    }
  }
  return 0;
}

int launch_gemm_tn256_pair(const TnLaunch& t0, const TnLaunch& t1, hipStream_t st) {
  const TnLaunch ts[2] = {t0, t1};
  return launch_gemm_tn256_multi(ts, 2, st);
}
