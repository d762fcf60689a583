// bf16 MFMA weight-gradient GEMM, "TN" form:
//     dW[Kd,N] = sum_m X[m,Kd] * dY[m,N]      and      db[N] = sum_m dY[m,N]
//
// Backward of every nn.Dense on the eps-net path (the value_and_grad of train_ncsn.py:282-283).
// Both operands are stored with the contraction index m as the ROW index, so MFMA fragments
// (8 consecutive m per lane) need a transposed read.  gfx950 has one: ds_read_b64_tr_b16.
//
//   * 128(Kd) x 128(N) output tile, 64 m-rows per K-tile, 4 waves (2x2) x 2x2 MFMA 32x32x16.
//   * operand tiles [64 m][128 cols] bf16 are DMA'd row-major with global_load_lds_dwordx4
//     (a 256-B row = 16 lanes), two LDS buffers, counted vmcnt, raw s_barrier (as gemm_nt).
//   * fragment = two ds_read_b64_tr_b16: within a 16-lane group lane i supplies the address of
//     row (i>>2), 8-byte column chunk (i&3) of a [4 m][16 col] block and receives column i of
//     those 4 rows (cdna_hip_programming.md section 2 / T10; verified by smd_probe_tr_read).
//   * 16-byte chunk c of LDS row r is stored at chunk c ^ ((r&3)<<2) (source-side swizzle +
//     matching XOR on the read): the 4 rows of one transpose block land in 4 different 64-B
//     bank quarters, so a 32-lane service group is conflict free.
//   * the bias gradient rides on the matrix cores: the workgroups of the first Kd-tile row run
//     one extra MFMA per B fragment with an all-ones A fragment, whose every output row is the
//     column sum of dY -- no separate reduction kernels.
//   * split-K over m (blockIdx.y) for the 128-wide weights writes fp32 partial tiles to slabs
//     with 16-byte stores; a second kernel adds the slabs in a fixed order (deterministic, no
//     atomics, no memsets).  Rows past Mrows are sourced from a caller-provided zero page.
//
// Fallback (tr_path = 0): explicit bf16 transposes into scratch + the NT kernel + column sums.
#include "smd_kernels.h"

namespace {

constexpr int BT = 128;          // output tile edge (both Kd and N)
constexpr int BKM = 64;          // m rows per K-tile
constexpr int TILE_BYTES = BKM * BT * 2;   // 16 KiB
constexpr int BUF_BYTES = 2 * TILE_BYTES;
constexpr int STAGE_LD = 132;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;

__device__ __forceinline__ void glds16(const bf16_t* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

union Frag8 {
  bf16x8_t v;
  s16x4_t h[2];
};

// All 8 transpose reads of one k-step + their wait in ONE asm statement (hipcc does not count
// asm loads; with the builtin form it drains vmcnt(0) -- and with it the prefetched LDS-DMA --
// before every read).  a0/a1/b0/b1: LDS byte addresses of the two A / two B fragments at
// k-step 0, row block 0; KOFF = ks*4096 selects the k-step, +1024 the second 4-row block.
template <int KOFF>
__device__ __forceinline__ void tr_read_kstep(unsigned a0, unsigned a1, unsigned b0, unsigned b1,
                                              Frag8 (&af)[2], Frag8 (&bfr)[2]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %2, %9 offset:%12\n\t"
      "ds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
      "ds_read_b64_tr_b16 %4, %10 offset:%12\n\t"
      "ds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
      "ds_read_b64_tr_b16 %6, %11 offset:%12\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:%13\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(af[0].h[0]), "=&v"(af[0].h[1]), "=&v"(af[1].h[0]), "=&v"(af[1].h[1]),
        "=&v"(bfr[0].h[0]), "=&v"(bfr[0].h[1]), "=&v"(bfr[1].h[0]), "=&v"(bfr[1].h[1])
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "i"(KOFF), "i"(KOFF + 1024)
      : "memory");
}

// The same eight reads WITHOUT the wait, and the wait as a statement of its own that names the registers it releases
// ("+v": later uses of the fragments are ordered behind it, and the asm stays where it is written): the reads of k-step
// k+1 are issued before the MFMAs of k-step k, so a one-wave-per-SIMD workgroup (CU-exclusive wgrads) no longer sits
// through an LDS round trip in front of every four MFMAs.  LDS reads return in order: with the next k-step's eight
// reads queued behind them, lgkmcnt(8) means "this k-step's fragments have arrived".
template <int KOFF>
__device__ __forceinline__ void tr_issue_kstep(unsigned a0, unsigned a1, unsigned b0, unsigned b1,
                                               Frag8 (&af)[2], Frag8 (&bfr)[2]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %2, %9 offset:%12\n\t"
      "ds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
      "ds_read_b64_tr_b16 %4, %10 offset:%12\n\t"
      "ds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
      "ds_read_b64_tr_b16 %6, %11 offset:%12\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:%13"
      : "=&v"(af[0].h[0]), "=&v"(af[0].h[1]), "=&v"(af[1].h[0]), "=&v"(af[1].h[1]),
        "=&v"(bfr[0].h[0]), "=&v"(bfr[0].h[1]), "=&v"(bfr[1].h[0]), "=&v"(bfr[1].h[1])
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "i"(KOFF), "i"(KOFF + 1024)
      : "memory");
}
template <int N>
__device__ __forceinline__ void tr_wait_kstep(Frag8 (&af)[2], Frag8 (&bfr)[2]) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(af[0].h[0]), "+v"(af[0].h[1]), "+v"(af[1].h[0]), "+v"(af[1].h[1]),
                 "+v"(bfr[0].h[0]), "+v"(bfr[0].h[1]), "+v"(bfr[1].h[0]), "+v"(bfr[1].h[1])
               : "i"(N)
               : "memory");
}

template <int... Es> struct IntSeq {};
typedef IntSeq<0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15> Seq16;
// 32x32 MFMA C layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
template <int... Es>
__device__ __forceinline__ void stage_tile(const f32x16_t& acc, float* stage, int row0, int col, IntSeq<Es...>) {
  ((stage[(row0 + (Es & 3) + 8 * (Es >> 2)) * STAGE_LD + col] = acc[Es]), ...);
}

struct TnArgs {
  const bf16_t* X; int ldx;
  const bf16_t* dY; int ldy;
  int Mrows, Kd, N;
  float* out; int ldo;            // final dW (nsplit == 1) ...
  float* bias_out;                // ... and db, may be null
  float* slab; size_t slab_stride; // nsplit > 1: partials slab[split][Kd*N (+N)], row stride N
  int tiles_n, ktiles_per_split, nsplit;
  const bf16_t* zero_page;
};

// Several wgrad problems with the same contraction length in one launch (the 128-wide weights of all encoder
// layers: one problem alone is a few hundred short workgroups; grouped, the K loops are long and the launch,
// prologue and slab-reduce costs are paid once).  blockIdx.x walks the concatenated tile lists.
#define SMD_TN_GROUP_MAX 8
struct TnGroupArgs {
  int ngroups;
  int tile_start[SMD_TN_GROUP_MAX + 1];
  TnArgs p[SMD_TN_GROUP_MAX];
};

// NS LDS K-tile buffers: 2 (64 KiB, two workgroups per CU) or 4 (128 KiB; three K-tiles in flight -- for the
// 128-wide weights every workgroup runs few MFMAs per K-tile and a 2-deep pipeline waits on the DMA latency).
// NW = waves that issue the LDS-DMA: 4 (the MFMA waves themselves) or 8 -- four extra LOADER waves that only stage their
// share of every K-tile and take part in the barriers: DMA issue then no longer queues behind the MFMA waves' own phases
// (+2.8 % train step; the raw HBM -> LDS rate itself does not depend on the wave count, tools/dma_waves.hip).
template <int NS, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_tn_128x128_kernel(TnGroupArgs ga) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * BUF_BYTES];

  int gi = 0;
  while (gi + 1 < ga.ngroups && (int)blockIdx.x >= ga.tile_start[gi + 1]) ++gi;
  const TnArgs a = ga.p[gi];
  const int tile = (int)blockIdx.x - ga.tile_start[gi];
  const int tk = tile / a.tiles_n, tn = tile - tk * a.tiles_n;
  const int kd0 = tk * BT, n0 = tn * BT;
  const int kt_begin = blockIdx.y * a.ktiles_per_split;
  const int total_kt = (a.Mrows + BKM - 1) / BKM;
  int kt_end = kt_begin + a.ktiles_per_split;
  kt_end = kt_end < total_kt ? kt_end : total_kt;      // launcher guarantees kt_begin < kt_end

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool mma = w < 4;                                                  // waves 4.. only load
  const int wr = (w & 3) >> 1, wc = w & 1;
  const bool do_bias = (a.bias_out != nullptr) && tk == 0 && wr == 0 && mma;     // wave-uniform

  // ---- DMA sources: wave w piece j covers LDS rows (w*4+j)*4 .. +4 ; 16 lanes per 256-B row.
  // LDS chunk (lane&15) of row r receives global chunk (lane&15) ^ ((r&3)<<2); r&3 == lane>>4.
  const int src_chunk = (lane & 15) ^ ((lane >> 4) << 2);
  int xcol = kd0 + src_chunk * 8;
  int ycol = n0 + src_chunk * 8;
  // keep the 16-B read inside the row: columns past the end only feed discarded outputs
  xcol = xcol + 8 <= a.ldx ? xcol : 0;
  ycol = ycol + 8 <= a.ldy ? ycol : 0;
  constexpr int RPW = BKM / NW;                   // K-tile rows staged per wave (16 or 8), in pieces of 4 rows
  constexpr int NJ = RPW / 4;
  const int piece_row = w * RPW + (lane >> 4);    // + j*4

  auto issue_tile = [&](int kt, int buf) {
    unsigned char* base = smem + buf * BUF_BYTES + w * (RPW * 256);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int m = kt * BKM + piece_row + j * 4;
      const bf16_t* src = m < a.Mrows ? a.X + (size_t)m * a.ldx + xcol : a.zero_page + (lane & 15) * 8;
      glds16(src, base + j * 1024);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int m = kt * BKM + piece_row + j * 4;
      const bf16_t* src = m < a.Mrows ? a.dY + (size_t)m * a.ldy + ycol : a.zero_page + (lane & 15) * 8;
      glds16(src, base + TILE_BYTES + j * 1024);
    }
  };

  // ---- transpose-read addressing (bytes within an operand tile), see header
  const int g = lane >> 4;                 // 16-lane group
  const int rsub = (lane >> 2) & 3;        // row within the 4-row transpose block == row & 3
  const int m_lane = 8 * (g >> 1) + rsub;  // + ks*16 + rd*4
  int a_col[2], b_col[2];                  // swizzled byte offset within the 256-B row
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = (wr * 64 + i * 32 + 16 * (g & 1) + 4 * (lane & 3)) * 2;
    const int cb = (wc * 64 + i * 32 + 16 * (g & 1) + 4 * (lane & 3)) * 2;
    a_col[i] = ((((ca >> 4) ^ (rsub << 2)) << 4) | (ca & 15));
    b_col[i] = ((((cb >> 4) ^ (rsub << 2)) << 4) | (cb & 15)) + TILE_BYTES;
  }

  f32x16_t acc[2][2], acc_b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc_b[i][e] = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  }
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

  const unsigned lds_base = (unsigned)(size_t)(lds_byte_t*)smem;
  // prologue: NS-1 tiles in flight (past the end the last tile is re-requested into a buffer nobody reads, which
  // keeps the vmcnt arithmetic uniform; drained before the epilogue reuses the LDS)
#pragma unroll
  for (int s2 = 0; s2 < NS - 1; ++s2) issue_tile(kt_begin + s2 < kt_end ? kt_begin + s2 : kt_end - 1, s2);
  int buf = 0, wbuf = NS - 1;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    if constexpr (NS == 2) {
      if (kt + 1 < kt_end) {
        issue_tile(kt + 1, wbuf);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      const int nxt = kt + NS - 1;
      issue_tile(nxt < kt_end ? nxt : kt_end - 1, wbuf);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * 2 * NJ) : "memory");      // NS-1 tiles stay in flight
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (mma) {
    const unsigned tb = lds_base + buf * BUF_BYTES + m_lane * 256;
    Frag8 af[2][2], bfr[2][2];       // two k-steps of fragments: one being multiplied, the next one arriving
#define SMD_TN_MMA(S)                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                               \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                               \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[S][i].v, bfr[S][j].v, acc[i][j], 0, 0, 0); \
    if (do_bias) {                                                                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                             \
        acc_b[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bfr[S][j].v, acc_b[j], 0, 0, 0); \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);
#define SMD_TN_ISSUE(KOFF, S) tr_issue_kstep<KOFF>(tb + a_col[0], tb + a_col[1], tb + b_col[0], tb + b_col[1], af[S], bfr[S])
    SMD_TN_ISSUE(0, 0);
    SMD_TN_ISSUE(4096, 1);
    tr_wait_kstep<8>(af[0], bfr[0]);
    SMD_TN_MMA(0)
    SMD_TN_ISSUE(8192, 0);
    tr_wait_kstep<8>(af[1], bfr[1]);
    SMD_TN_MMA(1)
    SMD_TN_ISSUE(12288, 1);
    tr_wait_kstep<8>(af[0], bfr[0]);
    SMD_TN_MMA(0)
    tr_wait_kstep<0>(af[1], bfr[1]);
    SMD_TN_MMA(1)
#undef SMD_TN_MMA
#undef SMD_TN_ISSUE
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    buf = buf + 1 == NS ? 0 : buf + 1;
    wbuf = wbuf + 1 == NS ? 0 : wbuf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the redundant tail requests
  __builtin_amdgcn_s_barrier();
  if (!mma) return;                                    // loader waves are done (barriers only count live waves)

  // ---- destination of this block's partial: final buffers or its split's slab
  float* dst = a.out;
  float* dst_b = a.bias_out;
  int ld = a.ldo;
  if (a.nsplit > 1) {
    dst = a.slab + (size_t)blockIdx.y * a.slab_stride;
    dst_b = dst + (size_t)a.Kd * a.N;
    ld = a.N;
  }
  const int kh = lane >> 5;
  if (do_bias && kh == 0) {                       // row 0 of the ones-MFMA = column sums
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wc * 64 + j * 32 + (lane & 31);
      if (col < a.N) dst_b[col] = acc_b[j][0];
    }
  }
  // ---- dW tile through LDS: 64 rows per pass, each lane stores 4 consecutive columns
  float* stage = reinterpret_cast<float*>(smem);
  const bool vec = (ld % 4 == 0) && ((((size_t)dst) & 15) == 0);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (p > 0) __syncthreads();
    if (wr == p) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          stage_tile(acc[i][j], stage, i * 32 + 4 * kh, wc * 64 + j * 32 + (lane & 31), Seq16{});
    }
    __syncthreads();
    const int c4 = (tid & 31) * 4;
    const int col = n0 + c4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rs = (tid >> 5) + 8 * i;
      const int row = kd0 + p * 64 + rs;
      if (row < a.Kd && col < a.N) {
        const float4 v = *reinterpret_cast<const float4*>(stage + rs * STAGE_LD + c4);
        float* o = dst + (size_t)row * ld + col;
        if (vec && col + 3 < a.N) {
          *reinterpret_cast<float4*>(o) = v;
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < a.N) o[e] = vv[e];
        }
      }
    }
  }
}

// out[i] = sum_s slab[s][i] (dW then db).  64 float4 columns x 4 slab-slices per block; each thread
// adds its slabs in a fixed order and the 4 slices are combined in a fixed order: deterministic.
struct RedProb { const float* slab; size_t stride; int nsplit; size_t n_w; float* out_w; int n_b; float* out_b; int block_start; };
struct RedGroupArgs { int n; RedProb p[SMD_TN_GROUP_MAX]; };
__global__ __launch_bounds__(256) void reduce_slabs_kernel(RedGroupArgs ga) {
  __shared__ float4 red[4][64];
  int g = 0;
  while (g + 1 < ga.n && (int)blockIdx.x >= ga.p[g + 1].block_start) ++g;
  const RedProb pr = ga.p[g];
  const float* __restrict__ slab = pr.slab;
  const size_t stride = pr.stride, n_w = pr.n_w;
  const int nsplit = pr.nsplit, n_b = pr.n_b;
  float* __restrict__ out_w = pr.out_w;
  float* __restrict__ out_b = pr.out_b;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const size_t total = n_w + (out_b ? (size_t)n_b : 0);
  const size_t i = ((size_t)((int)blockIdx.x - pr.block_start) * 64 + lane) * 4;
  float4 s0 = make_float4(0, 0, 0, 0), s1 = s0;
  const bool in = i < total;
  const bool vec = in && (i + 3 < total);
  if (vec) {
    int s = slice;
    for (; s + 4 < nsplit; s += 8) {
      const float4 u = *reinterpret_cast<const float4*>(slab + (size_t)s * stride + i);
      const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)(s + 4) * stride + i);
      s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
      s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
    }
    if (s < nsplit) {
      const float4 u = *reinterpret_cast<const float4*>(slab + (size_t)s * stride + i);
      s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
    }
  } else if (in) {
    float t[4] = {0, 0, 0, 0};
    for (int s = slice; s < nsplit; s += 4)
      for (int e = 0; e < 4; ++e)
        if (i + e < total) t[e] += slab[(size_t)s * stride + i + e];
    s0 = make_float4(t[0], t[1], t[2], t[3]);
  }
  red[slice][lane] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
  __syncthreads();
  if (slice == 0 && in) {
    const float4 a0 = red[0][lane], a1 = red[1][lane], a2 = red[2][lane], a3 = red[3][lane];
    const float r[4] = {(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                        (a0.w + a1.w) + (a2.w + a3.w)};
    for (int e = 0; e < 4; ++e) {
      const size_t k = i + e;
      if (k < n_w) out_w[k] = r[e];
      else if (k < total) out_b[k - n_w] = r[e];
    }
  }
}

// ---- bf16 transpose through LDS: out[c][r] = in[r][c]
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, int ld_in,
                                                             int rows, int cols, bf16_t* __restrict__ out,
                                                             int ld_out) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? in[(size_t)r * ld_in + c] : (bf16_t)0.0f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < ld_out) out[(size_t)c * ld_out + r] = r < rows ? tile[tx][i] : (bf16_t)0.0f;
  }
}

// ---- column sums (fallback-path bias gradients), two deterministic stages
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const bf16_t* __restrict__ dY, int ldy, int rows,
                                                            int cols, int rows_per_chunk,
                                                            float* __restrict__ partial) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r_begin = blockIdx.y * rows_per_chunk;
  int r_end = r_begin + rows_per_chunk;
  r_end = r_end < rows ? r_end : rows;
  if (c >= cols) return;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;      // 4 independent chains keep loads in flight
  int r = r_begin;
  for (; r + 3 < r_end; r += 4) {
    s0 += bf2f(dY[(size_t)r * ldy + c]);
    s1 += bf2f(dY[(size_t)(r + 1) * ldy + c]);
    s2 += bf2f(dY[(size_t)(r + 2) * ldy + c]);
    s3 += bf2f(dY[(size_t)(r + 3) * ldy + c]);
  }
  for (; r < r_end; ++r) s0 += bf2f(dY[(size_t)r * ldy + c]);
  partial[(size_t)blockIdx.y * cols + c] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void colsum_stage2_kernel(const float* __restrict__ partial, int nchunks,
                                                            int cols, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int k = 0;
  for (; k + 3 < nchunks; k += 4) {
    s0 += partial[(size_t)k * cols + c];
    s1 += partial[(size_t)(k + 1) * cols + c];
    s2 += partial[(size_t)(k + 2) * cols + c];
    s3 += partial[(size_t)(k + 3) * cols + c];
  }
  for (; k < nchunks; ++k) s0 += partial[(size_t)k * cols + c];
  out[c] = (s0 + s1) + (s2 + s3);
}

}  // namespace

int launch_transpose_bf16(const bf16_t* in, int ld_in, int rows, int cols, bf16_t* out, int ld_out,
                          hipStream_t st) {
  SMD_ARG_CHECK(in && out && rows > 0 && cols > 0 && ld_in >= cols && ld_out >= rows,
                "transpose_bf16: bad arguments");
  dim3 grid((cols + 63) / 64, (ld_out + 63) / 64);
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, st, in, ld_in, rows, cols, out, ld_out);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_colsum_bf16(const bf16_t* dY, int ldy, int rows, int cols, float* out, float* partial,
                       size_t partial_elems, hipStream_t st) {
  SMD_ARG_CHECK(dY && out && partial && rows > 0 && cols > 0, "colsum_bf16: bad arguments");
  int nchunks = (rows + 63) / 64;
  if (nchunks > 128) nchunks = 128;
  const int rows_per_chunk = (rows + nchunks - 1) / nchunks;
  nchunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
  SMD_ARG_CHECK(partial_elems >= (size_t)nchunks * cols, "colsum_bf16: workspace too small");
  dim3 grid((cols + 255) / 256, nchunks);
  hipLaunchKernelGGL(colsum_stage1_kernel, grid, dim3(256), 0, st, dY, ldy, rows, cols, rows_per_chunk,
                     partial);
  SMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_stage2_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, partial, nchunks,
                     cols, out);
  SMD_LAUNCH_CHECK();
  return 0;
}

// room for 4 split partials of a 2048x2048 weight (gemm_tn256) + bias partial rows
int smd_tn_pad_bytes(int static_lds_bytes) {
  if (smd_tuning_get("tn_exclusive_cu") != 1) return 0;      // 2: the exclusive launch's kernels WITHOUT the pad (A/B)
  const int pad = 160 * 1024 - static_lds_bytes;
  return pad > 0 ? pad : 0;
}

// Split-K factor of a 128x128-tile launch.  With CU-exclusive workgroups (tn_exclusive_cu) a launch runs in whole rounds of
// 256 workgroups, so 36 tiles x 15 splits = 540 workgroups (the old "about 512" rule) took THREE rounds of 9 K-tiles;
// 7 splits = 252 workgroups take one round of 19.  Cost in K-tile units: rounds x (K-tiles per split + a fixed prologue /
// slab-epilogue share) + the slab traffic the split adds; the smallest wins, ties go to fewer splits.
static int smd_tn_pick_split(int tiles, int total_kt, int max_split) {
  if (!smd_tuning_get("tn_split_model")) {
    int ns = (512 + tiles - 1) / tiles;
    return ns < 1 ? 1 : (ns > max_split ? max_split : ns);
  }
  const int per_round = smd_tuning_get("tn_exclusive_cu") ? 256 : 512;
  int best = 1;
  float best_cost = 1e30f;
  for (int ns = 1; ns <= max_split && ns <= total_kt; ++ns) {
    const int per = (total_kt + ns - 1) / ns;
    const int rounds = (tiles * ns + per_round - 1) / per_round;
    const float cost = (float)rounds * ((float)per + 5.0f) + (ns > 1 ? 0.35f * (float)ns : 0.0f);
    if (cost < best_cost - 1e-3f) { best_cost = cost; best = ns; }
  }
  return best;
}

// CU-exclusive launch: four MFMA waves + four loader waves (tn128_loader_waves = 0: the four MFMA waves load themselves).
// Knob "tn_mode" (A/B experiments) overrides the choice: NS*100 + NW*10 + pad, pad 0 = none, 1 = fill the
// CU's 160 KiB, 2 = pad the workgroup to 96 KiB (one wgrad workgroup per CU, small-LDS workgroups may still share it).
static int smd_tn_launch_128(int tiles, int nsplit, const TnGroupArgs& ga, bool exclusive_default, hipStream_t st) {
  int mode = smd_tuning_get("tn_mode");
  if (!mode) {
    if (exclusive_default) mode = 400 + (smd_tuning_get("tn128_loader_waves") ? 80 : 40) + (smd_tuning_get("tn_exclusive_cu") == 1 ? 1 : 0);
    else mode = 240;
  }
  const int ns = mode / 100, nw = (mode / 10) % 10, padc = mode % 10;
  const int lds = ns * BUF_BYTES;
  int pad = 0;
  if (padc == 1) pad = 160 * 1024 - lds;
  else if (padc == 2) pad = 96 * 1024 - lds;
  if (pad < 0) pad = 0;
  const dim3 grid(tiles, nsplit);
  // Only <4, 8> ships.  The two-buffer instantiations (and, at a lower rate, <4, 4>) disturb a dependent VALU -> v_rsq_f32 pair
  // of small LayerNorm workgroups that share their CU -- reproduced without the engine by tools/rsq_repro.hip, 53 of 3000
  // victim launches wrong next to <2, 8> even with the guarded instruction -- so they exist only in an
  // experiment build (-DSMD_TN_EXPERIMENTS: tools/build_rsq_repro.sh) and a knob that asks for them fails loudly here.
  if (ns == 4 && nw == 8) hipLaunchKernelGGL((gemm_tn_128x128_kernel<4, 8>), grid, dim3(512), pad, st, ga);
#ifdef SMD_TN_EXPERIMENTS
  else if (ns == 2 && nw == 4) hipLaunchKernelGGL((gemm_tn_128x128_kernel<2, 4>), grid, dim3(256), pad, st, ga);
  else if (ns == 2 && nw == 8) hipLaunchKernelGGL((gemm_tn_128x128_kernel<2, 8>), grid, dim3(512), pad, st, ga);
  else if (ns == 4 && nw == 4) hipLaunchKernelGGL((gemm_tn_128x128_kernel<4, 4>), grid, dim3(256), pad, st, ga);
#endif
  else {
    smd_set_error("gemm_tn: kernel variant %d (tn_mode / tn_exclusive_cu / tn128_loader_waves) is not in this build: only the "
                  "four-buffer kernel with loader waves ships, the others need -DSMD_TN_EXPERIMENTS", mode);
    return -1;
  }
  return 0;
}

size_t gemm_tn_slab_elems() { return (size_t)4 * 2048 * 2048 + (size_t)1024 * 1024; }

int launch_gemm_tn(const TnLaunch& t, hipStream_t st) {
  SMD_ARG_CHECK(t.X && t.dY && t.out, "gemm_tn: null operand");
  SMD_ARG_CHECK(t.Mrows > 0 && t.Kd > 0 && t.N > 0, "gemm_tn: bad shape");
  SMD_ARG_CHECK(t.ldx % 8 == 0 && t.ldy % 8 == 0 && t.ldx >= 8 && t.ldy >= 8 && t.ldo >= t.N,
                "gemm_tn: ldx/ldy must be multiples of 8, ldo >= N");
  if (t.tr_path) {
    int per256 = 0;
    if (const int ns256 = gemm_tn256_plan(t, &per256)) return launch_gemm_tn256(t, ns256, per256, st);
    SMD_ARG_CHECK(t.zero_page, "gemm_tn: needs a 128-element zeroed bf16 page");
    const int tiles_k = (t.Kd + BT - 1) / BT, tiles_n = (t.N + BT - 1) / BT;
    const int tiles = tiles_k * tiles_n;
    const int total_kt = (t.Mrows + BKM - 1) / BKM;
    const size_t stride = ((size_t)t.Kd * t.N + t.N + 3) / 4 * 4;
    int nsplit = 1;
    // workgroups wanted on the chip (split-K over m): 512 (two per CU); 256 for the 128-wide weights, where the
    // slab traffic of 32 splits costs more than the second workgroup per CU gains (kbench --tn128)
    int target = smd_tuning_get("tn128_target_wgs");
    if (target == 512 && tiles <= 16) target = 256;
    if (tiles < target && t.slab && t.ldo == t.N) {
      nsplit = smd_tuning_get("tn_split_model") ? smd_tn_pick_split(tiles, total_kt, 32) : (target + tiles - 1) / tiles;
      if (nsplit > 32) nsplit = 32;
      if (nsplit > total_kt) nsplit = total_kt;
      const size_t cap = t.slab_elems / stride;
      if ((size_t)nsplit > cap) nsplit = (int)cap;
      if (nsplit < 1) nsplit = 1;
    }
    const int per = (total_kt + nsplit - 1) / nsplit;
    nsplit = (total_kt + per - 1) / per;
    TnGroupArgs ga;
    ga.ngroups = 1; ga.tile_start[0] = 0; ga.tile_start[1] = tiles;
    TnArgs& a = ga.p[0];
    a.X = t.X; a.ldx = t.ldx; a.dY = t.dY; a.ldy = t.ldy; a.Mrows = t.Mrows; a.Kd = t.Kd; a.N = t.N;
    a.out = t.out; a.ldo = t.ldo; a.bias_out = t.bias_out; a.slab = t.slab; a.slab_stride = stride;
    a.tiles_n = tiles_n; a.ktiles_per_split = per; a.nsplit = nsplit; a.zero_page = t.zero_page;
    if (int rc = smd_tn_launch_128(tiles, nsplit, ga, smd_tuning_get("tn_exclusive_cu") || (per >= 6 && smd_tuning_get("gemm_tn_deep")), st)) return rc;
    SMD_LAUNCH_CHECK();
    if (nsplit > 1) {
      RedGroupArgs ra;
      ra.n = 1;
      ra.p[0] = RedProb{t.slab, stride, nsplit, (size_t)t.Kd * t.N, t.out, t.N, t.bias_out, 0};
      const size_t total = ra.p[0].n_w + (t.bias_out ? t.N : 0);
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ra);
      SMD_LAUNCH_CHECK();
    }
    return 0;
  }
  // ---- fallback: explicit transposed copies, then the NT kernel (contraction = Mrows padded to 64)
  const int Mp = (t.Mrows + 63) / 64 * 64;
  SMD_ARG_CHECK(t.scratch && t.scratch_elems >= (size_t)(t.Kd + t.N) * Mp, "gemm_tn: scratch too small for fallback");
  bf16_t* Xt = t.scratch;
  bf16_t* Yt = t.scratch + (size_t)t.Kd * Mp;
  int rc = launch_transpose_bf16(t.X, t.ldx, t.Mrows, t.Kd, Xt, Mp, st);
  if (rc) return rc;
  rc = launch_transpose_bf16(t.dY, t.ldy, t.Mrows, t.N, Yt, Mp, st);
  if (rc) return rc;
  GemmEpilogue ep;
  ep.out_f32 = t.out;
  ep.ld_out = t.ldo;
  rc = launch_gemm_nt(Xt, Mp, Yt, Mp, t.Kd, t.N, Mp, ep, st);
  if (rc) return rc;
  if (t.bias_out) {
    SMD_ARG_CHECK(t.slab && t.slab_elems >= (size_t)128 * t.N, "gemm_tn: fallback bias needs the slab workspace");
    rc = launch_colsum_bf16(t.dY, t.ldy, t.Mrows, t.N, t.bias_out, t.slab, t.slab_elems, st);
  }
  return rc;
}

// Grouped launch of the 128-wide kernel: n problems (<= 8 per launch) sharing Mrows, each with ldo == N; one
// split count for all, slabs carved from the first problem's workspace, one grouped reduce.
int launch_gemm_tn_grouped(const TnLaunch* probs, int n, hipStream_t st) {
  int i0 = 0;
  while (i0 < n) {
    const int cnt = (n - i0) < SMD_TN_GROUP_MAX ? (n - i0) : SMD_TN_GROUP_MAX;
    const TnLaunch& t0 = probs[i0];
    SMD_ARG_CHECK(t0.zero_page && t0.slab && t0.tr_path, "gemm_tn_grouped: needs zero page, slab workspace and tr_path = 1");
    const int total_kt = (t0.Mrows + BKM - 1) / BKM;
    TnGroupArgs ga;
    ga.ngroups = cnt;
    int tiles = 0;
    size_t slab_per_split = 0;
    for (int i = 0; i < cnt; ++i) {
      const TnLaunch& t = probs[i0 + i];
      SMD_ARG_CHECK(t.X && t.dY && t.out && t.Mrows == t0.Mrows && t.ldo == t.N && t.ldx % 8 == 0 && t.ldy % 8 == 0,
                    "gemm_tn_grouped: problem %d incompatible", i0 + i);
      ga.tile_start[i] = tiles;
      tiles += ((t.Kd + BT - 1) / BT) * ((t.N + BT - 1) / BT);
      slab_per_split += ((size_t)t.Kd * t.N + t.N + 3) / 4 * 4;
    }
    ga.tile_start[cnt] = tiles;
    int nsplit = smd_tn_pick_split(tiles, total_kt, 32);
    if (nsplit > 32) nsplit = 32;
    if (nsplit > total_kt) nsplit = total_kt;
    if ((size_t)nsplit * slab_per_split > t0.slab_elems) nsplit = (int)(t0.slab_elems / slab_per_split);
    if (nsplit < 1) nsplit = 1;            // one split writes the gradients directly, no slab needed
    const int per = (total_kt + nsplit - 1) / nsplit;
    nsplit = (total_kt + per - 1) / per;
    RedGroupArgs ra;
    ra.n = cnt;
    size_t slab_off = 0;
    int blocks = 0;
    for (int i = 0; i < cnt; ++i) {
      const TnLaunch& t = probs[i0 + i];
      const size_t stride = ((size_t)t.Kd * t.N + t.N + 3) / 4 * 4;
      TnArgs& a = ga.p[i];
      a.X = t.X; a.ldx = t.ldx; a.dY = t.dY; a.ldy = t.ldy; a.Mrows = t.Mrows; a.Kd = t.Kd; a.N = t.N;
      a.out = t.out; a.ldo = t.ldo; a.bias_out = t.bias_out; a.slab = t0.slab + slab_off; a.slab_stride = stride;
      a.tiles_n = (t.N + BT - 1) / BT; a.ktiles_per_split = per; a.nsplit = nsplit; a.zero_page = t0.zero_page;
      ra.p[i] = RedProb{a.slab, stride, nsplit, (size_t)t.Kd * t.N, t.out, t.N, t.bias_out, blocks};
      blocks += (int)(((size_t)t.Kd * t.N + (t.bias_out ? t.N : 0) + 255) / 256);
      slab_off += (size_t)nsplit * stride;
    }
    // tn_exclusive_cu != 0 (default 2): the four-buffer instantiation (1: padded to the CU's whole LDS, see smd_tn_pad_bytes())
    if (int rc = smd_tn_launch_128(tiles, nsplit, ga, smd_tuning_get("tn_exclusive_cu") != 0, st)) return rc;
    SMD_LAUNCH_CHECK();
    if (nsplit > 1) {
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, ra);
      SMD_LAUNCH_CHECK();
    }
    i0 += cnt;
  }
  return 0;
}

// ---- debug probe: what does ds_read_b64_tr_b16 return for a linear image with lane address lane*8 ?
namespace {
__global__ void probe_tr_read_kernel(const bf16_t* __restrict__ image, bf16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) bf16_t img[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) img[i] = image[i];
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)(lds_byte_t*)img + threadIdx.x * 8;
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  union { s16x4_t s; bf16x4_t b; } u;
  u.s = v;
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = u.b[j];
}
}  // namespace
int launch_probe_tr_read(const bf16_t* image, bf16_t* out, hipStream_t st) {
  SMD_ARG_CHECK(image && out, "probe_tr_read: null pointer");
  hipLaunchKernelGGL(probe_tr_read_kernel, dim3(1), dim3(64), 0, st, image, out);
  SMD_LAUNCH_CHECK();
  return 0;
}
