// Stand-alone reproducer for the co-residency event of DESIGN.md section 6 (no engine, no Python): a dependent
// VALU -> v_rsq_f32 pair of a small-LDS "victim" workgroup reads a stale source register in some lanes while a
// weight-gradient workgroup of the two-buffer gemm_tn_128x128 kernel is resident on the same CU.
//
//   victim     the row statistic of the 128-wide LayerNorm backward (csrc/encoder_fused.hip ln128_bwd_parts_kernel): a wave
//              holds 8 rows x 128 floats (2 per lane), sums x and x^2 over the wave (ds_bpermute butterflies), then
//              var = m2/128 - mean^2, and  v_add_f32 t, var, eps ; v_rsq_f32 r, t  back to back.  Every lane writes its r:
//              all 64 lanes of a row must hold the same bits, and the same bits as in a quiet run.
//              Forms: 0 = bare pair (what hipcc schedules), 1 = s_nop 7 in FRONT of v_rsq_f32 (the shipped smd_ln_rstd),
//              2 = s_nop 7 BEHIND it; 3 = the LIBRARY's own ln128_bwd_parts kernel from a build with the bare instruction
//              (tools/build_rsq_repro.sh), 4 = the shipped build's; 5..12 = an in-tool copy of that kernel that dumps what each lane
//              held (rsq argument, rsq result, first use, second reduction) in variants of the instruction sequence; 13 / 14 = the
//              library kernel with round 3's code generation (packed-fp32 arithmetic: SMD_SLP=1), bare / guarded.
//   aggressor  0 none (quiet), 1 the library's two-buffer 128x128 TN kernel (tn_mode 240), 2 the shipped four-buffer kernel
//              with loader waves (tn_mode 480), 3 a plain one-wave copy + reduce kernel with sc0 sc1 (system-scope) loads and
//              stores and 4 KiB of LDS -- the shape of RCCL's ring kernels, 4 the two-buffer kernel with four extra loader
//              waves (tn_mode 280).
// Two streams, the aggressor launched back to back on one, the victim + an on-device checker on the other; per
// (aggressor, form): victim launches, launches with at least one wrong element, wrong elements by lane.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/rsq_repro tools/rsq_repro.hip -ldl && tools/rsq_repro [victim launches per cell]
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
  } while (0)

constexpr int E_DIM = 128;
constexpr float LN_EPS = 1e-6f;

template <int FORM>
__device__ __forceinline__ float rstd_pair(float var, float eps) {
  float r;
  if constexpr (FORM == 0) asm volatile("v_add_f32 %0, %1, %2\n\tv_rsq_f32 %0, %0" : "=&v"(r) : "v"(var), "v"(eps));
  else if constexpr (FORM == 1) asm volatile("v_add_f32 %0, %1, %2\n\ts_nop 7\n\tv_rsq_f32 %0, %0\n\ts_nop 0" : "=&v"(r) : "v"(var), "v"(eps));
  else asm volatile("v_add_f32 %0, %1, %2\n\tv_rsq_f32 %0, %0\n\ts_nop 7" : "=&v"(r) : "v"(var), "v"(eps));
  return r;
}

template <int FORM>
__global__ __launch_bounds__(256) void victim_kernel(const float* __restrict__ x, float* __restrict__ out) {
  __shared__ float pad[1024];                               // 4 KiB like the kernel it stands for
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t r0 = (size_t)blockIdx.x * 32 + w * 8;
  float2 xv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const float2*>(x + (r0 + i) * E_DIM + lane * 2);
  float st[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) { st[2 * i] = xv[i].x + xv[i].y; st[2 * i + 1] = xv[i].x * xv[i].x + xv[i].y * xv[i].y; }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] += __shfl_xor(st[i], o, 64);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float mean = st[2 * i] * (1.0f / E_DIM);
    const float var = st[2 * i + 1] * (1.0f / E_DIM) - mean * mean;
    out[(r0 + i) * 64 + lane] = rstd_pair<FORM>(var, LN_EPS);
  }
  if (threadIdx.x == 1023) pad[0] = 0.f;                    // keep the allocation
}

// victim form 5 / 6: the library kernel's arithmetic (csrc/encoder_fused.hip ln128_bwd_parts_kernel<true, true, false>) copied here so that
// it can DUMP what each lane held: var_out = the argument of the row statistic (after the wave reduction), rs_out = 1/sqrt of
// it.  FORM 5: bare __builtin_amdgcn_rsqf, FORM 6: s_nop 7 in front.
template <int R_, int DPP>
__device__ __forceinline__ void wave_allreduce_sum(float (&v)[R_]) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
    for (int i = 0; i < R_; ++i) v[i] += __shfl_xor(v[i], o, 64);
  }
}
template <int FORM>
__global__ __launch_bounds__(256) void lnb_copy_kernel(const float* __restrict__ x, const float* __restrict__ parts, size_t part_stride,
                                                       const float* __restrict__ gamma, const float* dres, float* dx_f32,
                                                       float* __restrict__ partial, float* __restrict__ var_out, float* __restrict__ rs_out,
                                                       float* __restrict__ xh_out, float* __restrict__ t2_out) {
  __shared__ float red[4][2][E_DIM];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t r0 = (size_t)blockIdx.x * 32 + w * 8;
  const float2 g2 = *reinterpret_cast<const float2*>(gamma + lane * 2);
  float2 xv[8], dv[8], rv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const size_t o = (r0 + i) * E_DIM + lane * 2;
    xv[i] = *reinterpret_cast<const float2*>(x + o);
    const float2 p0 = *reinterpret_cast<const float2*>(parts + o), p1 = *reinterpret_cast<const float2*>(parts + part_stride + o);
    const float2 p2 = *reinterpret_cast<const float2*>(parts + 2 * part_stride + o), p3 = *reinterpret_cast<const float2*>(parts + 3 * part_stride + o);
    dv[i].x = (p0.x + p1.x) + (p2.x + p3.x);
    dv[i].y = (p0.y + p1.y) + (p2.y + p3.y);
    rv[i] = *reinterpret_cast<const float2*>(dres + o);
  }
  float st[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) { st[2 * i] = xv[i].x + xv[i].y; st[2 * i + 1] = xv[i].x * xv[i].x + xv[i].y * xv[i].y; }
  wave_allreduce_sum<16, 0>(st);
  float Px = 0.f, Py = 0.f, Qx = 0.f, Qy = 0.f;
  float tt[16], rs[8], vr[8];
  float2 xh[8], dxh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float mean = st[2 * i] * (1.0f / E_DIM);
    vr[i] = st[2 * i + 1] * (1.0f / E_DIM) - mean * mean + LN_EPS;
    if constexpr (FORM == 6) asm volatile("s_nop 7\n\tv_rsq_f32 %0, %1\n\ts_nop 0" : "=v"(rs[i]) : "v"(vr[i]));
    else if constexpr (FORM == 9 || FORM == 12) {           // no transcendental instruction at all: magic-number guess + 4 Newton steps (plain FMAs)
      float y = __uint_as_float(0x5f3759dfu - (__float_as_uint(vr[i]) >> 1));
#pragma unroll
      for (int it = 0; it < 4; ++it) y = y * (1.5f - 0.5f * vr[i] * y * y);
      rs[i] = y;
    } else rs[i] = __builtin_amdgcn_rsqf(vr[i]);
    if constexpr (FORM == 7) asm volatile("s_nop 7" : "+v"(rs[i]));                       // eight idle cycles BEHIND it
    if constexpr (FORM == 10) asm volatile("v_mov_b32 %0, %0" : "+v"(rs[i]));             // a plain (unpacked) VALU consumer first
    if constexpr (FORM == 11) asm volatile("s_nop 7\n\tv_mov_b32 %0, %0\n\ts_nop 1" : "+v"(rs[i]));
    if constexpr (FORM == 8 || FORM == 12) {  // the first uses as UNPACKED multiplies (hipcc forms v_pk_mul_f32 on the register pair);
      const float dx_ = xv[i].x - mean, dy_ = xv[i].y - mean;      // s_nop 1: the trans-forwarding wait state hipcc cannot see inside asm
      asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(xh[i].x) : "v"(dx_), "v"(rs[i]));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(xh[i].y) : "v"(dy_), "v"(rs[i]));
    } else {
      xh[i].x = (xv[i].x - mean) * rs[i];
      xh[i].y = (xv[i].y - mean) * rs[i];
    }
    Qx += dv[i].x; Qy += dv[i].y;
    Px += dv[i].x * xh[i].x; Py += dv[i].y * xh[i].y;
    dxh[i].x = dv[i].x * g2.x; dxh[i].y = dv[i].y * g2.y;
    tt[2 * i] = dxh[i].x + dxh[i].y;
    tt[2 * i + 1] = dxh[i].x * xh[i].x + dxh[i].y * xh[i].y;
  }
  wave_allreduce_sum<16, 0>(tt);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float t1 = tt[2 * i] * (1.0f / E_DIM), t2 = tt[2 * i + 1] * (1.0f / E_DIM);
    float2 o;
    o.x = rs[i] * (dxh[i].x - t1 - xh[i].x * t2) + rv[i].x;
    o.y = rs[i] * (dxh[i].y - t1 - xh[i].y * t2) + rv[i].y;
    *reinterpret_cast<float2*>(dx_f32 + (r0 + i) * E_DIM + lane * 2) = o;
    var_out[(r0 + i) * 64 + lane] = vr[i];
    rs_out[(r0 + i) * 64 + lane] = rs[i];
    xh_out[(r0 + i) * 64 + lane] = xh[i].x;
    t2_out[(r0 + i) * 64 + lane] = t2;
  }
  red[w][0][lane * 2] = Px; red[w][0][lane * 2 + 1] = Py;
  red[w][1][lane * 2] = Qx; red[w][1][lane * 2 + 1] = Qy;
  __syncthreads();
  {
    const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
    partial[((size_t)blockIdx.x * 2 + which) * E_DIM + c] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
  }
}

// hist[lane] += wrong elements of that lane; hist[64] += 1 when the launch had any
__global__ __launch_bounds__(256) void check_kernel(const float* __restrict__ out, const float* __restrict__ ref, size_t n,
                                                    unsigned* __restrict__ hist, int width, int per_lane) {
  __shared__ unsigned any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (__float_as_uint(out[i]) != __float_as_uint(ref[i])) { atomicAdd(&hist[(i % width) / per_lane], 1u); any = 1; }
  }
  __syncthreads();
  if (threadIdx.x == 0 && any) atomicAdd(&hist[65 + (blockIdx.x & 0)], 1u);      // blocks that saw one (diagnostic)
}
__global__ void mark_launch_kernel(unsigned* hist, unsigned* last_total) {
  unsigned t = 0;
  for (int i = 0; i < 64; ++i) t += hist[i];
  if (t != *last_total) { hist[64] += 1; *last_total = t; }
}

// aggressor 3: one wave per workgroup, 4 KiB LDS, system-scope loads / stores: dst = a + b (a ring reduce step)
typedef __attribute__((ext_vector_type(4))) float vf4;
__global__ __launch_bounds__(64) void ring_like_kernel(const vf4* __restrict__ a, const vf4* __restrict__ b, vf4* __restrict__ dst,
                                                       size_t n4) {
  __shared__ vf4 stage[256];
  for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 64) {
    vf4 u, v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(u) : "v"(a + i) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(b + i) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stage[threadIdx.x] = u + v;
    vf4 o = stage[threadIdx.x ^ 1];
    o.x += u.x;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + i), "v"(o) : "memory");
  }
}

typedef int (*tn_fn)(const void*, int, const void*, int, int, int, int, float*, int, float*, const void*, float*, int64_t, void*, int64_t,
                     int, void*);
typedef int (*tune_fn)(const char*, int);
typedef int64_t (*slab_fn)(void);
typedef const char* (*err_fn)(void);
typedef int (*lnb_fn)(const float*, const float*, int64_t, int, const float*, const float*, float*, void*, float*, void*);

int main(int argc, char** argv) {
  const int per_cell = argc > 1 ? atoi(argv[1]) : 5000;
  // the library next to this tool: <repo>/symbolic-music-diffusion_amd/csrc/libsmd_hip.so
  std::string self = argv[0];
  std::string dir = self.substr(0, self.find_last_of('/') == std::string::npos ? 0 : self.find_last_of('/'));
  if (dir.empty()) dir = ".";
  const std::string base = dir + "/../symbolic-music-diffusion_amd/csrc/";
  const std::string so_ship = base + "libsmd_hip.so";
  void* lib_ship = dlopen(so_ship.c_str(), RTLD_NOW | RTLD_LOCAL);
  // aggressors come from the experiment build (tools/build_rsq_repro.sh); the shipped library only holds <4,8>
  std::string so = base + "libsmd_hip_tnx.so";
  void* lib = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!lib) { so = so_ship; lib = lib_ship; fprintf(stderr, "rsq_repro: libsmd_hip_tnx.so not found: only the shipped aggressor is available\n"); }
  tn_fn gemm_tn = lib ? (tn_fn)dlsym(lib, "smd_gemm_bf16_tn") : nullptr;
  tune_fn tune = lib ? (tune_fn)dlsym(lib, "smd_set_tuning") : nullptr;
  slab_fn slab_elems = lib ? (slab_fn)dlsym(lib, "smd_gemm_tn_slab_elems") : nullptr;
  err_fn last_err = lib ? (err_fn)dlsym(lib, "smd_last_error") : nullptr;
  lnb_fn ln_ship = lib_ship ? (lnb_fn)dlsym(lib_ship, "smd_ln128_bwd_parts") : nullptr;
  void* lib_bare = dlopen((base + "libsmd_hip_tsm1.so").c_str(), RTLD_NOW | RTLD_LOCAL);
  lnb_fn ln_bare = lib_bare ? (lnb_fn)dlsym(lib_bare, "smd_ln128_bwd_parts") : nullptr;
  if (!ln_bare) fprintf(stderr, "rsq_repro: libsmd_hip_tsm1.so (bare v_rsq_f32 build) not found: victim form 3 skipped\n");
  void* lib_slpbare = dlopen((base + "libsmd_hip_slpbare.so").c_str(), RTLD_NOW | RTLD_LOCAL);
  lnb_fn ln_slpbare = lib_slpbare ? (lnb_fn)dlsym(lib_slpbare, "smd_ln128_bwd_parts") : nullptr;
  void* lib_slp = dlopen((base + "libsmd_hip_slp.so").c_str(), RTLD_NOW | RTLD_LOCAL);
  lnb_fn ln_slp = lib_slp ? (lnb_fn)dlsym(lib_slp, "smd_ln128_bwd_parts") : nullptr;
  if (!gemm_tn || !tune || !slab_elems) fprintf(stderr, "rsq_repro: %s not loadable (%s): library aggressors skipped\n", so.c_str(), dlerror());

  const int R = 8192;                                       // token rows of the benchmark batch
  std::vector<float> hx((size_t)R * E_DIM);
  uint32_t s = 12345u;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 8388608.0f - 1.0f) * 2.0f; }
  float *x, *out, *ref, *parts, *gamma, *dres, *lnpart;
  unsigned *hist, *last_total;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&out, (size_t)R * 128 * 4)); CK(hipMalloc(&ref, (size_t)R * 128 * 4));
  CK(hipMalloc(&parts, (size_t)4 * R * E_DIM * 4)); CK(hipMalloc(&gamma, E_DIM * 4)); CK(hipMalloc(&dres, (size_t)R * E_DIM * 4));
  CK(hipMalloc(&lnpart, (size_t)(R / 32) * 2 * E_DIM * 4));
  float *var_o, *rs_o, *var_ref, *rs_ref, *xh_o, *xh_ref, *t2_o, *t2_ref;
  unsigned *hist_var, *hist_rs, *lt2, *lt3, *hist_xh, *hist_t2, *lt4, *lt5;
  CK(hipMalloc(&xh_o, (size_t)R * 64 * 4)); CK(hipMalloc(&xh_ref, (size_t)R * 64 * 4)); CK(hipMalloc(&t2_o, (size_t)R * 64 * 4)); CK(hipMalloc(&t2_ref, (size_t)R * 64 * 4));
  CK(hipMalloc(&hist_xh, 128 * 4)); CK(hipMalloc(&hist_t2, 128 * 4)); CK(hipMalloc(&lt4, 4)); CK(hipMalloc(&lt5, 4));
  CK(hipMalloc(&var_o, (size_t)R * 64 * 4)); CK(hipMalloc(&rs_o, (size_t)R * 64 * 4)); CK(hipMalloc(&var_ref, (size_t)R * 64 * 4)); CK(hipMalloc(&rs_ref, (size_t)R * 64 * 4));
  CK(hipMalloc(&hist_var, 128 * 4)); CK(hipMalloc(&hist_rs, 128 * 4)); CK(hipMalloc(&lt2, 4)); CK(hipMalloc(&lt3, 4));
  {
    std::vector<float> hp((size_t)4 * R * E_DIM), hg(E_DIM), hd((size_t)R * E_DIM);
    for (auto& v : hp) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 8388608.0f - 1.0f) * 1e-6f; }
    for (auto& v : hg) { s = s * 1664525u + 1013904223u; v = 1.0f + ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.1f; }
    for (auto& v : hd) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 8388608.0f - 1.0f) * 1e-6f; }
    CK(hipMemcpy(parts, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(gamma, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dres, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&hist, 128 * 4)); CK(hipMalloc(&last_total, 4));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  // aggressor operands: the encoder's fc1 weight gradient (X = a2 [8192][128], dY = dz [8192][2048]) and its fc2 twin
  const int M = R, Kd = 128, N = 2048;
  void *X, *dY, *zero, *scratch;
  float *dW, *db, *slab = nullptr;
  CK(hipMalloc(&X, (size_t)M * Kd * 2)); CK(hipMalloc(&dY, (size_t)M * N * 2)); CK(hipMalloc(&zero, 256)); CK(hipMalloc(&scratch, 256));
  CK(hipMalloc(&dW, (size_t)Kd * N * 4)); CK(hipMalloc(&db, (size_t)N * 4));
  CK(hipMemset(X, 0x3c, (size_t)M * Kd * 2)); CK(hipMemset(dY, 0x3c, (size_t)M * N * 2)); CK(hipMemset(zero, 0, 256));
  const int64_t nslab = slab_elems ? slab_elems() : 0;
  if (nslab) CK(hipMalloc(&slab, (size_t)nslab * 4));
  vf4 *ra, *rb_, *rd;
  const size_t n4 = (size_t)8 << 20;                        // 128 MiB per ring buffer
  CK(hipMalloc(&ra, n4 * 16)); CK(hipMalloc(&rb_, n4 * 16)); CK(hipMalloc(&rd, n4 * 16));
  CK(hipMemset(ra, 0, n4 * 16)); CK(hipMemset(rb_, 0, n4 * 16));
  hipStream_t sa, sv;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));

  auto victim = [&](int form, float* dst) {
    if (form >= 5 && form <= 12) {
      float* vo = dst == ref ? var_ref : var_o;
      float* ro = dst == ref ? rs_ref : rs_o;
      float* xo = dst == ref ? xh_ref : xh_o;
      float* to = dst == ref ? t2_ref : t2_o;
#define LNB(F) hipLaunchKernelGGL(lnb_copy_kernel<F>, dim3(R / 32), dim3(256), 0, sv, x, parts, (size_t)R * E_DIM, gamma, dres, dst, lnpart, vo, ro, xo, to)
      switch (form) { case 5: LNB(5); break; case 6: LNB(6); break; case 7: LNB(7); break; case 8: LNB(8); break; case 9: LNB(9); break;
                      case 10: LNB(10); break; case 11: LNB(11); break; default: LNB(12); break; }
#undef LNB
      return;
    }
    if (form == 3 || form == 4 || form >= 13) {
      lnb_fn f = form == 3 ? ln_bare : (form == 4 ? ln_ship : (form == 13 ? ln_slpbare : ln_slp));
      if (f(x, parts, (int64_t)R * E_DIM, R, gamma, dres, dst, nullptr, lnpart, sv)) { fprintf(stderr, "victim launch failed\n"); exit(1); }
      return;
    }
    if (form == 0) hipLaunchKernelGGL(victim_kernel<0>, dim3(R / 32), dim3(256), 0, sv, x, dst);
    else if (form == 1) hipLaunchKernelGGL(victim_kernel<1>, dim3(R / 32), dim3(256), 0, sv, x, dst);
    else hipLaunchKernelGGL(victim_kernel<2>, dim3(R / 32), dim3(256), 0, sv, x, dst);
  };
  auto aggress = [&](int kind) -> bool {                    // one launch on stream sa
    if (kind == 3) { hipLaunchKernelGGL(ring_like_kernel, dim3(1024), dim3(64), 0, sa, ra, rb_, rd, n4); return true; }
    if (!gemm_tn) return false;
    const int rc = gemm_tn(X, Kd, dY, N, M, Kd, N, dW, N, db, zero, slab, nslab, scratch, 0, 1, sa);
    if (rc) { static int told = 0; if (!told++) fprintf(stderr, "aggressor launch failed: %s\n", last_err ? last_err() : "?"); return false; }
    return true;
  };
  const char* names[5] = {"none (quiet)", "two-buffer gemm_tn_128x128<2,4> (tn_mode 240)", "four-buffer + loader waves <4,8> (tn_mode 480, shipped)",
                          "one-wave copy+reduce, sc0 sc1, 4 KiB LDS (RCCL-like)", "two-buffer + loader waves <2,8> (tn_mode 280)"};
  const char* forms[15] = {"bare v_add -> v_rsq", "s_nop 7 in front (shipped)", "s_nop 7 behind", "library ln128_bwd_parts, bare v_rsq build",
                          "library ln128_bwd_parts, shipped build", "in-tool copy of ln128_bwd_parts, bare v_rsq, dumps", "in-tool copy, s_nop 7 in front, dumps",
                          "in-tool copy, s_nop 7 BEHIND v_rsq", "in-tool copy, first uses as unpacked v_mul_f32", "in-tool copy, NO transcendental (Newton)",
                          "in-tool copy, v_mov_b32 of the result first", "in-tool copy, s_nop 7 + v_mov_b32 + s_nop 1 behind",
                          "in-tool copy, NO transcendental AND unpacked first uses",
                          "library ln128_bwd_parts, packed-fp32 code generation (round 3) + bare v_rsq", "library ln128_bwd_parts, packed-fp32 code generation (round 3) + s_nop 7 guard"};
  printf("rsq_repro: %d victim launches per cell, victim = 256 workgroups x 4 waves x 8 rows of 128 floats\n", per_cell);
  const int only_form = argc > 2 ? atoi(argv[2]) : -1;
  for (int form = 0; form < 15; ++form) {
    if (only_form >= 0 && form != only_form) continue;
    if ((form == 3 && !ln_bare) || (form == 4 && !ln_ship) || (form == 13 && !ln_slpbare) || (form == 14 && !ln_slp)) continue;
    const int width = form >= 3 ? 128 : 64, per_lane = form >= 3 ? 2 : 1;
    victim(form, ref);
    CK(hipStreamSynchronize(sv));
    for (int kind = 0; kind < 5; ++kind) {
      if ((kind == 1 || kind == 2 || kind == 4) && !gemm_tn) continue;
      if (tune) {
        tune("tn_exclusive_cu", kind == 2 ? 2 : 0);
        tune("tn_mode", kind == 1 ? 240 : (kind == 2 ? 480 : (kind == 4 ? 280 : 0)));
      }
      CK(hipMemset(hist, 0, 128 * 4)); CK(hipMemset(last_total, 0, 4));
      CK(hipMemset(hist_var, 0, 128 * 4)); CK(hipMemset(hist_rs, 0, 128 * 4)); CK(hipMemset(lt2, 0, 4)); CK(hipMemset(lt3, 0, 4));
      CK(hipMemset(hist_xh, 0, 128 * 4)); CK(hipMemset(hist_t2, 0, 128 * 4)); CK(hipMemset(lt4, 0, 4)); CK(hipMemset(lt5, 0, 4));
      CK(hipDeviceSynchronize());
      int done = 0;
      while (done < per_cell) {
        const int burst = 250;
        if (kind) for (int i = 0; i < (kind == 3 ? 6 : 120); ++i) if (!aggress(kind)) break;     // ~3-4 ms of aggressor work
        for (int i = 0; i < burst; ++i) {
          victim(form, out);
          hipLaunchKernelGGL(check_kernel, dim3(64), dim3(256), 0, sv, out, ref, (size_t)R * width, hist, width, per_lane);
          hipLaunchKernelGGL(mark_launch_kernel, dim3(1), dim3(1), 0, sv, hist, last_total);
          if (form >= 5 && form <= 12) {
            hipLaunchKernelGGL(check_kernel, dim3(64), dim3(256), 0, sv, var_o, var_ref, (size_t)R * 64, hist_var, 64, 1);
            hipLaunchKernelGGL(mark_launch_kernel, dim3(1), dim3(1), 0, sv, hist_var, lt2);
            hipLaunchKernelGGL(check_kernel, dim3(64), dim3(256), 0, sv, rs_o, rs_ref, (size_t)R * 64, hist_rs, 64, 1);
            hipLaunchKernelGGL(mark_launch_kernel, dim3(1), dim3(1), 0, sv, hist_rs, lt3);
            hipLaunchKernelGGL(check_kernel, dim3(64), dim3(256), 0, sv, xh_o, xh_ref, (size_t)R * 64, hist_xh, 64, 1);
            hipLaunchKernelGGL(mark_launch_kernel, dim3(1), dim3(1), 0, sv, hist_xh, lt4);
            hipLaunchKernelGGL(check_kernel, dim3(64), dim3(256), 0, sv, t2_o, t2_ref, (size_t)R * 64, hist_t2, 64, 1);
            hipLaunchKernelGGL(mark_launch_kernel, dim3(1), dim3(1), 0, sv, hist_t2, lt5);
          }
        }
        CK(hipStreamSynchronize(sv));
        CK(hipStreamSynchronize(sa));
        done += burst;
      }
      unsigned h[128];
      CK(hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost));
      unsigned long total = 0;
      for (int i = 0; i < 64; ++i) total += h[i];
      unsigned q[4] = {0, 0, 0, 0};
      for (int i = 0; i < 64; ++i) q[i / 16] += h[i];
      printf("victim [%s] | aggressor [%s]: %u of %d launches wrong, %lu wrong elements; by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %u / %u / %u / %u\n",
             forms[form], names[kind], h[64], done, total, q[0], q[1], q[2], q[3]);
      if (form >= 5 && form <= 12) {
        unsigned hv[128], hr[128];
        CK(hipMemcpy(hv, hist_var, sizeof(hv), hipMemcpyDeviceToHost)); CK(hipMemcpy(hr, hist_rs, sizeof(hr), hipMemcpyDeviceToHost));
        unsigned qv[4] = {0, 0, 0, 0}, qr[4] = {0, 0, 0, 0};
        for (int i = 0; i < 64; ++i) { qv[i / 16] += hv[i]; qr[i / 16] += hr[i]; }
        printf("      rsq ARGUMENT (after the wave reduction) wrong in %u launches, lanes by quarter %u / %u / %u / %u;  1/sqrt wrong in %u launches, lanes by quarter %u / %u / %u / %u\n",
               hv[64], qv[0], qv[1], qv[2], qv[3], hr[64], qr[0], qr[1], qr[2], qr[3]);
        unsigned hx[128], ht[128];
        CK(hipMemcpy(hx, hist_xh, sizeof(hx), hipMemcpyDeviceToHost)); CK(hipMemcpy(ht, hist_t2, sizeof(ht), hipMemcpyDeviceToHost));
        unsigned qx[4] = {0, 0, 0, 0}, qt[4] = {0, 0, 0, 0};
        for (int i = 0; i < 64; ++i) { qx[i / 16] += hx[i]; qt[i / 16] += ht[i]; }
        printf("      xhat = (x - mean) * rs (first use of rs) wrong in %u launches, lanes by quarter %u / %u / %u / %u;  second wave reduction (t2) wrong in %u launches, lanes by quarter %u / %u / %u / %u\n",
               hx[64], qx[0], qx[1], qx[2], qx[3], ht[64], qt[0], qt[1], qt[2], qt[3]);
      }
      fflush(stdout);
    }
  }
  if (tune) { tune("tn_mode", 0); tune("tn_exclusive_cu", 2); }
  return 0;
}
