// Bulk jax.random draws on device (rng_threefry.h has the algorithm and its provenance).
// Every kernel addresses a WINDOW [offset, offset+count) of a logical array of n_total elements, so a rank that owns
// samples [b0, b1) of a sharded batch writes exactly the values a single process would.
#include "smd_kernels.h"
#include "rng_threefry.h"

namespace {

struct TfKeySrc {             // immediate key, or key_table[idx_add + idx_mul * *idx_ptr] (graph-replayable sampler)
  TfKey key;
  const uint32_t* table;
  const int32_t* idx_ptr;
  int idx_mul, idx_add;
  __device__ __forceinline__ TfKey get() const {
    if (!table) return key;
    const int i = idx_add + idx_mul * (idx_ptr ? *idx_ptr : 0);
    TfKey k; k.k0 = table[2 * i]; k.k1 = table[2 * i + 1];
    return k;
  }
};

enum { TF_BITS = 0, TF_UNIFORM = 1, TF_NORMAL = 2 };

template <int KIND>
__device__ __forceinline__ void tf_store(void* out, uint64_t pos, uint32_t bits, float lo, float hi) {
  if constexpr (KIND == TF_BITS) reinterpret_cast<uint32_t*>(out)[pos] = bits;
  else if constexpr (KIND == TF_UNIFORM) reinterpret_cast<float*>(out)[pos] = jax_uniform_from_bits(bits, lo, hi);
  else reinterpret_cast<float*>(out)[pos] = jax_normal_from_bits(bits);
}

// whole array: thread j < h evaluates block (j, j+h) once and writes both of its elements (two coalesced streams)
template <int KIND>
__global__ __launch_bounds__(256) void tf_full_kernel(void* __restrict__ out, uint64_t n, TfKeySrc ks, float lo, float hi) {
  const TfKey k = ks.get();
  const uint64_t h = (n + 1) >> 1;
  for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < h; j += (uint64_t)gridDim.x * 256) {
    uint32_t x0 = (uint32_t)j, x1 = (j + h < n) ? (uint32_t)(j + h) : 0u;
    threefry2x32(k, x0, x1);
    tf_store<KIND>(out, j, x0, lo, hi);
    if (j + h < n) tf_store<KIND>(out, j + h, x1, lo, hi);
  }
}

// window of a larger array: one block evaluation per element
template <int KIND>
__global__ __launch_bounds__(256) void tf_window_kernel(void* __restrict__ out, uint64_t n, uint64_t offset, uint64_t count,
                                                        TfKeySrc ks, float lo, float hi) {
  const TfKey k = ks.get();
  for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (uint64_t)gridDim.x * 256)
    tf_store<KIND>(out, e, jax_bits_at(k, offset + e, n), lo, hi);
}

__global__ __launch_bounds__(256) void tf_randint_kernel(int32_t* __restrict__ out, uint64_t n, uint64_t offset, uint64_t count,
                                                         TfKey key, int32_t minval, int32_t maxval) {
  const TfKey k1 = jax_split_at(key, 0, 2), k2 = jax_split_at(key, 1, 2);
  const int32_t mx = maxval > minval + 1 ? maxval : minval + 1;
  const uint32_t span = (uint32_t)(mx - minval);
  uint32_t mult = 65536u % span;
  mult = (mult * mult) % span;
  for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (uint64_t)gridDim.x * 256) {
    const uint32_t hi = jax_bits_at(k1, offset + e, n), lo = jax_bits_at(k2, offset + e, n);
    const uint32_t off = ((hi % span) * mult + (lo % span)) % span;
    out[e] = minval + (int32_t)off;
  }
}

template <int KIND>
int launch_tf(void* out, int64_t n_total, int64_t offset, int64_t count, const TfKeySrc& ks, float lo, float hi, hipStream_t st) {
  SMD_ARG_CHECK(out && n_total > 0 && n_total <= (1ll << 32) && offset >= 0 && count > 0 && offset + count <= n_total,
                "threefry: bad window n_total=%lld offset=%lld count=%lld (n_total <= 2^32)", (long long)n_total,
                (long long)offset, (long long)count);
  if (offset == 0 && count == n_total) {
    const uint64_t h = ((uint64_t)n_total + 1) >> 1;
    uint64_t blocks = (h + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(tf_full_kernel<KIND>, dim3((unsigned)blocks), dim3(256), 0, st, out, (uint64_t)n_total, ks, lo, hi);
  } else {
    uint64_t blocks = ((uint64_t)count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(tf_window_kernel<KIND>, dim3((unsigned)blocks), dim3(256), 0, st, out, (uint64_t)n_total,
                       (uint64_t)offset, (uint64_t)count, ks, lo, hi);
  }
  SMD_LAUNCH_CHECK();
  return 0;
}

TfKeySrc key_src(uint32_t k0, uint32_t k1, const uint32_t* table, const int32_t* idx_ptr, int idx_mul, int idx_add) {
  TfKeySrc s;
  s.key.k0 = k0; s.key.k1 = k1; s.table = table; s.idx_ptr = idx_ptr; s.idx_mul = idx_mul; s.idx_add = idx_add;
  return s;
}

}  // namespace

int launch_threefry_bits(uint32_t* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1,
                         hipStream_t st) {
  return launch_tf<TF_BITS>(out, n_total, offset, count, key_src(k0, k1, nullptr, nullptr, 0, 0), 0.f, 1.f, st);
}
int launch_threefry_uniform(float* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1,
                            float minval, float maxval, hipStream_t st) {
  return launch_tf<TF_UNIFORM>(out, n_total, offset, count, key_src(k0, k1, nullptr, nullptr, 0, 0), minval, maxval, st);
}
int launch_threefry_normal(float* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1,
                           const uint32_t* key_table, const int32_t* idx_ptr, int idx_mul, int idx_add, hipStream_t st) {
  return launch_tf<TF_NORMAL>(out, n_total, offset, count, key_src(k0, k1, key_table, idx_ptr, idx_mul, idx_add), 0.f, 1.f, st);
}
int launch_threefry_randint(int32_t* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1,
                            int32_t minval, int32_t maxval, hipStream_t st) {
  SMD_ARG_CHECK(out && n_total > 0 && n_total <= (1ll << 32) && offset >= 0 && count > 0 && offset + count <= n_total,
                "threefry_randint: bad window");
  uint64_t blocks = ((uint64_t)count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  TfKey k; k.k0 = k0; k.k1 = k1;
  hipLaunchKernelGGL(tf_randint_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, (uint64_t)n_total, (uint64_t)offset,
                     (uint64_t)count, k, minval, maxval);
  SMD_LAUNCH_CHECK();
  return 0;
}
