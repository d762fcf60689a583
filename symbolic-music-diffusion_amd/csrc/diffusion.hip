// HBM-bound elementwise kernels of the DDPM objective and reverse sampler.
//
//   noise_embed      NoiseEncoding.apply                      models/ncsn.py:28-41
//   q_sample         labels, alpha lookup, eps, x_t           utils/losses.py:271-296
//   mse_loss_grad    mean((eps-pred)^2) and d/dpred            utils/losses.py:304-308
//   reverse_step     one sample_with_beta iteration            utils/ebm_utils.py:327-394
//
// All per-step scalars (timestep t, optimiser step) are read from device memory so the whole
// step is hipGraph-replayable with frozen kernel arguments.
#include "smd_kernels.h"
#include "rng_threefry.h"
#include "rng.h"

namespace {

// ------------------------------------------------------------------ noise embedding
__global__ __launch_bounds__(256) void noise_embed_kernel(const float* __restrict__ s, int n, int channels,
                                                          bf16_t* __restrict__ out, int ld_out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int half = channels >> 1;
  if (idx >= n * half) return;
  const int r = idx / half, i = idx - r * half;
  // emb = exp(arange(half) * -(log(10000)/(half-1))) ; arg = (5000 * noise) * emb   (fp32, same order)
  const float f = expf((float)i * -(9.210340371976184f / (float)(half - 1)));
  const float arg = (5000.0f * s[r]) * f;
  float sn, cs;
  sincosf(arg, &sn, &cs);
  out[(size_t)r * ld_out + i] = f2bf(sn);
  out[(size_t)r * ld_out + half + i] = f2bf(cs);
  if ((channels & 1) && i == 0) out[(size_t)r * ld_out + channels - 1] = f2bf(0.0f);
}

// ------------------------------------------------------------------ q-sample
// VEC4: S*C is a multiple of 4 (16-byte loads / stores).  Otherwise (DenseDDPM on sliced latents: C = 42, 146 ...)
// the same groups of four Philox normals are used, moved element by element and cut at the end of the sample.
template <bool VEC4>
__global__ __launch_bounds__(256) void q_sample_kernel(QSampleArgs a) {
  const int SC = a.S * a.C;
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int e0 = g * 4;
  const int b = blockIdx.y;
  if (e0 >= SC) return;
  const uint32_t bglob = (uint32_t)b + a.sample_offset;
  const uint32_t step = a.step_ptr ? *a.step_ptr : 0u;
  const uint4 r = philox4x32_10(make_uint4(0u, bglob, SMD_STREAM_LABEL, step), a.key.seed_lo, a.key.seed_hi);
  int label;
  if (a.labels) {
    label = min(max(a.labels[b], 0), a.T);
  } else {
    // randint[1, T+1) (continuous_noise) or randint[0, T) (label_min = 0), utils/losses.py:272-275
    label = a.label_min + (int)(r.x % (uint32_t)a.T);
  }
  float alpha;
  if (a.dsm) {
    alpha = 0.f;                                      // unused: x_t = x0 + sigma * eps below
  } else if (a.alpha_in) {
    alpha = a.alpha_in[b];
  } else if (label > 0) {
    // jax-0.2.8 uniform(minval=ap[l-1], maxval=ap[l]) degenerates to minval
    alpha = a.alphas_prod_ext[label - 1];
  } else {
    // label 0 (only with continuous_noise=False): alphas_prod'[-1] = ap[T] < alphas_prod'[0] = 1, a real uniform draw
    const float lo = a.alphas_prod_ext[a.T];
    const float u = __uint_as_float((r.y >> 9) | 0x3F800000u) - 1.0f;
    alpha = fmaxf(lo, u * (1.0f - lo) + lo);
  }
  // denoising score matching (utils/losses.py:163-167): perturbed = batch + used_sigma * eps, conditioned on used_sigma
  const float sa = a.dsm ? 1.0f : sqrtf(alpha), sb = a.dsm ? a.alpha_in[b] : sqrtf(1.0f - alpha);
  const size_t base = (size_t)b * SC + e0;
  float4 eps;
  if (a.eps_in) {
    if constexpr (VEC4) {
      eps = *reinterpret_cast<const float4*>(a.eps_in + base);
    } else {
      eps.x = a.eps_in[base];
      eps.y = e0 + 1 < SC ? a.eps_in[base + 1] : 0.f;
      eps.z = e0 + 2 < SC ? a.eps_in[base + 2] : 0.f;
      eps.w = e0 + 3 < SC ? a.eps_in[base + 3] : 0.f;
    }
  } else {
    eps = philox_normal4((uint32_t)g, bglob, SMD_STREAM_EPS, step, a.key.seed_lo, a.key.seed_hi);
  }
  float4 x0;
  if constexpr (VEC4) {
    x0 = *reinterpret_cast<const float4*>(a.x0 + base);
  } else {
    x0.x = a.x0[base];
    x0.y = e0 + 1 < SC ? a.x0[base + 1] : 0.f;
    x0.z = e0 + 2 < SC ? a.x0[base + 2] : 0.f;
    x0.w = e0 + 3 < SC ? a.x0[base + 3] : 0.f;
  }
  const float xt[4] = {sa * x0.x + sb * eps.x, sa * x0.y + sb * eps.y, sa * x0.z + sb * eps.z,
                       sa * x0.w + sb * eps.w};
  const float ev[4] = {eps.x, eps.y, eps.z, eps.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = e0 + i;
    if (!VEC4 && e >= SC) break;
    const int srow = e / a.C, c = e - srow * a.C;
    a.xt_bf16[((size_t)b * a.S + srow) * a.Cp + c] = f2bf(xt[i]);
    if constexpr (!VEC4) a.eps_out[base + i] = ev[i];
  }
  if constexpr (VEC4) *reinterpret_cast<float4*>(a.eps_out + base) = eps;
  if (g == 0) a.s_out[b] = a.dsm ? sb : sa;
}

// The same for C % 4 == 0 and Cp % 4 == 0 (a group of four never straddles a row): QPT groups per thread with all x0 loads issued
// first, ONE 8-byte bf16 store and one 16-byte eps store per group, the row index from a float reciprocal (S C < 2^22) -- and the
// label draw / alpha lookup / square roots once per QPT groups.  Same Philox counters as q_sample_kernel: the same draws.
template <int QPT>
__global__ __launch_bounds__(256) void q_sample_quads_kernel(QSampleArgs a) {
  const int SC = a.S * a.C, nq = SC >> 2;
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * (256 * QPT) + threadIdx.x;
  const size_t sbase = (size_t)b * SC;
  float4 x0[QPT], ein[QPT];
#pragma unroll
  for (int k = 0; k < QPT; ++k) {
    const int q = q0 + k * 256;
    if (q < nq) {
      x0[k] = *reinterpret_cast<const float4*>(a.x0 + sbase + (size_t)q * 4);
      if (a.eps_in) ein[k] = *reinterpret_cast<const float4*>(a.eps_in + sbase + (size_t)q * 4);
    }
  }
  const uint32_t bglob = (uint32_t)b + a.sample_offset;
  const uint32_t step = a.step_ptr ? *a.step_ptr : 0u;
  int label;
  uint4 r = make_uint4(0u, 0u, 0u, 0u);
  if (a.labels) {
    label = min(max(a.labels[b], 0), a.T);
    if (label == 0 && !a.alpha_in && !a.dsm) r = philox4x32_10(make_uint4(0u, bglob, SMD_STREAM_LABEL, step), a.key.seed_lo, a.key.seed_hi);
  } else {
    r = philox4x32_10(make_uint4(0u, bglob, SMD_STREAM_LABEL, step), a.key.seed_lo, a.key.seed_hi);
    label = a.label_min + (int)(r.x % (uint32_t)a.T);                 // utils/losses.py:272-275
  }
  float alpha;
  if (a.dsm) alpha = 0.f;
  else if (a.alpha_in) alpha = a.alpha_in[b];
  else if (label > 0) alpha = a.alphas_prod_ext[label - 1];           // jax-0.2.8 uniform(minval > maxval) = minval
  else {
    const float lo = a.alphas_prod_ext[a.T];
    const float u = __uint_as_float((r.y >> 9) | 0x3F800000u) - 1.0f;
    alpha = fmaxf(lo, u * (1.0f - lo) + lo);
  }
  const float sa = a.dsm ? 1.0f : sqrtf(alpha), sb = a.dsm ? a.alpha_in[b] : sqrtf(1.0f - alpha);
  const float invC = 1.0f / (float)a.C;
#pragma unroll
  for (int k = 0; k < QPT; ++k) {
    const int q = q0 + k * 256;
    if (q >= nq) continue;
    const float4 eps = a.eps_in ? ein[k] : philox_normal4((uint32_t)q, bglob, SMD_STREAM_EPS, step, a.key.seed_lo, a.key.seed_hi);
    const int e = q * 4;
    int srow = (int)((float)e * invC);
    srow -= (srow * a.C > e);
    srow += ((srow + 1) * a.C <= e);
    const int c = e - srow * a.C;
    bf16x4_t t;
    // (explicit fma: both instantiations -- and a shard of the batch that takes the other one -- round identically)
    t[0] = f2bf(__builtin_fmaf(sb, eps.x, sa * x0[k].x)); t[1] = f2bf(__builtin_fmaf(sb, eps.y, sa * x0[k].y));
    t[2] = f2bf(__builtin_fmaf(sb, eps.z, sa * x0[k].z)); t[3] = f2bf(__builtin_fmaf(sb, eps.w, sa * x0[k].w));
    *reinterpret_cast<bf16x4_t*>(a.xt_bf16 + ((size_t)b * a.S + srow) * a.Cp + c) = t;
    *reinterpret_cast<float4*>(a.eps_out + sbase + (size_t)e) = eps;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.s_out[b] = a.dsm ? sb : sa;
}

template <int VEC> struct VecT;
template <> struct VecT<4> { typedef float4 type; };
template <> struct VecT<1> { typedef float type; };

template <int VEC>
__device__ __forceinline__ void ldv(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void stv(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else *p = v[0];
}

// ------------------------------------------------------------------ loss + gradient wrt pred
// one workgroup of 1024 threads per sample; VEC = 4: 16-byte loads / 8-byte bf16 stores (C and Cp multiples of 4)
template <int VEC>
__global__ __launch_bounds__(1024) void mse_loss_grad_kernel(const float* __restrict__ pred,
                                                             const float* __restrict__ eps, int S, int C, int Cp,
                                                             float inv_global_count,
                                                             float* __restrict__ loss_per_sample,
                                                             bf16_t* __restrict__ dpred,
                                                             const float* __restrict__ dsm_sigma) {
  __shared__ float red[16];
  const int b = blockIdx.x, SC = S * C;
  // DDPM: d = pred - eps, loss = mean d^2, dpred = 2 d / (Bg S C).  Denoising score matching (utils/losses.py:166-178):
  // target = -eps / sigma, loss = 0.5 sum (pred - target)^2 sigma^2 = 0.5 sum (sigma pred + eps)^2; with d = sigma pred + eps
  // the gradient of the batch MEAN is d sigma / Bg.
  const float sg = dsm_sigma ? dsm_sigma[b] : 1.0f;
  const float qs = dsm_sigma ? -1.0f : 1.0f;
  const float gscale = dsm_sigma ? sg * inv_global_count * (float)SC : 2.0f * inv_global_count;
  float acc = 0.f;
  for (int e = threadIdx.x * VEC; e < SC; e += 1024 * VEC) {
    float p[VEC], q[VEC];
    ldv<VEC>(pred + (size_t)b * SC + e, p);
    ldv<VEC>(eps + (size_t)b * SC + e, q);
    const int srow = e / C, c = e - srow * C;
    bf16_t* dst = dpred + ((size_t)b * S + srow) * Cp + c;
    if constexpr (VEC == 4) {
      bf16x4_t o;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float d = sg * p[v] - qs * q[v];
        acc += d * d;
        o[v] = f2bf(d * gscale);
      }
      *reinterpret_cast<bf16x4_t*>(dst) = o;
    } else {
      const float d = sg * p[0] - qs * q[0];
      acc += d * d;
      dst[0] = f2bf(d * gscale);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i];
    loss_per_sample[b] = dsm_sigma ? 0.5f * t : t / (float)SC;
  }
}

// ------------------------------------------------------------------ fused reverse step
// One workgroup per sample: 128 column threads x RG row groups (RG = 4 for sequence states: four rows of every column
// in flight instead of one; the per-column sums over the sequence axis are combined through LDS in a fixed order).
template <int VEC, int RG>
__global__ __launch_bounds__(128 * RG) void reverse_step_kernel(ReverseStepArgs a) {
  __shared__ float red[2][3];
  __shared__ float part[RG > 1 ? RG : 1][3][128 * VEC];
  const int b = blockIdx.x;
  const int ct = threadIdx.x & 127, rg = threadIdx.x >> 7;
  const int t = *a.t_ptr;
  if (t < 0 || t >= a.T) return;          // walked past t = 0 (or a bad timestep): a no-op, t is not advanced either
  const float* cf = a.coef + (size_t)t * 8;
  const float sqrt_recip = cf[0], sqrt_m1 = cf[1], mu1 = cf[2], mu2 = cf[3], sigma = cf[4];
  const float sqrt_ap = cf[6], sqrt_1m = cf[7];
  const bool noisy = t > 0;
  int slot = a.collection ? a.slot_table[t] : -1;
  if (slot > 40) slot = -1;            // the collection has 41 rows (utils/ebm_utils.py:320-322); out-of-range scatters are dropped
  const uint32_t bglob = (uint32_t)b + a.sample_offset;
  const size_t sample_base = (size_t)b * a.S * a.C;
  const uint64_t tf_base = (uint64_t)bglob * a.S * a.C;       // this sample's first element in the global jax array
  const uint32_t key_lo = a.key_ptr ? a.key_ptr[0] : a.key.seed_lo, key_hi = a.key_ptr ? a.key_ptr[1] : a.key.seed_hi;
  TfKey tf_nk{0, 0}, tf_ik{0, 0};
  if (a.tf_noise_keys) { tf_nk.k0 = a.tf_noise_keys[2 * (a.tf_t0 - t)]; tf_nk.k1 = a.tf_noise_keys[2 * (a.tf_t0 - t) + 1]; }
  if (a.tf_infill_keys) { tf_ik.k0 = a.tf_infill_keys[2 * (a.tf_t0 - t)]; tf_ik.k1 = a.tf_infill_keys[2 * (a.tf_t0 - t) + 1]; }
  float m_eps = 0.f, m_step = 0.f, m_z = 0.f;
  for (int cb = 0; cb < a.C; cb += 128 * VEC) {            // uniform trip count: the LDS combine below has barriers
    const int col0 = cb + ct * VEC;
    const bool live = col0 < a.C;
    float acc_e[VEC], acc_s[VEC], acc_z[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc_e[v] = acc_s[v] = acc_z[v] = 0.f;
    for (int s = rg; live && s < a.S; s += RG) {
      const int e = s * a.C + col0;
      const size_t idx = sample_base + e;
      float x[VEC], eh[VEC], z[VEC], nx[VEC];
      ldv<VEC>(a.x + idx, x);
      ldv<VEC>(a.eps_hat + idx, eh);
      if (noisy) {                                               // utils/ebm_utils.py:360-364
        if (a.z_in) {
          ldv<VEC>(a.z_in + idx, z);
        } else if (a.tf_noise_keys) {                            // jax.random.normal(noise_rng, state.shape) (:361-362)
#pragma unroll
          for (int v = 0; v < VEC; ++v) z[v] = jax_normal_from_bits(jax_bits_at(tf_nk, tf_base + e + v, (uint64_t)a.tf_n_total));
        } else {
          const float4 n4 = philox_normal4((uint32_t)(e >> 2), bglob, SMD_STREAM_Z, (uint32_t)t,
                                           key_lo, key_hi);
          if constexpr (VEC == 4) { z[0] = n4.x; z[1] = n4.y; z[2] = n4.z; z[3] = n4.w; }
          else z[0] = pick4(n4, e & 3);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) z[v] *= sigma;
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) z[v] = 0.f;
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float recon = sqrt_recip * x[v] - sqrt_m1 * eh[v];        // :371
        recon = fminf(fmaxf(recon, -1.0f), 1.0f);                 // :372
        nx[v] = mu1 * recon + mu2 * x[v] + z[v];                  // :373-374
      }
      if (a.infill_masks) {                                       // :342-348, :377
        float im[VEC], is[VEC], iz[VEC];
        ldv<VEC>(a.infill_masks + idx, im);
        ldv<VEC>(a.infill_samples + idx, is);
        if (noisy) {
          if (a.infill_z_in) {
            ldv<VEC>(a.infill_z_in + idx, iz);
          } else if (a.tf_infill_keys) {                         // jax.random.normal(infill_noise_rng, ...) (:343-345)
#pragma unroll
            for (int v = 0; v < VEC; ++v) iz[v] = jax_normal_from_bits(jax_bits_at(tf_ik, tf_base + e + v, (uint64_t)a.tf_n_total));
          } else {
            const float4 n4 = philox_normal4((uint32_t)(e >> 2), bglob, SMD_STREAM_INFILL, (uint32_t)t,
                                             key_lo, key_hi);
            if constexpr (VEC == 4) { iz[0] = n4.x; iz[1] = n4.y; iz[2] = n4.z; iz[3] = n4.w; }
            else iz[0] = pick4(n4, e & 3);
          }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float y = noisy ? sqrt_ap * is[v] + sqrt_1m * iz[v] : is[v];
          nx[v] = nx[v] * (1.0f - im[v]) + y * im[v];
        }
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float st = x[v] - nx[v];
        acc_e[v] += eh[v] * eh[v];
        acc_s[v] += st * st;
        acc_z[v] += z[v] * z[v];
      }
      stv<VEC>(a.x + idx, nx);
      if (slot >= 0) stv<VEC>(a.collection + ((size_t)slot * a.B) * a.S * a.C + idx, nx);
      if (a.x_bf16) {
        bf16_t* xb = a.x_bf16 + ((size_t)b * a.S + s) * a.Cp + col0;
        if constexpr (VEC == 4) {
          bf16x4_t p;
#pragma unroll
          for (int v = 0; v < 4; ++v) p[v] = f2bf(nx[v]);
          *reinterpret_cast<bf16x4_t*>(xb) = p;
        } else {
          xb[0] = f2bf(nx[0]);
        }
      }
    }
    if constexpr (RG > 1) {
      __syncthreads();
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        part[rg][0][ct * VEC + v] = acc_e[v]; part[rg][1][ct * VEC + v] = acc_s[v]; part[rg][2][ct * VEC + v] = acc_z[v];
      }
      __syncthreads();
      if (rg == 0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float se = 0.f, ss = 0.f, sz = 0.f;
#pragma unroll
          for (int g2 = 0; g2 < RG; ++g2) {
            se += part[g2][0][ct * VEC + v]; ss += part[g2][1][ct * VEC + v]; sz += part[g2][2][ct * VEC + v];
          }
          acc_e[v] = se; acc_s[v] = ss; acc_z[v] = sz;
        }
      }
    }
    // :381-383 sqrt(sum(v^2, axis=1) + 1e-10): axis 1 is the SEQUENCE axis for (B,S,C) states and the
    // channel axis for the 2-D (B,C) states of DenseDDPM (S == 1 here).
    if (rg == 0 && live)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (a.S > 1) {
        m_eps += sqrtf(acc_e[v] + 1e-10f);
        m_step += sqrtf(acc_s[v] + 1e-10f);
        m_z += sqrtf(acc_z[v] + 1e-10f);
      } else {
        m_eps += acc_e[v]; m_step += acc_s[v]; m_z += acc_z[v];
      }
    }
  }
  if (a.metrics_partial) {
    m_eps = wave_sum(m_eps); m_step = wave_sum(m_step); m_z = wave_sum(m_z);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0 && w < 2) { red[w][0] = m_eps; red[w][1] = m_step; red[w][2] = m_z; }
    __syncthreads();
    if (threadIdx.x < 3) {
      float v = red[0][threadIdx.x] + red[1][threadIdx.x];
      if (a.S == 1) v = sqrtf(v + 1e-10f);
      a.metrics_partial[((size_t)t * a.B + b) * 3 + threadIdx.x] = v;
    }
  }
  // *t_advance = t - 1 by the LAST workgroup to get here (every workgroup read t at its top, i.e. before its own
  // arrival): the timestep decrement of the captured sampling step without a launch of its own.  `arrive` is zero
  // between launches (the last arriver resets it with a device-scope atomic).
  if (a.t_advance && threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      __hip_atomic_exchange(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.t_advance, t - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void advance_t_kernel(int* t_ptr) { *t_ptr -= 1; }
__global__ void set_t_kernel(int* t_ptr, int v) { *t_ptr = v; }

// ------------------------------------------------------------------ Langevin update (annealed / consistent)
// utils/ebm_utils.py:131-164 (langevin_step of annealed_langevin_dynamics) and :231-253 (consistent):
//   next = state + alpha * grad + noise_coef * z ;  next = next (1 - mask) + (infill + infill_sigma * zi) mask
// One workgroup per sample, a thread per column walking the sequence axis, so the per-column sums over axis 1 of the
// metrics (:157-161) stay in registers; 2-D states (S == 1) reduce over the channel axis instead.
__global__ __launch_bounds__(256) void langevin_step_kernel(LangevinStepArgs a) {
  __shared__ float red[4][3];
  const int b = blockIdx.x;
  const uint32_t bglob = (uint32_t)b + a.sample_offset;
  const size_t sample_base = (size_t)b * a.S * a.C;
  const uint64_t tf_base = (uint64_t)bglob * a.S * a.C;
  TfKey nk{a.tf_noise_key[0], a.tf_noise_key[1]}, ik{a.tf_infill_key[0], a.tf_infill_key[1]};
  int kstep = 0;
  if (a.k_ptr) {          // table mode: this update's arguments from device memory (one captured launch serves every update)
    kstep = *a.k_ptr;
    if (kstep < 0 || kstep >= a.n_steps) return;
    const float4 row = *reinterpret_cast<const float4*>(a.step_table + (size_t)kstep * 4);
    a.alpha = row.x; a.noise_coef = row.y; a.infill_sigma = row.z;
    a.step = (uint32_t)kstep;
    if (a.key_table) {
      const uint4 kk = *reinterpret_cast<const uint4*>(a.key_table + (size_t)kstep * 4);
      nk = TfKey{kk.x, kk.y}; ik = TfKey{kk.z, kk.w};
    }
    const int slot = a.slot_table ? a.slot_table[kstep] : -1;
    a.collect_out = (slot >= 0 && a.collection) ? a.collection + (size_t)slot * a.B * a.S * a.C : nullptr;
    if (a.metrics_partial) a.metrics_partial += (size_t)kstep * a.B * 3;
    if (a.sigma_out && threadIdx.x == 0) a.sigma_out[b] = row.w;
    if (a.level_out && b == 0 && threadIdx.x == 0) {
      const int lv = (kstep + 1) / (a.steps_per_level > 0 ? a.steps_per_level : 1);
      *a.level_out = lv < a.n_levels ? lv : a.n_levels - 1;
    }
  }
  float m_g = 0.f, m_s = 0.f, m_z = 0.f;
  for (int c = threadIdx.x; c < a.C; c += 256) {
    float acc_g = 0.f, acc_s = 0.f, acc_z = 0.f;
    for (int s = 0; s < a.S; ++s) {
      const int e = s * a.C + c;
      const size_t idx = sample_base + e;
      const float x = a.x[idx], g = a.grad[idx];
      float z;
      if (a.z_in) z = a.z_in[idx];
      else if (a.use_threefry) z = jax_normal_from_bits(jax_bits_at(nk, tf_base + e, (uint64_t)a.tf_n_total));
      else z = pick4(philox_normal4((uint32_t)(e >> 2), bglob, SMD_STREAM_Z, a.step, a.key.seed_lo, a.key.seed_hi), e & 3);
      const float noise = a.noise_coef * z;
      const float stp = a.alpha * g;
      float nx = x + stp + noise;                                                  // :143 gradient ascent
      if (a.infill_masks) {                                                         // :137-138,146
        float zi;
        if (a.infill_z_in) zi = a.infill_z_in[idx];
        else if (a.use_threefry) zi = jax_normal_from_bits(jax_bits_at(ik, tf_base + e, (uint64_t)a.tf_n_total));
        else zi = pick4(philox_normal4((uint32_t)(e >> 2), bglob, SMD_STREAM_INFILL, a.step, a.key.seed_lo, a.key.seed_hi), e & 3);
        const float y = a.infill_samples[idx] + a.infill_sigma * zi;
        const float im = a.infill_masks[idx];
        nx = nx * (1.0f - im) + y * im;
      }
      acc_g += g * g; acc_s += stp * stp; acc_z += noise * noise;
      a.x[idx] = nx;
      if (a.collect_out) a.collect_out[idx] = nx;
    }
    if (a.S > 1) { m_g += sqrtf(acc_g + 1e-10f); m_s += sqrtf(acc_s + 1e-10f); m_z += sqrtf(acc_z + 1e-10f); }
    else { m_g += acc_g; m_s += acc_s; m_z += acc_z; }
  }
  if (a.metrics_partial) {
    m_g = wave_sum(m_g); m_s = wave_sum(m_s); m_z = wave_sum(m_z);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w][0] = m_g; red[w][1] = m_s; red[w][2] = m_z; }
    __syncthreads();
    if (threadIdx.x < 3) {
      float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
      if (a.S == 1) v = sqrtf(v + 1e-10f);
      a.metrics_partial[(size_t)b * 3 + threadIdx.x] = v;
    }
  }
  // table mode: *k_ptr = k + 1 by the LAST workgroup to get here (every workgroup read k at its top, before its own arrival)
  if (a.k_ptr && a.arrive && threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      __hip_atomic_exchange(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.k_ptr, kstep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(256) void cast_pad_bf16_kernel(const float* __restrict__ in, int rows, int cols,
                                                            bf16_t* __restrict__ out, int ld_out) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)rows * ld_out) return;
  const int r = (int)(idx / ld_out), c = (int)(idx - (size_t)r * ld_out);
  out[idx] = c < cols ? f2bf(in[(size_t)r * cols + c]) : f2bf(0.0f);
}

template <bool VEC4>
__global__ __launch_bounds__(256) void fill_normal_kernel(float* __restrict__ out, int per_sample, RngKey key,
                                                          uint32_t stream, uint32_t sample_offset) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (g * 4 >= per_sample) return;
  const float4 n = philox_normal4((uint32_t)g, (uint32_t)b + sample_offset, stream, 0u, key.seed_lo, key.seed_hi);
  float* dst = out + (size_t)b * per_sample + g * 4;
  if constexpr (VEC4) {
    *reinterpret_cast<float4*>(dst) = n;
  } else {                       // per_sample not a multiple of 4: same draws, scalar stores, cut at the sample's end
    const float v[4] = {n.x, n.y, n.z, n.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (g * 4 + i < per_sample) dst[i] = v[i];
  }
}

__global__ __launch_bounds__(256) void swish_bwd_bf16_kernel(const bf16_t* __restrict__ pre,
                                                             const bf16_t* __restrict__ dout,
                                                             bf16_t* __restrict__ din, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) din[i] = f2bf(bf2f(dout[i]) * swish_gradf_(bf2f(pre[i])));
}

}  // namespace

int launch_noise_embed(const float* s, int n, int channels, bf16_t* out, int ld_out, hipStream_t st) {
  SMD_ARG_CHECK(s && out && n > 0 && channels >= 4 && ld_out >= channels, "noise_embed: bad arguments");
  const int total = n * (channels / 2);
  hipLaunchKernelGGL(noise_embed_kernel, dim3((total + 255) / 256), dim3(256), 0, st, s, n, channels, out, ld_out);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_q_sample(const QSampleArgs& a, hipStream_t st) {
  SMD_ARG_CHECK(a.x0 && a.alphas_prod_ext && a.xt_bf16 && a.eps_out && a.s_out, "q_sample: null pointer");
  SMD_ARG_CHECK(a.B > 0 && a.S > 0 && a.C > 0 && a.Cp >= a.C && a.T > 0, "q_sample: bad shape");
  SMD_ARG_CHECK(a.label_min == 0 || a.label_min == 1, "q_sample: label_min=%d", a.label_min);
  SMD_ARG_CHECK(!a.dsm || a.alpha_in, "q_sample: the score-matching form needs the per-sample used_sigmas");
  const int groups = (a.S * a.C + 3) / 4;
  if (a.C % 4 == 0 && a.Cp % 4 == 0 && a.S * a.C < (1 << 22)) {
    // four groups per thread when that still leaves >= 4 workgroups per CU, else one
    if ((long)a.B * ((groups + 1023) / 1024) >= 1024) hipLaunchKernelGGL(q_sample_quads_kernel<4>, dim3((groups + 1023) / 1024, a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(q_sample_quads_kernel<1>, dim3((groups + 255) / 256, a.B), dim3(256), 0, st, a);
    SMD_LAUNCH_CHECK();
    return 0;
  }
  const dim3 grid((groups + 255) / 256, a.B);
  if ((a.S * a.C) % 4 == 0) hipLaunchKernelGGL(q_sample_kernel<true>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(q_sample_kernel<false>, grid, dim3(256), 0, st, a);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_mse_loss_grad(const float* pred, const float* eps, int B, int S, int C, int Cp, float inv_global_count,
                         float* loss_per_sample, bf16_t* dpred_bf16, hipStream_t st, const float* dsm_sigma) {
  SMD_ARG_CHECK(pred && eps && loss_per_sample && dpred_bf16 && B > 0 && S > 0 && C > 0 && Cp >= C,
                "mse_loss_grad: bad arguments");
  if (C % 4 == 0 && Cp % 4 == 0)
    hipLaunchKernelGGL(mse_loss_grad_kernel<4>, dim3(B), dim3(1024), 0, st, pred, eps, S, C, Cp, inv_global_count,
                       loss_per_sample, dpred_bf16, dsm_sigma);
  else
    hipLaunchKernelGGL(mse_loss_grad_kernel<1>, dim3(B), dim3(1024), 0, st, pred, eps, S, C, Cp, inv_global_count,
                       loss_per_sample, dpred_bf16, dsm_sigma);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_reverse_step(const ReverseStepArgs& a, hipStream_t st) {
  SMD_ARG_CHECK(a.x && a.eps_hat && a.coef && a.t_ptr, "reverse_step: null pointer");
  SMD_ARG_CHECK(a.B > 0 && a.S > 0 && a.C > 0 && (!a.x_bf16 || a.Cp >= a.C), "reverse_step: bad shape");
  SMD_ARG_CHECK((a.infill_masks != nullptr) == (a.infill_samples != nullptr), "reverse_step: infill needs samples and masks");
  SMD_ARG_CHECK(!a.collection || a.slot_table, "reverse_step: collection needs slot_table");
  SMD_ARG_CHECK(a.T > 0, "reverse_step: T=%d (number of timesteps: bounds the coefficient table)", a.T);
  SMD_ARG_CHECK(!a.t_advance || a.arrive, "reverse_step: t_advance needs the arrival counter");
  SMD_ARG_CHECK(!(a.tf_noise_keys || a.tf_infill_keys) || (a.tf_n_total >= (int64_t)(a.sample_offset + a.B) * a.S * a.C &&
                                                            a.tf_n_total <= (1ll << 32)),
                "reverse_step: tf_n_total=%lld must cover this rank's window and be <= 2^32", (long long)a.tf_n_total);
  const bool vec = a.C % 4 == 0 && (!a.x_bf16 || a.Cp % 4 == 0);
  if (a.S >= 4) {
    if (vec) hipLaunchKernelGGL((reverse_step_kernel<4, 4>), dim3(a.B), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((reverse_step_kernel<1, 4>), dim3(a.B), dim3(512), 0, st, a);
  } else {
    if (vec) hipLaunchKernelGGL((reverse_step_kernel<4, 1>), dim3(a.B), dim3(128), 0, st, a);
    else hipLaunchKernelGGL((reverse_step_kernel<1, 1>), dim3(a.B), dim3(128), 0, st, a);
  }
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_langevin_step(const LangevinStepArgs& a, hipStream_t st) {
  SMD_ARG_CHECK(a.x && a.grad && a.B > 0 && a.S > 0 && a.C > 0, "langevin_step: bad arguments");
  SMD_ARG_CHECK(!a.infill_masks || a.infill_samples, "langevin_step: infill needs both the samples and the masks");
  SMD_ARG_CHECK(!a.k_ptr || (a.step_table && a.arrive && a.n_steps > 0), "langevin_step: table mode needs step_table, arrive and n_steps");
  SMD_ARG_CHECK(!a.use_threefry || (a.tf_n_total >= ((int64_t)a.sample_offset + a.B) * a.S * a.C && a.tf_n_total <= (1ll << 32)),
                "langevin_step: tf_n_total=%lld must cover this rank's window and be <= 2^32", (long long)a.tf_n_total);
  hipLaunchKernelGGL(langevin_step_kernel, dim3(a.B), dim3(256), 0, st, a);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_set_t(int* t_ptr, int v, hipStream_t st) {
  SMD_ARG_CHECK(t_ptr, "set_t: null pointer");
  hipLaunchKernelGGL(set_t_kernel, dim3(1), dim3(1), 0, st, t_ptr, v);
  SMD_LAUNCH_CHECK();
  return 0;
}
int launch_advance_t(int* t_ptr, hipStream_t st) {
  SMD_ARG_CHECK(t_ptr, "advance_t: null pointer");
  hipLaunchKernelGGL(advance_t_kernel, dim3(1), dim3(1), 0, st, t_ptr);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_cast_pad_bf16(const float* in, int rows, int cols, bf16_t* out, int ld_out, hipStream_t st) {
  SMD_ARG_CHECK(in && out && rows > 0 && cols > 0 && ld_out >= cols, "cast_pad_bf16: bad arguments");
  const size_t total = (size_t)rows * ld_out;
  hipLaunchKernelGGL(cast_pad_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, rows, cols,
                     out, ld_out);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_fill_normal(float* out, int B, int per_sample, RngKey key, uint32_t stream, uint32_t sample_offset,
                       hipStream_t st) {
  SMD_ARG_CHECK(out && B > 0 && per_sample > 0, "fill_normal: bad arguments");
  const dim3 grid(((per_sample + 3) / 4 + 255) / 256, B);
  if (per_sample % 4 == 0)
    hipLaunchKernelGGL(fill_normal_kernel<true>, grid, dim3(256), 0, st, out, per_sample, key, stream, sample_offset);
  else
    hipLaunchKernelGGL(fill_normal_kernel<false>, grid, dim3(256), 0, st, out, per_sample, key, stream, sample_offset);
  SMD_LAUNCH_CHECK();
  return 0;
}

int launch_swish_bwd_bf16(const bf16_t* pre, const bf16_t* dout, bf16_t* din, size_t n, hipStream_t st) {
  SMD_ARG_CHECK(pre && dout && din && n > 0, "swish_bwd: bad arguments");
  hipLaunchKernelGGL(swish_bwd_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pre, dout, din, n);
  SMD_LAUNCH_CHECK();
  return 0;
}

// TransformerPositionalEncoding.apply, models/shared.py:36-48: pe[s] = [sin(s f_i) | cos(s f_i)]
namespace {
__global__ __launch_bounds__(256) void pos_encoding_kernel(float* __restrict__ pe, int S, int channels) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int half = channels >> 1;
  if (idx >= S * half) return;
  const int s = idx / half, i = idx - s * half;
  const float f = expf((float)i * -(9.210340371976184f / (float)(half - 1)));
  float sn, cs;
  sincosf((float)s * f, &sn, &cs);
  pe[(size_t)s * channels + i] = sn;
  pe[(size_t)s * channels + half + i] = cs;
  if ((channels & 1) && i == 0) pe[(size_t)s * channels + channels - 1] = 0.0f;
}
}  // namespace
int launch_pos_encoding(float* pe, int S, int channels, hipStream_t st) {
  SMD_ARG_CHECK(pe && S > 0 && channels >= 4, "pos_encoding: bad arguments");
  hipLaunchKernelGGL(pos_encoding_kernel, dim3((S * (channels / 2) + 255) / 256), dim3(256), 0, st, pe, S, channels);
  SMD_LAUNCH_CHECK();
  return 0;
}
