// Internal C++ launcher interface of the smd HIP library (not part of the C-ABI).
// Every launcher enqueues on `st`, touches only caller-owned device memory and returns the
// library error convention (0 ok, <0 argument error, >0 hipError_t).
#pragma once
#include "smd_common.h"

// ------------------------------------------------------------------ GEMM (gemm_nt.hip / gemm_tn.hip)
enum { SMD_ACT_NONE = 0, SMD_ACT_GELU = 1, SMD_ACT_SWISH = 2 };
enum { SMD_AUX_NONE = 0, SMD_AUX_GELU_GRAD = 1, SMD_AUX_SWISH_GRAD = 2 };

// Epilogue applied to the fp32 accumulator tile, in this order:
//   v = alpha*acc + bias[col]; pre_bf16 <- v; v = act(v); v *= aux'(aux[row][col]);
//   v += res_f32[row % res_row_mod][col] (+ res_bf16); out_f32 (= or +=) v; out_bf16 <- v
struct GemmEpilogue {
  float alpha = 1.0f;
  const float* bias = nullptr;       // [N]
  int act = SMD_ACT_NONE;
  bf16_t* pre_bf16 = nullptr;        // [M][ld_pre]  value before the activation (training: saved z)
  int ld_pre = 0;
  const bf16_t* aux = nullptr;       // [M][ld_aux]  pre-activation whose act'() multiplies (dgrad)
  int ld_aux = 0;
  int aux_mode = SMD_AUX_NONE;
  const float* res_f32 = nullptr;    // [*][ld_res]
  int ld_res = 0;
  int res_row_mod = 0;               // >0: residual row = row % res_row_mod (positional table)
  const bf16_t* res_bf16 = nullptr;  // [M][ld_resb]
  int ld_resb = 0;
  float* out_f32 = nullptr;          // [M][ld_out]
  int ld_out = 0;
  int accumulate = 0;                // out_f32 += v instead of =
  bf16_t* out_bf16 = nullptr;        // [M][ld_outb]
  int ld_outb = 0;
};

// C[M,N] = A[M,K] * Bt[N,K]^T   (both operands K-contiguous bf16; K % 64 == 0; lda,ldb % 8 == 0)
int launch_gemm_nt(const bf16_t* A, int lda, const bf16_t* Bt, int ldb, int M, int N, int K,
                   const GemmEpilogue& ep, hipStream_t st);

// 256x256x64 8-phase kernel (gemm_nt256.hip); launch_gemm_nt dispatches to it when eligible and enabled
bool gemm_nt256_eligible(int M, int N, int K, const GemmEpilogue& ep, int min_tiles = 192);
int launch_gemm_nt256(const bf16_t* A, int lda, const bf16_t* Bt, int ldb, int M, int N, int K,
                      const GemmEpilogue& ep, hipStream_t st);
// the same tile and pipeline on OCP e4m3 operands with one E8M0 (power-of-two) scale per row of A and of Bt (dword
// arrays, byte 0), v_mfma_scale_f32_32x32x64_f8f6f4: M, N % 256 == 0, K % 256 == 0, lda / ldb in elements (= bytes)
int launch_gemm_nt256_fp8(const unsigned char* A8, int lda, const uint32_t* scale_a, const unsigned char* Bt8, int ldb,
                          const uint32_t* scale_b, int M, int N, int K, const GemmEpilogue& ep, hipStream_t st);
// rows of a bf16 matrix -> e4m3 bytes + one E8M0 scale per row (weights of the e4m3 GEMM path): K % 512 == 0
int launch_quantize_rows_e4m3(const bf16_t* in, int ld, int rows, int K, unsigned char* out8, uint32_t* scale, hipStream_t st);
// process-wide kernel-selection knobs (benchmark A/B; defaults are the fast paths). Keys: "gemm_nt256", "gemm_nt256_variant", "gemm_tn256".
int smd_tuning_set(const char* key, int value);
int smd_tuning_get(const char* key);

// dW[Kd,N] = sum_m X[m,Kd] * dY[m,N] and (optionally) db[N] = sum_m dY[m,N]   (wgrad; both operands
// have the contraction index m as the row index).  X [Mrows][ldx] bf16, dY [Mrows][ldy] bf16,
// out fp32 [Kd][ldo].  tr_path 1: LDS transpose-read kernel (needs `zero_page`: 128 zeroed bf16, and
// `slab`: gemm_tn_slab_elems() floats for deterministic split-K partials); tr_path 0: explicit
// transposed copies through `scratch` (>= (Kd+N)*roundup(Mrows,64) bf16) + the NT kernel.
struct TnLaunch {
  const bf16_t* X = nullptr; int ldx = 0;
  const bf16_t* dY = nullptr; int ldy = 0;
  int Mrows = 0, Kd = 0, N = 0;
  float* out = nullptr; int ldo = 0;
  float* bias_out = nullptr;
  const bf16_t* zero_page = nullptr;
  float* slab = nullptr; size_t slab_elems = 0;
  bf16_t* scratch = nullptr; size_t scratch_elems = 0;
  int tr_path = 1;
};
size_t gemm_tn_slab_elems();
// Weight-gradient GEMMs normally run on the engine's side stream next to the backward chain.  With the TWO-buffer 128-wide
// kernel on their CU, the small 128-wide LayerNorm-backward kernels lost bitwise repeatability (round 3, DESIGN.md section 6:
// a v_rsq_f32 directly behind the VALU that writes its source reads the stale register in lanes 48..63).  Knob
// "tn_exclusive_cu": 2 (default) = the four-buffer kernels, unpadded (0 of 1499 repeats differ with any LayerNorm build);
// 1 = the same kernels with a dynamic-LDS pad that fills the CU's 160 KiB (round-2 default); 0 = the two-buffer kernel (A/B only).
int smd_tn_pad_bytes(int static_lds_bytes);
// 256x256 8-phase wgrad kernel (gemm_tn256.hip): plan returns nsplit (0 = not eligible)
int gemm_tn256_plan(const TnLaunch& t, int* ktiles_per_split);
int launch_gemm_tn256(const TnLaunch& t, int nsplit, int ktiles_per_split, hipStream_t st);
int launch_gemm_tn256_pair(const TnLaunch& t0, const TnLaunch& t1, hipStream_t st);   // two problems, one launch
#define SMD_TN256_MULTI_MAX 4
int launch_gemm_tn256_multi(const TnLaunch* ts, int n, hipStream_t st);                 // 1..4 problems, one launch (4 x 64 tiles: no split-K)
int launch_gemm_tn(const TnLaunch& t, hipStream_t st);
// n wgrad problems with the same Mrows (ldo == N each) in grouped launches of the 128-wide kernel + one reduce
int launch_gemm_tn_grouped(const TnLaunch* probs, int n, hipStream_t st);
int launch_transpose_bf16(const bf16_t* in, int ld_in, int rows, int cols, bf16_t* out, int ld_out,
                          hipStream_t st);
// out[n] = sum_m dY[m][n]  (bias gradient), deterministic two-stage reduction via `partial`
int launch_colsum_bf16(const bf16_t* dY, int ldy, int rows, int cols, float* out, float* partial,
                       size_t partial_elems, hipStream_t st);

// ------------------------------------------------------------------ normalisation (norm.hip)
// y = LN(x) * gamma + beta ; optional FiLM (scale*y + shift per sample) and swish; bf16 out.
// x is fp32 [rows][D] (ldx = D).  film_scale/shift: [rows/rows_per_sample][ld_film] fp32 or null.
// If t_ptr != null the FiLM row is *t_ptr for every sample (batch-uniform sampling tables).
struct LnArgs {
  const float* x = nullptr;
  const bf16_t* x_bf16 = nullptr;    // alternative bf16 input (exactly one of x / x_bf16)
  int rows = 0, D = 0;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  const float* film_scale = nullptr;
  const float* film_shift = nullptr;
  int ld_film = 0;
  int rows_per_sample = 1;
  const int* t_ptr = nullptr;
  int film_rows = 1 << 30;           // rows of the FiLM table when t_ptr indexes it (num_timesteps): *t_ptr is clamped to it
  int swish = 0;
  bf16_t* out = nullptr;             // [rows][D]
  // e4m3 copy of the output with one power-of-two scale per row (D = 1024 / 2048 row-group kernels only): out_f8
  // [rows][D] bytes = e4m3(y * 2^-e), out_scale [rows] dwords = E8M0 byte e + 127; `out` may then be null
  unsigned char* out_f8 = nullptr;
  uint32_t* out_scale = nullptr;
};
int launch_layernorm_fwd(const LnArgs& a, hipStream_t st);

// Backward of the same op.  dout bf16 [rows][D] is the gradient wrt `out`.
//   dx = LN-backward(dout) + (dres ? dres : 0), written as fp32 (dx) and/or bf16 (dx_bf16);
//   dres may alias dx (in-place residual-gradient stream).
//   dgamma/dbeta (+=, the flat gradient buffer is zeroed once per step) via per-group partials
//   in `partial`; dscale/dshift (fp32 [nsamples][ld_film]) = or += per-sample sums over the
//   sample's rows (the two FiLM uses inside one DenseResBlock share scale/shift).
// deferred dgamma/dbeta reduction of one LayerNorm backward (filled by launch_layernorm_bwd when requested)
struct LnReduceEntry { const float* partial; int ngroups; int D; float* dgamma; float* dbeta; int block_start; };
#define SMD_LN_REDUCE_MAX 24
struct LnReduceTable { int n; LnReduceEntry e[SMD_LN_REDUCE_MAX]; };
struct LnBwdArgs {
  LnArgs f;                          // the forward arguments (x, gamma, beta, film, swish); f.out unused
  const bf16_t* dout = nullptr;
  const float* dres = nullptr;
  const bf16_t* dres_bf16 = nullptr;  // bf16 residual gradient instead of dres (D in {1024, 2048})
  float* dx = nullptr;
  bf16_t* dx_bf16 = nullptr;
  float* dgamma = nullptr;
  float* dbeta = nullptr;
  float* dscale = nullptr;
  float* dshift = nullptr;
  int dfilm_accumulate = 0;
  LnReduceEntry* deferred = nullptr; // non-null: skip the reduce launch, describe it here (partial must stay intact)
  float* partial = nullptr;          // workspace >= ln_bwd_partial_elems(rows, D)
  size_t partial_elems = 0;
};
size_t ln_bwd_partial_elems(int rows, int D);
int launch_layernorm_bwd(const LnBwdArgs& a, hipStream_t st);
// one launch for the dgamma/dbeta (+=) reductions of up to many LayerNorms (24 per kernel launch)
int launch_ln_bwd_reduce_batched(const LnReduceEntry* entries, int n, hipStream_t st);

// ------------------------------------------------------------------ attention (attention.hip)
// qkv bf16 [B*S][3E] ([q|k|v], head h at cols h*d..), out bf16 [B*S][E]; S == 32, d in {8,16,32}
int launch_attention_fwd(const bf16_t* qkv, bf16_t* out, int B, int S, int E, int H, hipStream_t st);
int launch_attention_bwd(const bf16_t* qkv, const bf16_t* dout, bf16_t* dqkv, int B, int S, int E,
                         int H, hipStream_t st);

// ------------------------------------------------------------------ fused encoder half-layers (encoder_fused.hip)
// h_out = h_in + fc2(gelu(fc1(LN(h_in))))  for rows % 32 == 0; W1t [M][128], W2t [128][M] (the forward operand
// pack); optional saves for the backward pass: a2 = LN output [rows][128], z1 / u = pre / post GELU [rows][M]
int launch_mlp_block_fwd(const float* h_in, float* h_out, int rows, const float* gamma, const float* beta,
                         const bf16_t* W1t, const float* b1, const bf16_t* W2t, const float* b2, int M, bf16_t* save_a2,
                         bf16_t* save_z1, bf16_t* save_u, hipStream_t st);

// The MLP half-layer with the hidden dimension split over workgroups (4 quarters x groups of 4 / 2 / 1 samples: a
// quarter of the weight stream per CU).  Input a2 = ln2(h_mid) in bf16 (emitted by launch_attn_block_fwd); output four
// fp32 partial tiles part[k] = part + k * rows * 128 with h_out = (p0 + p1) + (p2 + p3) (quarter 0 carries b2 and the
// residual h_res), summed by the consumer: the next attention kernel or launch_ln128_parts.  Needs M % 512 == 0.
int mlp_hs_samples_per_group(int rows, int M);          // 4 / 2 / 1, or 0 when the shape is not supported
int launch_mlp_block_fwd_hs(const bf16_t* a2, const float* h_res, int rows, const bf16_t* W1t, const float* b1,
                            const bf16_t* W2t, const float* b2, int M, float* part, hipStream_t st);
// x = (p0 + p1) + (p2 + p3) per 128-wide row; x_out (nullable) <- x; ln_out (nullable) <- LayerNorm(x) in bf16
int launch_ln128_parts(const float* parts, size_t part_stride, int rows, const float* gamma, const float* beta, float* x_out,
                       bf16_t* ln_out, hipStream_t st);

// Backward of the MLP half-layer with the hidden activations recomputed from a2 (nothing but a2 saved by the forward):
// writes u = gelu(a2 W1 + b1) and dz = (dh W2^T) * gelu'(.) ([rows][M] bf16: operands of the two weight gradients) and
// four fp32 partial tiles of da2 = dz W1^T.  W1t [M][128] / W2 [M][128] / W1 [128][M]: fc1 forward pack, fc2 and fc1
// dgrad packs.  rows % 128 == 0, M % 512 == 0.
int launch_mlp_block_bwd_hs(const bf16_t* a2, const bf16_t* dh, int rows, const bf16_t* W1t, const bf16_t* W2, const bf16_t* W1,
                            const float* b1, int M, bf16_t* u, bf16_t* dz, float* part, hipStream_t st);
// LayerNorm (D = 128) backward on dout = (p0 + p1) + (p2 + p3): dx = LN-bwd + dres -> fp32 (may alias dres) / bf16;
// partial: [rows/32][2][128] dgamma / dbeta group sums (reduced by launch_ln_bwd_reduce_batched)
int launch_ln128_bwd_parts(const float* x, const float* parts, size_t part_stride, int rows, const float* gamma, const float* dres,
                           float* dx_f32, bf16_t* dx_bf16, float* partial, hipStream_t st);

// optional extras of the attention half-layer kernel: partial-sum input and the ln2 of its output
struct AttnBlockExtra {
  const float* h_parts = nullptr;   // input x = (p0 + p1) + (p2 + p3), parts h_parts + k * part_stride (h_in unused)
  size_t part_stride = 0;
  float* h_comb = nullptr;          // x written out (nullable)
  const float* gamma2 = nullptr;    // LayerNorm of the output rows -> a2_out (bf16 [rows][128], nullable)
  const float* beta2 = nullptr;
  bf16_t* a2_out = nullptr;
};

// h_out = h_in + out_proj(attention(qkv(LN(h_in)))), S = 32 tokens per sample, E = 128; Wqkv_t [384][128]
// (q | k | v rows, head h = rows h*d..), Wo_t [128][128]; optional saves: a1 = LN output, qkv (q unscaled), o
int launch_attn_block_fwd(const float* h_in, float* h_out, int rows, const float* gamma, const float* beta,
                          const bf16_t* Wqkv_t, const float* b_qkv, const bf16_t* Wo_t, const float* b_o, int num_heads,
                          bf16_t* save_a1, bf16_t* save_qkv, bf16_t* save_o, hipStream_t st,
                          const AttnBlockExtra* extra = nullptr);

// backward of the same half-layer between the two LayerNorms: dqkv (written for the qkv wgrad) and da1 = gradient
// wrt the LN1 output, from dh_mid (bf16), the saved qkv and the dgrad operand packs Wo [128][128], Wqkv [128][384]
int launch_attn_block_bwd(const bf16_t* dh_mid, const bf16_t* qkv, const bf16_t* Wo, const bf16_t* Wqkv, bf16_t* dqkv,
                          bf16_t* da1, int rows, int num_heads, hipStream_t st);

// The same launch WITH the two LayerNorm backwards that bracket the attention in the backward pass (hidden-split MLP dataflow):
//   dh_mid = LN2-backward((p0 + p1) + (p2 + p3) of da2_parts; h_mid, gamma2) + dh     -> dh_mid_out (bf16), partial2
//   ... attention half-layer backward from dh_mid ...                                  -> dqkv (, da1)
//   dh     = LN1-backward(bf16(da1); h, gamma1) + dh_mid                               -> dh (fp32, IN PLACE), dh_out (bf16), partial1
// partial1 / partial2: [rows/32][2][128] dgamma | dbeta sums per sample (reduced by launch_ln_bwd_reduce_batched).
struct AttnBwdLnArgs {
  const bf16_t* qkv = nullptr;        // [R][384] saved q | k | v
  const bf16_t* Wo = nullptr;         // dgrad operand packs W [in][out]
  const bf16_t* Wqkv = nullptr;
  bf16_t* dqkv = nullptr;             // [R][384]
  bf16_t* da1 = nullptr;              // [R][128], optional
  const float* h_mid = nullptr;       // [R][128] input of LayerNorm 2
  const float* da2_parts = nullptr;   // four partial tiles, part_stride floats apart
  size_t part_stride = 0;
  const float* gamma2 = nullptr;
  float* dh = nullptr;                // [R][128] fp32 residual-stream gradient (read, then overwritten)
  bf16_t* dh_mid_out = nullptr;       // [R][128]
  float* partial2 = nullptr;
  const float* h = nullptr;           // [R][128] input of LayerNorm 1
  const float* gamma1 = nullptr;
  bf16_t* dh_out = nullptr;           // [R][128]
  float* partial1 = nullptr;
};
int launch_attn_block_bwd_ln(const AttnBwdLnArgs& a, int rows, int num_heads, hipStream_t st);

// ------------------------------------------------------------------ diffusion elementwise (diffusion.hip)
// sinusoidal noise embedding, reference models/ncsn.py:28-41: s[n] -> bf16 [n][channels]
int launch_noise_embed(const float* s, int n, int channels, bf16_t* out, int ld_out, hipStream_t st);

struct RngKey { uint32_t seed_lo, seed_hi; };

// q-sample (reference utils/losses.py:271-296).  x0 fp32 [B][S*C]; tables on device.
struct QSampleArgs {
  const float* x0 = nullptr;
  int B = 0, S = 0, C = 0, Cp = 0;       // Cp: padded row length of xt_bf16
  int T = 0;
  const float* alphas_prod_ext = nullptr; // [T+1] = [1, cumprod(1-beta)]
  const int* labels = nullptr;            // [B] explicit labels in [0,T] or null -> Philox
  int label_min = 1;                      // Philox labels in [label_min, label_min + T): 1 = continuous_noise, 0 = not
  const float* alpha_in = nullptr;        // [B] explicit used_alphas (utils/losses.py:283-286) or null -> from the label
  int dsm = 0;                            // 1: denoising score matching (:163-165): alpha_in holds used_sigmas,
                                          //    x_t = x0 + sigma * eps and s_out = sigma
  const float* eps_in = nullptr;          // explicit eps or null -> Philox
  RngKey key{0, 0};
  const uint32_t* step_ptr = nullptr;     // device step counter (RNG stream offset), may be null
  uint32_t sample_offset = 0;             // global index of sample 0 (data-parallel shard offset)
  bf16_t* xt_bf16 = nullptr;              // [B*S][Cp] zero padded
  float* eps_out = nullptr;               // [B*S][C]
  float* s_out = nullptr;                 // [B] sqrt(alpha) noise level
};
int launch_q_sample(const QSampleArgs& a, hipStream_t st);

// loss + dpred: loss_b = mean_{s,c} (eps-pred)^2 ; dpred = 2 (pred-eps) / (Bglobal*S*C) -> bf16 [B*S][Cp]
int launch_mse_loss_grad(const float* pred, const float* eps, int B, int S, int C, int Cp,
                         float inv_global_count, float* loss_per_sample, bf16_t* dpred_bf16,
                         hipStream_t st, const float* dsm_sigma = nullptr);   // dsm_sigma [B]: the score-matching form

// fused reverse step (reference utils/ebm_utils.py:327-394)
struct ReverseStepArgs {
  float* x = nullptr;                 // [B][S][C] state, updated in place
  const float* eps_hat = nullptr;     // [B][S][C]
  int B = 0, S = 0, C = 0, Cp = 0, T = 0;
  const float* coef = nullptr;        // [T][8]: sqrt_recip, sqrt_m1, mu1, mu2, sigma, alpha_prod, sqrt_ap, sqrt_1m
  const int* t_ptr = nullptr;         // device timestep; t outside [0, T) makes the launch a no-op
  int* t_advance = nullptr;           // non-null: the last workgroup stores t - 1 here (normally == t_ptr)
  unsigned* arrive = nullptr;         // arrival counter for t_advance: zero before the first launch, reset by the kernel
  const float* z_in = nullptr;        // explicit N(0,1) draw [B][S][C] or null -> Philox
  RngKey key{0, 0};
  const uint32_t* key_ptr = nullptr;  // device-resident key [2] (overrides `key`: a captured step serves later runs with other seeds)
  uint32_t sample_offset = 0;
  const float* infill_samples = nullptr;  // [B][S][C] or null
  const float* infill_masks = nullptr;
  const float* infill_z_in = nullptr;
  // jax.random streams drawn in the kernel (rng_threefry.h): per-iteration keys [iters][2], row (tf_t0 - t); the state
  // is the window [sample_offset*S*C, ...) of a global array of tf_n_total elements.  Overrides Philox when set.
  const uint32_t* tf_noise_keys = nullptr;
  const uint32_t* tf_infill_keys = nullptr;
  int64_t tf_n_total = 0;
  int tf_t0 = 0;
  bf16_t* x_bf16 = nullptr;           // [B*S][Cp] next network input (zero padded)
  float* metrics_partial = nullptr;   // [T][B][3] (grad, step, noise) sums over c of sqrt(sum_s v^2 + 1e-10)
  float* collection = nullptr;        // [41][B][S][C] or null
  const int* slot_table = nullptr;    // [T] slot for timestep t or -1
};
int launch_reverse_step(const ReverseStepArgs& a, hipStream_t st);
int launch_advance_t(int* t_ptr, hipStream_t st);   // *t_ptr -= 1
int launch_set_t(int* t_ptr, int v, hipStream_t st);   // *t_ptr = v (restart of a walk on the sampler's own stream)

// Langevin update of annealed_langevin_dynamics / consistent_langevin_dynamics (utils/ebm_utils.py:131-164, 231-253)
struct LangevinStepArgs {
  float* x = nullptr;                 // [B][S][C] state, updated in place
  const float* grad = nullptr;        // [B][S][C] model(state, sigma)
  int B = 0, S = 0, C = 0;
  float alpha = 0.f;                  // step size
  float noise_coef = 0.f;             // sqrt(2 alpha) (annealed) or beta * next_sigma (consistent)
  const float* z_in = nullptr;        // explicit N(0,1) draw or null
  RngKey key{0, 0};
  uint32_t step = 0;                  // Philox counter word of this update
  uint32_t sample_offset = 0;
  int use_threefry = 0;               // jax.random.normal(step_rng / infill_rng) drawn in the kernel from the two keys below
  uint32_t tf_noise_key[2] = {0, 0};
  uint32_t tf_infill_key[2] = {0, 0};
  int64_t tf_n_total = 0;
  const float* infill_samples = nullptr;
  const float* infill_masks = nullptr;
  const float* infill_z_in = nullptr;
  float infill_sigma = 0.f;           // y = infill_samples + sigma * N(0,1)
  float* metrics_partial = nullptr;   // [B][3] (grad, step, noise): sums over c of sqrt(sum_s v^2 + 1e-10)
  float* collect_out = nullptr;       // [B][S][C] copy of the new state or null
  // table mode (include/smd_hip.h smd_langevin_io): step-dependent arguments from device tables indexed by *k_ptr
  const float* step_table = nullptr;  // [n][4] alpha, noise_coef, infill_sigma, sigma of the next update
  const int32_t* slot_table = nullptr;
  const uint32_t* key_table = nullptr;   // [n][4] threefry noise key | infill key
  int32_t* k_ptr = nullptr;
  uint32_t* arrive = nullptr;
  float* collection = nullptr;
  float* sigma_out = nullptr;
  int n_steps = 0;
  int32_t* level_out = nullptr;       // FiLM-table row of the next forward: min((k + 1) / steps_per_level, n_levels - 1)
  int steps_per_level = 1, n_levels = 1;
};
int launch_langevin_step(const LangevinStepArgs& a, hipStream_t st);

// fp32 [rows][cols] -> bf16 [rows][ld_out] zero padded
int launch_cast_pad_bf16(const float* in, int rows, int cols, bf16_t* out, int ld_out, hipStream_t st);
// Philox standard normals (for init state), element e of sample b -> counter (b+sample_offset, e/4)
int launch_fill_normal(float* out, int B, int per_sample, RngKey key, uint32_t stream,
                       uint32_t sample_offset, hipStream_t st);

// ------------------------------------------------------------------ jax.random-compatible draws (rng_jax.hip)
// window [offset, offset+count) of a logical n_total-element array (n_total <= 2^32); values identical to
// jax.random.{bits,uniform,normal,randint}(key, (n_total,)) of jax 0.2.8 -- see rng_threefry.h
int launch_threefry_bits(uint32_t* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1,
                         hipStream_t st);
int launch_threefry_uniform(float* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1,
                            float minval, float maxval, hipStream_t st);
// key_table != null: key = key_table[idx_add + idx_mul * *idx_ptr] (read on device: graph-replayable)
int launch_threefry_normal(float* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1,
                           const uint32_t* key_table, const int32_t* idx_ptr, int idx_mul, int idx_add, hipStream_t st);
int launch_threefry_randint(int32_t* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1,
                            int32_t minval, int32_t maxval, hipStream_t st);

// ------------------------------------------------------------------ optimiser (optim.hip)
struct AdamArgs {
  float* params = nullptr;            // fp32 master [n]
  const float* grads = nullptr;
  float* m = nullptr;
  float* v = nullptr;
  float* ema = nullptr;               // may be null
  size_t n = 0;
  float lr0 = 1e-3f, lr_gamma = 0.98f;
  int lr_interval = 10000;
  float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f, grad_clip = 1.0f, mu = 0.999f;
  float grad_scale = 1.0f;            // applied to grads before everything (1/world for DP sum)
  uint32_t* step_ptr = nullptr;       // device step counter, incremented by the kernel tail
  float* norm_partial = nullptr;      // workspace [>= 1024]
  float* metrics_out = nullptr;       // [4]: grad norm before clip, after clip, lr, step
};
int launch_grad_sumsq(const AdamArgs& a, hipStream_t st);
int launch_adam_clip_ema(const AdamArgs& a, hipStream_t st);
// engine path (optim.hip): norm partials in fixed slots, one-workgroup prepare (norm -> clip factor, LR, bias corrections in
// consts[0..3]; *step_ptr += 1), and clip + Adam + EMA + bf16 re-cast of the Dense kernels in one sweep over a tile table
#define SMD_NORM_SLOTS 2048        // one workgroup of the partial-sum pass per slot: 1536 for the output-stage slice, 512 for the stem
#define SMD_NORM_HEAD_SLOTS 1536
#define SMD_OPT_DENSE_MAX 60
#define SMD_OPT_FLAT_MAX 64
struct OptDense { uint32_t w_off, K, N, W_off, ldw, Wt_off, ldwt, blk_start; };   // one (K, N) Dense kernel: 64 x 64 tiles from blk_start
struct OptFlat { uint32_t off, len, blk_start; };                                 // biases / LayerNorm parameters: 1024 per block
struct OptTable {
  int n_dense = 0, n_flat = 0;
  uint32_t flat_blk0 = 0, total_blocks = 0;
  OptDense d[SMD_OPT_DENSE_MAX];
  OptFlat f[SMD_OPT_FLAT_MAX];
};
int launch_grad_sumsq_slots(const float* g, size_t n, float* partial, int nslots, hipStream_t st);
int launch_opt_prepare(const AdamArgs& a, int nslots, float* consts, hipStream_t st);
int launch_adam_recast(const AdamArgs& a, const float* consts, bf16_t* wpack, const OptTable& t, hipStream_t st, int max_blocks = 0);

// master fp32 kernel (K_in, N_out) -> bf16 W [Kp][ldw] (zero padded) and Wt [N_out][ldwt]
int launch_recast_weight(const float* w, int K_in, int N_out, bf16_t* W, int ldw, bf16_t* Wt, int ldwt,
                         hipStream_t st);

// all Dense kernels in one launch (table passed by value: <= SMD_RECAST_MAX weights per launch)
#define SMD_RECAST_MAX 60
struct RecastEntry { uint32_t w_off, K, N, W_off, ldw, Wt_off, ldwt, tile_start; };
struct RecastTable { int n; RecastEntry e[SMD_RECAST_MAX]; };
int launch_recast_all(const float* params, bf16_t* wpack, const RecastTable& t, int total_tiles, hipStream_t st);
int launch_probe_tr_read(const bf16_t* image, bf16_t* out, hipStream_t st);   // gemm_tn.hip debug probe
int launch_pos_encoding(float* pe, int S, int channels, hipStream_t st);      // models/shared.py:36-48
// generic small helpers
int launch_swish_bwd_bf16(const bf16_t* pre, const bf16_t* dout, bf16_t* din, size_t n, hipStream_t st);
