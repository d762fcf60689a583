/* smd_hip.h -- C-ABI of libsmd_hip.so, the MI355X (gfx950) DDPM train + sample engine.
 *
 * The reference (magenta/symbolic-music-diffusion) has no FFI: its seam for this path is a set
 * of Python callables.  Each entry point below names the reference callable
 * (file:line under the reference repo root) whose arithmetic it replaces; INTEGRATION.md shows the
 * ctypes stub a maintainer would add on the reference side.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.  All `void*`/`float*` data pointers are
 *     DEVICE pointers borrowed from the caller (never freed, never retained past the handle's
 *     bindings).  `stream` is a hipStream_t passed as void*.  Calls only enqueue work.
 *   - return value: 0 ok; < 0 argument / state error; > 0 a hipError_t.  smd_last_error() returns
 *     a thread-local message for the last non-zero return.
 *   - bf16 buffers are uint16 storage (`smd_bf16`).  GEMM operands are row-major with the
 *     contraction dimension padded to a multiple of 64 (zero filled).
 *   - entry points are re-entrant and stream-ordered.  The only process-wide mutable state is a thread-local error string
 *     (and the kernel-selection table behind smd_set_tuning() of smd_hip_lab.h: benchmark A/B knobs, the defaults are the
 *     shipped paths); everything else lives in the caller's buffers or in an smd_engine handle.
 *   - THIS header is the stable surface a maintainer binds.  What the repository's own laboratory uses --
 *     tuning knobs, debug views of saved activations, hardware probes -- is declared in smd_hip_lab.h and may change freely.
 */
#ifndef SMD_HIP_H_
#define SMD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMD_ABI_VERSION 6   /* 6: smd_attn_block_bwd_ln (the attention backward with both LayerNorm backwards in the launch), "fused_attn_bwd" 2; 2: smd_ddpm_reverse_step takes T; hidden-split MLP, fp8, loss-side and Langevin entries; 3: smd_langevin_io table mode, debug snapshots; 4: one-sweep optimiser, "opt_overlap", smd_engine_join_update; 5: smd_build_id, smd_engine_sample_step_part, smd_engine_forward_train / backward_from; the lab hooks (tuning knobs, debug tensors, probes) moved to smd_hip_lab.h -- this header is the stable surface */

typedef uint16_t smd_bf16;
typedef struct smd_engine smd_engine;

const char* smd_last_error(void);
int smd_abi_version(void);
/* 16 hex digits: a hash of the sources, headers and compiler flags this binary was built from (build.py source_id()).  The
 * host refuses to run a library whose id differs from its tree's (a stale .so next to edited sources). */
const char* smd_build_id(void);

/* ---- model description: the kwargs of train_ncsn.py:321-326 + data shape ---------------------- */
typedef struct smd_model_desc {
  int32_t arch;            /* 0 TransformerDDPM (models/ncsn.py:138-179), 1 DenseDDPM (:122-135) */
  int32_t data_channels;   /* C */
  int32_t seq_len;         /* S (32; 1 for DenseDDPM) */
  int32_t num_layers;      /* --num_layers       train_ncsn.py:69 */
  int32_t num_heads;       /* --num_heads        train_ncsn.py:70 */
  int32_t num_mlp_layers;  /* --num_mlp_layers   train_ncsn.py:71 */
  int32_t mlp_dims;        /* --mlp_dims         train_ncsn.py:72 */
  int32_t embed_channels;  /* 128, models/ncsn.py:151 */
  int32_t film_channels;   /* 128, models/ncsn.py:174 */
  int32_t num_timesteps;   /* --num_sigmas       train_ncsn.py:81 */
} smd_model_desc;

/* ---- engine handle ---------------------------------------------------------------------------- */
/* replaces train_ncsn.create_model (train_ncsn.py:193-203): builds the parameter layout only */
int smd_engine_create(const smd_model_desc* desc, smd_engine** out);
void smd_engine_destroy(smd_engine* e);

/* parameter pytree as a flat fp32 buffer: tensor i = (name, element offset, rows, cols[0 = 1-D]) */
int smd_engine_num_tensors(const smd_engine* e);
int smd_engine_tensor_info(const smd_engine* e, int i, const char** name, int64_t* offset, int32_t* rows,
                           int32_t* cols);
int64_t smd_engine_param_count(const smd_engine* e);
int64_t smd_engine_head_param_offset(const smd_engine* e);  /* first parameter of the output stage */
int64_t smd_engine_wpack_elems(const smd_engine* e);        /* bf16 elements of the GEMM operand pack */
int64_t smd_engine_workspace_bytes(const smd_engine* e, int batch, int training);
int64_t smd_engine_film_table_floats(const smd_engine* e);
int smd_engine_padded_channels(const smd_engine* e);
/* Per-handle options (set before the first bind unless noted; unknown keys return < 0).  Defaults are the shipped paths:
 *   "tr_path" 1          weight gradients with transposing LDS reads (0: explicit-transpose fallback)
 *   "side_wgrad" 0/1     weight-gradient GEMMs (and the gradient memset, FiLM chains) on the engine's low-priority side
 *                        stream, joined inside smd_engine_loss_backward; the Python host turns it on for training handles
 *   "group_wgrad" 2, "pair_wgrad" 1, "film_side" 1, "film_side_fwd" 1, "tail_on_main" 1   launch grouping of those GEMMs
 *   "fused_encoder" 1, "fused_attn_bwd" 2 (2: + both LayerNorm backwards in the launch, 1: attention only), "mlp_hs" 1   fused encoder half-layers / hidden-split MLP (0: separate launches)
 *   "resgrad_bf16" 1     DenseResBlock residual-gradient chain in bf16
 *   "trunk_bf16" 2       2048-wide residual stream in bf16 (2: inference + training, 1: inference only, 0: fp32)
 *   "fp8" 0/1            e4m3 DenseResBlock forward GEMMs (BASELINE config 5); "w8_dirty" 1: operand pack changed elsewhere
 *   "label_min" 1/0      Philox labels in [1, T] (continuous_noise) or [0, T);  "loss_kind" 0/1  DDPM / score matching
 *   "grad_memset" 2      0 never / 1 always / 2 only with the tr_path = 0 fallback: zero the gradient buffer before a step
 *                        (every gradient element is written, not accumulated, by the default kernels)
 *   "nt256_min_tiles" 0  > 0: Dense layers take the 256x256 GEMM from this many output tiles up (default rule: 192)
 *   "fp8_dgrad" 1        fp8 mode, training: the DenseResBlock dgrad GEMMs on e4m3 operands too
 *   "dp_layer_events" 0  1: per-encoder-layer gradient-complete events in the stem backward (smd_engine_wait_grad_bucket)
 *   "opt_overlap" 0      smd_engine_optimizer_step placement (any time).  Bit 0: the update of the output-stage slice (parameters
 *                        >= smd_engine_head_param_offset, 86 % of the bytes) runs on the side stream and is NOT complete in
 *                        stream order when the call returns: the next smd_engine_loss_backward of the same handle waits for it
 *                        right before its first output-stage kernel; any other reader of params / m / v / ema / wpack (another
 *                        handle on the same buffers, a host copy) calls smd_engine_join_update(handle, its stream) first.
 *                        Bit 1: smd_engine_loss_backward(stage 0) reduces that slice's gradient-norm partials on the side
 *                        stream under the encoder backward; the caller must not change the gradient before the optimiser step
 *                        (leave it off around an all-reduce).  Opt-in (the Python host leaves 0 unless SMD_OPT_OVERLAP is set): +1 % at best on a
 *                        process's first training handle, and bit 0 halved the step rate of later handles of the same process
 *                        (DESIGN.md section 6). */
int smd_engine_set_option(smd_engine* e, const char* key, int value);

int smd_engine_bind_params(smd_engine* e, float* params, smd_bf16* wpack);
int smd_engine_bind_train(smd_engine* e, float* grads, float* adam_m, float* adam_v, float* ema /*nullable*/,
                          uint32_t* step_counter, float* metrics /*[4]: |g|, |g| clipped, lr, step*/);
int smd_engine_bind_workspace(smd_engine* e, void* workspace, int64_t bytes, int batch, int training,
                              void* stream);
/* coef [T][8] = (sqrt(1/ap), sqrt(1-ap)/sqrt(ap), mu1, mu2, sigma, ap, sqrt(ap), sqrt(1-ap)) per t
 * (utils/ebm_utils.py:332-358); sqrt_ap [T]; alphas_prod_ext [T+1] = [1, cumprod(1-beta)]
 * (utils/losses.py:277-281); film_tables: smd_engine_film_table_floats() floats or NULL */
int smd_engine_bind_schedule(smd_engine* e, const float* coef, const float* sqrt_ap,
                             const float* alphas_prod_ext, float* film_tables);
int smd_engine_refresh_weights(smd_engine* e, void* stream);   /* fp32 master -> bf16 operand pack */

/* nn.Model.__call__: model(x:(B,S,C), noise_level:(B,)) -> eps_hat (models/ncsn.py:141-179, 125-135) */
int smd_engine_forward(smd_engine* e, const float* x, const float* noise_level, float* eps_out, void* stream);
/* model(x, level) for a BATCH-UNIFORM noise level given as a device-side row index into the FiLM tables built by
 * smd_engine_prepare_sampler (row r = noise level sqrt_ap[r] of the bound schedule): no FiLM-generator launches and nothing
 * step-dependent on the host, so an (eps-net forward + Langevin update) pair can be captured once and replayed
 * (utils/ebm_utils.py:140,242 -- the noise level is the same for the whole batch inside both Langevin samplers).  out may be
 * NULL: the result then stays in the engine's own buffer, smd_engine_pred(). */
int smd_engine_forward_level(smd_engine* e, const float* x, const int32_t* level_ptr, float* out, void* stream);

/* diffusion_loss + value_and_grad (utils/losses.py:250-308, train_ncsn.py:282-283).
 * labels/eps_in NULL -> on-device Philox draws keyed by (seed, global sample index, step).
 * inv_global_count = 1 / (global_batch * S * C).  stage 0 = everything, 1 = forward + output-stage
 * backward (gradients of parameters >= head offset are final), 2 = remaining backward. */
int smd_engine_loss_backward(smd_engine* e, const float* x0, const int32_t* labels, const float* eps_in,
                             uint32_t seed_lo, uint32_t seed_hi, uint32_t sample_offset,
                             float inv_global_count, int stage, void* stream);
/* jax.value_and_grad over an ARBITRARY objective (train_ncsn.py:279-283, `objective` is any callable of the model's output):
 * smd_engine_forward_train = model(x, noise_level) in the bound TRAINING workspace, every activation the backward needs saved
 * (eps_out [B][S][C] fp32, nullable: smd_engine_pred); smd_engine_backward_from = the backward pass of that forward from
 * dpred = d objective / d eps_hat ([B][S][C] fp32), parameter gradients WRITTEN to the bound gradient buffer.  stage as in
 * smd_engine_loss_backward (0 everything, 1 output stage, 2 the rest; dpred is read by stages 0 and 1).  The host binds the pair
 * to torch autograd (ops.py smd_amd::eps_forward_train), so train_step accepts objectives other than the two fused ones. */
int smd_engine_forward_train(smd_engine* e, const float* x, const float* noise_level, float* eps_out, void* stream);
int smd_engine_backward_from(smd_engine* e, const float* dpred, int stage, void* stream);
/* continuous_noise=False (utils/losses.py:272-286; option "label_min" = 0): labels are drawn in [0, T) and label 0 takes
 * a real uniform used_alpha in [alphas_prod[T], 1).  used_alphas ([B] device floats, or NULL) replaces the label -> alpha
 * lookup of the following smd_engine_loss_backward calls (parity mode: the reference's own jax.random.uniform draws). */
int smd_engine_set_used_alphas(smd_engine* e, const float* used_alphas);
const float* smd_engine_loss_per_sample(const smd_engine* e);   /* [B] device pointer */
const float* smd_engine_pred(const smd_engine* e);              /* [B*S][C] device pointer */

/* clip_grads + Adam + stepped LR + EMA + bf16 re-cast (train_ncsn.py:284-287,340-342,364-365) */
typedef struct smd_train_hyper {
  float lr0, lr_gamma;
  int32_t lr_interval;
  float beta1, beta2, eps, grad_clip, mu;
  float grad_scale;        /* multiplies the gradients first (1/world_size after a SUM all-reduce) */
} smd_train_hyper;
int smd_engine_optimizer_step(smd_engine* e, const smd_train_hyper* h, void* stream);
/* Data-parallel gradient buckets of the stem slice [0, head offset) in BACKWARD order: bucket b = encoder layer L-1-b (the last
 * bucket also holds in_proj).  With option "dp_layer_events" = 1, smd_engine_loss_backward (stage 2 or 0) records an event per
 * bucket as soon as its gradients are final; smd_engine_wait_grad_bucket makes `stream` (the communication stream) wait for it,
 * so the collective of layer l starts while the layers below are still in their backward pass.  The last bucket is final when
 * the call returns (in the caller's stream order).  (The reference has no counterpart: jax.pmap's pmean at train_ncsn.py:282.) */
int smd_engine_num_grad_buckets(const smd_engine* e);
int smd_engine_grad_bucket(const smd_engine* e, int bucket, int64_t* offset, int64_t* length);
int smd_engine_wait_grad_bucket(smd_engine* e, int bucket, void* stream);
/* Makes `stream` wait for an output-stage update deferred by "opt_overlap" bit 0 (no-op when none is pending). */
int smd_engine_join_update(smd_engine* e, void* stream);

/* diffusion_dynamics (utils/ebm_utils.py:280-405), one sample_with_beta iteration per call */
typedef struct smd_sample_io {
  float* x;                        /* [B][S][C] state, in place */
  int32_t* t_ptr;                  /* device timestep; decremented by the call */
  const float* z_in;               /* explicit N(0,1) draw or NULL (Philox) */
  uint32_t seed_lo, seed_hi, sample_offset;
  const float* infill_samples;     /* NULL unless infilling */
  const float* infill_masks;
  const float* infill_z_in;
  float* metrics_partial;          /* [T][B][3] or NULL */
  float* collection;               /* [41][B][S][C] or NULL */
  const int32_t* slot_table;       /* [T] collection slot for timestep t, -1 = none */
  /* jax.random streams drawn inside the fused step (utils/ebm_utils.py:342-345,360-362): per-iteration key tables
   * [iterations][2] (uint32 pairs, row tf_t0 - t), NULL = Philox / explicit z.  The state is rows
   * [sample_offset, sample_offset + B) of a global (N, S, C) array of tf_n_total elements. */
  const uint32_t* tf_noise_keys;
  const uint32_t* tf_infill_keys;
  int64_t tf_n_total;
  int32_t tf_t0;
  /* Philox key in DEVICE memory ([2] words: seed_lo, seed_hi), NULL = the seed_lo / seed_hi fields above.  A captured
   * step that reads its key here (and its timestep from t_ptr) can be replayed for a later sampling run with another rng. */
  const uint32_t* key_ptr;
} smd_sample_io;
int smd_engine_prepare_sampler(smd_engine* e, void* stream);
int smd_engine_init_state(smd_engine* e, float* x, uint32_t seed_lo, uint32_t seed_hi, uint32_t sample_offset,
                          void* stream);
/* explicit initial state (parity mode / infill / interpolation): refreshes the bf16 network input */
int smd_engine_load_state(smd_engine* e, const float* x, void* stream);
int smd_engine_sample_step(smd_engine* e, const smd_sample_io* io, void* stream);
/* The two halves of that iteration, for a software pipeline over independent chains (the batch of diffusion_dynamics is a set of
 * independent samples): part 1 = the network's stem (models/ncsn.py:152-171: in_proj .. up projection; it reads the state left by
 * the previous reverse update and does not depend on t), part 2 = the output stage (models/ncsn.py:173-178) + the fused reverse
 * update (utils/ebm_utils.py:327-394), which advances *t_ptr.  part 1 then part 2 on one stream == smd_engine_sample_step.  The
 * host runs chain A as (2, 1) and chain B as (1, 2) on two streams inside one captured graph, so one chain's latency-bound
 * encoder kernels always sit beside the other's 2048-wide GEMMs (two free-running chains drift INTO phase: DESIGN.md section 5). */
int smd_engine_sample_step_part(smd_engine* e, const smd_sample_io* io, int part, void* stream);

/* One Langevin update of annealed_langevin_dynamics / consistent_langevin_dynamics (utils/ebm_utils.py:131-164, 231-253):
 *   next = x + alpha * grad + noise_coef * z;  with infill: next = next (1 - mask) + (infill_samples + infill_sigma * zi) mask.
 * grad = model(state, sigma) comes from smd_engine_forward.  z / zi: explicit arrays, else jax.random.normal(step_rng) /
 * (infill_rng) from the two threefry keys (use_threefry; the state is rows [sample_offset, +B) of a global array of
 * tf_n_total elements), else Philox keyed by (seed, global sample index, step).  metrics_partial [B][3] receives the
 * per-sample sums behind grad_norm / step_norm / noise_norm (:157-161); collect_out a copy of the new state. */
typedef struct smd_langevin_io {
  float* x;
  const float* grad;
  float alpha, noise_coef;
  const float* z_in;
  uint32_t seed_lo, seed_hi, step, sample_offset;
  int32_t use_threefry;
  uint32_t tf_noise_key[2], tf_infill_key[2];
  int64_t tf_n_total;
  const float* infill_samples;
  const float* infill_masks;
  const float* infill_z_in;
  float infill_sigma;
  float* metrics_partial;
  float* collect_out;
  /* Table mode (graph-captured loops; ABI version 3): with k_ptr != NULL the update takes its step-dependent arguments from
   * device memory, indexed by the device-side update counter k = *k_ptr, so that ONE captured (forward + update) pair can be
   * replayed for a whole annealed / consistent run:  alpha, noise_coef, infill_sigma = step_table[k][0..2]; the Philox counter
   * word and (use_threefry) the two keys key_table[k][0..1 | 2..3] are those of update k; metrics_partial is the base of an
   * [n_steps][B][3] array (row k is written); the new state is copied to collection + slot_table[k] * B*S*C when that slot
   * is >= 0; sigma_out[b] receives step_table[k][3], the noise level of the NEXT forward pass (level_out its table row); the launch's last workgroup
   * stores k + 1 to k_ptr (arrive: one zeroed uint32).  k outside [0, n_steps) makes the launch a no-op. */
  const float* step_table;
  const int32_t* slot_table;
  const uint32_t* key_table;
  int32_t* k_ptr;
  uint32_t* arrive;
  float* collection;
  float* sigma_out;
  int32_t n_steps;
  int32_t* level_out;        /* one int32: receives min((k + 1) / steps_per_level, n_levels - 1), the FiLM-table row of the next forward */
  int32_t steps_per_level, n_levels;
} smd_langevin_io;
int smd_langevin_step(const smd_langevin_io* io, int B, int S, int C, void* stream);
/* *t_ptr = t on `stream`: restarts a reverse walk (the device-side timestep of smd_engine_sample_step) without a framework fill */
int smd_set_timestep(int32_t* t_ptr, int32_t t, void* stream);


/* ---- e4m3 (OCP fp8) path: BASELINE config 5.  Operands are e4m3 bytes with one power-of-two (E8M0) scale per row,
 * contracted by v_mfma_scale_f32_32x32x64_f8f6f4; engine option "fp8" = 1 (before bind_workspace) routes the
 * DenseResBlock forward GEMMs (models/shared.py:65,69) through it. ------------------------------------------------ */
/* rows of a bf16 matrix [rows][ld] (K used) -> out8 [rows][K] e4m3(v * 2^-e), scale[rows] = E8M0 byte e + 127 */
int smd_quantize_rows_e4m3(const smd_bf16* in, int ld, int rows, int K, uint8_t* out8, uint32_t* scale, void* stream);
/* C[M,N] = (2^sa[m] A8[m,:]) . (2^sb[n] Bt8[n,:]) + bias (+ fp32 residual) -> fp32 and/or bf16; M, N, K % 256 == 0 */
int smd_gemm_e4m3_nt(const uint8_t* A8, int lda, const uint32_t* scale_a, const uint8_t* Bt8, int ldb,
                     const uint32_t* scale_b, int M, int N, int K, const float* bias, const float* residual, int ld_res,
                     float* out_f32, int ld_out, smd_bf16* out_bf16, int ld_outb, void* stream);
/* smd_layernorm_fwd (D = 1024 / 2048, FiLM + swish or neither) writing the e4m3 copy + row scales (and bf16 if non-NULL) */
int smd_layernorm_fwd_e4m3(const float* x, int rows, int D, const float* gamma, const float* beta, const float* film_scale,
                           const float* film_shift, int ld_film, int rows_per_sample, int swish, uint8_t* out8,
                           uint32_t* out_scale, smd_bf16* out_bf16 /*nullable*/, void* stream);

/* ---- single kernels (unit-testable ops) ------------------------------------------------------- */
enum { SMD_EPI_NONE = 0, SMD_EPI_GELU = 1, SMD_EPI_SWISH = 2 };
/* C[M,N] = act(A[M,K] Bt[N,K]^T + bias) (+ residual); nn.Dense, models/ncsn.py:155 etc. */
int smd_gemm_bf16_nt(const smd_bf16* A, int lda, const smd_bf16* Bt, int ldb, int M, int N, int K,
                     const float* bias, int act, const float* residual, int ld_res, float* out_f32, int ld_out,
                     smd_bf16* out_bf16, int ld_outb, void* stream);
/* Fused encoder MLP half-layer, models/ncsn.py:163-168: h_out = h_in + Dense(gelu(Dense(LN(h_in)))) for the
 * 128-wide residual stream, rows % 32 == 0.  W1t [hidden][128] and W2t [128][hidden] are bf16 with the contraction
 * index contiguous; save_* (optional) receive LN output, pre-GELU and post-GELU activations for the backward. */
int smd_mlp_block_fwd(const float* h_in, float* h_out, int rows, const float* gamma, const float* beta,
                      const smd_bf16* W1t, const float* b1, const smd_bf16* W2t, const float* b2, int hidden,
                      smd_bf16* save_a2, smd_bf16* save_z1, smd_bf16* save_u, void* stream);
/* MLP half-layer (models/ncsn.py:163-168) with the hidden dimension split over 4 workgroups per group of 4 / 2 / 1
 * samples: a quarter of the weight stream per CU.  a2 = ln2(h_mid) in bf16 (smd_attn_block_fwd_ex emits it);
 * output: four fp32 partial tiles part[k] = part + k*rows*128 with h_out = (p0 + p1) + (p2 + p3) -- quarter 0 carries
 * b2 + the residual h_res -- to be summed by the consumer (smd_attn_block_fwd_ex / smd_ln128_parts).  hidden % 512 == 0. */
int smd_mlp_block_fwd_hs(const smd_bf16* a2, const float* h_res, int rows, const smd_bf16* W1t, const float* b1,
                         const smd_bf16* W2t, const float* b2, int hidden, float* part, void* stream);
/* backward of that half-layer between ln2 and the residual add, hidden activations recomputed from a2:
 * u = gelu(a2 W1 + b1) and dz = (dh W2^T) gelu'(.) are written ([rows][hidden] bf16, the operands of the fc2 / fc1 weight
 * gradients), da2 = dz W1^T leaves as four fp32 partial tiles (summed by smd_ln128_bwd_parts).  W1t [hidden][128]: fc1
 * forward pack; W2 [hidden][128], W1 [128][hidden]: dgrad packs of fc2 / fc1.  rows % 128 == 0, hidden % 512 == 0. */
int smd_mlp_block_bwd_hs(const smd_bf16* a2, const smd_bf16* dh, int rows, const smd_bf16* W1t, const smd_bf16* W2,
                         const smd_bf16* W1, const float* b1, int hidden, smd_bf16* u, smd_bf16* dz, float* part, void* stream);
/* LayerNorm (D = 128) backward with dout = (p0 + p1) + (p2 + p3) (fp32 partial tiles): dx = LN-backward + dres (nullable)
 * -> dx_f32 (nullable, may alias dres) / dx_bf16 (nullable); partial [rows/32][2][128] = per-group dgamma / dbeta sums */
int smd_ln128_bwd_parts(const float* x, const float* parts, int64_t part_stride, int rows, const float* gamma,
                        const float* dres, float* dx_f32, smd_bf16* dx_bf16, float* partial, void* stream);
/* x = (p0 + p1) + (p2 + p3) per 128-wide row (parts + k*part_stride); x_out (nullable) <- x, ln_out (nullable) <-
 * LayerNorm(x) bf16: the encoder's final norm (models/ncsn.py:170) on a hidden-split MLP output */
int smd_ln128_parts(const float* parts, int64_t part_stride, int rows, const float* gamma, const float* beta, float* x_out,
                    smd_bf16* ln_out, void* stream);
/* smd_attn_block_fwd with (a) the input given as four partial tiles (h_parts != NULL: h_in unused; h_comb nullable
 * receives their sum) and (b) the LayerNorm of the OUTPUT rows (ln2 of the same layer) written as bf16 a2_out (nullable) */
int smd_attn_block_fwd_ex(const float* h_in, const float* h_parts, int64_t part_stride, float* h_comb, float* h_out, int rows,
                          const float* gamma, const float* beta, const smd_bf16* Wqkv_t, const float* b_qkv,
                          const smd_bf16* Wo_t, const float* b_o, int num_heads, const float* gamma2, const float* beta2,
                          smd_bf16* a2_out, smd_bf16* save_a1, smd_bf16* save_qkv, smd_bf16* save_o, void* stream);
/* Fused encoder attention half-layer, models/ncsn.py:159-162: h_out = h_in + out(softmax(q k^T / sqrt(d)) v) with
 * q,k,v = Dense(LN(h_in)); 32 tokens per sample, 128-wide stream, num_heads in {4, 8, 16}.  Wqkv_t [384][128]
 * (q | k | v rows), Wo_t [128][128], bf16, contraction contiguous; save_qkv receives the unscaled q. */
int smd_attn_block_fwd(const float* h_in, float* h_out, int rows, const float* gamma, const float* beta,
                       const smd_bf16* Wqkv_t, const float* b_qkv, const smd_bf16* Wo_t, const float* b_o, int num_heads,
                       smd_bf16* save_a1, smd_bf16* save_qkv, smd_bf16* save_o, void* stream);
/* Backward of that half-layer between its LayerNorms: dO = dh_mid Wo^T, attention backward (softmax recomputed
 * from the saved qkv), da1 = dqkv Wqkv^T.  Wo [128 in][128 out], Wqkv [128 in][384 out]: bf16, out index contiguous. */
int smd_attn_block_bwd(const smd_bf16* dh_mid, const smd_bf16* qkv, const smd_bf16* Wo, const smd_bf16* Wqkv,
                       smd_bf16* dqkv, smd_bf16* da1, int rows, int num_heads, void* stream);
/* The backward of models/ncsn.py:159-164 from the MLP's input gradient to the layer's input gradient in ONE launch: LayerNorm-2
 * backward on da2 given as four partial tiles (part_stride floats apart, summed (p0 + p1) + (p2 + p3)) + the residual gradient dh,
 * the attention half-layer backward (smd_attn_block_bwd), LayerNorm-1 backward + residual.  dh [rows][128] fp32 is read and then
 * overwritten IN PLACE with the gradient that leaves the layer; dh_mid_out / dh_out: its bf16 copies between the half-layers (the
 * operand of out_proj's weight gradient) and at the exit; partial1 / partial2 [rows/32][2][128]: dgamma | dbeta sums per sample
 * of LayerNorm 1 / 2 (reduce with smd_ln_bwd_reduce); da1 optional. */
int smd_attn_block_bwd_ln(const smd_bf16* qkv, const smd_bf16* Wo, const smd_bf16* Wqkv, smd_bf16* dqkv, smd_bf16* da1,
                          const float* h_mid, const float* da2_parts, int64_t part_stride, const float* gamma2, float* dh,
                          smd_bf16* dh_mid_out, float* partial2, const float* h, const float* gamma1, smd_bf16* dh_out,
                          float* partial1, int rows, int num_heads, void* stream);
/* dW[Kd,N] = X[M,Kd]^T dY[M,N] and db[N] = colsum(dY) (weight + bias gradient of nn.Dense).
 * tr_path 1: zero_page = 128 zeroed bf16, slab = smd_gemm_tn_slab_elems() floats (split-K partials);
 * tr_path 0: scratch = (Kd+N)*roundup(M,64) bf16 for explicit transposes (+ slab for the bias). */
int smd_gemm_bf16_tn(const smd_bf16* X, int ldx, const smd_bf16* dY, int ldy, int M, int Kd, int N, float* out,
                     int ldo, float* bias_out, const smd_bf16* zero_page, float* slab, int64_t slab_elems,
                     smd_bf16* scratch, int64_t scratch_elems, int tr_path, void* stream);
int64_t smd_gemm_tn_slab_elems(void);
/* flax.nn.LayerNorm (+ FiLM + swish), models/shared.py:62-68 */
int smd_layernorm_fwd(const float* x, int rows, int D, const float* gamma, const float* beta,
                      const float* film_scale, const float* film_shift, int ld_film, int rows_per_sample,
                      int swish, smd_bf16* out, void* stream);
/* the engine's form: the input row as fp32 (x) or bf16 (x_bf16: the 2048-wide trunk), exactly one of them non-NULL */
int smd_layernorm_fwd_ex(const float* x, const smd_bf16* x_bf16, int rows, int D, const float* gamma, const float* beta,
                         const float* film_scale, const float* film_shift, int ld_film, int rows_per_sample, int swish,
                         smd_bf16* out, void* stream);
int smd_layernorm_bwd(const float* x, int rows, int D, const float* gamma, const float* beta,
                      const float* film_scale, const float* film_shift, int ld_film, int rows_per_sample,
                      int swish, const smd_bf16* dout, float* dx, float* dgamma, float* dbeta, float* dscale,
                      float* dshift, float* partial, int64_t partial_elems, void* stream);
/* flax.nn.SelfAttention core, models/ncsn.py:161 */
/* (every LayerNorm backward WRITES dgamma / dbeta -- the sum over its row groups -- it does not accumulate into them)
 * the engine's form of the LayerNorm backward: optional fp32 residual gradient `dres` added to dx (may alias dx: the
 * in-place residual-gradient stream), optional bf16 copy of dx, bf16 or fp32 input */
int smd_layernorm_bwd_ex(const float* x, const smd_bf16* x_bf16, int rows, int D, const float* gamma, const float* beta,
                         const smd_bf16* dout, const float* dres, float* dx, smd_bf16* dx_bf16, float* dgamma, float* dbeta,
                         float* partial, int64_t partial_elems, void* stream);
/* ... and the ResBlock form (models/shared.py:62-68): FiLM + swish, bf16 or fp32 input, the residual gradient in fp32
 * (`dres`) or bf16 (`dres_bf16`, D in {1024, 2048}), fp32 and/or bf16 dx; dfilm_accumulate: dscale/dshift += */
int smd_layernorm_bwd_film(const float* x, const smd_bf16* x_bf16, int rows, int D, const float* gamma, const float* beta,
                           const float* film_scale, const float* film_shift, int ld_film, int rows_per_sample, int swish,
                           const smd_bf16* dout, const float* dres, const smd_bf16* dres_bf16, float* dx,
                           smd_bf16* dx_bf16, float* dgamma, float* dbeta, float* dscale, float* dshift,
                           int dfilm_accumulate, float* partial, int64_t partial_elems, void* stream);
int smd_attention_fwd(const smd_bf16* qkv, smd_bf16* out, int B, int S, int E, int H, void* stream);
int smd_attention_bwd(const smd_bf16* qkv, const smd_bf16* dout, smd_bf16* dqkv, int B, int S, int E, int H,
                      void* stream);
/* NoiseEncoding.apply, models/ncsn.py:28-41 */
int smd_noise_embed(const float* noise_level, int n, int channels, smd_bf16* out, int ld_out, void* stream);
/* The loss-side and optimiser kernels without an engine handle (the engine calls the same launchers).
 * smd_q_sample: utils/losses.py:271-296.  x0 [B][S][C] fp32 -> xt_bf16 [B*S][Cp] (the columns >= C are not written),
 *   eps_out [B][S][C], noise_level_out [B] = sqrt(used_alpha).  labels NULL -> Philox labels in
 *   [label_min, label_min + T) keyed by (seed, b + sample_offset, *step_ptr); used_alphas NULL -> alphas_prod_ext[label - 1]
 *   (label 0: a uniform draw in [alphas_prod_ext[T], 1)); eps_in NULL -> Philox normals.  S*C need not be a multiple of 4.
 * smd_mse_fwd_bwd: :304-305 and d(mean loss)/d pred: loss_per_sample [B], dpred_bf16 [B*S][Cp] = 2 (pred - eps) *
 *   inv_global_count.
 * smd_adam_clip_ema: train_ncsn.py:284-287,340-342,364-365 on flat fp32 buffers of n elements: gradient norm -> clip ->
 *   Adam with the stepped LR evaluated from *step_ptr (incremented by the kernel) -> EMA (ema may be NULL).
 *   norm_partial: >= 1024 floats of scratch; metrics_out [4] = norm before clip, after clip, lr, step. */
int smd_q_sample(const float* x0, int B, int S, int C, int Cp, int T, const float* alphas_prod_ext, const int32_t* labels,
                 int label_min, const float* used_alphas, const float* eps_in, uint32_t seed_lo, uint32_t seed_hi,
                 const uint32_t* step_ptr, uint32_t sample_offset, smd_bf16* xt_bf16, float* eps_out,
                 float* noise_level_out, void* stream);
int smd_mse_fwd_bwd(const float* pred, const float* eps, int B, int S, int C, int Cp, float inv_global_count,
                    float* loss_per_sample, smd_bf16* dpred_bf16, void* stream);
int smd_adam_clip_ema(float* params, const float* grads, float* m, float* v, float* ema, int64_t n,
                      const smd_train_hyper* h, uint32_t* step_ptr, float* norm_partial, float* metrics_out, void* stream);
/* Philox4x32-10 normals: out[b][e], counter (e/4, b + sample_offset, stream_id, 0) */
int smd_rng_normal(float* out, int B, int per_sample, uint32_t seed_lo, uint32_t seed_hi, uint32_t stream_id,
                   uint32_t sample_offset, void* stream);
/* jax.random-compatible draws (jax 0.2.8 threefry2x32 conventions; replaces jax.random.{randint,uniform,normal} at
 * utils/losses.py:272-294, utils/ebm_utils.py:343-345,361-362, train_ncsn.py:539-540).  Each call fills the window
 * [offset, offset+count) of a logical array of n_total <= 2^32 elements with exactly the values
 * jax.random.X(key=(k0,k1), shape=(n_total,)) holds there, so sharded callers reproduce the single-process stream.
 * smd_threefry_normal: key_table != NULL takes the key from key_table[idx_add + idx_mul * *idx_ptr] on device
 * (uint32 pairs), which keeps a captured reverse-sampling step replayable. */
int smd_threefry_bits(uint32_t* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1, void* stream);
int smd_threefry_uniform(float* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1,
                         float minval, float maxval, void* stream);
int smd_threefry_normal(float* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1,
                        const uint32_t* key_table, const int32_t* idx_ptr, int idx_mul, int idx_add, void* stream);
int smd_threefry_randint(int32_t* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1,
                         int32_t minval, int32_t maxval, void* stream);
int smd_cast_pad_bf16(const float* in, int rows, int cols, smd_bf16* out, int ld_out, void* stream);
/* one reverse step on explicit eps_hat (the elementwise part of utils/ebm_utils.py:327-394); coef is the [T][8]
 * table, *t_ptr outside [0, T) makes the call a no-op */
int smd_ddpm_reverse_step(float* x, const float* eps_hat, int B, int S, int C, const float* coef, int T,
                          const int32_t* t_ptr, const float* z_in, uint32_t seed_lo, uint32_t seed_hi,
                          uint32_t sample_offset, float* metrics_partial, float* collection,
                          const int32_t* slot_table, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMD_HIP_H_ */
