// Host-side orchestration of the eps-net forward / backward / optimiser / reverse-sampler launches.
// No device allocation happens here: every buffer is borrowed from the caller (torch's caching
// allocator on the Python side) and bound through the C-ABI (include/smd_hip.h).
#pragma once
#include <string>
#include <vector>

#include "smd_kernels.h"

struct SmdModelDesc {
  int arch = 0;              // 0 TransformerDDPM (models/ncsn.py:138-179), 1 DenseDDPM (:122-135)
  int data_channels = 512;   // C
  int seq_len = 32;          // S (1 for DenseDDPM)
  int num_layers = 6;
  int num_heads = 8;
  int num_mlp_layers = 2;
  int mlp_dims = 2048;
  int embed_channels = 128;  // models/ncsn.py:151
  int film_channels = 128;   // models/ncsn.py:174
  int num_timesteps = 1000;
};

struct TensorInfo {
  std::string name;
  int64_t offset;            // element offset in the flat fp32 parameter buffer
  int rows, cols;            // cols == 0 for 1-D tensors
};

struct DenseP {              // one nn.Dense: kernel (K,N) + bias (N)
  int K = 0, N = 0, Kp = 0, Np = 0;
  int64_t w_off = 0, b_off = 0;        // flat fp32 offsets
  int64_t W_off = 0, Wt_off = 0;       // bf16 pack offsets: W [K][Np] (dgrad operand), Wt [N][Kp] (forward)
};
struct LnP {
  int D = 0;
  int64_t g_off = 0, b_off = 0;
};
struct EncLayerP { LnP ln1, ln2; DenseP qkv, out, fc1, fc2; };
struct FilmResP { DenseP f1, f2, ss, r1, r2; LnP ln1, ln2; };

struct TrainHyper {
  float lr0 = 1e-3f, lr_gamma = 0.98f;
  int lr_interval = 10000;
  float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f, grad_clip = 1.0f, mu = 0.999f;
  float grad_scale = 1.0f;
};

struct SampleStepIO {
  float* x = nullptr;                 // [B][S][C] fp32 state (in place)
  int* t_ptr = nullptr;               // device timestep, decremented by the step
  const float* z_in = nullptr;        // explicit noise or null (Philox)
  uint32_t seed_lo = 0, seed_hi = 0, sample_offset = 0;
  const float* infill_samples = nullptr;
  const float* infill_masks = nullptr;
  const float* infill_z_in = nullptr;
  float* metrics_partial = nullptr;   // [T][B][3]
  float* collection = nullptr;        // [41][B][S][C]
  const int* slot_table = nullptr;    // [T]
  const uint32_t* tf_noise_keys = nullptr;   // jax.random key tables (see ReverseStepArgs)
  const uint32_t* tf_infill_keys = nullptr;
  int64_t tf_n_total = 0;
  int tf_t0 = 0;
  const uint32_t* key_ptr = nullptr;         // device-resident Philox key (overrides seed_lo / seed_hi)
};

class SmdEngine {
 public:
  explicit SmdEngine(const SmdModelDesc& d);
  ~SmdEngine();
  SmdEngine(const SmdEngine&) = delete;
  SmdEngine& operator=(const SmdEngine&) = delete;
  const SmdModelDesc& desc() const { return d_; }
  const std::vector<TensorInfo>& tensors() const { return tensors_; }
  int64_t param_count() const { return n_params_; }
  int64_t wpack_elems() const { return n_wpack_; }
  int64_t head_param_offset() const { return head_off_; }   // params >= this belong to the output stage
  int64_t workspace_bytes(int batch, int training) const;
  // K tables [T][2M] fp32 + bf16 scratch for the T-row FiLM generator GEMMs (emb, f1, p)
  int64_t film_table_floats() const {
    return (int64_t)nblocks() * d_.num_timesteps * 2 * d_.mlp_dims + (int64_t)d_.num_timesteps * 9 * d_.film_channels / 2 + 64;
  }
  int padded_channels() const { return Cp_; }

  int bind_params(float* params, bf16_t* wpack);
  int bind_train(float* grads, float* m, float* v, float* ema, uint32_t* step_ptr, float* metrics);
  int bind_workspace(void* ws, int64_t bytes, int batch, int training, hipStream_t st);
  // coef [T][8] reverse-step constants, sqrt_ap [T], alphas_prod_ext [T+1] = [1, cumprod(1-beta)],
  // film_tables: film_table_floats() floats (sampling only, may be null for training)
  int bind_schedule(const float* coef, const float* sqrt_ap, const float* alphas_prod_ext, float* film_tables);

  int refresh_weights(hipStream_t st);                       // fp32 master -> bf16 pack
  // model(x, cond): x fp32 [B][S][C], noise_level [B] -> eps_hat fp32 [B][S][C]
  int forward(const float* x, const float* noise_level, float* eps_out, hipStream_t st);
  // the same with a BATCH-UNIFORM noise level given as a device-side row index into the FiLM tables of prepare_sampler()
  // (row r = noise level sqrt_ap[r] of the bound schedule): no FiLM generator launches, replayable from a graph
  int forward_level(const float* x, const int* level_ptr, float* eps_out, hipStream_t st);
  // one diffusion_loss forward + backward on the bound batch; stage: 0 = all, 1 = loss+forward+output
  // stage backward (grads of params >= head_param_offset complete), 2 = stem backward
  int loss_backward(const float* x0, const int* labels, const float* eps_in, uint32_t seed_lo, uint32_t seed_hi,
                    uint32_t sample_offset, float inv_global_count, int stage, hipStream_t st);
  int optimizer_step(const TrainHyper& h, hipStream_t st);
  // Optimiser placement (optimizer_step is ONE sweep -- clip + Adam + EMA + bf16 re-cast -- in every mode):
  //   bit 0  the update of the OUTPUT-STAGE slice (parameters >= head_param_offset: 86 % of the bytes, not needed before the
  //          `up` projection of the next forward pass) runs on the engine's side stream behind the stem slice's norm; the next
  //          run_network() of THIS handle waits for it right before the first output-stage kernel, every other reader of the
  //          parameters / Adam state / operand pack (another handle, the host, a checkpoint) must call join_update() first;
  //   bit 1  loss_backward(stage 0) reduces the output-stage slice's gradient-norm partials on the side stream as soon as that
  //          slice is final (under the encoder backward) instead of in optimizer_step -- valid only when nothing changes the
  //          gradient between the two calls (no all-reduce: the data-parallel path leaves this bit off).
  // 0 (default): everything on the caller's stream, complete in stream order when optimizer_step returns.
  int opt_overlap = 0;
  int opt_side_blocks = 0;              // > 0: workgroups of the deferred output-stage sweep (grid-stride: throttles its HBM rate); 0 = one per tile
  void set_opt_fused(bool on) { opt_fused_user_ = on; }
  // Data-parallel gradient buckets of the STEM slice in backward order (option "dp_layer_events" = 1): loss_backward(stage 2 / 0)
  // records one event per encoder layer l = L-1 .. 1 as soon as that layer's parameter gradients are final (its grouped weight
  // gradients on the side stream + its two LayerNorm reductions), so the caller can start that layer's collective while the
  // layers below are still in their backward pass; the last bucket (in_proj + layer 0) is final when the call returns.
  int dp_layer_events = 0;
  int num_grad_buckets() const { return d_.arch == 0 ? d_.num_layers : 1; }
  int grad_bucket(int b, int64_t* off, int64_t* len) const;
  int wait_grad_bucket(int b, hipStream_t s);      // `s` waits for bucket b's event (no-op for the last bucket / events off)
  int join_update(hipStream_t st);      // make `st` wait for a deferred output-stage update (no-op when none is pending)
  int prepare_sampler(hipStream_t st);                       // FiLM tables for every timestep
  int forward_train(const float* x, const float* noise_level, float* eps_out, hipStream_t st);   // model(x, cond), activations saved
  int backward_from(const float* dpred, int stage, hipStream_t st);                                 // backward of that pass from d/d eps_hat
  int sample_step(const SampleStepIO& io, hipStream_t st, int part = 0);   // eps-net forward + fused reverse step
  int init_state(float* x, uint32_t seed_lo, uint32_t seed_hi, uint32_t sample_offset, hipStream_t st);
  int load_state(const float* x, hipStream_t st);            // explicit state -> bf16 network input
  // device pointers of a few internals (tests / metrics)
  const float* loss_per_sample() const { return W.loss; }
  const float* pred() const { return W.pred; }
  const float* noise_levels() const { return W.s; }
  int tr_path = 1;                                            // wgrad: 1 LDS transpose-read kernel, 0 fallback
  // Weight-gradient GEMMs on a low-priority side stream (owned by the engine, created on first use): nothing
  // on the backward chain consumes dW before the optimiser, so the HBM-heavy wgrad + slab reduce overlap the
  // dgrad / LayerNorm chain.  Every gradient buffer a side wgrad reads has its own slot (no reuse inside a
  // step); the side stream is joined at the end of loss_backward().  0 = single stream.
  int set_side_stream(int enable);
  void mark_w8_dirty() { w8_dirty_ = true; }
  // diffusion_loss(continuous_noise=False), utils/losses.py:272-286: labels in [0, T) and, for label 0, a real uniform
  // used_alpha in [alphas_prod[T], 1).  used_alphas (device, [B], may be null) overrides the label -> alpha lookup of the
  // NEXT loss_backward calls until it is reset to null.
  int label_min = 1;
  // 0: diffusion_loss (utils/losses.py:250-308).  1: denoising_score_matching_loss (:129-179) on the same network:
  // used_alphas then holds the per-sample used_sigmas (required), the network is conditioned on sigma, and the loss /
  // gradient are 0.5 sum (sigma score + eps)^2 and its batch mean's derivative.
  int loss_kind = 0;
  void set_used_alphas(const float* a) { used_alphas_ = a; }
  // Debugging aid (tools/det_first_diff.py): when a buffer is set, backward_stem copies the state of the SHARED, per-layer
  // overwritten gradient buffers behind every encoder-layer kernel into it (per layer, in execution order: da2 partial
  // tiles after mlp_hs_bwd, dh after the ln2 backward, dA_E after attn_block_bwd, dh after the ln1 backward; plus the layer's
  // incoming dh and its saved h_mid), so that two
  // identical steps can be compared kernel by kernel.  bytes >= debug_snapshot_bytes(); null switches it off.
  // layout: the encoder's per-layer segments (34 * rows * E bytes a layer, last layer first), then 2 K + 1 copies of the output
  // stage's shared dX buffer (bf16 [rows][M]) in execution order: behind out_proj's dgrad, then for k = K-1 .. 0 behind the
  // dgrad of block k's second and first Dense
  int64_t debug_stem_snapshot_bytes() const { return (int64_t)d_.num_layers * 34 * rows() * d_.embed_channels; }
  int64_t debug_snapshot_bytes() const { return debug_stem_snapshot_bytes() + (int64_t)(2 * nblocks() + 1) * rows() * d_.mlp_dims * 2; }
  int set_debug_snapshots(void* buf, int64_t bytes);
  // Debugging aid (tests/test_gpu_engine.py layer-by-layer check): device pointer, shape and element type (0 fp32, 1 bf16) of
  // an activation the TRAINING forward pass saved in the bound workspace -- "x_bf16", "h"[l], "h_mid"[l], "a1"[l], "qkv"[l],
  // "o"[l], "a2"[l], "h_last", "af", "y"[k], "ya1"[k], "o1"[k], "ya2"[k], "emb", "f1"[k], "p"[k], "ss"[k], "ao", "pred", "s"; the
  // operands of the weight-gradient GEMMs of the last backward pass: "dpred", "dyb"[k], "do1"[k], "dss_bf16"[k], "dp"[k], "df1"[k],
  // "dhb"[i], "dqkv"[l], "dz1"[l], "u"[l].
  int debug_tensor(const char* name, int index, const void** ptr, int64_t* rows, int64_t* cols, int* dtype) const;
  int film_side = 1;                                          // FiLM-generator wgrads deferred to the side stream too
  // The 2048-wide trunk y (models/ncsn.py:171-176) lives in bf16 instead of fp32: the residual operand and the output of
  // every fc2 GEMM and the LayerNorm inputs (forward and backward) shrink by half (-64 MB per DenseResBlock; the fp32
  // round trip is what makes the residual-form GEMM 83 us instead of 62).  Measured: eps_hat 5.7e-3 -> 6.3e-3 against the
  // oracle, gradient parity unchanged (5.7e-3); sample step +9.6 %, train step +2.2 %.
  int trunk_bf16 = 2;      // 0: fp32 trunk everywhere; 1: bf16 in inference workspaces only; 2: training too (default)
  bool trunk_bf16_on() const { return d_.mlp_dims % 8 == 0 && (training_ ? trunk_bf16 == 2 : trunk_bf16 >= 1); }
  int sample_split = 0;        // split sampler passes (sample_step part 1 / 2): half-blocks of the output stage that belong to part 1
  int nt256_min_tiles = 0; // > 0: Dense layers take the 256x256 GEMM from this many tiles up (concurrent sampling chains: 128)
  int grad_memset = 2;     // 0 never, 1 always, 2 (default): only with the tr_path = 0 fallback wgrads
  int tail_on_main = 1;    // the last grouped wgrad launch of a step runs on the caller's stream (which would idle) while
                           // the side stream drains its backlog
  int film_side_fwd = 1;   // training: FiLM generators (forward) and their backward chain run on the side stream
  int pair_wgrad = 1;                                         // the 2048x2048 wgrads of the DenseResBlocks in grouped launches
  int wgrad256_group = 4;  // problems per 256x256-kernel wgrad launch: 4 (default: 256 tiles, no split over m, no slab reduce),
                           // 2 = one launch per DenseResBlock with two m-splits (round 2)
  int group_wgrad = 2;                                        // 128-wide weight gradients in grouped launches: 2 = one per encoder
                                                              // layer as soon as its backward is enqueued (+4.6 % train), 1 = all at
                                                              // the end of the backward (+3 %), 0 = one launch + reduce each
  int fused_attn_bwd = 2;                                     // attn_block_bwd kernel: 2 = with the LayerNorm backwards either side of it in the
                                                              // launch (hidden-split dataflow), 1 = attention only, 0 = three separate launches
  int resgrad_bf16 = 1;   // ResBlock residual-gradient chain kept in bf16 (the GEMM operand copy) instead of fp32 + bf16
  int fused_encoder = 1;                                      // encoder_fused.hip half-layer kernels (0: separate launches)
  int side_wgrad = 0;
  // e4m3 (OCP fp8) operands with per-row E8M0 scales for the DenseResBlock GEMMs of the FORWARD pass (77 % of its flops),
  // v_mfma_scale_f32_32x32x64_f8f6f4; everything else, and the whole backward pass, stays bf16.  Set before bind_workspace.
  int fp8 = 0;
  // fp8 mode, training: the four DenseResBlock DGRAD GEMMs (dX = dY W^T, 8192 x 2048 x 2048 each) also run on e4m3
  // operands -- dY quantised per row (one E8M0 scale per token row) right before the GEMM, the dgrad layout of the weights
  // quantised per row next to the forward layout; the weight gradients and everything 128-wide stay bf16.
  int fp8_dgrad = 1;
  int mlp_hs = 1;              // hidden-split fused MLP half-layers (forward; backward with recompute); 0: the older paths

 private:
  int nblocks() const { return d_.arch == 0 ? d_.num_mlp_layers : d_.num_layers; }
  int rows() const { return batch_ * d_.seq_len; }
  void build_layout();
  int run_network(const int* t_ptr, hipStream_t st, int part = 0);          // x_bf16 (+ s or table row) -> pred
  int backward_head(hipStream_t st);
  int backward_stem(hipStream_t st);
  int dense_fwd(const DenseP& p, const bf16_t* A, int lda, int M, GemmEpilogue ep, hipStream_t st);
  int dense_bwd(const DenseP& p, const bf16_t* X, int ldx, const bf16_t* dY, int ldy, int M, bf16_t* dX,
                int ld_dx, const bf16_t* aux, int ld_aux, int aux_mode, hipStream_t st, bool allow_side = false);
  int wgrad(const DenseP& p, const bf16_t* X, int ldx, const bf16_t* dY, int ldy, int M, bool allow_side, hipStream_t st);
  int join_side(hipStream_t st);
  int finish_backward(hipStream_t st, bool stem_ran);
  int ln_bwd(LnBwdArgs& b, hipStream_t st);
  int flush_ln_reduce(hipStream_t st);
  int flush_grouped_wgrads(hipStream_t st, bool on_caller_stream = false);
  std::vector<TnLaunch> deferred_wgrads_;
  std::vector<TnLaunch> pending256_;          // 256x256-kernel wgrad problems waiting for their group (wgrad256_group)
  int flush_pending256(hipStream_t st);
  std::vector<LnReduceEntry> ln_pending_;
  size_t ln_slot_off_ = 0;
  float* P(int64_t off) const { return params_ + off; }
  float* G(int64_t off) const { return grads_ + off; }

  SmdModelDesc d_;
  std::vector<TensorInfo> tensors_;
  int64_t n_params_ = 0, n_wpack_ = 0, head_off_ = 0;
  int Cp_ = 0;
  // parameter handles
  DenseP in_proj_, up_, out_proj_;
  LnP ln_f_, ln_o_;
  std::vector<EncLayerP> enc_;
  std::vector<FilmResP> blk_;
  std::vector<DenseP*> all_dense_;

  // bound buffers
  float* params_ = nullptr;
  bf16_t* wpack_ = nullptr;
  float *grads_ = nullptr, *m_ = nullptr, *v_ = nullptr, *ema_ = nullptr, *metrics_ = nullptr;
  uint32_t* step_ptr_ = nullptr;
  const float* used_alphas_ = nullptr;
  char* dbg_snap_ = nullptr;
  const float* coef_ = nullptr;
  const float* sqrt_ap_ = nullptr;
  const float* alphas_prod_ext_ = nullptr;
  float* film_tables_ = nullptr;
  int batch_ = 0, training_ = 0;
  hipStream_t side_ = nullptr;                 // low-priority wgrad stream (owned)
  std::vector<hipEvent_t> events_;             // recycled per loss_backward
  size_t next_event_ = 0;
  bool side_pending_ = false;
  bool w8_dirty_ = true;                       // the e4m3 weight copies are older than the bf16 operand pack
  bool hs_train_ = false;                      // the forward pass of this step used the hidden-split MLP dataflow
  hipEvent_t take_event();
  OptTable opt_stem_, opt_head_;               // tile tables of the fused optimiser sweep (parameters < / >= head_off_)
  void build_opt_tables();
  bool opt_fused_ok_ = true;
  bool opt_fused_user_ = true;                 // option "opt_fused": false forces the three-pass fallback (A/B runs)
  bool head_norm_ready_ = false;               // the head slots of norm_partial hold this step's output-stage partials (side stream)
  bool head_pending_ = false;                  // an output-stage update is in flight on the side stream
  hipEvent_t head_done_ev_ = nullptr;          // recorded behind it (owned; not from the recycled pool)
  std::vector<hipEvent_t> bucket_ev_;          // dp_layer_events: one per early stem bucket (owned)
  int buckets_recorded_ = 0;
  hipEvent_t stem_done_ev_ = nullptr;          // dp_layer_events: behind the whole backward (fallback of wait_grad_bucket; owned)
  bool stem_done_valid_ = false;

  struct Work {
    // inputs / outputs of the network
    bf16_t* x_bf16 = nullptr;     // [R][Cp]
    float* pe = nullptr;          // [S][E]
    float* pred = nullptr;        // [R][C]
    float* s = nullptr;           // [B] noise levels
    // encoder (index l only in training mode; inference reuses slot 0)
    std::vector<float*> h, h_mid;         // [R][E]
    std::vector<bf16_t*> a1, qkv, o, a2, z1, u;
    float* h_last = nullptr;
    bf16_t* af = nullptr;
    // output stage
    std::vector<float*> y;                // [R][M] trunk (K+1 in training, 1 in inference)
    std::vector<bf16_t*> ya1, o1, ya2;
    std::vector<unsigned char*> ya1_f8, ya2_f8;   // e4m3 copies of the two FiLM-LayerNorm outputs + row scales (fp8 mode)
    std::vector<uint32_t*> sa1, sa2;
    unsigned char* w8 = nullptr;          // [K blocks][r1, r2][M][M] e4m3 weights (rows = output features)
    uint32_t* w8s = nullptr;              // [K blocks][r1, r2][M] row scales
    unsigned char* w8d = nullptr;         // the same for the dgrad layout (rows = input features), training only
    uint32_t* w8ds = nullptr;
    unsigned char* dy8 = nullptr;         // [R][M] e4m3 copy of the gradient entering a dgrad GEMM + its row scales
    uint32_t* sdy = nullptr;
    bf16_t* ao = nullptr;
    bf16_t* emb = nullptr;                // [B][F]
    std::vector<bf16_t*> zf1, f1, p;      // [B][4F]
    std::vector<float*> ss;               // [B][2M]
    // training
    float* eps = nullptr;                 // [R][C]
    float* loss = nullptr;                // [B]
    bf16_t* dpred = nullptr;              // [R][Cp]
    float* dy = nullptr;                  // [R][M]
    // gradient buffers that are wgrad operands have one slot per use (side-stream wgrads read them late)
    std::vector<bf16_t*> dyb;             // [K+1] x [R][M]: trunk gradient entering block k (k = K: from ln_o)
    bf16_t* dA_M = nullptr;               // [R][M]
    std::vector<bf16_t*> do1;             // [K] x [R][M]
    float* dh = nullptr;                  // [R][E]
    std::vector<bf16_t*> dhb;             // [2L+1] x [R][E]: residual-stream gradient versions
    bf16_t* dA_E = nullptr;               // [R][E]
    std::vector<bf16_t*> dqkv;            // [L] x [R][3E]
    bf16_t* do_ = nullptr;                // [R][E]
    std::vector<bf16_t*> dz1;             // [L] x [R][M]
    std::vector<float*> dss;              // [B][2M]
    std::vector<bf16_t*> dss_bf16;        // [K] x [B][2M]
    std::vector<bf16_t*> dp;              // [K] x [B][4F]
    std::vector<bf16_t*> df1;             // [K] x [B][4F]
    float* ln_partial = nullptr;
    size_t ln_partial_elems = 0;
    float* tn_slab = nullptr;             // split-K partial tiles of the wgrad kernel (main stream)
    float* tn_slab_side = nullptr;        // the same for wgrads issued on the side stream
    size_t tn_slab_elems = 0;
    float* norm_partial = nullptr;        // [SMD_NORM_SLOTS]: slots [0, SMD_NORM_HEAD_SLOTS) output-stage slice, the rest stem slice
    float* opt_consts = nullptr;          // [8]: clip multiplier, lr, 1/(1-b1^t), 1/(1-b2^t) of the current update
    bf16_t* zero_page = nullptr;          // [128]
    unsigned* step_arrive = nullptr;      // [64] arrival counter of the fused reverse step
    float* mlp_part = nullptr;            // [4][R][E] partial tiles of the hidden-split MLP kernels
    bf16_t* tn_scratch = nullptr;         // fallback wgrad transposes
    size_t tn_scratch_elems = 0;
  } W;
  int64_t plan(void* base, int batch, int training, Work* w) const;
};
