// extern "C" surface of libsmd_hip.so -- see include/smd_hip.h for the contract.
#include "../../include/smd_hip.h"
#include "../../include/smd_hip_lab.h"

#include <new>

#include "engine.h"

const char* smd_get_error();

struct smd_engine {
  SmdEngine impl;
  explicit smd_engine(const SmdModelDesc& d) : impl(d) {}
};

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline const bf16_t* B(const smd_bf16* p) { return reinterpret_cast<const bf16_t*>(p); }
static inline bf16_t* B(smd_bf16* p) { return reinterpret_cast<bf16_t*>(p); }

#define NEED(e)                                        \
  do {                                                 \
    if (!(e)) {                                        \
      smd_set_error("%s: null engine handle", __func__); \
      return -1;                                       \
    }                                                  \
  } while (0)

extern "C" {

const char* smd_last_error(void) { return smd_get_error(); }
int smd_abi_version(void) { return SMD_ABI_VERSION; }

int smd_engine_create(const smd_model_desc* d, smd_engine** out) {
  SMD_ARG_CHECK(d && out, "smd_engine_create: null argument");
  SMD_ARG_CHECK(d->arch == 0 || d->arch == 1, "smd_engine_create: arch=%d (0 TransformerDDPM, 1 DenseDDPM)", d->arch);
  SMD_ARG_CHECK(d->data_channels > 0 && d->num_layers > 0 && d->mlp_dims > 0 && d->num_timesteps > 0,
                "smd_engine_create: non-positive dimension");
  SMD_ARG_CHECK(d->embed_channels == 128 && d->film_channels == 128,
                "smd_engine_create: embed/film channels are fixed at 128 (models/ncsn.py:151,174)");
  SMD_ARG_CHECK(d->mlp_dims % 256 == 0 && d->mlp_dims <= 4096, "smd_engine_create: mlp_dims=%d must be a multiple of 256 <= 4096", d->mlp_dims);
  if (d->arch == 0) {
    SMD_ARG_CHECK(d->seq_len == 32, "smd_engine_create: TransformerDDPM kernels are specialised for seq_len 32 (got %d)", d->seq_len);
    SMD_ARG_CHECK(d->num_heads > 0 && 128 % d->num_heads == 0 && (128 / d->num_heads == 8 || 128 / d->num_heads == 16 || 128 / d->num_heads == 32),
                  "smd_engine_create: num_heads=%d unsupported (4, 8, 16)", d->num_heads);
    SMD_ARG_CHECK(d->num_mlp_layers > 0, "smd_engine_create: num_mlp_layers must be positive");
  } else {
    SMD_ARG_CHECK(d->seq_len == 1, "smd_engine_create: DenseDDPM takes (B, C) inputs: seq_len must be 1");
  }
  SmdModelDesc m;
  m.arch = d->arch; m.data_channels = d->data_channels; m.seq_len = d->seq_len; m.num_layers = d->num_layers;
  m.num_heads = d->num_heads; m.num_mlp_layers = d->num_mlp_layers; m.mlp_dims = d->mlp_dims;
  m.embed_channels = d->embed_channels; m.film_channels = d->film_channels; m.num_timesteps = d->num_timesteps;
  smd_engine* e = new (std::nothrow) smd_engine(m);
  SMD_ARG_CHECK(e, "smd_engine_create: out of host memory");
  *out = e;
  return 0;
}
void smd_engine_destroy(smd_engine* e) { delete e; }

int smd_engine_num_tensors(const smd_engine* e) { return e ? (int)e->impl.tensors().size() : -1; }
int smd_engine_tensor_info(const smd_engine* e, int i, const char** name, int64_t* offset, int32_t* rows,
                           int32_t* cols) {
  NEED(e);
  SMD_ARG_CHECK(i >= 0 && i < (int)e->impl.tensors().size(), "tensor_info: index %d out of range", i);
  const TensorInfo& t = e->impl.tensors()[i];
  if (name) *name = t.name.c_str();
  if (offset) *offset = t.offset;
  if (rows) *rows = t.rows;
  if (cols) *cols = t.cols;
  return 0;
}
int64_t smd_engine_param_count(const smd_engine* e) { return e ? e->impl.param_count() : -1; }
int64_t smd_engine_head_param_offset(const smd_engine* e) { return e ? e->impl.head_param_offset() : -1; }
int64_t smd_engine_wpack_elems(const smd_engine* e) { return e ? e->impl.wpack_elems() : -1; }
int64_t smd_engine_workspace_bytes(const smd_engine* e, int batch, int training) {
  return (e && batch > 0) ? e->impl.workspace_bytes(batch, training) : -1;
}
int64_t smd_engine_film_table_floats(const smd_engine* e) { return e ? e->impl.film_table_floats() : -1; }
int smd_engine_padded_channels(const smd_engine* e) { return e ? e->impl.padded_channels() : -1; }
int smd_engine_set_option(smd_engine* e, const char* key, int value) {
  NEED(e);
  SMD_ARG_CHECK(key, "set_option: null key");
  if (std::string(key) == "tr_path") { e->impl.tr_path = value ? 1 : 0; return 0; }
  if (std::string(key) == "side_wgrad") return e->impl.set_side_stream(value);
  if (std::string(key) == "fused_attn_bwd") { e->impl.fused_attn_bwd = value < 0 ? 0 : (value > 2 ? 2 : value); return 0; }
  if (std::string(key) == "resgrad_bf16") { e->impl.resgrad_bf16 = value ? 1 : 0; return 0; }
  if (std::string(key) == "film_side") { e->impl.film_side = value; return 0; }
  if (std::string(key) == "pair_wgrad") { e->impl.pair_wgrad = value ? 1 : 0; return 0; }
  if (std::string(key) == "wgrad256_group") { e->impl.wgrad256_group = value < 1 ? 1 : (value > 4 ? 4 : value); return 0; }
  if (std::string(key) == "group_wgrad") { e->impl.group_wgrad = value; return 0; }
  if (std::string(key) == "fused_encoder") { e->impl.fused_encoder = value; return 0; }
  if (std::string(key) == "trunk_bf16") { e->impl.trunk_bf16 = value; return 0; }
  if (std::string(key) == "sample_split") { e->impl.sample_split = value < 0 ? 0 : value; return 0; }
  if (std::string(key) == "nt256_min_tiles") { e->impl.nt256_min_tiles = value; return 0; }
  if (std::string(key) == "grad_memset") { e->impl.grad_memset = value; return 0; }
  if (std::string(key) == "tail_on_main") { e->impl.tail_on_main = value ? 1 : 0; return 0; }
  if (std::string(key) == "film_side_fwd") { e->impl.film_side_fwd = value ? 1 : 0; return 0; }
  if (std::string(key) == "loss_kind") { e->impl.loss_kind = value ? 1 : 0; return 0; }
  if (std::string(key) == "label_min") { e->impl.label_min = value ? 1 : 0; return 0; }
  if (std::string(key) == "mlp_hs") { e->impl.mlp_hs = value ? 1 : 0; return 0; }
  if (std::string(key) == "fp8") { e->impl.fp8 = value ? 1 : 0; return 0; }
  if (std::string(key) == "fp8_dgrad") { e->impl.fp8_dgrad = value ? 1 : 0; return 0; }
  if (std::string(key) == "opt_overlap") { e->impl.opt_overlap = value & 3; return 0; }
  if (std::string(key) == "dp_layer_events") { e->impl.dp_layer_events = value ? 1 : 0; return 0; }
  if (std::string(key) == "opt_fused") { e->impl.set_opt_fused(value != 0); return 0; }     // 0: norm / update / re-cast as three passes (A/B)
  if (std::string(key) == "opt_side_blocks") { e->impl.opt_side_blocks = value < 0 ? 0 : value; return 0; }
  if (std::string(key) == "w8_dirty") { e->impl.mark_w8_dirty(); return 0; }   // the bf16 operand pack changed under another handle
  smd_set_error("set_option: unknown key '%s'", key);
  return -1;
}

int smd_engine_bind_params(smd_engine* e, float* params, smd_bf16* wpack) { NEED(e); return e->impl.bind_params(params, B(wpack)); }
int smd_engine_bind_train(smd_engine* e, float* grads, float* m, float* v, float* ema, uint32_t* step, float* metrics) {
  NEED(e);
  return e->impl.bind_train(grads, m, v, ema, step, metrics);
}
int smd_engine_bind_workspace(smd_engine* e, void* ws, int64_t bytes, int batch, int training, void* stream) {
  NEED(e);
  return e->impl.bind_workspace(ws, bytes, batch, training, S(stream));
}
int smd_engine_bind_schedule(smd_engine* e, const float* coef, const float* sqrt_ap, const float* ape, float* tables) {
  NEED(e);
  return e->impl.bind_schedule(coef, sqrt_ap, ape, tables);
}
int smd_engine_refresh_weights(smd_engine* e, void* stream) { NEED(e); return e->impl.refresh_weights(S(stream)); }
int smd_engine_forward(smd_engine* e, const float* x, const float* s, float* out, void* stream) {
  NEED(e);
  return e->impl.forward(x, s, out, S(stream));
}
int smd_engine_forward_level(smd_engine* e, const float* x, const int32_t* level_ptr, float* out, void* stream) {
  NEED(e);
  return e->impl.forward_level(x, level_ptr, out, S(stream));
}
int smd_engine_loss_backward(smd_engine* e, const float* x0, const int32_t* labels, const float* eps_in,
                             uint32_t seed_lo, uint32_t seed_hi, uint32_t sample_offset, float inv_global_count,
                             int stage, void* stream) {
  NEED(e);
  return e->impl.loss_backward(x0, labels, eps_in, seed_lo, seed_hi, sample_offset, inv_global_count, stage, S(stream));
}
int smd_engine_forward_train(smd_engine* e, const float* x, const float* noise_level, float* eps_out, void* stream) {
  NEED(e);
  return e->impl.forward_train(x, noise_level, eps_out, S(stream));
}
int smd_engine_backward_from(smd_engine* e, const float* dpred, int stage, void* stream) {
  NEED(e);
  return e->impl.backward_from(dpred, stage, S(stream));
}
int smd_engine_set_used_alphas(smd_engine* e, const float* used_alphas) {
  NEED(e);
  e->impl.set_used_alphas(used_alphas);
  return 0;
}
int64_t smd_engine_debug_snapshot_bytes(const smd_engine* e) { return e ? e->impl.debug_snapshot_bytes() : -1; }
int smd_engine_debug_snapshots(smd_engine* e, void* buf, int64_t bytes) {
  NEED(e);
  return e->impl.set_debug_snapshots(buf, bytes);
}
int smd_engine_debug_tensor(const smd_engine* e, const char* name, int index, const void** ptr, int64_t* rows, int64_t* cols,
                            int32_t* dtype) {
  NEED(e);
  int dt = 0;
  const int rc = e->impl.debug_tensor(name, index, ptr, rows, cols, &dt);
  if (rc == 0 && dtype) *dtype = dt;
  return rc;
}
const float* smd_engine_loss_per_sample(const smd_engine* e) { return e ? e->impl.loss_per_sample() : nullptr; }
const float* smd_engine_pred(const smd_engine* e) { return e ? e->impl.pred() : nullptr; }

int smd_engine_num_grad_buckets(const smd_engine* e) { return e ? e->impl.num_grad_buckets() : -1; }
int smd_engine_grad_bucket(const smd_engine* e, int bucket, int64_t* offset, int64_t* length) {
  NEED(e);
  return e->impl.grad_bucket(bucket, offset, length);
}
int smd_engine_wait_grad_bucket(smd_engine* e, int bucket, void* stream) { NEED(e); return e->impl.wait_grad_bucket(bucket, S(stream)); }
int smd_engine_join_update(smd_engine* e, void* stream) { NEED(e); return e->impl.join_update(S(stream)); }
int smd_engine_optimizer_step(smd_engine* e, const smd_train_hyper* h, void* stream) {
  NEED(e);
  SMD_ARG_CHECK(h, "optimizer_step: null hyper-parameters");
  TrainHyper t;
  t.lr0 = h->lr0; t.lr_gamma = h->lr_gamma; t.lr_interval = h->lr_interval; t.beta1 = h->beta1; t.beta2 = h->beta2;
  t.eps = h->eps; t.grad_clip = h->grad_clip; t.mu = h->mu; t.grad_scale = h->grad_scale;
  return e->impl.optimizer_step(t, S(stream));
}
int smd_engine_prepare_sampler(smd_engine* e, void* stream) { NEED(e); return e->impl.prepare_sampler(S(stream)); }
int smd_engine_init_state(smd_engine* e, float* x, uint32_t lo, uint32_t hi, uint32_t off, void* stream) {
  NEED(e);
  return e->impl.init_state(x, lo, hi, off, S(stream));
}
int smd_engine_load_state(smd_engine* e, const float* x, void* stream) {
  NEED(e);
  return e->impl.load_state(x, S(stream));
}
static int sample_step_part(smd_engine* e, const smd_sample_io* io, int part, void* stream) {
  NEED(e);
  SMD_ARG_CHECK(io, "sample_step: null io");
  SampleStepIO s;
  s.x = io->x; s.t_ptr = io->t_ptr; s.z_in = io->z_in; s.seed_lo = io->seed_lo; s.seed_hi = io->seed_hi;
  s.sample_offset = io->sample_offset; s.infill_samples = io->infill_samples; s.infill_masks = io->infill_masks;
  s.infill_z_in = io->infill_z_in; s.metrics_partial = io->metrics_partial; s.collection = io->collection;
  s.slot_table = io->slot_table;
  s.tf_noise_keys = io->tf_noise_keys; s.tf_infill_keys = io->tf_infill_keys; s.tf_n_total = io->tf_n_total; s.tf_t0 = io->tf_t0;
  s.key_ptr = io->key_ptr;
  return e->impl.sample_step(s, S(stream), part);
}
int smd_engine_sample_step(smd_engine* e, const smd_sample_io* io, void* stream) { return sample_step_part(e, io, 0, stream); }
int smd_engine_sample_step_part(smd_engine* e, const smd_sample_io* io, int part, void* stream) {
  SMD_ARG_CHECK(part == 1 || part == 2, "smd_engine_sample_step_part: part=%d (1 stem, 2 output stage + reverse update)", part);
  return sample_step_part(e, io, part, stream);
}

// ------------------------------------------------------------------ single kernels
int smd_set_tuning(const char* key, int value) { return smd_tuning_set(key, value); }
int smd_gemm_bf16_nt(const smd_bf16* A, int lda, const smd_bf16* Bt, int ldb, int M, int N, int K, const float* bias,
                     int act, const float* residual, int ld_res, float* out_f32, int ld_out, smd_bf16* out_bf16,
                     int ld_outb, void* stream) {
  GemmEpilogue ep;
  ep.bias = bias; ep.act = act; ep.res_f32 = residual; ep.ld_res = ld_res;
  ep.out_f32 = out_f32; ep.ld_out = ld_out; ep.out_bf16 = B(out_bf16); ep.ld_outb = ld_outb;
  return launch_gemm_nt(B(A), lda, B(Bt), ldb, M, N, K, ep, S(stream));
}
int smd_quantize_rows_e4m3(const smd_bf16* in, int ld, int rows, int K, uint8_t* out8, uint32_t* scale, void* stream) {
  return launch_quantize_rows_e4m3(B(in), ld, rows, K, out8, scale, S(stream));
}
int smd_gemm_e4m3_nt(const uint8_t* A8, int lda, const uint32_t* scale_a, const uint8_t* Bt8, int ldb, const uint32_t* scale_b, int M,
                     int N, int K, const float* bias, const float* residual, int ld_res, float* out_f32, int ld_out,
                     smd_bf16* out_bf16, int ld_outb, void* stream) {
  GemmEpilogue ep;
  ep.bias = bias; ep.res_f32 = residual; ep.ld_res = ld_res; ep.out_f32 = out_f32; ep.ld_out = ld_out; ep.out_bf16 = B(out_bf16);
  ep.ld_outb = ld_outb;
  return launch_gemm_nt256_fp8(A8, lda, scale_a, Bt8, ldb, scale_b, M, N, K, ep, S(stream));
}
int smd_layernorm_fwd_e4m3(const float* x, int rows, int D, const float* gamma, const float* beta, const float* film_scale,
                           const float* film_shift, int ld_film, int rows_per_sample, int swish, uint8_t* out8, uint32_t* out_scale,
                           smd_bf16* out_bf16, void* stream) {
  LnArgs a;
  a.x = x; a.rows = rows; a.D = D; a.gamma = gamma; a.beta = beta; a.film_scale = film_scale; a.film_shift = film_shift;
  a.ld_film = ld_film; a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; a.swish = swish; a.out = B(out_bf16);
  a.out_f8 = out8; a.out_scale = out_scale;
  return launch_layernorm_fwd(a, S(stream));
}
int smd_mlp_block_fwd(const float* h_in, float* h_out, int rows, const float* gamma, const float* beta, const smd_bf16* W1t,
                      const float* b1, const smd_bf16* W2t, const float* b2, int hidden, smd_bf16* save_a2, smd_bf16* save_z1,
                      smd_bf16* save_u, void* stream) {
  return launch_mlp_block_fwd(h_in, h_out, rows, gamma, beta, B(W1t), b1, B(W2t), b2, hidden, B(save_a2), B(save_z1), B(save_u),
                              S(stream));
}
int smd_mlp_block_fwd_hs(const smd_bf16* a2, const float* h_res, int rows, const smd_bf16* W1t, const float* b1,
                         const smd_bf16* W2t, const float* b2, int hidden, float* part, void* stream) {
  return launch_mlp_block_fwd_hs(B(a2), h_res, rows, B(W1t), b1, B(W2t), b2, hidden, part, S(stream));
}
int smd_mlp_block_bwd_hs(const smd_bf16* a2, const smd_bf16* dh, int rows, const smd_bf16* W1t, const smd_bf16* W2, const smd_bf16* W1,
                         const float* b1, int hidden, smd_bf16* u, smd_bf16* dz, float* part, void* stream) {
  return launch_mlp_block_bwd_hs(B(a2), B(dh), rows, B(W1t), B(W2), B(W1), b1, hidden, B(u), B(dz), part, S(stream));
}
int smd_ln128_bwd_parts(const float* x, const float* parts, int64_t part_stride, int rows, const float* gamma, const float* dres,
                        float* dx_f32, smd_bf16* dx_bf16, float* partial, void* stream) {
  return launch_ln128_bwd_parts(x, parts, (size_t)part_stride, rows, gamma, dres, dx_f32, B(dx_bf16), partial, S(stream));
}
int smd_ln128_parts(const float* parts, int64_t part_stride, int rows, const float* gamma, const float* beta, float* x_out,
                    smd_bf16* ln_out, void* stream) {
  return launch_ln128_parts(parts, (size_t)part_stride, rows, gamma, beta, x_out, B(ln_out), S(stream));
}
int smd_attn_block_fwd_ex(const float* h_in, const float* h_parts, int64_t part_stride, float* h_comb, float* h_out, int rows,
                          const float* gamma, const float* beta, const smd_bf16* Wqkv_t, const float* b_qkv,
                          const smd_bf16* Wo_t, const float* b_o, int num_heads, const float* gamma2, const float* beta2,
                          smd_bf16* a2_out, smd_bf16* save_a1, smd_bf16* save_qkv, smd_bf16* save_o, void* stream) {
  AttnBlockExtra ex;
  ex.h_parts = h_parts; ex.part_stride = (size_t)part_stride; ex.h_comb = h_comb; ex.gamma2 = gamma2; ex.beta2 = beta2;
  ex.a2_out = B(a2_out);
  return launch_attn_block_fwd(h_in, h_out, rows, gamma, beta, B(Wqkv_t), b_qkv, B(Wo_t), b_o, num_heads, B(save_a1), B(save_qkv),
                               B(save_o), S(stream), &ex);
}
int smd_attn_block_fwd(const float* h_in, float* h_out, int rows, const float* gamma, const float* beta, const smd_bf16* Wqkv_t,
                       const float* b_qkv, const smd_bf16* Wo_t, const float* b_o, int num_heads, smd_bf16* save_a1,
                       smd_bf16* save_qkv, smd_bf16* save_o, void* stream) {
  return launch_attn_block_fwd(h_in, h_out, rows, gamma, beta, B(Wqkv_t), b_qkv, B(Wo_t), b_o, num_heads, B(save_a1), B(save_qkv),
                               B(save_o), S(stream));
}
int smd_attn_block_bwd(const smd_bf16* dh_mid, const smd_bf16* qkv, const smd_bf16* Wo, const smd_bf16* Wqkv, smd_bf16* dqkv,
                       smd_bf16* da1, int rows, int num_heads, void* stream) {
  return launch_attn_block_bwd(B(dh_mid), B(qkv), B(Wo), B(Wqkv), B(dqkv), B(da1), rows, num_heads, S(stream));
}
int smd_attn_block_bwd_ln(const smd_bf16* qkv, const smd_bf16* Wo, const smd_bf16* Wqkv, smd_bf16* dqkv, smd_bf16* da1,
                          const float* h_mid, const float* da2_parts, int64_t part_stride, const float* gamma2, float* dh,
                          smd_bf16* dh_mid_out, float* partial2, const float* h, const float* gamma1, smd_bf16* dh_out,
                          float* partial1, int rows, int num_heads, void* stream) {
  AttnBwdLnArgs a;
  a.qkv = B(qkv); a.Wo = B(Wo); a.Wqkv = B(Wqkv); a.dqkv = B(dqkv); a.da1 = B(da1);
  a.h_mid = h_mid; a.da2_parts = da2_parts; a.part_stride = (size_t)part_stride; a.gamma2 = gamma2; a.dh = dh;
  a.dh_mid_out = B(dh_mid_out); a.partial2 = partial2; a.h = h; a.gamma1 = gamma1; a.dh_out = B(dh_out); a.partial1 = partial1;
  return launch_attn_block_bwd_ln(a, rows, num_heads, S(stream));
}
int smd_gemm_bf16_tn(const smd_bf16* X, int ldx, const smd_bf16* dY, int ldy, int M, int Kd, int N, float* out, int ldo,
                     float* bias_out, const smd_bf16* zero_page, float* slab, int64_t slab_elems, smd_bf16* scratch,
                     int64_t scratch_elems, int tr_path, void* stream) {
  TnLaunch t;
  t.X = B(X); t.ldx = ldx; t.dY = B(dY); t.ldy = ldy; t.Mrows = M; t.Kd = Kd; t.N = N; t.out = out; t.ldo = ldo;
  t.bias_out = bias_out; t.zero_page = B(zero_page); t.slab = slab; t.slab_elems = (size_t)slab_elems;
  t.scratch = B(scratch); t.scratch_elems = (size_t)scratch_elems; t.tr_path = tr_path;
  return launch_gemm_tn(t, S(stream));
}
int64_t smd_gemm_tn_slab_elems(void) { return (int64_t)gemm_tn_slab_elems(); }
int smd_layernorm_fwd(const float* x, int rows, int D, const float* gamma, const float* beta, const float* film_scale,
                      const float* film_shift, int ld_film, int rows_per_sample, int swish, smd_bf16* out, void* stream) {
  LnArgs a;
  a.x = x; a.rows = rows; a.D = D; a.gamma = gamma; a.beta = beta; a.film_scale = film_scale; a.film_shift = film_shift;
  a.ld_film = ld_film; a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; a.swish = swish; a.out = B(out);
  return launch_layernorm_fwd(a, S(stream));
}
int smd_layernorm_fwd_ex(const float* x, const smd_bf16* x_bf16, int rows, int D, const float* gamma, const float* beta,
                         const float* film_scale, const float* film_shift, int ld_film, int rows_per_sample, int swish, smd_bf16* out,
                         void* stream) {
  LnArgs a;
  a.x = x; a.x_bf16 = B(x_bf16); a.rows = rows; a.D = D; a.gamma = gamma; a.beta = beta; a.film_scale = film_scale;
  a.film_shift = film_shift; a.ld_film = ld_film; a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; a.swish = swish;
  a.out = B(out);
  return launch_layernorm_fwd(a, S(stream));
}
int smd_layernorm_bwd(const float* x, int rows, int D, const float* gamma, const float* beta, const float* film_scale,
                      const float* film_shift, int ld_film, int rows_per_sample, int swish, const smd_bf16* dout,
                      float* dx, float* dgamma, float* dbeta, float* dscale, float* dshift, float* partial,
                      int64_t partial_elems, void* stream) {
  LnBwdArgs b;
  b.f.x = x; b.f.rows = rows; b.f.D = D; b.f.gamma = gamma; b.f.beta = beta; b.f.film_scale = film_scale;
  b.f.film_shift = film_shift; b.f.ld_film = ld_film; b.f.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
  b.f.swish = swish;
  b.dout = B(dout); b.dx = dx; b.dgamma = dgamma; b.dbeta = dbeta; b.dscale = dscale; b.dshift = dshift;
  b.partial = partial; b.partial_elems = (size_t)partial_elems;
  return launch_layernorm_bwd(b, S(stream));
}
int smd_layernorm_bwd_ex(const float* x, const smd_bf16* x_bf16, int rows, int D, const float* gamma, const float* beta,
                         const smd_bf16* dout, const float* dres, float* dx, smd_bf16* dx_bf16, float* dgamma, float* dbeta,
                         float* partial, int64_t partial_elems, void* stream) {
  LnBwdArgs b;
  b.f.x = x; b.f.x_bf16 = B(x_bf16); b.f.rows = rows; b.f.D = D; b.f.gamma = gamma; b.f.beta = beta;
  b.dout = B(dout); b.dres = dres; b.dx = dx; b.dx_bf16 = B(dx_bf16); b.dgamma = dgamma; b.dbeta = dbeta;
  b.partial = partial; b.partial_elems = (size_t)partial_elems;
  return launch_layernorm_bwd(b, S(stream));
}
int smd_layernorm_bwd_film(const float* x, const smd_bf16* x_bf16, int rows, int D, const float* gamma, const float* beta,
                           const float* film_scale, const float* film_shift, int ld_film, int rows_per_sample, int swish,
                           const smd_bf16* dout, const float* dres, const smd_bf16* dres_bf16, float* dx,
                           smd_bf16* dx_bf16, float* dgamma, float* dbeta, float* dscale, float* dshift,
                           int dfilm_accumulate, float* partial, int64_t partial_elems, void* stream) {
  LnBwdArgs b;
  b.f.x = x; b.f.x_bf16 = B(x_bf16); b.f.rows = rows; b.f.D = D; b.f.gamma = gamma; b.f.beta = beta;
  b.f.film_scale = film_scale; b.f.film_shift = film_shift; b.f.ld_film = ld_film; b.f.rows_per_sample = rows_per_sample;
  b.f.swish = swish;
  b.dout = B(dout); b.dres = dres; b.dres_bf16 = B(dres_bf16); b.dx = dx; b.dx_bf16 = B(dx_bf16);
  b.dgamma = dgamma; b.dbeta = dbeta; b.dscale = dscale; b.dshift = dshift; b.dfilm_accumulate = dfilm_accumulate;
  b.partial = partial; b.partial_elems = (size_t)partial_elems;
  return launch_layernorm_bwd(b, S(stream));
}
int smd_attention_fwd(const smd_bf16* qkv, smd_bf16* out, int Bn, int Sn, int E, int H, void* stream) {
  return launch_attention_fwd(B(qkv), B(out), Bn, Sn, E, H, S(stream));
}
int smd_attention_bwd(const smd_bf16* qkv, const smd_bf16* dout, smd_bf16* dqkv, int Bn, int Sn, int E, int H, void* stream) {
  return launch_attention_bwd(B(qkv), B(dout), B(dqkv), Bn, Sn, E, H, S(stream));
}
int smd_noise_embed(const float* s, int n, int channels, smd_bf16* out, int ld_out, void* stream) {
  return launch_noise_embed(s, n, channels, B(out), ld_out, S(stream));
}
int smd_q_sample(const float* x0, int Bn, int Sn, int C, int Cp, int T, const float* alphas_prod_ext, const int32_t* labels,
                 int label_min, const float* used_alphas, const float* eps_in, uint32_t seed_lo, uint32_t seed_hi,
                 const uint32_t* step_ptr, uint32_t sample_offset, smd_bf16* xt_bf16, float* eps_out, float* noise_level_out,
                 void* stream) {
  QSampleArgs q;
  q.x0 = x0; q.B = Bn; q.S = Sn; q.C = C; q.Cp = Cp; q.T = T; q.alphas_prod_ext = alphas_prod_ext;
  q.labels = labels; q.label_min = label_min; q.alpha_in = used_alphas; q.eps_in = eps_in;
  q.key = RngKey{seed_lo, seed_hi}; q.step_ptr = step_ptr; q.sample_offset = sample_offset;
  q.xt_bf16 = B(xt_bf16); q.eps_out = eps_out; q.s_out = noise_level_out;
  return launch_q_sample(q, S(stream));
}
int smd_mse_fwd_bwd(const float* pred, const float* eps, int Bn, int Sn, int C, int Cp, float inv_global_count,
                    float* loss_per_sample, smd_bf16* dpred_bf16, void* stream) {
  return launch_mse_loss_grad(pred, eps, Bn, Sn, C, Cp, inv_global_count, loss_per_sample, B(dpred_bf16), S(stream));
}
int smd_adam_clip_ema(float* params, const float* grads, float* m, float* v, float* ema, int64_t n, const smd_train_hyper* h,
                      uint32_t* step_ptr, float* norm_partial, float* metrics_out, void* stream) {
  SMD_ARG_CHECK(h && n > 0, "adam_clip_ema: bad arguments");
  AdamArgs a;
  a.params = params; a.grads = grads; a.m = m; a.v = v; a.ema = ema; a.n = (size_t)n;
  a.lr0 = h->lr0; a.lr_gamma = h->lr_gamma; a.lr_interval = h->lr_interval; a.beta1 = h->beta1; a.beta2 = h->beta2;
  a.eps = h->eps; a.grad_clip = h->grad_clip; a.mu = h->mu; a.grad_scale = h->grad_scale;
  a.step_ptr = step_ptr; a.norm_partial = norm_partial; a.metrics_out = metrics_out;
  int rc = launch_grad_sumsq(a, S(stream));
  return rc ? rc : launch_adam_clip_ema(a, S(stream));
}
int smd_set_timestep(int32_t* t_ptr, int32_t t, void* stream) { return launch_set_t(t_ptr, t, S(stream)); }
int smd_langevin_step(const smd_langevin_io* io, int Bn, int Sn, int C, void* stream) {
  SMD_ARG_CHECK(io, "langevin_step: null io");
  LangevinStepArgs a;
  a.x = io->x; a.grad = io->grad; a.B = Bn; a.S = Sn; a.C = C; a.alpha = io->alpha; a.noise_coef = io->noise_coef;
  a.z_in = io->z_in; a.key = RngKey{io->seed_lo, io->seed_hi}; a.step = io->step; a.sample_offset = io->sample_offset;
  a.use_threefry = io->use_threefry;
  a.tf_noise_key[0] = io->tf_noise_key[0]; a.tf_noise_key[1] = io->tf_noise_key[1];
  a.tf_infill_key[0] = io->tf_infill_key[0]; a.tf_infill_key[1] = io->tf_infill_key[1];
  a.tf_n_total = io->tf_n_total;
  a.infill_samples = io->infill_samples; a.infill_masks = io->infill_masks; a.infill_z_in = io->infill_z_in;
  a.infill_sigma = io->infill_sigma; a.metrics_partial = io->metrics_partial; a.collect_out = io->collect_out;
  a.step_table = io->step_table; a.slot_table = io->slot_table; a.key_table = io->key_table; a.k_ptr = io->k_ptr;
  a.arrive = io->arrive; a.collection = io->collection; a.sigma_out = io->sigma_out; a.n_steps = io->n_steps;
  a.level_out = io->level_out; a.steps_per_level = io->steps_per_level; a.n_levels = io->n_levels;
  return launch_langevin_step(a, S(stream));
}
int smd_rng_normal(float* out, int Bn, int per_sample, uint32_t lo, uint32_t hi, uint32_t stream_id, uint32_t off, void* stream) {
  return launch_fill_normal(out, Bn, per_sample, RngKey{lo, hi}, stream_id, off, S(stream));
}
int smd_threefry_bits(uint32_t* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1, void* stream) {
  return launch_threefry_bits(out, n_total, offset, count, k0, k1, S(stream));
}
int smd_threefry_uniform(float* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1, float minval,
                         float maxval, void* stream) {
  return launch_threefry_uniform(out, n_total, offset, count, k0, k1, minval, maxval, S(stream));
}
int smd_threefry_normal(float* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1,
                        const uint32_t* key_table, const int32_t* idx_ptr, int idx_mul, int idx_add, void* stream) {
  return launch_threefry_normal(out, n_total, offset, count, k0, k1, key_table, idx_ptr, idx_mul, idx_add, S(stream));
}
int smd_threefry_randint(int32_t* out, int64_t n_total, int64_t offset, int64_t count, uint32_t k0, uint32_t k1,
                         int32_t minval, int32_t maxval, void* stream) {
  return launch_threefry_randint(out, n_total, offset, count, k0, k1, minval, maxval, S(stream));
}
int smd_cast_pad_bf16(const float* in, int rows, int cols, smd_bf16* out, int ld_out, void* stream) {
  return launch_cast_pad_bf16(in, rows, cols, B(out), ld_out, S(stream));
}
int smd_ddpm_reverse_step(float* x, const float* eps_hat, int Bn, int Sn, int C, const float* coef, int T, const int32_t* t_ptr,
                          const float* z_in, uint32_t lo, uint32_t hi, uint32_t off, float* metrics_partial,
                          float* collection, const int32_t* slot_table, void* stream) {
  ReverseStepArgs a;
  a.x = x; a.eps_hat = eps_hat; a.B = Bn; a.S = Sn; a.C = C; a.Cp = C; a.coef = coef; a.T = T; a.t_ptr = t_ptr; a.z_in = z_in;
  a.key = RngKey{lo, hi}; a.sample_offset = off; a.metrics_partial = metrics_partial; a.collection = collection;
  a.slot_table = slot_table;
  return launch_reverse_step(a, S(stream));
}

int smd_probe_tr_read(const smd_bf16* image, smd_bf16* out, void* stream) { return launch_probe_tr_read(B(image), B(out), S(stream)); }

// ---- lab probe: a stream restricted by a raw CU mask (hipExtStreamCreateWithCUMask).  What the mask can express on this part is
// documented in include/smd_hip_lab.h (tools/cumask_probe.hip: bit i belongs to XCC i % 8; an XCC whose share is empty is NOT masked).
int smd_probe_stream_create_cu_mask(const uint32_t* mask_words, int n_words, void** stream_out) {
  SMD_ARG_CHECK(mask_words && stream_out && n_words >= 1 && n_words <= 8, "smd_probe_stream_create_cu_mask: bad argument");
  hipStream_t st = nullptr;
  const hipError_t err = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask_words);
  if (err != hipSuccess) { smd_set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(err)); return (int)err; }
  *stream_out = reinterpret_cast<void*>(st);
  return 0;
}
int smd_probe_stream_destroy(void* stream) {
  SMD_ARG_CHECK(stream, "smd_probe_stream_destroy: null stream");
  const hipError_t err = hipStreamDestroy(S(stream));
  if (err != hipSuccess) { smd_set_error("hipStreamDestroy: %s", hipGetErrorString(err)); return (int)err; }
  return 0;
}

// lab probe: where a one-wave kernel runs and the shader clock it sees.  out[0] = XCC id, out[1] = HW_ID, out[2..3] = s_memtime
// ticks (shader cycles) and out[4..5] = s_memrealtime ticks (100 MHz) spent spinning for ~spin_us
__global__ __launch_bounds__(64) void probe_clock_kernel(uint32_t* __restrict__ out, int spin_us) {
  const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);         // HW_REG_HW_ID
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t c0 = __builtin_amdgcn_s_memtime();
  uint64_t r1 = r0;
  while (r1 - r0 < (uint64_t)spin_us * 100) { __builtin_amdgcn_s_sleep(4); r1 = __builtin_amdgcn_s_memrealtime(); }
  const uint64_t c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    uint32_t* o = out + 8 * blockIdx.x;
    o[0] = xcc; o[1] = hw;
    o[2] = (uint32_t)(c1 - c0); o[3] = (uint32_t)((c1 - c0) >> 32);
    o[4] = (uint32_t)(r1 - r0); o[5] = (uint32_t)((r1 - r0) >> 32);
    o[6] = (uint32_t)r0; o[7] = (uint32_t)(r0 >> 32);
  }
}
// lab probe: every XCD reads all `bytes` of `p` (workgroup b runs on XCD b % 8 and reads slice b / 8): the buffer is then
// resident in all eight L2s -- what a weight prefetch ahead of a latency-bound kernel would achieve
__global__ __launch_bounds__(256) void l2_warm_kernel(const uint4* __restrict__ p, size_t n16, int slices, uint32_t* __restrict__ sink) {
  const int slice = blockIdx.x >> 3;
  const size_t per = (n16 + slices - 1) / slices;
  const size_t lo = (size_t)slice * per, hi = lo + per < n16 ? lo + per : n16;
  uint32_t acc = 0;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x9E3779B9u && sink) *sink = acc;     // never true in practice: keeps the loads
}
int smd_probe_l2_warm(const void* p, int64_t bytes, uint32_t* sink, void* stream) {
  SMD_ARG_CHECK(p && bytes >= 16 && bytes % 16 == 0, "smd_probe_l2_warm: bad argument");
  const size_t n16 = (size_t)bytes / 16;
  int slices = (int)((n16 + 2047) / 2048);          // <= 32 KiB per workgroup
  if (slices > 256) slices = 256;
  hipLaunchKernelGGL(l2_warm_kernel, dim3(8 * slices), dim3(256), 0, S(stream), reinterpret_cast<const uint4*>(p), n16, slices, sink);
  SMD_LAUNCH_CHECK();
  return 0;
}
int smd_probe_clock(uint32_t* out, int blocks, int spin_us, void* stream) {
  SMD_ARG_CHECK(out && blocks > 0 && spin_us >= 0, "smd_probe_clock: bad argument");
  hipLaunchKernelGGL(probe_clock_kernel, dim3(blocks), dim3(64), 0, S(stream), out, spin_us);
  SMD_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
