// SmdEngine: launch sequences for the eps-net (reference models/ncsn.py:122-179), the DDPM
// objective (utils/losses.py:250-308), the optimiser step (train_ncsn.py:260-288) and one
// reverse-diffusion iteration (utils/ebm_utils.py:327-394).  See engine.h.
#include "engine.h"

#include <algorithm>
#include <cstdlib>
#include <cmath>

#include <cstdarg>
#include <cstdio>
#include <cstring>

// ------------------------------------------------------------------ error string (thread-local)
static thread_local char g_err[512] = "";
void smd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* smd_get_error() { return g_err; }

#define RC(x)            \
  do {                   \
    int rc__ = (x);      \
    if (rc__) return rc__; \
  } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

int launch_pos_encoding(float* pe, int S, int channels, hipStream_t st);

// ------------------------------------------------------------------ layout
SmdEngine::SmdEngine(const SmdModelDesc& d) : d_(d) { build_layout(); }
SmdEngine::~SmdEngine() {
  if (side_) (void)hipStreamSynchronize(side_);        // a deferred update may still be reading the caller's buffers
  if (head_done_ev_) (void)hipEventDestroy(head_done_ev_);
  for (hipEvent_t e : bucket_ev_) (void)hipEventDestroy(e);
  if (stem_done_ev_) (void)hipEventDestroy(stem_done_ev_);
  for (hipEvent_t e : events_) (void)hipEventDestroy(e);
  if (side_) (void)hipStreamDestroy(side_);
}

int SmdEngine::set_side_stream(int enable) {
  if (enable && !side_) {
    int lo = 0, hi = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);      // lo = least priority (numerically largest)
    const char* pr = getenv("SMD_SIDE_PRIORITY");            // "normal": same priority as the main stream (debug / A-B)
    if (e == hipSuccess) e = (pr && !strcmp(pr, "normal")) ? hipStreamCreateWithFlags(&side_, hipStreamNonBlocking)
                                                           : hipStreamCreateWithPriority(&side_, hipStreamNonBlocking, lo);
    if (e != hipSuccess) { smd_set_error("set_side_stream: %s", hipGetErrorString(e)); side_ = nullptr; return (int)e; }
  }
  side_wgrad = enable ? 1 : 0;
  return 0;
}

hipEvent_t SmdEngine::take_event() {
  if (next_event_ == events_.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    events_.push_back(e);
  }
  return events_[next_event_++];
}

// dW / db of one Dense.  On the side stream when enabled: the side stream first waits for everything the main
// stream has enqueued so far (dY is complete), then runs the wgrad with its own slab workspace.
int SmdEngine::wgrad(const DenseP& p, const bf16_t* X, int ldx, const bf16_t* dY, int ldy, int M, bool allow_side,
                     hipStream_t st) {
  TnLaunch t;
  t.X = X; t.ldx = ldx; t.dY = dY; t.ldy = ldy; t.Mrows = M; t.Kd = p.K; t.N = p.N;
  t.out = G(p.w_off); t.ldo = p.N; t.bias_out = G(p.b_off);
  t.zero_page = W.zero_page; t.slab = W.tn_slab; t.slab_elems = W.tn_slab_elems;
  t.scratch = W.tn_scratch; t.scratch_elems = W.tn_scratch_elems; t.tr_path = tr_path;
  if (allow_side && group_wgrad && tr_path && t.ldo == t.N) {
    int per = 0;
    if (gemm_tn256_plan(t, &per) == 0) {       // a 128-wide-kernel problem: launched with its peers at the end
      deferred_wgrads_.push_back(t);
      return 0;
    }
  }
  if (!(allow_side && side_wgrad && side_ && tr_path)) return launch_gemm_tn(t, st);
  t.slab = W.tn_slab_side;
  if (pair_wgrad) {                      // a 256x256-kernel problem: wait for its partners (the Dense layers of the DenseResBlocks)
    int per = 0;
    if (gemm_tn256_plan(t, &per) > 0) {
      pending256_.push_back(t);
      if ((int)pending256_.size() < wgrad256_group) return 0;
      return flush_pending256(st);
    }
  }
  hipEvent_t ev = take_event();
  SMD_ARG_CHECK(ev, "wgrad: cannot create an event");
  hipError_t e = hipEventRecord(ev, st);
  if (e == hipSuccess) e = hipStreamWaitEvent(side_, ev, 0);
  if (e != hipSuccess) { smd_set_error("wgrad: event: %s", hipGetErrorString(e)); return (int)e; }
  side_pending_ = true;
  return launch_gemm_tn(t, side_);
}

// The collected 256x256-kernel weight gradients as ONE side-stream launch behind everything the main stream has enqueued so
// far (four 2048 x 2048 problems = 256 tiles: no split over m, no slabs, no reduce; fewer: two or four m-splits)
int SmdEngine::flush_pending256(hipStream_t st) {
  if (pending256_.empty()) return 0;
  hipEvent_t ev = take_event();
  SMD_ARG_CHECK(ev, "wgrad: cannot create an event");
  hipError_t e2 = hipEventRecord(ev, st);
  if (e2 == hipSuccess) e2 = hipStreamWaitEvent(side_, ev, 0);
  if (e2 != hipSuccess) { smd_set_error("wgrad: event: %s", hipGetErrorString(e2)); return (int)e2; }
  side_pending_ = true;
  int rc = 0;
  for (size_t i = 0; i < pending256_.size() && rc == 0; i += SMD_TN256_MULTI_MAX) {
    const int n = (int)std::min<size_t>(SMD_TN256_MULTI_MAX, pending256_.size() - i);
    rc = launch_gemm_tn256_multi(pending256_.data() + i, n, side_);
  }
  pending256_.clear();
  return rc;
}

// The deferred 128-wide weight gradients (every operand is a saved activation or a per-use gradient slot, so
// they can wait): grouped launches of <= 8 problems + one slab reduce each, on the side stream when enabled.
int SmdEngine::flush_grouped_wgrads(hipStream_t st, bool on_caller_stream) {
  if (deferred_wgrads_.empty()) return 0;
  hipStream_t ls = st;
  float* slab = W.tn_slab;
  if (side_wgrad && side_ && !on_caller_stream) {
    hipEvent_t ev = take_event();
    SMD_ARG_CHECK(ev, "flush_grouped_wgrads: cannot create an event");
    hipError_t e = hipEventRecord(ev, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(side_, ev, 0);
    if (e != hipSuccess) { smd_set_error("flush_grouped_wgrads: event: %s", hipGetErrorString(e)); return (int)e; }
    ls = side_;
    slab = W.tn_slab_side;
    side_pending_ = true;
  }
  for (TnLaunch& t : deferred_wgrads_) t.slab = slab;
  // a grouped launch shares one contraction length: batch-row problems (FiLM generators) and token-row problems apart
  std::stable_sort(deferred_wgrads_.begin(), deferred_wgrads_.end(),
                   [](const TnLaunch& x, const TnLaunch& y) { return x.Mrows < y.Mrows; });
  int rc = 0;
  for (size_t i = 0; i < deferred_wgrads_.size() && rc == 0;) {
    size_t j = i;
    while (j < deferred_wgrads_.size() && deferred_wgrads_[j].Mrows == deferred_wgrads_[i].Mrows) ++j;
    rc = launch_gemm_tn_grouped(deferred_wgrads_.data() + i, (int)(j - i), ls);
    i = j;
  }
  deferred_wgrads_.clear();
  return rc;
}

int SmdEngine::join_side(hipStream_t st) {
  if (!pending256_.empty()) RC(flush_pending256(st));      // an incomplete group of 256x256 problems
  if (!side_pending_) { next_event_ = 0; return 0; }
  hipEvent_t ev = take_event();
  SMD_ARG_CHECK(ev, "join_side: cannot create an event");
  hipError_t e = hipEventRecord(ev, side_);
  if (e == hipSuccess) e = hipStreamWaitEvent(st, ev, 0);
  if (e != hipSuccess) { smd_set_error("join_side: %s", hipGetErrorString(e)); return (int)e; }
  side_pending_ = false;
  next_event_ = 0;
  return 0;
}

void SmdEngine::build_layout() {
  const int C = d_.data_channels, E = d_.embed_channels, M = d_.mlp_dims, F = d_.film_channels;
  Cp_ = round_up(C, 64);
  int64_t off = 0, woff = 0;
  auto add_dense = [&](const std::string& name, DenseP& p, int K, int N) {
    p.K = K; p.N = N; p.Kp = round_up(K, 64); p.Np = round_up(N, 64);
    p.w_off = off; tensors_.push_back({name + ".kernel", off, K, N}); off += (int64_t)K * N;
    p.b_off = off; tensors_.push_back({name + ".bias", off, N, 0}); off += N;
    p.W_off = woff; woff += (int64_t)K * p.Np;
    p.Wt_off = woff; woff += (int64_t)N * p.Kp;
    woff = (woff + 127) / 128 * 128;
  };
  auto add_ln = [&](const std::string& name, LnP& p, int D) {
    p.D = D;
    p.g_off = off; tensors_.push_back({name + ".scale", off, D, 0}); off += D;
    p.b_off = off; tensors_.push_back({name + ".bias", off, D, 0}); off += D;
  };
  auto add_block = [&](int k) {
    FilmResP& b = blk_[k];
    const std::string f = "film." + std::to_string(k), r = "res." + std::to_string(k);
    add_dense(f + ".fc1", b.f1, F, 4 * F);
    add_dense(f + ".fc2", b.f2, 4 * F, 4 * F);
    add_dense(f + ".ss", b.ss, 4 * F, 2 * M);
    add_ln(r + ".ln1", b.ln1, M);
    add_dense(r + ".fc1", b.r1, M, M);
    add_ln(r + ".ln2", b.ln2, M);
    add_dense(r + ".fc2", b.r2, M, M);
  };
  blk_.resize(nblocks());
  if (d_.arch == 0) {
    enc_.resize(d_.num_layers);
    add_dense("in_proj", in_proj_, C, E);
    for (int l = 0; l < d_.num_layers; ++l) {
      const std::string p = "enc." + std::to_string(l);
      add_ln(p + ".ln1", enc_[l].ln1, E);
      add_dense(p + ".attn.qkv", enc_[l].qkv, E, 3 * E);
      add_dense(p + ".attn.out", enc_[l].out, E, E);
      add_ln(p + ".ln2", enc_[l].ln2, E);
      add_dense(p + ".mlp.fc1", enc_[l].fc1, E, M);
      add_dense(p + ".mlp.fc2", enc_[l].fc2, M, E);
    }
    head_off_ = off;
    add_ln("ln_f", ln_f_, E);
    add_dense("up", up_, E, M);
  } else {
    add_dense("in_proj", in_proj_, C, M);
    head_off_ = off;
  }
  for (int k = 0; k < nblocks(); ++k) add_block(k);
  add_ln("ln_o", ln_o_, M);
  add_dense("out_proj", out_proj_, M, C);
  n_params_ = off;
  n_wpack_ = woff;
  all_dense_.push_back(&in_proj_);
  for (auto& e : enc_) { all_dense_.push_back(&e.qkv); all_dense_.push_back(&e.out); all_dense_.push_back(&e.fc1); all_dense_.push_back(&e.fc2); }
  if (d_.arch == 0) all_dense_.push_back(&up_);
  for (auto& b : blk_) { all_dense_.push_back(&b.f1); all_dense_.push_back(&b.f2); all_dense_.push_back(&b.ss); all_dense_.push_back(&b.r1); all_dense_.push_back(&b.r2); }
  all_dense_.push_back(&out_proj_);
  build_opt_tables();
}

// Work list of the fused optimiser sweep: every Dense kernel as 64 x 64 tiles, everything between two kernels (biases,
// LayerNorm scale / bias) as flat runs of 1024 elements; one table per slice so the two slices can go to different streams.
void SmdEngine::build_opt_tables() {
  for (int part = 0; part < 2; ++part) {
    OptTable& t = part == 0 ? opt_stem_ : opt_head_;
    t = OptTable();
    const int64_t lo = part == 0 ? 0 : head_off_, hi = part == 0 ? head_off_ : n_params_;
    uint32_t blk = 0;
    std::vector<std::pair<int64_t, int64_t>> dense_ranges;
    for (const DenseP* p : all_dense_) {
      if (p->w_off < lo || p->w_off >= hi) continue;
      // a very deep DenseDDPM: three-pass fallback.  Every user of the tables (optimizer_step, loss_backward's early norm) tests
      // opt_fused_ok_ first, so the tables may stay half built; the flat runs are counted by add_flat below
      if (t.n_dense == SMD_OPT_DENSE_MAX) { opt_fused_ok_ = false; return; }
      OptDense& e = t.d[t.n_dense++];
      e.w_off = (uint32_t)p->w_off; e.K = (uint32_t)p->K; e.N = (uint32_t)p->N;
      e.W_off = (uint32_t)p->W_off; e.ldw = (uint32_t)p->Np; e.Wt_off = (uint32_t)p->Wt_off; e.ldwt = (uint32_t)p->Kp;
      e.blk_start = blk;
      blk += (uint32_t)(((p->K + 63) / 64) * ((p->N + 63) / 64));
      dense_ranges.emplace_back(p->w_off, p->w_off + (int64_t)p->K * p->N);
    }
    t.flat_blk0 = blk;
    int64_t cur = lo;
    auto add_flat = [&](int64_t a, int64_t b) {
      if (b <= a) return;
      if (t.n_flat == SMD_OPT_FLAT_MAX) { opt_fused_ok_ = false; return; }
      OptFlat& f = t.f[t.n_flat++];
      f.off = (uint32_t)a; f.len = (uint32_t)(b - a); f.blk_start = blk;
      blk += (uint32_t)((b - a + 1023) / 1024);
    };
    for (auto& r : dense_ranges) { add_flat(cur, r.first); cur = r.second; }      // all_dense_ is in parameter order
    add_flat(cur, hi);
    t.total_blocks = blk;
  }
}

// ------------------------------------------------------------------ workspace planner
namespace {
struct Carver {
  char* base;
  int64_t off = 0;
  template <typename T> T* take(size_t elems) {
    off = (off + 255) / 256 * 256;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += (int64_t)(elems * sizeof(T));
    return p;
  }
};
}  // namespace

int64_t SmdEngine::plan(void* base, int batch, int training, Work* w) const {
  const int S = d_.seq_len, C = d_.data_channels, E = d_.embed_channels, M = d_.mlp_dims, F = d_.film_channels;
  const size_t R = (size_t)batch * S, B = (size_t)batch;
  const int L = d_.arch == 0 ? d_.num_layers : 0, K = nblocks();
  const int nl = training ? L : (L > 0 ? 1 : 0);   // saved activations per layer only when training
  const int nk = training ? K : 1;
  Carver c{reinterpret_cast<char*>(base)};
  Work t;
  t.zero_page = c.take<bf16_t>(128);
  t.step_arrive = c.take<unsigned>(64);               // arrival counter of the fused reverse step (zeroed with the workspace)
  if (d_.arch == 0) {
    t.mlp_part = c.take<float>((size_t)4 * R * E);
  }
  t.x_bf16 = c.take<bf16_t>(R * Cp_);
  t.pe = c.take<float>((size_t)S * E);
  t.pred = c.take<float>(R * C);
  t.s = c.take<float>(B);
  t.h.resize(nl); t.h_mid.resize(nl); t.a1.resize(nl); t.qkv.resize(nl); t.o.resize(nl);
  t.a2.resize(nl); t.z1.resize(nl); t.u.resize(nl);
  for (int l = 0; l < nl; ++l) {
    t.h[l] = c.take<float>(R * E);
    t.h_mid[l] = training ? c.take<float>(R * E) : t.h[l];      // inference: residual stream in place
    t.a1[l] = c.take<bf16_t>(R * E);
    t.qkv[l] = c.take<bf16_t>(R * 3 * E);
    t.o[l] = c.take<bf16_t>(R * E);
    t.a2[l] = training ? c.take<bf16_t>(R * E) : t.a1[l];
    t.z1[l] = training ? c.take<bf16_t>(R * M) : nullptr;
    t.u[l] = c.take<bf16_t>(R * M);
  }
  if (L > 0) {
    t.h_last = training ? c.take<float>(R * E) : t.h[0];
    t.af = c.take<bf16_t>(R * E);
  }
  t.y.resize(training ? K + 1 : 1);
  for (auto& p : t.y) p = c.take<float>(R * M);
  t.ya1.resize(nk); t.o1.resize(nk); t.ya2.resize(nk);
  t.zf1.resize(nk); t.f1.resize(nk); t.p.resize(nk); t.ss.resize(nk);
  for (int k = 0; k < nk; ++k) {
    t.ya1[k] = c.take<bf16_t>(R * M);
    t.o1[k] = c.take<bf16_t>(R * M);
    t.ya2[k] = training ? c.take<bf16_t>(R * M) : t.ya1[k];
    t.zf1[k] = training ? c.take<bf16_t>(B * 4 * F) : nullptr;
    t.f1[k] = c.take<bf16_t>(B * 4 * F);
    t.p[k] = c.take<bf16_t>(B * 4 * F);
    t.ss[k] = c.take<float>(B * 2 * M);
  }
  if (fp8) {
    t.ya1_f8.resize(nk); t.ya2_f8.resize(nk); t.sa1.resize(nk); t.sa2.resize(nk);
    for (int k = 0; k < nk; ++k) {
      t.ya1_f8[k] = c.take<unsigned char>(R * M);
      t.ya2_f8[k] = training ? c.take<unsigned char>(R * M) : t.ya1_f8[k];
      t.sa1[k] = c.take<uint32_t>(R);
      t.sa2[k] = training ? c.take<uint32_t>(R) : t.sa1[k];
    }
    t.w8 = c.take<unsigned char>((size_t)K * 2 * M * M);
    t.w8s = c.take<uint32_t>((size_t)K * 2 * M);
    if (training) {
      t.w8d = c.take<unsigned char>((size_t)K * 2 * M * M);
      t.w8ds = c.take<uint32_t>((size_t)K * 2 * M);
      t.dy8 = c.take<unsigned char>(R * M);
      t.sdy = c.take<uint32_t>(R);
    }
  }
  t.ao = training ? c.take<bf16_t>(R * M) : t.ya1[0];
  t.emb = c.take<bf16_t>(B * F);
  if (training) {
    t.eps = c.take<float>(R * C);
    t.loss = c.take<float>(B);
    t.dpred = c.take<bf16_t>(R * Cp_);
    t.dy = c.take<float>(R * M);
    t.dyb.resize(K + 1);
    for (auto& p : t.dyb) p = c.take<bf16_t>(R * M);
    t.dA_M = c.take<bf16_t>(R * M);
    t.do1.resize(K);
    for (auto& p : t.do1) p = c.take<bf16_t>(R * M);
    t.dss.resize(K);
    for (auto& p : t.dss) p = c.take<float>(B * 2 * M);
    t.dss_bf16.resize(K); t.dp.resize(K); t.df1.resize(K);      // per block: their wgrads run late (side stream)
    for (int k = 0; k < K; ++k) {
      t.dss_bf16[k] = c.take<bf16_t>(B * 2 * M);
      t.dp[k] = c.take<bf16_t>(B * 4 * F);
      t.df1[k] = c.take<bf16_t>(B * 4 * F);
    }
    if (L > 0) {
      t.dh = c.take<float>(R * E);
      t.dhb.resize(2 * L + 1);
      for (auto& p : t.dhb) p = c.take<bf16_t>(R * E);
      t.dA_E = c.take<bf16_t>(R * E);
      t.dqkv.resize(L);
      for (auto& p : t.dqkv) p = c.take<bf16_t>(R * 3 * E);
      t.do_ = c.take<bf16_t>(R * E);
      t.dz1.resize(L);
      for (auto& p : t.dz1) p = c.take<bf16_t>(R * M);
    }
    {   // one partial slot per LayerNorm backward (their dgamma/dbeta reductions are batched at the end)
      const size_t groups = S >= 32 ? (R + 31) / 32 : R;
      t.ln_partial_elems = (size_t)(2 * K + 1) * groups * 2 * M + (size_t)(L > 0 ? 2 * L + 1 : 0) * groups * 2 * E + 2 * (size_t)M;
    }
    t.ln_partial = c.take<float>(t.ln_partial_elems);
    t.tn_slab_elems = gemm_tn_slab_elems();
    t.tn_slab = c.take<float>(t.tn_slab_elems);
    t.tn_slab_side = c.take<float>(t.tn_slab_elems);
    t.norm_partial = c.take<float>(SMD_NORM_SLOTS);
    t.opt_consts = c.take<float>(8);
    const size_t Mp = (R + 63) / 64 * 64;
    t.tn_scratch_elems = tr_path ? 0 : (size_t)2 * (2 * M) * Mp;
    t.tn_scratch = t.tn_scratch_elems ? c.take<bf16_t>(t.tn_scratch_elems) : nullptr;
  }
  if (w) *w = t;
  return (c.off + 255) / 256 * 256;
}

int64_t SmdEngine::workspace_bytes(int batch, int training) const { return plan(nullptr, batch, training, nullptr); }

// ------------------------------------------------------------------ binding
int SmdEngine::bind_params(float* params, bf16_t* wpack) {
  SMD_ARG_CHECK(params && wpack, "bind_params: null pointer");
  params_ = params; wpack_ = wpack;
  return 0;
}
int SmdEngine::bind_train(float* grads, float* m, float* v, float* ema, uint32_t* step_ptr, float* metrics) {
  SMD_ARG_CHECK(grads && m && v && step_ptr && metrics, "bind_train: null pointer");
  grads_ = grads; m_ = m; v_ = v; ema_ = ema; step_ptr_ = step_ptr; metrics_ = metrics;
  return 0;
}
int SmdEngine::bind_workspace(void* ws, int64_t bytes, int batch, int training, hipStream_t st) {
  SMD_ARG_CHECK(ws && batch > 0, "bind_workspace: null workspace or batch=%d", batch);
  const int64_t need = plan(nullptr, batch, training, nullptr);
  SMD_ARG_CHECK(bytes >= need, "bind_workspace: %lld bytes given, %lld needed", (long long)bytes, (long long)need);
  RC(join_update(st));                     // a deferred update reads the optimiser constants of the OLD workspace
  plan(ws, batch, training, &W);
  batch_ = batch; training_ = training;
  w8_dirty_ = true;
  hipError_t e = hipMemsetAsync(ws, 0, (size_t)need, st);   // zero pads / zero page / padded operand columns
  if (e != hipSuccess) { smd_set_error("bind_workspace: memset: %s", hipGetErrorString(e)); return (int)e; }
  if (d_.arch == 0) RC(launch_pos_encoding(W.pe, d_.seq_len, d_.embed_channels, st));
  return 0;
}
int SmdEngine::bind_schedule(const float* coef, const float* sqrt_ap, const float* alphas_prod_ext,
                             float* film_tables) {
  SMD_ARG_CHECK(coef && sqrt_ap && alphas_prod_ext, "bind_schedule: null pointer");
  coef_ = coef; sqrt_ap_ = sqrt_ap; alphas_prod_ext_ = alphas_prod_ext; film_tables_ = film_tables;
  return 0;
}

int SmdEngine::grad_bucket(int b, int64_t* off, int64_t* len) const {
  SMD_ARG_CHECK(b >= 0 && b < num_grad_buckets() && off && len, "grad_bucket: index %d of %d", b, num_grad_buckets());
  if (d_.arch != 0) { *off = 0; *len = head_off_; return 0; }
  const int L = d_.num_layers, l = L - 1 - b;                  // bucket b = encoder layer L-1-b; the last one also holds in_proj
  const int64_t lo = l == 0 ? 0 : enc_[l].ln1.g_off;
  const int64_t hi = l + 1 < L ? enc_[l + 1].ln1.g_off : head_off_;
  *off = lo; *len = hi - lo;
  return 0;
}

int SmdEngine::wait_grad_bucket(int b, hipStream_t s) {
  SMD_ARG_CHECK(b >= 0 && b < num_grad_buckets(), "wait_grad_bucket: index %d of %d", b, num_grad_buckets());
  if (b >= buckets_recorded_) {
    // no per-layer event for this bucket (the last bucket; or an option combination under which the stem backward records
    // none, e.g. group_wgrad != 2): it is final when the whole backward is -- the event loss_backward leaves behind it
    if (!stem_done_ev_ || !stem_done_valid_) return 0;           // nothing recorded: the caller orders the stream itself
    hipError_t e0 = hipStreamWaitEvent(s, stem_done_ev_, 0);
    if (e0 != hipSuccess) { smd_set_error("wait_grad_bucket: %s", hipGetErrorString(e0)); return (int)e0; }
    return 0;
  }
  hipError_t e = hipStreamWaitEvent(s, bucket_ev_[b], 0);
  if (e != hipSuccess) { smd_set_error("wait_grad_bucket: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

int SmdEngine::join_update(hipStream_t st) {
  if (!head_pending_) return 0;
  hipError_t e = hipStreamWaitEvent(st, head_done_ev_, 0);
  if (e != hipSuccess) { smd_set_error("join_update: %s", hipGetErrorString(e)); return (int)e; }
  head_pending_ = false;
  return 0;
}

int SmdEngine::refresh_weights(hipStream_t st) {
  SMD_ARG_CHECK(params_ && wpack_, "refresh_weights: parameters not bound");
  RC(join_update(st));
  w8_dirty_ = true;
  size_t i = 0;
  while (i < all_dense_.size()) {                 // one launch per <= SMD_RECAST_MAX weights (normally one)
    RecastTable t;
    t.n = 0;
    uint32_t tiles = 0;
    for (; i < all_dense_.size() && t.n < SMD_RECAST_MAX; ++i) {
      const DenseP* p = all_dense_[i];
      RecastEntry& e = t.e[t.n++];
      e.w_off = (uint32_t)p->w_off; e.K = (uint32_t)p->K; e.N = (uint32_t)p->N;
      e.W_off = (uint32_t)p->W_off; e.ldw = (uint32_t)p->Np; e.Wt_off = (uint32_t)p->Wt_off; e.ldwt = (uint32_t)p->Kp;
      e.tile_start = tiles;
      tiles += (uint32_t)(((p->K + 63) / 64) * ((p->N + 63) / 64));
    }
    RC(launch_recast_all(params_, wpack_, t, (int)tiles, st));
  }
  return 0;
}

// ------------------------------------------------------------------ dense helpers
int SmdEngine::dense_fwd(const DenseP& p, const bf16_t* A, int lda, int M, GemmEpilogue ep, hipStream_t st) {
  ep.bias = P(p.b_off);
  if (nt256_min_tiles > 0 && gemm_nt256_eligible(M, p.N, p.Kp, ep, nt256_min_tiles))
    return launch_gemm_nt256(A, lda, wpack_ + p.Wt_off, p.Kp, M, p.N, p.Kp, ep, st);
  return launch_gemm_nt(A, lda, wpack_ + p.Wt_off, p.Kp, M, p.N, p.Kp, ep, st);
}

int SmdEngine::dense_bwd(const DenseP& p, const bf16_t* X, int ldx, const bf16_t* dY, int ldy, int M, bf16_t* dX,
                         int ld_dx, const bf16_t* aux, int ld_aux, int aux_mode, hipStream_t st, bool allow_side) {
  // dW = X^T dY and db = colsum(dY) (side stream when allowed) ; dX = dY W^T (* act'(aux)) on the main chain
  RC(wgrad(p, X, ldx, dY, ldy, M, allow_side, st));
  if (dX) {
    GemmEpilogue ep;
    ep.out_bf16 = dX; ep.ld_outb = ld_dx;
    ep.aux = aux; ep.ld_aux = ld_aux; ep.aux_mode = aux_mode;
    RC(launch_gemm_nt(dY, ldy, wpack_ + p.W_off, p.Np, M, p.K, p.Np, ep, st));
  }
  return 0;
}

// ------------------------------------------------------------------ forward
// part 0: the whole network; 1: the stem only (x_bf16 -> trunk input: in_proj, encoder layers, final norm, up projection --
// nothing in it depends on the noise level); 2: the output stage only (DenseResBlocks with their FiLM rows, output norm +
// Dense).  The split passes serve the sampler's software pipeline (sample_step): one chain's stem beside another's head.
int SmdEngine::run_network(const int* t_ptr, hipStream_t st, int part) {
  SMD_ARG_CHECK(params_ && wpack_ && batch_ > 0, "run_network: engine not bound");
  SMD_ARG_CHECK(part == 0 || (part >= 1 && part <= 2 && !training_ && t_ptr), "run_network: split passes belong to the table-driven sampler");
  const int S = d_.seq_len, C = d_.data_channels, E = d_.embed_channels, M = d_.mlp_dims, F = d_.film_channels;
  const int R = rows(), B = batch_, K = nblocks();
  const bool tr = training_ != 0;
  SMD_ARG_CHECK(!(tr && t_ptr), "run_network: table-driven FiLM is inference only");
  SMD_ARG_CHECK(!t_ptr || film_tables_, "run_network: sampler tables not bound");

  // Training: the FiLM generators of all blocks depend on the noise levels only and are three launch-latency-bound GEMMs
  // of 256 rows per block (9 us each for a few MFLOP).  With the side stream on they run THERE, underneath the encoder
  // layers, and the first DenseResBlock waits for them (film_side_fwd; the inference sampler reads tables instead).
  hipEvent_t film_ready = nullptr;
  if (tr && !t_ptr && film_side_fwd && side_wgrad && side_ && d_.arch == 0) {
    RC(launch_noise_embed(W.s, B, F, W.emb, F, st));
    hipEvent_t ev = take_event();
    film_ready = take_event();
    SMD_ARG_CHECK(ev && film_ready, "run_network: cannot create an event");
    hipError_t e = hipEventRecord(ev, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(side_, ev, 0);
    if (e != hipSuccess) { smd_set_error("run_network: event: %s", hipGetErrorString(e)); return (int)e; }
    side_pending_ = true;
    for (int k = 0; k < K; ++k) {
      const FilmResP& b = blk_[k];
      { GemmEpilogue ep; ep.act = SMD_ACT_SWISH; ep.out_bf16 = W.f1[k]; ep.ld_outb = 4 * F; ep.pre_bf16 = W.zf1[k]; ep.ld_pre = 4 * F;
        RC(dense_fwd(b.f1, W.emb, F, B, ep, side_)); }
      { GemmEpilogue ep; ep.out_bf16 = W.p[k]; ep.ld_outb = 4 * F; RC(dense_fwd(b.f2, W.f1[k], 4 * F, B, ep, side_)); }
      { GemmEpilogue ep; ep.out_f32 = W.ss[k]; ep.ld_out = 2 * M; RC(dense_fwd(b.ss, W.p[k], 4 * F, B, ep, side_)); }
    }
    e = hipEventRecord(film_ready, side_);
    if (e != hipSuccess) { smd_set_error("run_network: event: %s", hipGetErrorString(e)); return (int)e; }
  }

  float* y0 = W.y[0];
  // bf16 trunk: inference (trunk_bf16 >= 1) and training (trunk_bf16 == 2); the same buffers, half of each
  const bool tb = trunk_bf16_on();
  auto ybk = [&](int k) { return reinterpret_cast<bf16_t*>(W.y[tr ? k : 0]); };
  bf16_t* yb = ybk(0);
  if (part == 2) {
    // output stage only: the trunk input is where an earlier stem pass of this handle left it
  } else if (d_.arch == 0) {
    {  // in_proj + positional encoding (models/ncsn.py:152-157)
      GemmEpilogue ep;
      ep.res_f32 = W.pe; ep.ld_res = E; ep.res_row_mod = S;
      ep.out_f32 = W.h[0]; ep.ld_out = E;
      RC(dense_fwd(in_proj_, W.x_bf16, Cp_, R, ep, st));
    }
    // Hidden-split MLP dataflow (inference): the attention kernel of layer l sums the four partial tiles the MLP of
    // layer l-1 left, and emits a2 = ln2(h_mid) for its own MLP; the final norm sums the last layer's tiles.
    // training needs groups of 4 samples (the backward kernel's shape) and the fused attention backward
    const bool hs = mlp_hs && fused_encoder && S == 32 && E == 128 && mlp_hs_samples_per_group(R, M) > 0 &&
                    (!tr || (R % 128 == 0 && fused_encoder == 1 && tr_path));
    if (tr) hs_train_ = hs;
    bool parts_pending = false;
    for (int l = 0; l < d_.num_layers; ++l) {   // models/ncsn.py:158-168
      const int i = tr ? l : 0;
      float* h_in = W.h[i];
      float* h_mid = W.h_mid[i];
      float* h_out = tr ? (l + 1 < d_.num_layers ? W.h[l + 1] : W.h_last) : W.h[0];
      const EncLayerP& p = enc_[l];
      LnArgs ln;
      ln.rows = R; ln.D = E;
      const bool layer_hs = hs && p.qkv.Kp == E && p.out.Kp == E && p.fc1.Kp == E && p.fc2.Kp == M;
      if (fused_encoder && S == 32 && E == 128 && p.qkv.Kp == E && p.out.Kp == E) {
        // LN1 + QKV + softmax(q k^T) v + out_proj + residual in one launch (saved activations are small: also training)
        AttnBlockExtra ex;
        if (parts_pending) { ex.h_parts = W.mlp_part; ex.part_stride = (size_t)R * E; if (tr) ex.h_comb = h_in; }
        if (layer_hs) { ex.gamma2 = P(p.ln2.g_off); ex.beta2 = P(p.ln2.b_off); ex.a2_out = W.a2[i]; }
        // the in-place inference stream: with a partial-sum input the kernel never reads h_in, h_mid may be h[0]
        RC(launch_attn_block_fwd(parts_pending ? nullptr : h_in, h_mid, R, P(p.ln1.g_off), P(p.ln1.b_off), wpack_ + p.qkv.Wt_off,
                                 P(p.qkv.b_off), wpack_ + p.out.Wt_off, P(p.out.b_off), d_.num_heads, tr ? W.a1[i] : nullptr,
                                 tr ? W.qkv[i] : nullptr, tr ? W.o[i] : nullptr, st, &ex));
        parts_pending = false;
      } else {
        ln.x = h_in; ln.gamma = P(p.ln1.g_off); ln.beta = P(p.ln1.b_off); ln.out = W.a1[i];
        RC(launch_layernorm_fwd(ln, st));
        { GemmEpilogue ep; ep.out_bf16 = W.qkv[i]; ep.ld_outb = 3 * E; RC(dense_fwd(p.qkv, W.a1[i], E, R, ep, st)); }
        RC(launch_attention_fwd(W.qkv[i], W.o[i], B, S, E, d_.num_heads, st));
        { GemmEpilogue ep; ep.res_f32 = h_in; ep.ld_res = E; ep.out_f32 = h_mid; ep.ld_out = E;
          RC(dense_fwd(p.out, W.o[i], E, R, ep, st)); }
      }
      if (layer_hs) {
        RC(launch_mlp_block_fwd_hs(W.a2[i], h_mid, R, wpack_ + p.fc1.Wt_off, P(p.fc1.b_off), wpack_ + p.fc2.Wt_off,
                                   P(p.fc2.b_off), M, W.mlp_part, st));
        parts_pending = true;
      } else if (fused_encoder && (!tr || fused_encoder == 2) && S == 32 && E == 128 && M % 128 == 0 && p.fc1.Kp == E && p.fc2.Kp == M) {
        // LN2 + fc1 + GELU + fc2 + residual in one launch; the 2048-wide hidden stays in registers.  Inference
        // only by default: with the three saved activations the fused kernel is store-bound (8-byte stores
        // from the MFMA C layout) and no faster than the separate GEMMs (option fused_encoder = 2 forces it).
        RC(launch_mlp_block_fwd(h_mid, h_out, R, P(p.ln2.g_off), P(p.ln2.b_off), wpack_ + p.fc1.Wt_off, P(p.fc1.b_off),
                                wpack_ + p.fc2.Wt_off, P(p.fc2.b_off), M, tr ? W.a2[i] : nullptr, tr ? W.z1[i] : nullptr,
                                tr ? W.u[i] : nullptr, st));
      } else {
        ln.x = h_mid; ln.gamma = P(p.ln2.g_off); ln.beta = P(p.ln2.b_off); ln.out = W.a2[i];
        RC(launch_layernorm_fwd(ln, st));
        { GemmEpilogue ep; ep.act = SMD_ACT_GELU; ep.out_bf16 = W.u[i]; ep.ld_outb = M;
          if (tr) { ep.pre_bf16 = W.z1[i]; ep.ld_pre = M; }
          RC(dense_fwd(p.fc1, W.a2[i], E, R, ep, st)); }
        { GemmEpilogue ep; ep.res_f32 = h_mid; ep.ld_res = E; ep.out_f32 = h_out; ep.ld_out = E;
          RC(dense_fwd(p.fc2, W.u[i], M, R, ep, st)); }
      }
    }
    RC(join_update(st));     // a deferred output-stage update of the previous step (opt_overlap): everything below reads its parameters
    {  // models/ncsn.py:170-171
      if (parts_pending) {
        RC(launch_ln128_parts(W.mlp_part, (size_t)R * E, R, P(ln_f_.g_off), P(ln_f_.b_off), tr ? W.h_last : nullptr, W.af, st));
      } else {
        LnArgs ln;
        ln.x = W.h_last; ln.rows = R; ln.D = E; ln.gamma = P(ln_f_.g_off); ln.beta = P(ln_f_.b_off); ln.out = W.af;
        RC(launch_layernorm_fwd(ln, st));
      }
      GemmEpilogue ep;
      if (tb) { ep.out_bf16 = yb; ep.ld_outb = M; } else { ep.out_f32 = y0; ep.ld_out = M; }
      RC(dense_fwd(up_, W.af, E, R, ep, st));
    }
  } else {  // DenseDDPM stem, models/ncsn.py:129
    GemmEpilogue ep;
    if (tb) { ep.out_bf16 = yb; ep.ld_outb = M; } else { ep.out_f32 = y0; ep.ld_out = M; }
    RC(dense_fwd(in_proj_, W.x_bf16, Cp_, R, ep, st));
    RC(join_update(st));
  }
  if (part == 1 && sample_split <= 0) return 0;

  // DenseResBlocks (models/shared.py:61-75) each with its own FiLM generator (models/ncsn.py:47-61,
  // 173-175 / 130-132).  Per-sample noise levels generate scale/shift here; the sampler reads the
  // per-timestep tables built by prepare_sampler() instead.
  if (!t_ptr && !film_ready) RC(launch_noise_embed(W.s, B, F, W.emb, F, st));
  if (film_ready) {
    hipError_t e = hipStreamWaitEvent(st, film_ready, 0);
    if (e != hipSuccess) { smd_set_error("run_network: event: %s", hipGetErrorString(e)); return (int)e; }
  }
  const bool f8 = fp8 && W.w8 && R % 256 == 0 && M % 256 == 0 && (M == 1024 || M == 2048);
  if (f8 && w8_dirty_) {           // e4m3 copies of the ResBlock weights (per output row), once per weight refresh
    for (int k = 0; k < K; ++k) {
      const FilmResP& b = blk_[k];
      RC(launch_quantize_rows_e4m3(wpack_ + b.r1.Wt_off, b.r1.Kp, M, M, W.w8 + (size_t)k * 2 * M * M, W.w8s + (size_t)k * 2 * M, st));
      RC(launch_quantize_rows_e4m3(wpack_ + b.r2.Wt_off, b.r2.Kp, M, M, W.w8 + (size_t)k * 2 * M * M + (size_t)M * M,
                                   W.w8s + (size_t)k * 2 * M + M, st));
      if (tr && fp8_dgrad && W.w8d) {      // dgrad layout W[in][out]: one scale per input feature row
        RC(launch_quantize_rows_e4m3(wpack_ + b.r1.W_off, b.r1.Np, M, M, W.w8d + (size_t)k * 2 * M * M, W.w8ds + (size_t)k * 2 * M, st));
        RC(launch_quantize_rows_e4m3(wpack_ + b.r2.W_off, b.r2.Np, M, M, W.w8d + (size_t)k * 2 * M * M + (size_t)M * M,
                                     W.w8ds + (size_t)k * 2 * M + M, st));
      }
    }
    w8_dirty_ = false;
  }
  // split passes: part 1 also runs the first `sample_split` half-blocks (LayerNorm + Dense) of the output stage, part 2 the rest
  auto runs_hb = [&](int hb) { return part == 0 || (part == 1 ? hb < sample_split : hb >= sample_split); };
  for (int k = 0; k < K; ++k) {
    const int i = tr ? k : 0;
    const FilmResP& b = blk_[k];
    float* y_in = tr ? W.y[k] : W.y[0];
    float* y_out = tr ? W.y[k + 1] : W.y[0];
    if (!runs_hb(2 * k) && !runs_hb(2 * k + 1)) continue;
    const float* scale;
    const int ld_film = 2 * M;
    if (t_ptr) {
      scale = film_tables_ + (size_t)k * d_.num_timesteps * 2 * M;
    } else if (film_ready) {
      scale = W.ss[i];
    } else {
      { GemmEpilogue ep; ep.act = SMD_ACT_SWISH; ep.out_bf16 = W.f1[i]; ep.ld_outb = 4 * F;
        if (tr) { ep.pre_bf16 = W.zf1[i]; ep.ld_pre = 4 * F; }
        RC(dense_fwd(b.f1, W.emb, F, B, ep, st)); }
      { GemmEpilogue ep; ep.out_bf16 = W.p[i]; ep.ld_outb = 4 * F; RC(dense_fwd(b.f2, W.f1[i], 4 * F, B, ep, st)); }
      { GemmEpilogue ep; ep.out_f32 = W.ss[i]; ep.ld_out = 2 * M; RC(dense_fwd(b.ss, W.p[i], 4 * F, B, ep, st)); }
      scale = W.ss[i];
    }
    LnArgs ln;
    ln.rows = R; ln.D = M; ln.film_scale = scale; ln.film_shift = scale + M; ln.ld_film = ld_film;
    ln.rows_per_sample = S; ln.t_ptr = t_ptr; ln.film_rows = d_.num_timesteps; ln.swish = 1;
    if (f8) {
      // e4m3 forward GEMMs: the LayerNorm writes the A operand as e4m3 + row scales (and, when training, the bf16 copy
      // the weight gradient contracts), the weights were quantised per output row above
      const size_t wo = (size_t)k * 2 * M * M, so = (size_t)k * 2 * M;
      if (runs_hb(2 * k)) {
        if (tb) { ln.x = nullptr; ln.x_bf16 = ybk(k); } else ln.x = y_in;
        ln.gamma = P(b.ln1.g_off); ln.beta = P(b.ln1.b_off);
        ln.out = tr ? W.ya1[i] : nullptr; ln.out_f8 = W.ya1_f8[i]; ln.out_scale = W.sa1[i];
        RC(launch_layernorm_fwd(ln, st));
        { GemmEpilogue ep; ep.bias = P(b.r1.b_off); ep.out_bf16 = W.o1[i]; ep.ld_outb = M;
          RC(launch_gemm_nt256_fp8(W.ya1_f8[i], M, W.sa1[i], W.w8 + wo, M, W.w8s + so, R, M, M, ep, st)); }
      }
      if (runs_hb(2 * k + 1)) {
        ln.x = nullptr; ln.x_bf16 = W.o1[i]; ln.gamma = P(b.ln2.g_off); ln.beta = P(b.ln2.b_off);
        ln.out = tr ? W.ya2[i] : nullptr; ln.out_f8 = W.ya2_f8[i]; ln.out_scale = W.sa2[i];
        RC(launch_layernorm_fwd(ln, st));
        { GemmEpilogue ep; ep.bias = P(b.r2.b_off);
          if (tb) { ep.res_bf16 = ybk(k); ep.ld_resb = M; ep.out_bf16 = ybk(k + 1); ep.ld_outb = M; }
          else { ep.res_f32 = y_in; ep.ld_res = M; ep.out_f32 = y_out; ep.ld_out = M; }
          RC(launch_gemm_nt256_fp8(W.ya2_f8[i], M, W.sa2[i], W.w8 + wo + (size_t)M * M, M, W.w8s + so + M, R, M, M, ep, st)); }
      }
      continue;
    }
    if (runs_hb(2 * k)) {
      if (tb) { ln.x = nullptr; ln.x_bf16 = ybk(k); } else ln.x = y_in;
      ln.gamma = P(b.ln1.g_off); ln.beta = P(b.ln1.b_off); ln.out = W.ya1[i];
      RC(launch_layernorm_fwd(ln, st));
      { GemmEpilogue ep; ep.out_bf16 = W.o1[i]; ep.ld_outb = M; RC(dense_fwd(b.r1, W.ya1[i], M, R, ep, st)); }
    }
    if (runs_hb(2 * k + 1)) {
      ln.x = nullptr; ln.x_bf16 = W.o1[i]; ln.gamma = P(b.ln2.g_off); ln.beta = P(b.ln2.b_off); ln.out = W.ya2[i];
      RC(launch_layernorm_fwd(ln, st));
      { GemmEpilogue ep;
        if (tb) { ep.res_bf16 = ybk(k); ep.ld_resb = M; ep.out_bf16 = ybk(k + 1); ep.ld_outb = M; }
        else { ep.res_f32 = y_in; ep.ld_res = M; ep.out_f32 = y_out; ep.ld_out = M; }
        RC(dense_fwd(b.r2, W.ya2[i], M, R, ep, st)); }
    }
  }
  if (part == 1) return 0;
  {  // models/ncsn.py:177-178 / 133-134
    LnArgs ln;
    if (tb) ln.x_bf16 = ybk(K); else ln.x = tr ? W.y[K] : W.y[0];
    ln.rows = R; ln.D = M; ln.gamma = P(ln_o_.g_off); ln.beta = P(ln_o_.b_off);
    ln.out = W.ao;
    RC(launch_layernorm_fwd(ln, st));
    GemmEpilogue ep; ep.out_f32 = W.pred; ep.ld_out = C;
    RC(dense_fwd(out_proj_, W.ao, M, R, ep, st));
  }
  return 0;
}

int SmdEngine::forward(const float* x, const float* noise_level, float* eps_out, hipStream_t st) {
  SMD_ARG_CHECK(x && noise_level && eps_out, "forward: null pointer");
  SMD_ARG_CHECK(batch_ > 0 && !training_, "forward: bind an inference workspace first");
  const int R = rows(), C = d_.data_channels;
  RC(launch_cast_pad_bf16(x, R, C, W.x_bf16, Cp_, st));
  hipError_t e = hipMemcpyAsync(W.s, noise_level, sizeof(float) * batch_, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { smd_set_error("forward: memcpy: %s", hipGetErrorString(e)); return (int)e; }
  RC(run_network(nullptr, st));
  e = hipMemcpyAsync(eps_out, W.pred, sizeof(float) * (size_t)R * C, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { smd_set_error("forward: memcpy: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

int SmdEngine::forward_level(const float* x, const int* level_ptr, float* eps_out, hipStream_t st) {
  SMD_ARG_CHECK(x && level_ptr, "forward_level: null pointer");
  SMD_ARG_CHECK(batch_ > 0 && !training_ && film_tables_, "forward_level: bind an inference workspace and the sampler tables first");
  const int R = rows(), C = d_.data_channels;
  RC(launch_cast_pad_bf16(x, R, C, W.x_bf16, Cp_, st));
  RC(run_network(level_ptr, st));
  if (!eps_out) return 0;                 // the caller reads the engine's own output buffer (smd_engine_pred)
  hipError_t e = hipMemcpyAsync(eps_out, W.pred, sizeof(float) * (size_t)R * C, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { smd_set_error("forward_level: memcpy: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

// ------------------------------------------------------------------ backward
static LnArgs ln_args(const float* x, const bf16_t* xb, int rows, const LnP& p, float* params) {
  LnArgs a;
  a.x = x; a.x_bf16 = xb; a.rows = rows; a.D = p.D; a.gamma = params + p.g_off; a.beta = params + p.b_off;
  return a;
}

// LayerNorm backward with its own partial slot; the dgamma/dbeta reduction is deferred to flush_ln_reduce()
int SmdEngine::ln_bwd(LnBwdArgs& b, hipStream_t st) {
  const int gr = b.f.film_scale ? b.f.rows_per_sample : 32;
  const size_t need = (size_t)((b.f.rows + gr - 1) / gr) * 2 * b.f.D;
  SMD_ARG_CHECK(ln_slot_off_ + need <= W.ln_partial_elems, "ln_bwd: partial workspace exhausted");
  b.partial = W.ln_partial + ln_slot_off_;
  b.partial_elems = need;
  ln_slot_off_ += need;
  ln_pending_.emplace_back();
  b.deferred = &ln_pending_.back();
  // SMD_DEBUG_SYNC (bit 1: drain the main stream before every LayerNorm backward, 2: drain the side stream, 4: drain
  // after it) exists for tools/det_check.py -- see DESIGN.md section 6, "known issue"
  static const int dbg = getenv("SMD_DEBUG_SYNC") ? atoi(getenv("SMD_DEBUG_SYNC")) : 0;
  if (dbg & 1) (void)hipStreamSynchronize(st);          // debug: main chain drained before the LN backward
  if ((dbg & 2) && side_) (void)hipStreamSynchronize(side_);   // debug: side stream idle during the LN backward
  const int rc = launch_layernorm_bwd(b, st);
  if (dbg & 4) (void)hipStreamSynchronize(st);          // debug: LN backward complete before anything else is enqueued
  return rc;
}
int SmdEngine::flush_ln_reduce(hipStream_t st) {
  if (ln_pending_.empty()) return 0;
  const int rc = launch_ln_bwd_reduce_batched(ln_pending_.data(), (int)ln_pending_.size(), st);
  ln_pending_.clear();
  return rc;
}

int SmdEngine::backward_head(hipStream_t st) {
  deferred_wgrads_.clear();
  ln_slot_off_ = 0;
  ln_pending_.clear();
  ln_pending_.reserve(64);
  const int S = d_.seq_len, E = d_.embed_channels, M = d_.mlp_dims, F = d_.film_channels;
  const int R = rows(), B = batch_, K = nblocks();
  // the 8-wave LayerNorm backward reads the residual gradient as bf16 (M = 2048 only)
  const bool use_bf16_chain = resgrad_bf16 && M == 2048;
  // out_proj (models/ncsn.py:178): X = ao, dY = dpred
  RC(dense_bwd(out_proj_, W.ao, M, W.dpred, Cp_, R, W.dA_M, M, nullptr, 0, SMD_AUX_NONE, st, true));
  // debug snapshots: copy i of the shared dX buffer, stream-ordered behind the dgrad that wrote it
  auto hsnap = [&](int i) {
    if (!dbg_snap_) return;
    const size_t RM2 = (size_t)R * M * 2;
    (void)hipMemcpyAsync(dbg_snap_ + debug_stem_snapshot_bytes() + (size_t)i * RM2, W.dA_M, RM2, hipMemcpyDeviceToDevice, st);
  };
  hsnap(0);
  {
    LnBwdArgs b;
    const bool tbk = trunk_bf16_on();
    b.f = tbk ? ln_args(nullptr, reinterpret_cast<bf16_t*>(W.y[K]), R, ln_o_, params_) : ln_args(W.y[K], nullptr, R, ln_o_, params_);
    b.dout = W.dA_M; b.dx = use_bf16_chain ? nullptr : W.dy; b.dx_bf16 = W.dyb[K];
    b.dgamma = G(ln_o_.g_off); b.dbeta = G(ln_o_.b_off);
    RC(ln_bwd(b, st));
  }
  // fp8 mode: dX = dY W^T of the two 2048 x 2048 layers of a block on e4m3 operands (quantise dY per token row, then the
  // scaled-MFMA GEMM against the e4m3 copy of the dgrad pack); the weight gradient is issued first, as in dense_bwd
  const bool f8d = fp8 && fp8_dgrad && W.w8d && W.dy8 && R % 256 == 0 && M % 256 == 0 && (M == 1024 || M == 2048) && !w8_dirty_;
  auto res_bwd = [&](const DenseP& p, int widx, const bf16_t* X, const bf16_t* dY) -> int {
    if (!f8d) return dense_bwd(p, X, M, dY, M, R, W.dA_M, M, nullptr, 0, SMD_AUX_NONE, st, true);
    RC(wgrad(p, X, M, dY, M, R, true, st));
    RC(launch_quantize_rows_e4m3(dY, M, R, M, W.dy8, W.sdy, st));
    GemmEpilogue ep;
    ep.out_bf16 = W.dA_M; ep.ld_outb = M;
    return launch_gemm_nt256_fp8(W.dy8, M, W.sdy, W.w8d + (size_t)widx * M * M, M, W.w8ds + (size_t)widx * M, R, M, M, ep, st);
  };
  for (int k = K - 1; k >= 0; --k) {
    const FilmResP& p = blk_[k];
    // fc2 of the res block: y[k+1] = ya2 W + b + y[k]
    RC(res_bwd(p.r2, 2 * k + 1, W.ya2[k], W.dyb[k + 1]));
    hsnap(1 + 2 * (K - 1 - k));
    {
      LnBwdArgs b;
      b.f = ln_args(nullptr, W.o1[k], R, p.ln2, params_);
      b.f.film_scale = W.ss[k]; b.f.film_shift = W.ss[k] + M; b.f.ld_film = 2 * M; b.f.rows_per_sample = S;
      b.f.swish = 1;
      b.dout = W.dA_M; b.dx_bf16 = W.do1[k];
      b.dgamma = G(p.ln2.g_off); b.dbeta = G(p.ln2.b_off);
      b.dscale = W.dss[k]; b.dshift = W.dss[k] + M; b.dfilm_accumulate = 0;
      RC(ln_bwd(b, st));
    }
    RC(res_bwd(p.r1, 2 * k, W.ya1[k], W.do1[k]));
    hsnap(2 + 2 * (K - 1 - k));
    {
      LnBwdArgs b;
      b.f = trunk_bf16_on() ? ln_args(nullptr, reinterpret_cast<bf16_t*>(W.y[k]), R, p.ln1, params_)
                            : ln_args(W.y[k], nullptr, R, p.ln1, params_);
      b.f.film_scale = W.ss[k]; b.f.film_shift = W.ss[k] + M; b.f.ld_film = 2 * M; b.f.rows_per_sample = S;
      b.f.swish = 1;
      b.dout = W.dA_M; b.dx_bf16 = W.dyb[k];
      if (use_bf16_chain) b.dres_bf16 = W.dyb[k + 1];          // y[k+1] = y[k] + block(y[k]): dy[k] = dy[k+1] + ...
      else { b.dres = W.dy; b.dx = W.dy; }
      b.dgamma = G(p.ln1.g_off); b.dbeta = G(p.ln1.b_off);
      b.dscale = W.dss[k]; b.dshift = W.dss[k] + M; b.dfilm_accumulate = 1;
      RC(ln_bwd(b, st));
    }
    // FiLM generator (models/ncsn.py:52-61).  Nothing on the main chain consumes its gradients (the noise embedding has
    // no parameters behind it): with the side stream on, the whole chain -- cast, two dgrads, three wgrads -- leaves the
    // main stream behind an event (dscale / dshift of this block are final after the LayerNorm backward above)
    hipStream_t fs = st;
    if (film_side_fwd && film_side && side_wgrad && side_ && tr_path) {     // (the tr_path = 0 fallback shares one scratch)
      hipEvent_t ev = take_event();
      SMD_ARG_CHECK(ev, "backward_head: cannot create an event");
      hipError_t e = hipEventRecord(ev, st);
      if (e == hipSuccess) e = hipStreamWaitEvent(side_, ev, 0);
      if (e != hipSuccess) { smd_set_error("backward_head: event: %s", hipGetErrorString(e)); return (int)e; }
      fs = side_;
      side_pending_ = true;
    }
    RC(launch_cast_pad_bf16(W.dss[k], B, 2 * M, W.dss_bf16[k], 2 * M, fs));
    RC(dense_bwd(p.ss, W.p[k], 4 * F, W.dss_bf16[k], 2 * M, B, W.dp[k], 4 * F, nullptr, 0, SMD_AUX_NONE, fs, film_side != 0));
    RC(dense_bwd(p.f2, W.f1[k], 4 * F, W.dp[k], 4 * F, B, W.df1[k], 4 * F, W.zf1[k], 4 * F, SMD_AUX_SWISH_GRAD, fs, film_side != 0));
    RC(dense_bwd(p.f1, W.emb, F, W.df1[k], 4 * F, B, nullptr, 0, nullptr, 0, SMD_AUX_NONE, fs, film_side != 0));
  }
  if (d_.arch == 0) {
    // up (models/ncsn.py:171) and ln_f (:170)
    RC(dense_bwd(up_, W.af, E, W.dyb[0], M, R, W.dA_E, E, nullptr, 0, SMD_AUX_NONE, st, true));
    LnBwdArgs b;
    b.f = ln_args(W.h_last, nullptr, R, ln_f_, params_);
    b.dout = W.dA_E; b.dx = W.dh; b.dx_bf16 = W.dhb[2 * d_.num_layers];
    b.dgamma = G(ln_f_.g_off); b.dbeta = G(ln_f_.b_off);
    RC(ln_bwd(b, st));
  }
  return 0;
}

int SmdEngine::set_debug_snapshots(void* buf, int64_t bytes) {
  SMD_ARG_CHECK(!buf || (batch_ > 0 && bytes >= debug_snapshot_bytes()), "set_debug_snapshots: %lld bytes given, %lld needed",
                (long long)bytes, (long long)(batch_ > 0 ? debug_snapshot_bytes() : -1));
  dbg_snap_ = reinterpret_cast<char*>(buf);
  return 0;
}

int SmdEngine::debug_tensor(const char* name, int index, const void** ptr, int64_t* rows_out, int64_t* cols_out, int* dtype) const {
  SMD_ARG_CHECK(name && ptr && rows_out && cols_out && dtype, "debug_tensor: null argument");
  SMD_ARG_CHECK(batch_ > 0 && training_, "debug_tensor: bind a training workspace first");
  const int64_t R = rows(), B = batch_, E = d_.embed_channels, M = d_.mlp_dims, F = d_.film_channels;
  const int L = d_.arch == 0 ? d_.num_layers : 0, K = nblocks();
  const std::string n(name);
  auto set = [&](const void* p, int64_t r, int64_t c, int dt) { *ptr = p; *rows_out = r; *cols_out = c; *dtype = dt; return 0; };
  auto layer = [&](int hi) { return index >= 0 && index < hi; };
  if (n == "x_bf16") return set(W.x_bf16, R, Cp_, 1);
  if (n == "pred") return set(W.pred, R, d_.data_channels, 0);
  if (n == "s") return set(W.s, B, 1, 0);
  if (n == "emb") return set(W.emb, B, F, 1);
  if (n == "ao") return set(W.ao, R, M, 1);
  if (n == "af" && L > 0) return set(W.af, R, E, 1);
  if (n == "h_last" && L > 0) return set(W.h_last, R, E, 0);
  if (n == "h" && layer(L)) return set(W.h[index], R, E, 0);
  if (n == "h_mid" && layer(L)) return set(W.h_mid[index], R, E, 0);
  if (n == "a1" && layer(L)) return set(W.a1[index], R, E, 1);
  if (n == "qkv" && layer(L)) return set(W.qkv[index], R, 3 * E, 1);
  if (n == "o" && layer(L)) return set(W.o[index], R, E, 1);
  if (n == "a2" && layer(L)) return set(W.a2[index], R, E, 1);
  if (n == "y" && layer(K + 1)) return set(W.y[index], R, M, trunk_bf16_on() ? 1 : 0);
  if (n == "ya1" && layer(K)) return set(W.ya1[index], R, M, 1);
  if (n == "o1" && layer(K)) return set(W.o1[index], R, M, 1);
  if (n == "ya2" && layer(K)) return set(W.ya2[index], R, M, 1);
  if (n == "f1" && layer(K)) return set(W.f1[index], B, 4 * F, 1);
  if (n == "p" && layer(K)) return set(W.p[index], B, 4 * F, 1);
  if (n == "ss" && layer(K)) return set(W.ss[index], B, 2 * M, 0);
  // gradient activations of the last loss_backward: every dY / X operand of a weight-gradient GEMM has its own slot (the side
  // stream reads them late), so each weight gradient can be re-derived from exactly the operands the engine used
  if (n == "dpred") return set(W.dpred, R, Cp_, 1);
  if (n == "dyb" && layer(K + 1)) return set(W.dyb[index], R, M, 1);
  if (n == "do1" && layer(K)) return set(W.do1[index], R, M, 1);
  if (n == "dss" && layer(K)) return set(W.dss[index], B, 2 * M, 0);
  if (n == "zf1" && layer(K)) return set(W.zf1[index], B, 4 * F, 1);
  if (n == "dss_bf16" && layer(K)) return set(W.dss_bf16[index], B, 2 * M, 1);
  if (n == "dp" && layer(K)) return set(W.dp[index], B, 4 * F, 1);
  if (n == "df1" && layer(K)) return set(W.df1[index], B, 4 * F, 1);
  if (n == "dhb" && layer(2 * L + 1)) return set(W.dhb[index], R, E, 1);
  if (n == "dqkv" && layer(L)) return set(W.dqkv[index], R, 3 * E, 1);
  if (n == "dz1" && layer(L)) return set(W.dz1[index], R, M, 1);
  if (n == "u" && layer(L)) return set(W.u[index], R, M, 1);
  smd_set_error("debug_tensor: unknown tensor '%s'[%d]", name, index);
  return -1;
}

int SmdEngine::backward_stem(hipStream_t st) {
  buckets_recorded_ = 0;
  const int S = d_.seq_len, E = d_.embed_channels, M = d_.mlp_dims;
  const int R = rows(), B = batch_;
  // debug snapshots (set_debug_snapshots): segment `seg` of layer l <- src, stream-ordered behind the kernel that wrote it
  const size_t RE = (size_t)R * E;
  auto snap = [&](int l, int seg, const void* src) {
    if (!dbg_snap_) return;
    static const size_t off[6] = {0, 16, 20, 22, 26, 30}, len[6] = {16, 4, 2, 4, 4, 4};        // in units of R*E bytes
    char* dst = dbg_snap_ + ((size_t)(d_.num_layers - 1 - l) * 34 + off[seg]) * RE;
    (void)hipMemcpyAsync(dst, src, len[seg] * RE, hipMemcpyDeviceToDevice, st);
  };
  if (d_.arch != 0) {
    return dense_bwd(in_proj_, W.x_bf16, Cp_, W.dyb[0], M, R, nullptr, 0, nullptr, 0, SMD_AUX_NONE, st, true);
  }
  // residual-stream gradient versions: dhb[2l+2] enters layer l, dhb[2l+1] after its ln2, dhb[2l] after its ln1
  for (int l = d_.num_layers - 1; l >= 0; --l) {
    const EncLayerP& p = enc_[l];
    bf16_t* dh_in = W.dhb[2 * l + 2];
    bf16_t* dh_mid = W.dhb[2 * l + 1];
    bf16_t* dh_out = W.dhb[2 * l];
    snap(l, 4, W.dh);
    snap(l, 5, W.h_mid[l]);
    // fused_attn_bwd = 2: the two LayerNorm backwards of the layer run inside the attention backward's launch (debug snapshots
    // want the intermediate buffers of the separate kernels: they take the older path)
    const bool attn_fused = fused_encoder && fused_attn_bwd && S == 32 && E == 128 && p.out.Np == E && p.qkv.Np == 3 * E;
    const bool ln_fused = hs_train_ && attn_fused && fused_attn_bwd == 2 && !dbg_snap_;
    float* partial2 = nullptr;
    if (hs_train_) {
      // fused backward with the hidden activations recomputed from a2: writes u and dz1 (wgrad operands) and four
      // partial tiles of da2; the ln2 backward sums them (launch-boundary reduce)
      RC(launch_mlp_block_bwd_hs(W.a2[l], dh_in, R, wpack_ + p.fc1.Wt_off, wpack_ + p.fc2.W_off, wpack_ + p.fc1.W_off, P(p.fc1.b_off),
                                 M, W.u[l], W.dz1[l], W.mlp_part, st));
      snap(l, 0, W.mlp_part);
      RC(wgrad(p.fc2, W.u[l], M, dh_in, E, R, true, st));
      RC(wgrad(p.fc1, W.a2[l], E, W.dz1[l], M, R, true, st));
      const size_t need = (size_t)(R / 32) * 2 * E;
      SMD_ARG_CHECK(ln_slot_off_ + need <= W.ln_partial_elems, "backward_stem: LayerNorm partial workspace exhausted");
      partial2 = W.ln_partial + ln_slot_off_;
      ln_slot_off_ += need;
      if (!ln_fused) RC(launch_ln128_bwd_parts(W.h_mid[l], W.mlp_part, (size_t)R * E, R, P(p.ln2.g_off), W.dh, W.dh, dh_mid, partial2, st));
      LnReduceEntry en;
      en.partial = partial2; en.ngroups = R / 32; en.D = E; en.dgamma = G(p.ln2.g_off); en.dbeta = G(p.ln2.b_off); en.block_start = 0;
      ln_pending_.push_back(en);
      snap(l, 1, W.dh);
    } else {
      // mlp.fc2: h_out = u W2 + b + h_mid ; dz1 = (dh W2^T) * gelu'(z1)
      RC(dense_bwd(p.fc2, W.u[l], M, dh_in, E, R, W.dz1[l], M, W.z1[l], M, SMD_AUX_GELU_GRAD, st, true));
      RC(dense_bwd(p.fc1, W.a2[l], E, W.dz1[l], M, R, W.dA_E, E, nullptr, 0, SMD_AUX_NONE, st, true));
      LnBwdArgs b;
      b.f = ln_args(W.h_mid[l], nullptr, R, p.ln2, params_);
      b.dout = W.dA_E; b.dres = W.dh; b.dx = W.dh; b.dx_bf16 = dh_mid;
      b.dgamma = G(p.ln2.g_off); b.dbeta = G(p.ln2.b_off);
      RC(ln_bwd(b, st));
    }
    if (ln_fused) {
      // LayerNorm-2 backward + out_proj dgrad + attention backward + qkv dgrad + LayerNorm-1 backward in one launch
      const size_t need = (size_t)(R / 32) * 2 * E;
      SMD_ARG_CHECK(ln_slot_off_ + need <= W.ln_partial_elems, "backward_stem: LayerNorm partial workspace exhausted");
      float* partial1 = W.ln_partial + ln_slot_off_;
      ln_slot_off_ += need;
      AttnBwdLnArgs x;
      x.qkv = W.qkv[l]; x.Wo = wpack_ + p.out.W_off; x.Wqkv = wpack_ + p.qkv.W_off; x.dqkv = W.dqkv[l]; x.da1 = nullptr;
      x.h_mid = W.h_mid[l]; x.da2_parts = W.mlp_part; x.part_stride = (size_t)R * E; x.gamma2 = P(p.ln2.g_off); x.dh = W.dh;
      x.dh_mid_out = dh_mid; x.partial2 = partial2; x.h = W.h[l]; x.gamma1 = P(p.ln1.g_off); x.dh_out = dh_out; x.partial1 = partial1;
      RC(launch_attn_block_bwd_ln(x, R, d_.num_heads, st));
      RC(wgrad(p.out, W.o[l], E, dh_mid, E, R, true, st));           // (dh_mid is produced by the launch above)
      RC(wgrad(p.qkv, W.a1[l], E, W.dqkv[l], 3 * E, R, true, st));
      LnReduceEntry en;
      en.partial = partial1; en.ngroups = R / 32; en.D = E; en.dgamma = G(p.ln1.g_off); en.dbeta = G(p.ln1.b_off); en.block_start = 0;
      ln_pending_.push_back(en);
    } else if (attn_fused) {
      // out_proj dgrad + attention backward + qkv dgrad in one launch; the two wgrads stay GEMMs
      RC(wgrad(p.out, W.o[l], E, dh_mid, E, R, true, st));
      RC(launch_attn_block_bwd(dh_mid, W.qkv[l], wpack_ + p.out.W_off, wpack_ + p.qkv.W_off, W.dqkv[l], W.dA_E, R,
                               d_.num_heads, st));
      snap(l, 2, W.dA_E);
      RC(wgrad(p.qkv, W.a1[l], E, W.dqkv[l], 3 * E, R, true, st));
    } else {
      RC(dense_bwd(p.out, W.o[l], E, dh_mid, E, R, W.do_, E, nullptr, 0, SMD_AUX_NONE, st, true));
      RC(launch_attention_bwd(W.qkv[l], W.do_, W.dqkv[l], B, S, E, d_.num_heads, st));
      RC(dense_bwd(p.qkv, W.a1[l], E, W.dqkv[l], 3 * E, R, W.dA_E, E, nullptr, 0, SMD_AUX_NONE, st, true));
    }
    if (!ln_fused) {
      LnBwdArgs b;
      b.f = ln_args(W.h[l], nullptr, R, p.ln1, params_);
      b.dout = W.dA_E; b.dres = W.dh; b.dx = W.dh; b.dx_bf16 = dh_out;
      b.dgamma = G(p.ln1.g_off); b.dbeta = G(p.ln1.b_off);
      RC(ln_bwd(b, st));
      snap(l, 3, W.dh);
    }
    // this layer's four 128-wide wgrads as one side-stream launch; layer 0's wait for in_proj's (the 4-tile in_proj
    // problem alone was a 17 us launch + a reduce of its own at the very end of the step)
    if (group_wgrad == 2 && l > 0) RC(flush_grouped_wgrads(st));
    if (dp_layer_events && group_wgrad == 2 && l > 0) {
      // this layer's parameter gradients are final once its two LayerNorm reductions (main stream) and its grouped weight
      // gradients (side stream) have run: one event behind both
      RC(flush_ln_reduce(st));
      if (!pending256_.empty()) RC(flush_pending256(st));      // a 256-wide encoder wgrad (embed_channels % 256 == 0) parked for grouping
      const int b = d_.num_layers - 1 - l;
      while ((int)bucket_ev_.size() <= b) {
        hipEvent_t ev = nullptr;
        SMD_ARG_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess, "backward_stem: cannot create an event");
        bucket_ev_.push_back(ev);
      }
      hipStream_t es = st;
      if (side_wgrad && side_) {
        hipEvent_t em = take_event();
        SMD_ARG_CHECK(em, "backward_stem: cannot create an event");
        hipError_t e = hipEventRecord(em, st);
        if (e == hipSuccess) e = hipStreamWaitEvent(side_, em, 0);
        if (e != hipSuccess) { smd_set_error("backward_stem: event: %s", hipGetErrorString(e)); return (int)e; }
        es = side_;
        side_pending_ = true;
      }
      hipError_t e = hipEventRecord(bucket_ev_[b], es);
      if (e != hipSuccess) { smd_set_error("backward_stem: event: %s", hipGetErrorString(e)); return (int)e; }
      buckets_recorded_ = b + 1;
    }
  }
  return dense_bwd(in_proj_, W.x_bf16, Cp_, W.dhb[0], E, R, nullptr, 0, nullptr, 0, SMD_AUX_NONE, st, true);
}

int SmdEngine::loss_backward(const float* x0, const int* labels, const float* eps_in, uint32_t seed_lo,
                             uint32_t seed_hi, uint32_t sample_offset, float inv_global_count, int stage,
                             hipStream_t st) {
  SMD_ARG_CHECK(training_ && alphas_prod_ext_, "loss_backward: bind a training workspace and the schedule first");
  SMD_ARG_CHECK(stage >= 0 && stage <= 3, "loss_backward: stage=%d", stage);
  SMD_ARG_CHECK(stage == 3 || grads_, "loss_backward: bind the optimiser state first");
  const int S = d_.seq_len, C = d_.data_channels;
  if (stage == 0 || stage == 1 || stage == 3) {     // 3 = loss only (eval_step, train_ncsn.py:206-221)
    SMD_ARG_CHECK(x0, "loss_backward: null batch");
    hipEvent_t grads_zeroed = nullptr;
    // Every gradient element is WRITTEN (never accumulated into) exactly once per step on the default paths -- weight and
    // bias gradients by their GEMM or its slab reduce, LayerNorm scale / bias by the batched reduce -- so the 102 MB memset
    // is only kept for the fallback paths (grad_memset = 1 forces it; the poison test in tests/test_gpu_bench_config.py
    // fills the buffer with NaN before a step and requires a finite, oracle-matching gradient).
    const bool need_memset = grad_memset == 1 || (grad_memset == 2 && !(tr_path == 1));
    if (stage != 3 && need_memset) {
      // zero the gradient buffer; with the side stream on, the 106 MB memset runs there underneath the forward pass
      // (ordered after everything enqueued so far, i.e. after the previous optimiser step) and the main stream
      // picks its completion up just before the first gradient is written
      hipStream_t ms = st;
      if (side_wgrad && side_) {
        hipEvent_t ev0 = take_event();
        grads_zeroed = take_event();
        SMD_ARG_CHECK(ev0 && grads_zeroed, "loss_backward: cannot create an event");
        hipError_t e0 = hipEventRecord(ev0, st);
        if (e0 == hipSuccess) e0 = hipStreamWaitEvent(side_, ev0, 0);
        if (e0 != hipSuccess) { smd_set_error("loss_backward: %s", hipGetErrorString(e0)); return (int)e0; }
        ms = side_;
        side_pending_ = true;
      }
      hipError_t e = hipMemsetAsync(grads_, 0, sizeof(float) * (size_t)n_params_, ms);
      if (e == hipSuccess && grads_zeroed) e = hipEventRecord(grads_zeroed, side_);
      if (e != hipSuccess) { smd_set_error("loss_backward: memset: %s", hipGetErrorString(e)); return (int)e; }
    }
    QSampleArgs q;
    q.x0 = x0; q.B = batch_; q.S = S; q.C = C; q.Cp = Cp_; q.T = d_.num_timesteps;
    q.alphas_prod_ext = alphas_prod_ext_;
    q.labels = labels; q.eps_in = eps_in; q.key = RngKey{seed_lo, seed_hi};
    q.label_min = label_min; q.alpha_in = used_alphas_; q.dsm = loss_kind;
    SMD_ARG_CHECK(!loss_kind || used_alphas_, "loss_backward: the score-matching loss needs set_used_alphas(used_sigmas)");
    q.step_ptr = step_ptr_; q.sample_offset = sample_offset;
    q.xt_bf16 = W.x_bf16; q.eps_out = W.eps; q.s_out = W.s;
    RC(launch_q_sample(q, st));
    RC(run_network(nullptr, st));
    RC(launch_mse_loss_grad(W.pred, W.eps, batch_, S, C, Cp_, inv_global_count, W.loss, W.dpred, st, loss_kind ? W.s : nullptr));
    if (grads_zeroed) {
      hipError_t e = hipStreamWaitEvent(st, grads_zeroed, 0);
      if (e != hipSuccess) { smd_set_error("loss_backward: %s", hipGetErrorString(e)); return (int)e; }
    }
    if (stage != 3) RC(backward_head(st));
    // opt_overlap bit 1 (single-process step, no all-reduce between this call and the optimiser): the output-stage slice of
    // the gradient is final once its LayerNorm partials are reduced and its weight gradients are on the side stream, so its
    // share of the global-norm partials is reduced THERE, underneath the encoder backward
    const bool early_norm = stage == 0 && (opt_overlap & 2) && opt_fused_ok_ && opt_fused_user_ && side_wgrad && side_ && tr_path && head_off_ < n_params_;
    head_norm_ready_ = false;
    if (stage == 1 || early_norm) RC(flush_ln_reduce(st));       // output-stage gradients must be final before the DP all-reduce
    // the output stage's deferred wgrads (out_proj, up, FiLM generators) go to the side stream now: some of their operands
    // were produced THERE (the FiLM backward chain), and the step's last grouped launch -- which runs on the caller's
    // stream (tail_on_main) -- must only hold problems whose operands the caller's stream produced
    if (stage != 3) RC(flush_grouped_wgrads(st));
    if (early_norm) {
      if (!pending256_.empty()) RC(flush_pending256(st));
      hipEvent_t ev = take_event();
      SMD_ARG_CHECK(ev, "loss_backward: cannot create an event");
      hipError_t e = hipEventRecord(ev, st);
      if (e == hipSuccess) e = hipStreamWaitEvent(side_, ev, 0);
      if (e != hipSuccess) { smd_set_error("loss_backward: event: %s", hipGetErrorString(e)); return (int)e; }
      side_pending_ = true;
      RC(launch_grad_sumsq_slots(grads_ + head_off_, (size_t)(n_params_ - head_off_), W.norm_partial, SMD_NORM_HEAD_SLOTS, side_));
      head_norm_ready_ = true;
    }
  }
  if (stage == 0 || stage == 2) {
    RC(backward_stem(st));
    RC(flush_ln_reduce(st));                       // one launch for every pending LayerNorm dgamma/dbeta
    RC(flush_grouped_wgrads(st, tail_on_main != 0));
  }
  return finish_backward(st, stage == 0 || stage == 2);
}

// The tail both backward entry points share: join the side stream (every gradient is complete on `st` when this returns; after a
// stage-1 call: the output stage's) and, with dp_layer_events on and the stem backward in this call, leave the event that
// smd_engine_wait_grad_bucket falls back to for a bucket without a per-layer event.
int SmdEngine::finish_backward(hipStream_t st, bool stem_ran) {
  stem_done_valid_ = false;
  RC(join_side(st));
  if (dp_layer_events && stem_ran) {
    if (!stem_done_ev_) SMD_ARG_CHECK(hipEventCreateWithFlags(&stem_done_ev_, hipEventDisableTiming) == hipSuccess, "backward: cannot create an event");
    hipError_t e = hipEventRecord(stem_done_ev_, st);
    if (e != hipSuccess) { smd_set_error("backward: event: %s", hipGetErrorString(e)); return (int)e; }
    stem_done_valid_ = true;
  }
  return 0;
}

// The two halves of jax.value_and_grad over an ARBITRARY objective (train_ncsn.py:279-283): model(x, noise_level) in the training
// workspace with every activation the backward needs saved, then the backward pass from d objective / d eps_hat.
int SmdEngine::forward_train(const float* x, const float* noise_level, float* eps_out, hipStream_t st) {
  SMD_ARG_CHECK(x && noise_level, "forward_train: null pointer");
  SMD_ARG_CHECK(training_ && batch_ > 0, "forward_train: bind a training workspace first");
  const int R = rows(), C = d_.data_channels;
  RC(launch_cast_pad_bf16(x, R, C, W.x_bf16, Cp_, st));
  hipError_t e = hipMemcpyAsync(W.s, noise_level, sizeof(float) * batch_, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { smd_set_error("forward_train: memcpy: %s", hipGetErrorString(e)); return (int)e; }
  RC(run_network(nullptr, st));
  if (!eps_out) return 0;
  e = hipMemcpyAsync(eps_out, W.pred, sizeof(float) * (size_t)R * C, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { smd_set_error("forward_train: memcpy: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

int SmdEngine::backward_from(const float* dpred, int stage, hipStream_t st) {
  SMD_ARG_CHECK(training_ && grads_ && batch_ > 0, "backward_from: bind a training workspace and the optimiser state first");
  SMD_ARG_CHECK(stage >= 0 && stage <= 2, "backward_from: stage=%d", stage);
  SMD_ARG_CHECK(stage == 2 || dpred, "backward_from: null gradient");
  stem_done_valid_ = false;          // the event of an EARLIER backward orders nothing about this one (wait_grad_bucket's fallback)
  if (stage != 2) {
    const bool need_memset = grad_memset == 1 || (grad_memset == 2 && !(tr_path == 1));
    if (need_memset) {
      hipError_t e = hipMemsetAsync(grads_, 0, sizeof(float) * (size_t)n_params_, st);
      if (e != hipSuccess) { smd_set_error("backward_from: memset: %s", hipGetErrorString(e)); return (int)e; }
    }
    RC(launch_cast_pad_bf16(dpred, rows(), d_.data_channels, W.dpred, Cp_, st));
    RC(backward_head(st));
    head_norm_ready_ = false;
    if (stage == 1) RC(flush_ln_reduce(st));
    RC(flush_grouped_wgrads(st));
  }
  if (stage != 1) {
    RC(backward_stem(st));
    RC(flush_ln_reduce(st));
    RC(flush_grouped_wgrads(st, tail_on_main != 0));
  }
  return finish_backward(st, stage != 1);
}

int SmdEngine::optimizer_step(const TrainHyper& h, hipStream_t st) {
  SMD_ARG_CHECK(grads_ && params_, "optimizer_step: not bound");
  SMD_ARG_CHECK(training_ && W.norm_partial, "optimizer_step: bind a training workspace first");
  RC(join_update(st));
  AdamArgs a;
  a.params = params_; a.grads = grads_; a.m = m_; a.v = v_; a.ema = ema_; a.n = (size_t)n_params_;
  a.lr0 = h.lr0; a.lr_gamma = h.lr_gamma; a.lr_interval = h.lr_interval;
  a.beta1 = h.beta1; a.beta2 = h.beta2; a.eps = h.eps; a.grad_clip = h.grad_clip; a.mu = h.mu;
  a.grad_scale = h.grad_scale;
  a.step_ptr = step_ptr_; a.norm_partial = W.norm_partial; a.metrics_out = metrics_;
  if (!opt_fused_ok_ || !opt_fused_user_) {   // table overflow (a very deep DenseDDPM): norm, update, re-cast as three passes
    head_norm_ready_ = false;
    RC(launch_grad_sumsq(a, st));
    RC(launch_adam_clip_ema(a, st));
    return refresh_weights(st);
  }
  // norm partials: slots [0, SMD_NORM_HEAD_SLOTS) = output-stage slice (already reduced on the side stream by loss_backward(stage 0)
  // with opt_overlap bit 1 -- the join at the end of that call ordered `st` behind it), the rest = stem slice
  const size_t n_head = (size_t)(n_params_ - head_off_), n_stem = (size_t)head_off_;
  if (!head_norm_ready_) {
    if (n_head) RC(launch_grad_sumsq_slots(grads_ + head_off_, n_head, W.norm_partial, SMD_NORM_HEAD_SLOTS, st));
    else { hipError_t e = hipMemsetAsync(W.norm_partial, 0, SMD_NORM_HEAD_SLOTS * sizeof(float), st); if (e != hipSuccess) { smd_set_error("optimizer_step: %s", hipGetErrorString(e)); return (int)e; } }
  }
  head_norm_ready_ = false;
  if (n_stem) RC(launch_grad_sumsq_slots(grads_, n_stem, W.norm_partial + SMD_NORM_HEAD_SLOTS, SMD_NORM_SLOTS - SMD_NORM_HEAD_SLOTS, st));
  else { hipError_t e = hipMemsetAsync(W.norm_partial + SMD_NORM_HEAD_SLOTS, 0, (SMD_NORM_SLOTS - SMD_NORM_HEAD_SLOTS) * sizeof(float), st); if (e != hipSuccess) { smd_set_error("optimizer_step: %s", hipGetErrorString(e)); return (int)e; } }
  RC(launch_opt_prepare(a, SMD_NORM_SLOTS, W.opt_consts, st));
  RC(launch_adam_recast(a, W.opt_consts, wpack_, opt_stem_, st));
  w8_dirty_ = true;
  if ((opt_overlap & 1) && side_wgrad && side_ && opt_head_.total_blocks) {
    // the output stage's 86 % of the bytes: on the side stream, underneath the next forward pass's encoder
    hipEvent_t ev = take_event();
    if (!head_done_ev_ && hipEventCreateWithFlags(&head_done_ev_, hipEventDisableTiming) != hipSuccess) head_done_ev_ = nullptr;
    SMD_ARG_CHECK(ev && head_done_ev_, "optimizer_step: cannot create an event");
    hipError_t e = hipEventRecord(ev, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(side_, ev, 0);
    if (e != hipSuccess) { smd_set_error("optimizer_step: event: %s", hipGetErrorString(e)); return (int)e; }
    RC(launch_adam_recast(a, W.opt_consts, wpack_, opt_head_, side_, opt_side_blocks));
    e = hipEventRecord(head_done_ev_, side_);
    if (e != hipSuccess) { smd_set_error("optimizer_step: event: %s", hipGetErrorString(e)); return (int)e; }
    head_pending_ = true;
    next_event_ = 0;
    return 0;
  }
  return launch_adam_recast(a, W.opt_consts, wpack_, opt_head_, st);
}

// ------------------------------------------------------------------ sampler
int SmdEngine::prepare_sampler(hipStream_t st) {
  SMD_ARG_CHECK(film_tables_ && sqrt_ap_, "prepare_sampler: bind_schedule (with film tables) first");
  // Noise level is batch-uniform inside diffusion_dynamics (utils/ebm_utils.py:367-369): the FiLM
  // scale/shift of every block depend on t only, so all T rows are generated once here.
  const int T = d_.num_timesteps, M = d_.mlp_dims, F = d_.film_channels, K = nblocks();
  float* tables_end = film_tables_ + (size_t)K * T * 2 * M;
  bf16_t* emb = reinterpret_cast<bf16_t*>(tables_end);
  bf16_t* f1 = emb + (size_t)T * F;
  bf16_t* p = f1 + (size_t)T * 4 * F;
  RC(launch_noise_embed(sqrt_ap_, T, F, emb, F, st));
  for (int k = 0; k < K; ++k) {
    const FilmResP& b = blk_[k];
    { GemmEpilogue ep; ep.act = SMD_ACT_SWISH; ep.out_bf16 = f1; ep.ld_outb = 4 * F; RC(dense_fwd(b.f1, emb, F, T, ep, st)); }
    { GemmEpilogue ep; ep.out_bf16 = p; ep.ld_outb = 4 * F; RC(dense_fwd(b.f2, f1, 4 * F, T, ep, st)); }
    { GemmEpilogue ep; ep.out_f32 = film_tables_ + (size_t)k * T * 2 * M; ep.ld_out = 2 * M;
      RC(dense_fwd(b.ss, p, 4 * F, T, ep, st)); }
  }
  return 0;
}

int SmdEngine::init_state(float* x, uint32_t seed_lo, uint32_t seed_hi, uint32_t sample_offset, hipStream_t st) {
  SMD_ARG_CHECK(x && batch_ > 0, "init_state: not bound");
  const int per = d_.seq_len * d_.data_channels;
  RC(launch_fill_normal(x, batch_, per, RngKey{seed_lo, seed_hi}, /*SMD_STREAM_INIT*/ 3u, sample_offset, st));
  return launch_cast_pad_bf16(x, rows(), d_.data_channels, W.x_bf16, Cp_, st);
}

int SmdEngine::load_state(const float* x, hipStream_t st) {
  SMD_ARG_CHECK(x && batch_ > 0, "load_state: not bound");
  return launch_cast_pad_bf16(x, rows(), d_.data_channels, W.x_bf16, Cp_, st);
}

int SmdEngine::sample_step(const SampleStepIO& io, hipStream_t st, int part) {
  SMD_ARG_CHECK(io.x && io.t_ptr, "sample_step: null state / t pointer");
  SMD_ARG_CHECK(!training_ && coef_ && film_tables_, "sample_step: bind an inference workspace and the schedule tables first");
  SMD_ARG_CHECK(part >= 0 && part <= 2, "sample_step: part=%d (0 whole step, 1 stem, 2 output stage + reverse update)", part);
  RC(run_network(io.t_ptr, st, part));
  if (part == 1) return 0;
  ReverseStepArgs a;
  a.x = io.x; a.eps_hat = W.pred;
  a.B = batch_; a.S = d_.seq_len; a.C = d_.data_channels; a.Cp = Cp_; a.T = d_.num_timesteps;
  a.coef = coef_; a.t_ptr = io.t_ptr; a.z_in = io.z_in; a.key = RngKey{io.seed_lo, io.seed_hi};
  a.sample_offset = io.sample_offset;
  a.infill_samples = io.infill_samples; a.infill_masks = io.infill_masks; a.infill_z_in = io.infill_z_in;
  a.tf_noise_keys = io.tf_noise_keys; a.tf_infill_keys = io.tf_infill_keys; a.tf_n_total = io.tf_n_total; a.tf_t0 = io.tf_t0;
  a.key_ptr = io.key_ptr;
  a.x_bf16 = W.x_bf16; a.metrics_partial = io.metrics_partial; a.collection = io.collection;
  a.slot_table = io.slot_table;
  a.t_advance = io.t_ptr; a.arrive = W.step_arrive;      // *t_ptr -= 1 by the step's last workgroup (no launch of its own)
  return launch_reverse_step(a, st);
}
