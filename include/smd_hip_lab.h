/* smd_hip_lab.h -- laboratory hooks of libsmd_hip.so: kernel-selection knobs for A/B runs, debug views of the engine's saved
 * activations, hardware probes.  Used by tests/ and tools/ of this repository only; NOT part of the stable surface (smd_hip.h), no
 * reference callable stands behind any of these, and they may change without an ABI version bump. */
#ifndef SMD_HIP_LAB_H_
#define SMD_HIP_LAB_H_

#include "smd_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Debugging aid (tools/det_first_diff.py): with a buffer of smd_engine_debug_snapshot_bytes() bytes set, the encoder backward
 * copies the shared per-layer gradient buffers (da2 partial tiles, dh, dA_E, dh) behind each of its kernels into it, so two
 * identical steps can be compared kernel by kernel; behind those 34 * rows * E bytes per encoder layer follow 2 K + 1 copies of
 * the output stage's shared dX buffer (bf16 [rows][mlp_dims]; behind out_proj's dgrad, then for block K-1 .. 0 behind the dgrad
 * of its second and its first Dense).  NULL switches it off (the default). */
int64_t smd_engine_debug_snapshot_bytes(const smd_engine* e);
int smd_engine_debug_snapshots(smd_engine* e, void* buf, int64_t bytes);

/* Debugging aid (layer-by-layer parity): device pointer, shape and element type (0 fp32, 1 bf16) of an activation the training
 * forward pass saved in the bound workspace: "x_bf16", "h"/"h_mid"/"a1"/"qkv"/"o"/"a2" [encoder layer], "h_last", "af",
 * "y" [0..K], "ya1"/"o1"/"ya2"/"f1"/"p"/"ss" [block], "emb", "ao", "pred", "s"; and of the last backward pass the operands of
 * every weight-gradient GEMM: "dpred", "dyb" [0..K], "do1"/"dss"/"dss_bf16"/"dp"/"df1"/"zf1" [block], "dhb" [0..2L], "dqkv"/"dz1"/"u" [encoder
 * layer].  Valid until the next call on the handle. */
int smd_engine_debug_tensor(const smd_engine* e, const char* name, int index, const void** ptr, int64_t* rows, int64_t* cols,
                            int32_t* dtype);

/* process-wide kernel-selection knob for benchmark A/B runs (defaults = fast paths).
 * "gemm_nt256": 1 = large Dense GEMMs use the 256x256 8-phase kernel (default), 0 = 128-wide tiles only,
 *               2 = every shape with M,N % 256 == 0 and K % 128 == 0 (tests);
 * "gemm_nt256_variant": schedule variant of that kernel, 0 (default) .. 3, all computing the same result; the
 *               ablation variants used by tools/kbench.py exist only in a -DSMD_ABLATIONS build;
 * "ln_bwd_narrow": 1 (default) = the 128-wide LayerNorm backward runs on the 16-lanes-per-row kernel, 0 = one row per wave;
 * "tn_exclusive_cu": which 128-wide weight-gradient kernel the side stream runs and whether its workgroups own their CU:
 *               2 (default) = the four-buffer kernel (+ loader waves), unpadded: small-LDS workgroups of the main stream may
 *               share its CU (bitwise repeatable: 0 of 1499 repeated steps differ); 1 = the same kernels
 *               padded to the CU's whole LDS so that nothing shares their CU (the round-2 default, 2 % slower);
 *               0 = the two-buffer kernel, unpadded: NOT IN THE SHIPPED LIBRARY (the launch fails) -- with it on their CU the
 *               128-wide LayerNorm backward kernels intermittently compute a wrong row statistic (tools/rsq_repro.hip); it and
 *               the other experiment instantiations exist in a -DSMD_TN_EXPERIMENTS build only;
 * "tn_mode":    0 (default) = as "tn_exclusive_cu" says; NS*100 + NW*10 + pad picks buffers / issuing waves / pad (0 none,
 *               1 whole CU, 2 96 KiB) of that kernel explicitly; anything but 48x needs the experiment build;
 * "gemm_tn256": 1 = 2048-wide weight gradients use the 256x256 8-phase kernel (default), 0 = 128-wide tiles,
 *               2 = also on small grids (tests);
 * "tn_split_model": 1 = split-K of the 128-wide weight gradients chosen for whole rounds of 256 workgroups;
 * "tn128_loader_waves": 1 = the 128-wide weight-gradient kernel runs four extra waves that only issue LDS-DMA (0 needs the
 *               experiment build);
 * "gemm_nt_form": 0 (default) = the 128-column kernel picks its tile form from the shape; 1 .. 6 force <64,2>, <128,2>,
 *               <128,2,two K-groups>, <64,2,two K-groups>, <128,3>, <64,3,two K-groups> (tools/gemm_nt_forms_ab.py);
 *               "gemm_nt_form_wk": the same forms for the wide-K, few-column shapes only (N <= 512, K >= 2048: out_proj);
 * further keys ("ln_bwd_wide", "ln_fwd_wide", "ln_bwd_narrow", "gemm_nt_deep", "gemm_nt_kg", "mlp_variant", ...) select
 * between equivalent kernels for A/B runs; unknown keys return < 0. */
int smd_set_tuning(const char* key, int value);

/* lab probe: a HIP stream (returned as void*) created with hipExtStreamCreateWithCUMask(mask_words[0 .. n_words), one bit per CU).
 * Measured on MI355X / ROCm 7.2 (tools/cumask_probe.hip, profiles/r5a_cumask_probe.txt): mask bit i belongs to XCC i % 8; an XCC
 * whose share of the mask is EMPTY is not masked at all (a single set bit leaves 7 x 32 + 1 = 225 CUs), so a stream cannot be kept
 * off an XCD -- only a fraction of EVERY XCC's CUs can be selected (bits 0..127 = half of each).  Graph launches honour the launch
 * stream's mask like plain launches.  tools/chain_phase.py used it to show that halving every XCC between the two sampling chains
 * buys nothing; the product does not use it. */
int smd_probe_stream_create_cu_mask(const uint32_t* mask_words, int n_words, void** stream_out);
int smd_probe_stream_destroy(void* stream);
/* lab probe: `blocks` one-wave workgroups spin for ~spin_us; out[8 * b + ..] = XCC id, HW_ID, shader-clock ticks (2 words),
 * 100 MHz ticks (2 words), start time in 100 MHz ticks (2 words) -- where a stream runs and at which clock */
int smd_probe_clock(uint32_t* out, int blocks, int spin_us, void* stream);

/* lab probe: every XCD reads all `bytes` of `p`, leaving it resident in all eight L2s (sink: one writable dword or NULL) */
int smd_probe_l2_warm(const void* p, int64_t bytes, uint32_t* sink, void* stream);

/* lane-level probe of ds_read_b64_tr_b16 (debug): out[64][4] = values read from a 2 KiB linear image */
int smd_probe_tr_read(const smd_bf16* image_1024, smd_bf16* out_256, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMD_HIP_LAB_H_ */
