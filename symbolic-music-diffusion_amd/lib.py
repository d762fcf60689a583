"""ctypes binding of the C-ABI in include/smd_hip.h.  Fails loudly: there is no CPU fallback.

``get_lib()`` loads ``csrc/libsmd_hip.so`` (building it with hipcc first if it is missing or its embedded build id is
not the id of this tree's sources -- build.py -- and hipcc exists) and refuses a library of another build id.  Every wrapper raises ``ValueError`` for negative return codes (argument / state
errors -- the reference raises Python asserts there, e.g. models/ncsn.py:32,40,50,154) and
``RuntimeError`` for HIP errors.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import List, Optional

from . import build as _build

_LIB: Optional[C.CDLL] = None

c_f32p = C.POINTER(C.c_float)
c_void = C.c_void_p
c_i32 = C.c_int32
c_u32 = C.c_uint32
c_i64 = C.c_int64


class ModelDesc(C.Structure):
    _fields_ = [(n, c_i32) for n in ("arch", "data_channels", "seq_len", "num_layers", "num_heads",
                                     "num_mlp_layers", "mlp_dims", "embed_channels", "film_channels",
                                     "num_timesteps")]


class TrainHyper(C.Structure):
    _fields_ = [("lr0", C.c_float), ("lr_gamma", C.c_float), ("lr_interval", c_i32), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("grad_clip", C.c_float), ("mu", C.c_float),
                ("grad_scale", C.c_float)]


class SampleIO(C.Structure):
    _fields_ = [("x", c_void), ("t_ptr", c_void), ("z_in", c_void), ("seed_lo", c_u32), ("seed_hi", c_u32),
                ("sample_offset", c_u32), ("infill_samples", c_void), ("infill_masks", c_void),
                ("infill_z_in", c_void), ("metrics_partial", c_void), ("collection", c_void),
                ("slot_table", c_void), ("tf_noise_keys", c_void), ("tf_infill_keys", c_void), ("tf_n_total", c_i64),
                ("tf_t0", c_i32), ("key_ptr", c_void)]


class LangevinIO(C.Structure):
    _fields_ = [("x", c_void), ("grad", c_void), ("alpha", C.c_float), ("noise_coef", C.c_float), ("z_in", c_void),
                ("seed_lo", c_u32), ("seed_hi", c_u32), ("step", c_u32), ("sample_offset", c_u32), ("use_threefry", c_i32),
                ("tf_noise_key", c_u32 * 2), ("tf_infill_key", c_u32 * 2), ("tf_n_total", c_i64),
                ("infill_samples", c_void), ("infill_masks", c_void), ("infill_z_in", c_void), ("infill_sigma", C.c_float),
                ("metrics_partial", c_void), ("collect_out", c_void),
                ("step_table", c_void), ("slot_table", c_void), ("key_table", c_void), ("k_ptr", c_void), ("arrive", c_void),
                ("collection", c_void), ("sigma_out", c_void), ("n_steps", c_i32), ("level_out", c_void),
                ("steps_per_level", c_i32), ("n_levels", c_i32)]


ABI_VERSION = 6          # SMD_ABI_VERSION of include/smd_hip.h this table was written against

# name -> (restype, argtypes).  Pointers are passed as integers (tensor.data_ptr()) via c_void_p.
_SIGS = {
    "smd_last_error": (C.c_char_p, []),
    "smd_abi_version": (C.c_int, []),
    "smd_build_id": (C.c_char_p, []),
    "smd_engine_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(c_void)]),
    "smd_engine_destroy": (None, [c_void]),
    "smd_engine_num_tensors": (C.c_int, [c_void]),
    "smd_engine_tensor_info": (C.c_int, [c_void, C.c_int, C.POINTER(C.c_char_p), C.POINTER(c_i64),
                                         C.POINTER(c_i32), C.POINTER(c_i32)]),
    "smd_engine_param_count": (c_i64, [c_void]),
    "smd_engine_head_param_offset": (c_i64, [c_void]),
    "smd_engine_wpack_elems": (c_i64, [c_void]),
    "smd_engine_workspace_bytes": (c_i64, [c_void, C.c_int, C.c_int]),
    "smd_engine_film_table_floats": (c_i64, [c_void]),
    "smd_engine_padded_channels": (C.c_int, [c_void]),
    "smd_engine_set_option": (C.c_int, [c_void, C.c_char_p, C.c_int]),
    "smd_engine_bind_params": (C.c_int, [c_void, c_void, c_void]),
    "smd_engine_bind_train": (C.c_int, [c_void] * 7),
    "smd_engine_bind_workspace": (C.c_int, [c_void, c_void, c_i64, C.c_int, C.c_int, c_void]),
    "smd_engine_bind_schedule": (C.c_int, [c_void] * 5),
    "smd_engine_refresh_weights": (C.c_int, [c_void, c_void]),
    "smd_engine_forward": (C.c_int, [c_void] * 5),
    "smd_engine_forward_level": (C.c_int, [c_void] * 5),
    "smd_engine_loss_backward": (C.c_int, [c_void, c_void, c_void, c_void, c_u32, c_u32, c_u32, C.c_float,
                                           C.c_int, c_void]),
    "smd_engine_set_used_alphas": (C.c_int, [c_void, c_void]),
    "smd_engine_forward_train": (C.c_int, [c_void] * 5),
    "smd_engine_backward_from": (C.c_int, [c_void, c_void, C.c_int, c_void]),
    "smd_engine_debug_snapshot_bytes": (c_i64, [c_void]),
    "smd_engine_debug_snapshots": (C.c_int, [c_void, c_void, c_i64]),
    "smd_engine_loss_per_sample": (c_void, [c_void]),
    "smd_engine_pred": (c_void, [c_void]),
    "smd_engine_optimizer_step": (C.c_int, [c_void, C.POINTER(TrainHyper), c_void]),
    "smd_engine_join_update": (C.c_int, [c_void, c_void]),
    "smd_engine_debug_tensor": (C.c_int, [c_void, C.c_char_p, C.c_int, C.POINTER(c_void), C.POINTER(c_i64), C.POINTER(c_i64),
                                          C.POINTER(c_i32)]),
    "smd_engine_num_grad_buckets": (C.c_int, [c_void]),
    "smd_engine_grad_bucket": (C.c_int, [c_void, C.c_int, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "smd_engine_wait_grad_bucket": (C.c_int, [c_void, C.c_int, c_void]),
    "smd_engine_prepare_sampler": (C.c_int, [c_void, c_void]),
    "smd_engine_init_state": (C.c_int, [c_void, c_void, c_u32, c_u32, c_u32, c_void]),
    "smd_engine_load_state": (C.c_int, [c_void, c_void, c_void]),
    "smd_engine_sample_step": (C.c_int, [c_void, C.POINTER(SampleIO), c_void]),
    "smd_engine_sample_step_part": (C.c_int, [c_void, C.POINTER(SampleIO), C.c_int, c_void]),
    "smd_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "smd_set_timestep": (C.c_int, [c_void, c_i32, c_void]),
    "smd_gemm_bf16_nt": (C.c_int, [c_void, C.c_int, c_void, C.c_int, C.c_int, C.c_int, C.c_int, c_void, C.c_int,
                                   c_void, C.c_int, c_void, C.c_int, c_void, C.c_int, c_void]),
    "smd_quantize_rows_e4m3": (C.c_int, [c_void, C.c_int, C.c_int, C.c_int, c_void, c_void, c_void]),
    "smd_gemm_e4m3_nt": (C.c_int, [c_void, C.c_int, c_void, c_void, C.c_int, c_void, C.c_int, C.c_int, C.c_int, c_void, c_void,
                                   C.c_int, c_void, C.c_int, c_void, C.c_int, c_void]),
    "smd_layernorm_fwd_e4m3": (C.c_int, [c_void, C.c_int, C.c_int, c_void, c_void, c_void, c_void, C.c_int, C.c_int, C.c_int,
                                         c_void, c_void, c_void, c_void]),
    "smd_mlp_block_fwd": (C.c_int, [c_void, c_void, C.c_int, c_void, c_void, c_void, c_void, c_void, c_void, C.c_int, c_void,
                                    c_void, c_void, c_void]),
    "smd_mlp_block_fwd_hs": (C.c_int, [c_void, c_void, C.c_int, c_void, c_void, c_void, c_void, C.c_int, c_void, c_void]),
    "smd_mlp_block_bwd_hs": (C.c_int, [c_void, c_void, C.c_int, c_void, c_void, c_void, c_void, C.c_int, c_void, c_void, c_void, c_void]),
    "smd_ln128_bwd_parts": (C.c_int, [c_void, c_void, c_i64, C.c_int, c_void, c_void, c_void, c_void, c_void, c_void]),
    "smd_ln128_parts": (C.c_int, [c_void, c_i64, C.c_int, c_void, c_void, c_void, c_void, c_void]),
    "smd_attn_block_fwd_ex": (C.c_int, [c_void, c_void, c_i64, c_void, c_void, C.c_int, c_void, c_void, c_void, c_void, c_void,
                                        c_void, C.c_int, c_void, c_void, c_void, c_void, c_void, c_void, c_void]),
    "smd_attn_block_fwd": (C.c_int, [c_void, c_void, C.c_int, c_void, c_void, c_void, c_void, c_void, c_void, C.c_int, c_void,
                                     c_void, c_void, c_void]),
    "smd_attn_block_bwd": (C.c_int, [c_void, c_void, c_void, c_void, c_void, c_void, C.c_int, C.c_int, c_void]),
    "smd_attn_block_bwd_ln": (C.c_int, [c_void, c_void, c_void, c_void, c_void, c_void, c_void, C.c_int64, c_void, c_void, c_void, c_void,
                                        c_void, c_void, c_void, c_void, C.c_int, C.c_int, c_void]),
    "smd_gemm_bf16_tn": (C.c_int, [c_void, C.c_int, c_void, C.c_int, C.c_int, C.c_int, C.c_int, c_void, C.c_int,
                                   c_void, c_void, c_void, c_i64, c_void, c_i64, C.c_int, c_void]),
    "smd_gemm_tn_slab_elems": (c_i64, []),
    "smd_layernorm_fwd": (C.c_int, [c_void, C.c_int, C.c_int, c_void, c_void, c_void, c_void, C.c_int, C.c_int,
                                    C.c_int, c_void, c_void]),
    "smd_layernorm_fwd_ex": (C.c_int, [c_void, c_void, C.c_int, C.c_int, c_void, c_void, c_void, c_void, C.c_int, C.c_int,
                                       C.c_int, c_void, c_void]),
    "smd_layernorm_bwd": (C.c_int, [c_void, C.c_int, C.c_int, c_void, c_void, c_void, c_void, C.c_int, C.c_int,
                                    C.c_int, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_i64, c_void]),
    "smd_layernorm_bwd_ex": (C.c_int, [c_void, c_void, C.c_int, C.c_int, c_void, c_void, c_void, c_void, c_void, c_void, c_void,
                                       c_void, c_void, c_i64, c_void]),
    "smd_layernorm_bwd_film": (C.c_int, [c_void, c_void, C.c_int, C.c_int, c_void, c_void, c_void, c_void, C.c_int, C.c_int,
                                         C.c_int, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void,
                                         C.c_int, c_void, c_i64, c_void]),
    "smd_attention_fwd": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, c_void]),
    "smd_attention_bwd": (C.c_int, [c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, c_void]),
    "smd_noise_embed": (C.c_int, [c_void, C.c_int, C.c_int, c_void, C.c_int, c_void]),
    "smd_threefry_bits": (C.c_int, [c_void, c_i64, c_i64, c_i64, c_u32, c_u32, c_void]),
    "smd_threefry_uniform": (C.c_int, [c_void, c_i64, c_i64, c_i64, c_u32, c_u32, C.c_float, C.c_float, c_void]),
    "smd_threefry_normal": (C.c_int, [c_void, c_i64, c_i64, c_i64, c_u32, c_u32, c_void, c_void, C.c_int, C.c_int, c_void]),
    "smd_threefry_randint": (C.c_int, [c_void, c_i64, c_i64, c_i64, c_u32, c_u32, C.c_int32, C.c_int32, c_void]),
    "smd_q_sample": (C.c_int, [c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void, c_void, C.c_int, c_void, c_void,
                               c_u32, c_u32, c_void, c_u32, c_void, c_void, c_void, c_void]),
    "smd_mse_fwd_bwd": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_void, c_void, c_void]),
    "smd_adam_clip_ema": (C.c_int, [c_void, c_void, c_void, c_void, c_void, C.c_int64, C.POINTER(TrainHyper), c_void, c_void,
                                    c_void, c_void]),
    "smd_langevin_step": (C.c_int, [C.POINTER(LangevinIO), C.c_int, C.c_int, C.c_int, c_void]),
    "smd_rng_normal": (C.c_int, [c_void, C.c_int, C.c_int, c_u32, c_u32, c_u32, c_u32, c_void]),
    "smd_cast_pad_bf16": (C.c_int, [c_void, C.c_int, C.c_int, c_void, C.c_int, c_void]),
    "smd_ddpm_reverse_step": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, c_void, C.c_int, c_void, c_void, c_u32,
                                        c_u32, c_u32, c_void, c_void, c_void, c_void]),
    "smd_probe_tr_read": (C.c_int, [c_void, c_void, c_void]),
    "smd_probe_stream_create_cu_mask": (C.c_int, [C.POINTER(c_u32), C.c_int, C.POINTER(c_void)]),
    "smd_probe_stream_destroy": (C.c_int, [c_void]),
    "smd_probe_clock": (C.c_int, [c_void, C.c_int, C.c_int, c_void]),
    "smd_probe_l2_warm": (C.c_int, [c_void, c_i64, c_void, c_void]),
}

HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "smd_hip.h")


LAB_HEADER_PATH = os.path.join(os.path.dirname(HEADER_PATH), "smd_hip_lab.h")


def declared_symbols(lab: bool = True) -> List[str]:
    """Every function include/smd_hip.h (and, with ``lab``, include/smd_hip_lab.h) declares (used by the no-GPU export test)."""
    out = set()
    for path in (HEADER_PATH, LAB_HEADER_PATH) if lab else (HEADER_PATH,):
        text = open(path).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(smd_[a-z0-9_]+)\s*\(", text))
    return sorted(out)


def get_lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB_PATH
    if _build.is_stale() and not (os.environ.get("SMD_LIB_SUFFIX") and os.path.exists(path)):   # experiment builds are never rebuilt implicitly
        try:
            _build.build_library(verbose=False)
        except Exception as e:  # no hipcc, or compile failure
            if not os.path.exists(path):
                raise RuntimeError(
                    f"libsmd_hip.so is missing and could not be built ({e}). The HIP extension is "
                    "mandatory: there is no CPU fallback. Run `python -m smd_amd.build`.") from e
    # torch first: its bundled libamdhip64 must be the HIP runtime of the process.  Loading the library before torch pulls
    # in /opt/rocm's runtime under the same SONAME instead, and the first launch then fails with "no ROCm-capable device".
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)   # AttributeError here = the .so is older than the header
        fn.restype = res
        fn.argtypes = args
    if lib.smd_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libsmd_hip.so ABI {lib.smd_abi_version()} != {ABI_VERSION}; rebuild")
    have, want = lib.smd_build_id().decode(), _build.source_id()
    if have != want and os.environ.get("SMD_LIB_SUFFIX"):
        # an experiment library (tools/*.sh build these with their own SMD_EXTRA_DEFS / SMD_SLP, which feed the id): never rebuilt
        # implicitly and not expected to match the loading process's flags -- say so and go on
        import warnings
        warnings.warn(f"{path}: experiment library with build id {have} (this environment's tree id is {want})")
    elif have != want and os.environ.get("SMD_ALLOW_STALE_LIB") != "1":
        raise RuntimeError(f"{path} was built from other sources (build id {have}, this tree is {want}) and could not be "
                           "rebuilt here: run `python -m smd_amd.build` where hipcc exists (SMD_ALLOW_STALE_LIB=1 overrides)")
    _LIB = lib
    return lib


_TUNING_EPOCH = 0


def set_tuning(key: str, value: int) -> None:
    """smd_set_tuning (process-wide kernel-selection knob) + a bump of the epoch that cached captured steps are keyed on."""
    global _TUNING_EPOCH
    check(get_lib().smd_set_tuning(key.encode(), int(value)), f"set_tuning {key}")
    _TUNING_EPOCH += 1


def tuning_epoch() -> int:
    return _TUNING_EPOCH


def check(rc: int, what: str = "") -> None:
    if rc == 0:
        return
    msg = get_lib().smd_last_error()
    text = f"{what}: {msg.decode() if msg else 'unknown error'} (rc={rc})"
    if rc < 0:
        raise ValueError(text)
    raise RuntimeError(text)
