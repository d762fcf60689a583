"""Checkpoints of the (optimizer, ema, early_stop) tuple saved at train_ncsn.py:395-399.

File naming follows flax.training.checkpoints (``checkpoint_<step>`` in the model dir, ``keep``
newest kept, the step being the eval counter ``sampling_step``).  The payload is safetensors with
the parameter pytree under the engine's stable names: ``target/params/<name>``,
``state/param_states/<name>/{grad_ema,grad_sq_ema}``, ``state/step``, ``ema/params/<name>``; the
EarlyStopping fields ride in the metadata.

``fmt="flax"`` writes -- and ``restore_checkpoint`` / ``load_ema_params`` recognise and read -- the reference's own
file instead: flax.serialization msgpack of the (optimizer, ema, early_stop) state dict with the flax.nn parameter
tree (flax_io.py), so upstream-trained weights drop in and our checkpoints load upstream.
"""
from __future__ import annotations

import glob
import json
import os
import re
from typing import Dict, Optional, Tuple

import numpy as np  # noqa: F401  (type names in annotations)
import torch
from safetensors.torch import load_file, save_file

from . import flax_io
from .train_utils import EarlyStopping


def _named(engine, flat: torch.Tensor, prefix: str) -> Dict[str, torch.Tensor]:
    return {f"{prefix}/{k}": v.detach().cpu().contiguous() for k, v in engine.named_views(flat).items()}


def _np_named(engine, flat: torch.Tensor) -> Dict[str, "np.ndarray"]:
    return {k: v.detach().cpu().numpy() for k, v in engine.named_views(flat).items()}


def _rotate(ckpt_dir: str, prefix: str, keep: int) -> None:
    found = sorted((p for p in glob.glob(os.path.join(ckpt_dir, prefix + "*")) if re.findall(r"(\d+)$", p)),
                   key=lambda p: int(re.findall(r"(\d+)$", p)[0]))
    for old in found[:-keep] if keep > 0 else []:
        os.remove(old)


def save_checkpoint(ckpt_dir: str, target, step: int, keep: int = 50, prefix: str = "checkpoint_",
                    fmt: str = "safetensors", flax_naming: str = "shared", flax_attention_class: str = "MultiHeadDotProductAttention") -> str:
    """target = (optimizer, ema, early_stop) like the reference call (train_ncsn.py:397-399)."""
    optimizer, ema, early_stop = target
    eng = optimizer.engine
    os.makedirs(ckpt_dir, exist_ok=True)
    if fmt == "flax":
        ema_flat = ema.params if ema is not None and getattr(ema, "params", None) is not None else eng.params
        sd = flax_io.checkpoint_state_dict(
            eng.cfg, _np_named(eng, eng.params), _np_named(eng, eng.m), _np_named(eng, eng.v), int(eng.step_counter.item()),
            _np_named(eng, ema_flat), float(getattr(ema, "mu", 0.0)), early_stop.state_dict() if early_stop else {},
            rule=flax_naming, attention_class=flax_attention_class)
        path = os.path.join(ckpt_dir, f"{prefix}{step}")
        flax_io.write_file(path, sd)
        _rotate(ckpt_dir, prefix, keep)
        return path
    if fmt != "safetensors":
        raise ValueError(f"checkpoint format must be 'safetensors' or 'flax', got {fmt!r}")
    tensors = _named(eng, eng.params, "target/params")
    for k, v in eng.named_views(eng.m).items():
        tensors[f"state/param_states/{k}/grad_ema"] = v.detach().cpu().contiguous()
    for k, v in eng.named_views(eng.v).items():
        tensors[f"state/param_states/{k}/grad_sq_ema"] = v.detach().cpu().contiguous()
    tensors["state/step"] = eng.step_counter.detach().cpu().to(torch.int32)
    ema_flat = ema.params if ema is not None else eng.params        # ema=False: untouched init params upstream
    tensors.update(_named(eng, ema_flat, "ema/params"))
    meta = {"early_stop": json.dumps(early_stop.state_dict() if early_stop else {}),
            "ema_mu": str(getattr(ema, "mu", 0.0)), "format": "smd_amd-1",
            "trunk_dtype": str(getattr(eng, "trunk_dtype", "bf16")), "gemm_dtype": str(eng.cfg.dtype),
            "fp8_dgrad": str(int(getattr(eng, "fp8_dgrad", 1)) if eng.cfg.dtype == "fp8" else 0)}
    path = os.path.join(ckpt_dir, f"{prefix}{step}")
    save_file(tensors, path + ".tmp", metadata=meta)
    os.replace(path + ".tmp", path)
    _rotate(ckpt_dir, prefix, keep)
    return path


def latest_checkpoint(ckpt_dir: str, prefix: str = "checkpoint_") -> Optional[str]:
    found = [p for p in glob.glob(os.path.join(ckpt_dir, prefix + "*")) if re.findall(r"(\d+)$", p)]
    return max(found, key=lambda p: int(re.findall(r"(\d+)$", p)[0])) if found else None


def restore_checkpoint(ckpt_dir: str, engine, load_optimizer_state: bool = True) -> Tuple[bool, EarlyStopping]:
    """Loads the newest checkpoint into ``engine`` (params, and Adam/EMA buffers when training is
    enabled).  Returns (found, early_stop)."""
    path = ckpt_dir if os.path.isfile(ckpt_dir) else latest_checkpoint(ckpt_dir)
    if path is None:
        return False, EarlyStopping()
    if flax_io.is_flax_file(path):
        template = {name: tuple(shape) for name, _off, shape in engine.tensor_table}
        params, m, v, step, ema_params, _mu, es = flax_io.split_state_dict(flax_io.read_file(path), engine.cfg, template)
        engine.load_named(params)
        if load_optimizer_state and engine.grads is not None:
            for flat, src in ((engine.m, m), (engine.v, v)):
                for k, dst in engine.named_views(flat).items():
                    dst.copy_(torch.from_numpy(src[k]).to(dst.device))
            engine.step_counter.fill_(step)
            if engine.ema is not None:
                for k, dst in engine.named_views(engine.ema).items():
                    dst.copy_(torch.from_numpy(ema_params[k]).to(dst.device))
        known = {f for f in ("min_delta", "patience", "best_metric", "patience_count", "should_stop")}
        return True, EarlyStopping(**{k: x for k, x in es.items() if k in known})
    tensors = load_file(path)
    engine.load_named({k[len("target/params/"):]: v for k, v in tensors.items() if k.startswith("target/params/")})
    if load_optimizer_state and engine.grads is not None:
        for flat, suffix in ((engine.m, "grad_ema"), (engine.v, "grad_sq_ema")):
            for k, v in engine.named_views(flat).items():
                v.copy_(tensors[f"state/param_states/{k}/{suffix}"].to(v.device))
        engine.step_counter.copy_(tensors["state/step"].to(engine.device))
        if engine.ema is not None:
            for k, v in engine.named_views(engine.ema).items():
                v.copy_(tensors[f"ema/params/{k}"].to(v.device))
    from safetensors import safe_open
    with safe_open(path, framework="pt") as f:
        meta = f.metadata() or {}
    es = json.loads(meta.get("early_stop", "{}"))
    return True, EarlyStopping(**es) if es else EarlyStopping()


def load_ema_params(ckpt_dir: str, engine) -> bool:
    path = ckpt_dir if os.path.isfile(ckpt_dir) else latest_checkpoint(ckpt_dir)
    if path is None:
        return False
    if flax_io.is_flax_file(path):
        template = {name: tuple(shape) for name, _off, shape in engine.tensor_table}
        engine.load_named(flax_io.split_state_dict(flax_io.read_file(path), engine.cfg, template)[4])
        return True
    tensors = load_file(path)
    engine.load_named({k[len("ema/params/"):]: v for k, v in tensors.items() if k.startswith("ema/params/")})
    return True
