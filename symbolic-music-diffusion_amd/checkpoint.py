"""Checkpoints of the (optimizer, ema, early_stop) tuple saved at train_ncsn.py:395-399.

File naming follows flax.training.checkpoints (``checkpoint_<step>`` in the model dir, ``keep``
newest kept, the step being the eval counter ``sampling_step``).  The payload is safetensors with
the parameter pytree under the engine's stable names: ``target/params/<name>``,
``state/param_states/<name>/{grad_ema,grad_sq_ema}``, ``state/step``, ``ema/params/<name>``; the
EarlyStopping fields ride in the metadata.  Reading/writing flax-0.3.0 msgpack state dicts is a
"next" row.
"""
from __future__ import annotations

import glob
import json
import os
import re
from typing import Dict, Optional, Tuple

import torch
from safetensors.torch import load_file, save_file

from .train_utils import EarlyStopping


def _named(engine, flat: torch.Tensor, prefix: str) -> Dict[str, torch.Tensor]:
    return {f"{prefix}/{k}": v.detach().cpu().contiguous() for k, v in engine.named_views(flat).items()}


def save_checkpoint(ckpt_dir: str, target, step: int, keep: int = 50, prefix: str = "checkpoint_") -> str:
    """target = (optimizer, ema, early_stop) like the reference call."""
    optimizer, ema, early_stop = target
    eng = optimizer.engine
    os.makedirs(ckpt_dir, exist_ok=True)
    tensors = _named(eng, eng.params, "target/params")
    for k, v in eng.named_views(eng.m).items():
        tensors[f"state/param_states/{k}/grad_ema"] = v.detach().cpu().contiguous()
    for k, v in eng.named_views(eng.v).items():
        tensors[f"state/param_states/{k}/grad_sq_ema"] = v.detach().cpu().contiguous()
    tensors["state/step"] = eng.step_counter.detach().cpu().to(torch.int32)
    ema_flat = ema.params if ema is not None else eng.params        # ema=False: untouched init params upstream
    tensors.update(_named(eng, ema_flat, "ema/params"))
    meta = {"early_stop": json.dumps(early_stop.state_dict() if early_stop else {}),
            "ema_mu": str(getattr(ema, "mu", 0.0)), "format": "smd_amd-1"}
    path = os.path.join(ckpt_dir, f"{prefix}{step}")
    save_file(tensors, path + ".tmp", metadata=meta)
    os.replace(path + ".tmp", path)
    found = sorted(glob.glob(os.path.join(ckpt_dir, prefix + "*")),
                   key=lambda p: int(re.findall(r"(\d+)$", p)[0]) if re.findall(r"(\d+)$", p) else -1)
    for old in found[:-keep] if keep > 0 else []:
        os.remove(old)
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
    tensors = load_file(path)
    engine.load_named({k[len("ema/params/"):]: v for k, v in tensors.items() if k.startswith("ema/params/")})
    return True
