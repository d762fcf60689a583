"""Training-state helpers with the reference's names and semantics (utils/train_utils.py)."""
from __future__ import annotations

import logging
import math
from dataclasses import dataclass, replace
from typing import Any, Dict, Optional

import torch

log = logging.getLogger("smd_amd")


@dataclass(frozen=True)
class EarlyStopping:
    """utils/train_utils.py:26-59 (same fields, same update rule)."""
    min_delta: float = 0
    patience: int = 0
    best_metric: float = float("inf")
    patience_count: int = 0
    should_stop: bool = False

    def update(self, metric):
        if math.isinf(self.best_metric) or self.best_metric - metric > self.min_delta:
            return True, replace(self, best_metric=metric, patience_count=0)
        should_stop = self.patience_count >= self.patience or self.should_stop
        return False, replace(self, patience_count=self.patience_count + 1, should_stop=should_stop)

    def state_dict(self) -> Dict[str, Any]:
        return dict(min_delta=self.min_delta, patience=self.patience, best_metric=self.best_metric,
                    patience_count=self.patience_count, should_stop=self.should_stop)


class EMAHelper:
    """utils/train_utils.py:62-78.  ``params`` is the flat EMA buffer.  When it is the training
    engine's own EMA buffer the update p_ema = mu p_ema + (1-mu) p has already been applied by the
    fused clip+Adam+EMA kernel of that step and ``update`` is a no-op; a detached helper (EMA the
    engine does not track) is updated here with torch ops on the GPU."""

    def __init__(self, mu: float, params: torch.Tensor, fused: bool = False):
        self.mu = mu
        self.params = params
        self.fused = fused

    def update(self, model) -> "EMAHelper":
        if not self.fused:
            self.params.mul_(self.mu).add_(model.params, alpha=1.0 - self.mu)
        return self


def log_metrics(metrics, step, total_steps, epoch=None, summary_writer=None, verbose=True):
    """utils/train_utils.py:81-118: same stdout line format; ``summary_writer`` needs .scalar()."""
    if hasattr(metrics, "resolve"):
        metrics = metrics.resolve()
    metrics_str = ""
    for metric in metrics:
        value = float(metrics[metric])
        if metric == "lr":
            metrics_str += "{} {:5.4f} | ".format(metric, value)
        else:
            metrics_str += "{} {:5.2f} | ".format(metric, value)
        if summary_writer is not None:
            writer_step = step
            if epoch is not None:
                writer_step = total_steps * epoch + step
            summary_writer.scalar(metric, value, writer_step)
    epoch_str = "| epoch {:3d} ".format(epoch) if epoch is not None else ""
    if verbose:
        log.info("{}| {:5d}/{:5d} steps | {}".format(epoch_str, step, total_steps, metrics_str))


def report_model(model):
    """utils/train_utils.py:121-131."""
    n = model.num_parameters()
    log.info("Number of trainable paramters: {:,}".format(n))
    log.info("Memory footprint: %dMB", n * 4 / 2 ** 20)


class JsonlWriter:
    """TensorBoard is not installed on the target image: scalars go to <dir>/scalars.jsonl with
    the reference's tag names (loss, grad, lr, batch/s, ms/batch; slope, step, alpha, noise)."""

    def __init__(self, path: str):
        import os
        os.makedirs(path, exist_ok=True)
        self._f = open(os.path.join(path, "scalars.jsonl"), "a")

    def scalar(self, tag: str, value, step: int):
        import json
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()
