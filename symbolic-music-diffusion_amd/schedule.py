"""Noise schedule and the per-timestep constant tables uploaded once to the GPU.

``create_noise_schedule`` keeps the reference signature (utils/ebm_utils.py:62-86).  The tables are
computed in float32 on the host exactly as the reference's jnp float32 expressions
(utils/ebm_utils.py:313-358, utils/losses.py:277-281) so they are bit-exact against the oracle.
"""
from __future__ import annotations

import numpy as np

COLLECTION_STEPS = 40   # utils/ebm_utils.py:320


def create_noise_schedule(sigma_begin=1.0, sigma_end=1e-2, L=10, schedule="geometric") -> np.ndarray:
    """utils/ebm_utils.py:62-86 (float32 like jnp)."""
    f = np.float32
    if schedule == "geometric":
        s = np.exp(np.linspace(np.log(f(sigma_begin)), np.log(f(sigma_end)), L, dtype=np.float32))
    elif schedule == "linear":
        s = np.linspace(f(sigma_begin), f(sigma_end), L, dtype=np.float32)
    elif schedule == "fibonacci":
        v = [1e-6, 2e-6]
        for _ in range(L - 2):
            v.append(v[-1] + v[-2])
        s = np.array(v, dtype=np.float32)
    else:
        raise ValueError(f"Unsupported schedule: {schedule}")
    return s.astype(np.float32)


def alphas_cumprod(betas: np.ndarray) -> np.ndarray:
    """cumprod(1 - betas), float32 (utils/ebm_utils.py:315-316)."""
    betas = np.asarray(betas, dtype=np.float32)
    return np.cumprod((np.float32(1.0) - betas).astype(np.float32), dtype=np.float32)


def reverse_coefficient_table(betas: np.ndarray) -> np.ndarray:
    """[T][8] float32: sqrt(1/ap), sqrt(1-ap)*sqrt(1/ap), mu1, mu2, sigma, ap, sqrt(ap), sqrt(1-ap)
    (utils/ebm_utils.py:332-358, sigma = exp(0.5*log(max(var,1e-20))) as at :356-364)."""
    f = np.float32
    betas = np.asarray(betas, dtype=np.float32)
    alphas = (f(1) - betas).astype(np.float32)
    ap = alphas_cumprod(betas)
    ap_prev = np.concatenate([np.ones((1,), np.float32), ap[:-1]])
    sqrt_recip = np.sqrt(f(1) / ap, dtype=np.float32)
    sqrt_m1 = (np.sqrt(f(1) - ap, dtype=np.float32) * sqrt_recip).astype(np.float32)
    mu1 = (betas * np.sqrt(ap_prev, dtype=np.float32) / (f(1) - ap)).astype(np.float32)
    mu2 = ((f(1) - ap_prev) * np.sqrt(alphas, dtype=np.float32) / (f(1) - ap)).astype(np.float32)
    var = (betas * (f(1) - ap_prev) / (f(1) - ap)).astype(np.float32)
    var_c = np.maximum(var, f(1e-20)).astype(np.float32)
    sigma = np.exp(f(0.5) * np.log(var_c, dtype=np.float32), dtype=np.float32)
    out = np.stack([sqrt_recip, sqrt_m1, mu1, mu2, sigma, ap, np.sqrt(ap, dtype=np.float32),
                    np.sqrt(f(1) - ap, dtype=np.float32)], axis=1)
    return np.ascontiguousarray(out.astype(np.float32))


def collection_index_table(T: int) -> np.ndarray:
    """linspace(1, T, 40).astype(int32), utils/ebm_utils.py:324-325."""
    return np.linspace(np.float32(1), np.float32(T), COLLECTION_STEPS, dtype=np.float32).astype(np.int32)


def collection_slot_table(T: int) -> np.ndarray:
    """slot[t] = collection row written after the step at timestep t, or -1
    (image_idx = T - t + 1 matched against collection_idx, utils/ebm_utils.py:387-394).
    Slot 1 is never written and the final state (t=0) is never collected -- reference quirk."""
    table = collection_index_table(T)
    slot = np.full((T,), -1, dtype=np.int32)
    for t in range(T):
        hit = np.nonzero(table == (T - t + 1))[0]
        if hit.size:
            # short schedules (T < 40) repeat linspace entries and the summed index can pass the last row: the reference's
            # jax.ops.index_update then scatters out of bounds, which XLA drops
            s = int(np.sum(hit)) + 1
            slot[t] = s if s <= COLLECTION_STEPS else -1
    return slot
