"""create_model's initial parameters as the reference draws them (train_ncsn.py:193-203): ``module.init_by_shape(model_rng, ...)``
of the old flax.nn API with every Dense / DenseGeneral kernel = lecun_normal(), biases 0, LayerNorm scale 1 / bias 0.

With ``--rng_impl=threefry`` every NOISE stream of a run is already the reference's (jax_random.py); this module makes the
step-0 WEIGHTS the reference's too, so that "same seed => same training run" starts from the same point.  Restated from the
flax 0.3.0 / jax 0.2.8 sources (not installable here: unpinned until tests/golden/make_jax_goldens.py can run):

  flax.nn.base     child frame rng  = _fold_in_str(parent.rng, child_name)      child_name = auto name "<Class>_<n>" or explicit
                   parameter key    = _fold_in_str(frame.rng, param_name)       ("kernel"; zeros / ones initialisers ignore theirs)
                   _fold_in_str(rng, s) = jax.random.fold_in(rng, uint32(first 4 bytes of sha1(s), big endian))
  jax.random       fold_in(key, d)  = threefry_2x32(key, PRNGKey(d)) = one block on the counter pair (0, d)
  initializers     lecun_normal()   = variance_scaling(1.0, "fan_in", "truncated_normal"):
                                      truncated_normal(key, -2, 2, shape) * sqrt(1 / fan_in) / 0.87962566103423978
                   truncated_normal = sqrt(2) * erf_inv(uniform(key, shape, minval=erf(-2/sqrt2), maxval=erf(2/sqrt2))),
                                      clipped to the open interval (nextafter(-2, +inf), nextafter(2, -inf))
  DenseGeneral     (attention q / k / v / out) draws its kernel with the FLATTENED shape (E, H*d) / (H*d, E), then reshapes.

The auto-naming rule is the one flax_io writes and detects ("shared": one counter for all children of a module, parameter-less
modules take a number too; the attention module is named ``MultiHeadDotProductAttention_<n>`` because ``nn.SelfAttention`` is a
``partial`` of that class and keeps its ``__name__`` -- see flax_io.py); both are arguments because they are the unverified
part: a different name folds a different key into every q / k / v / out kernel, which would still be a valid lecun_normal draw
but not the reference's.  Host NumPy, once per model: ~5 s for the 26.6 M parameters of the base network (callers that restore
a checkpoint next skip it: ``create_model(..., init=False)``).
"""
from __future__ import annotations

import hashlib
from typing import Dict, Tuple

import numpy as np

from . import flax_io as _fio
from .jax_random import ThreefryKey, _block

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_SQRT2 = np.float32(np.sqrt(2.0))
# float32 erf(-2 / sqrt(2)), erf(2 / sqrt(2)) (XLA evaluates lax.erf in float32; these are the correctly rounded values)
_ERF_LO, _ERF_HI = np.float32(-0.9544997361036416), np.float32(0.9544997361036416)
_TRUNC_STD = 0.87962566103423978


def fold_in(key: ThreefryKey, data: int) -> ThreefryKey:
    """jax.random.fold_in: threefry_2x32(key, PRNGKey(data)); the two output words are the new key."""
    y0, y1 = _block(key.k0, key.k1, 0, int(data) & 0xFFFFFFFF)
    return ThreefryKey(y0, y1)


def fold_in_str(key: ThreefryKey, s: str) -> ThreefryKey:
    """flax.nn.base._fold_in_str."""
    return fold_in(key, int.from_bytes(hashlib.sha1(s.encode("utf-8")).digest()[:4], byteorder="big"))


def _threefry_np(k0: int, k1: int, x0: np.ndarray, x1: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    ks = (np.uint32(k0), np.uint32(k1), np.uint32(k0 ^ k1 ^ 0x1BD11BDA))
    with np.errstate(over="ignore"):
        x0 = x0.astype(np.uint32) + ks[0]
        x1 = x1.astype(np.uint32) + ks[1]
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 += x1
                x1 = ((x1 << np.uint32(r)) | (x1 >> np.uint32(32 - r))) ^ x0
            x0 += ks[(i + 1) % 3]
            x1 += ks[(i + 2) % 3] + np.uint32(i + 1)
    return x0, x1


def random_bits(key: ThreefryKey, n: int) -> np.ndarray:
    """jax.random._random_bits(key, 32, (n,)): counters iota(n) split into halves (odd n: one zero pad), outputs concatenated."""
    h = (n + 1) // 2
    cnt = np.arange(2 * h, dtype=np.uint32)
    if n % 2:
        cnt[-1] = 0
    y0, y1 = _threefry_np(key.k0, key.k1, cnt[:h], cnt[h:])
    return np.concatenate([y0, y1])[:n]


_ERFINV_LT5 = (2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503, -0.00417768164,
               0.246640727, 1.50140941)
_ERFINV_GE5 = (-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613, 0.00943887047,
               1.00167406, 2.83297682)


def _erfinv_f32(x: np.ndarray) -> np.ndarray:
    """XLA's float32 erf_inv polynomial (the one csrc/rng_threefry.h evaluates on the device)."""
    f = np.float32
    w = -np.log((f(1) - x) * (f(1) + x), dtype=np.float32)
    lt = w < f(5)
    ww = np.where(lt, w - f(2.5), np.sqrt(np.maximum(w, f(0)), dtype=np.float32) - f(3)).astype(np.float32)
    p = np.where(lt, f(_ERFINV_LT5[0]), f(_ERFINV_GE5[0])).astype(np.float32)
    for a, b in zip(_ERFINV_LT5[1:], _ERFINV_GE5[1:]):
        p = np.where(lt, f(a), f(b)).astype(np.float32) + p * ww
    return (p * x).astype(np.float32)


def truncated_normal(key: ThreefryKey, n: int) -> np.ndarray:
    """jax.random.truncated_normal(key, -2, 2, (n,), float32)."""
    bits = random_bits(key, n)
    u01 = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    u = np.maximum(_ERF_LO, u01 * (_ERF_HI - _ERF_LO) + _ERF_LO).astype(np.float32)       # uniform(minval=a, maxval=b)
    out = _SQRT2 * _erfinv_f32(u)
    lo = np.nextafter(np.float32(-2), np.float32(np.inf), dtype=np.float32)
    hi = np.nextafter(np.float32(2), np.float32(-np.inf), dtype=np.float32)
    return np.clip(out, lo, hi).astype(np.float32)


def lecun_normal(key: ThreefryKey, shape: Tuple[int, int]) -> np.ndarray:
    """jax.nn.initializers.lecun_normal()(key, (fan_in, fan_out)) in float32."""
    fan_in = int(shape[0])
    std = np.float32(np.sqrt(np.float32(1.0) / np.float32(fan_in), dtype=np.float32) / np.float32(_TRUNC_STD))
    return (truncated_normal(key, int(shape[0]) * int(shape[1])) * std).astype(np.float32).reshape(shape)


def init_params(cfg, model_rng: ThreefryKey, template: Dict[str, Tuple[int, ...]], rule: str = "shared",
                attention_class: str = "MultiHeadDotProductAttention") -> Dict[str, np.ndarray]:
    """Engine-named initial parameters of ``create_model(model_rng, ...)`` (train_ncsn.py:193-203) for the network ``cfg``.
    ``template``: engine tensor name -> shape (Engine.tensor_table / the oracle's param_spec)."""
    out = {k: np.zeros(s, dtype=np.float32) for k, s in template.items()}
    for k in out:
        if k.endswith(".scale"):
            out[k][...] = 1.0
    for path, our, shape, cols in _fio._walk(_fio.module_tree(cfg, attention_class), rule):
        if path[-1] != "kernel":
            continue
        key = model_rng
        for name in path:                           # module names, then the parameter name
            key = fold_in_str(key, name)
        if len(shape) == 3:                         # DenseGeneral: q / k / v (E, H, d) drawn as (E, H*d); out (H, d, E) as (H*d, E)
            flat = (int(shape[0]), int(shape[1] * shape[2])) if path[-2] != "out" else (int(shape[0] * shape[1]), int(shape[2]))
        else:
            flat = (int(shape[0]), int(shape[1]))
        w = lecun_normal(key, flat)
        dst = out[our]
        if cols is None:
            dst[...] = w.reshape(dst.shape)
        else:
            dst[..., cols[0]:cols[1]] = w.reshape(dst.shape[:-1] + (cols[1] - cols[0],))
    return out
