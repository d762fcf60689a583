"""jax.random-compatible keys and draws (jax 0.2.8 threefry2x32 conventions) for the DDPM path.

With a ``ThreefryKey`` the engine consumes exactly the random streams the reference does:

  train_ncsn.py:318-319,358      PRNGKey(seed) -> split(num=3) -> per-step split
  utils/losses.py:271-294        split(rng, 3); labels = randint(label_rng); eps = normal(sample_rng)
  train_ncsn.py:536-540          init_rng, ld_rng = split(rng); init = normal(init_rng, (N, S, C))
  utils/ebm_utils.py:329-362     three splits per iteration; z = normal(noise_rng), infill noise = normal(infill_rng)

Key algebra (a handful of Threefry blocks per call) runs on the host in plain Python integers; every array draw
runs on the device through the C-ABI (csrc/rng_jax.hip, csrc/rng_threefry.h).  The algorithm is pinned by the
Random123 vectors and by the keys / normals printed in JAX's documentation (tests/golden/jax_random_kat.json).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as _lib

_M32 = 0xFFFFFFFF
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


@dataclass(frozen=True)
class ThreefryKey:
    """A jax.random key: two uint32 words."""
    k0: int
    k1: int

    def __post_init__(self):
        object.__setattr__(self, "k0", int(self.k0) & _M32)
        object.__setattr__(self, "k1", int(self.k1) & _M32)

    @property
    def seed(self) -> int:            # lets a ThreefryKey seed the engine's Philox streams too (never used by this module)
        return (self.k0 << 32) | self.k1

    def __iter__(self):
        return iter((self.k0, self.k1))


def PRNGKey(seed: int) -> ThreefryKey:
    """jax.random.PRNGKey with 32-bit integers (x64 off): [0, seed mod 2^32]."""
    return ThreefryKey(0, int(seed) & _M32)


def _block(k0: int, k1: int, x0: int, x1: int) -> Tuple[int, int]:
    ks = (k0, k1, k0 ^ k1 ^ 0x1BD11BDA)
    x0 = (x0 + ks[0]) & _M32
    x1 = (x1 + ks[1]) & _M32
    for i in range(5):
        for r in _ROT[i % 2]:
            x0 = (x0 + x1) & _M32
            x1 = (((x1 << r) | (x1 >> (32 - r))) & _M32) ^ x0
        x0 = (x0 + ks[(i + 1) % 3]) & _M32
        x1 = (x1 + ks[(i + 2) % 3] + i + 1) & _M32
    return x0, x1


def random_bits_host(key: ThreefryKey, n: int) -> List[int]:
    """random_bits(key, 32, (n,)) for small n (key algebra): counter halves x0 | x1, outputs y0 | y1."""
    h = (n + 1) // 2
    out = [0] * (2 * h)
    for j in range(h):
        y0, y1 = _block(key.k0, key.k1, j, j + h if j + h < n else 0)
        out[j], out[h + j] = y0, y1
    return out[:n]


def split(key: ThreefryKey, num: int = 2) -> Tuple[ThreefryKey, ...]:
    """jax.random.split."""
    b = random_bits_host(key, 2 * num)
    return tuple(ThreefryKey(b[2 * j], b[2 * j + 1]) for j in range(num))


# ------------------------------------------------------------------ device draws
def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _window(shape: Sequence[int], n_total: Optional[int], offset: int) -> Tuple[int, int, int]:
    count = int(np.prod(shape))
    total = count if n_total is None else int(n_total)
    if offset < 0 or offset + count > total:
        raise ValueError(f"window offset={offset} count={count} outside n_total={total}")
    return total, int(offset), count


def normal(key: ThreefryKey, shape: Sequence[int], device, *, n_total: Optional[int] = None, offset: int = 0,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """jax.random.normal(key, shape_global)[window] as float32 on ``device``; the window is the flat element range
    [offset, offset + prod(shape)) of a logical array with n_total elements (default: the whole array)."""
    total, off, count = _window(shape, n_total, offset)
    if out is None:
        out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.get_lib().smd_threefry_normal(out.data_ptr(), total, off, count, key.k0, key.k1, None, None, 0, 0,
                                                      _stream()), "threefry_normal")
    return out


def uniform(key: ThreefryKey, shape: Sequence[int], device, minval: float = 0.0, maxval: float = 1.0, *,
            n_total: Optional[int] = None, offset: int = 0) -> torch.Tensor:
    total, off, count = _window(shape, n_total, offset)
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.get_lib().smd_threefry_uniform(out.data_ptr(), total, off, count, key.k0, key.k1, float(minval),
                                                       float(maxval), _stream()), "threefry_uniform")
    return out


def randint(key: ThreefryKey, shape: Sequence[int], minval: int, maxval: int, device, *, n_total: Optional[int] = None,
            offset: int = 0) -> torch.Tensor:
    """jax.random.randint(key, shape, minval, maxval) (int32)."""
    total, off, count = _window(shape, n_total, offset)
    out = torch.empty(tuple(shape), dtype=torch.int32, device=device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.get_lib().smd_threefry_randint(out.data_ptr(), total, off, count, key.k0, key.k1, int(minval),
                                                       int(maxval), _stream()), "threefry_randint")
    return out


def bits(key: ThreefryKey, n: int, device) -> torch.Tensor:
    out = torch.empty((n,), dtype=torch.int32, device=device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.get_lib().smd_threefry_bits(out.data_ptr(), n, 0, n, key.k0, key.k1, _stream()), "threefry_bits")
    return out


# ------------------------------------------------------------------ the reference's draw sequences
def diffusion_loss_draws(rng: ThreefryKey, local_shape: Sequence[int], num_sigmas: int, device, *,
                         continuous_noise: bool = True, sample_offset: int = 0,
                         global_batch: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(labels, eps) of utils/losses.py:271-294 for this rank's rows [sample_offset, sample_offset + B) of a
    global batch.  With continuous_noise=True the uniform of :283-286 is degenerate (minval > maxval returns minval)
    and draws nothing that matters: the engine's label -> alpha-bar table already holds alphas_prod_ext[label - 1].
    With continuous_noise=False label 0 is possible and its uniform is real: see ``diffusion_loss_used_alphas``."""
    _rng, label_rng, sample_rng = split(rng, 3)
    B = int(local_shape[0])
    gb = B if global_batch is None else int(global_batch)
    per = int(np.prod(local_shape[1:]))
    labels = randint(label_rng, (B,), int(continuous_noise), num_sigmas + int(continuous_noise), device, n_total=gb,
                     offset=sample_offset)
    eps = normal(sample_rng, tuple(local_shape), device, n_total=gb * per, offset=sample_offset * per)
    return labels, eps


def diffusion_loss_used_alphas(rng: ThreefryKey, labels: torch.Tensor, alphas_prod_ext: torch.Tensor, *,
                               sample_offset: int = 0, global_batch: Optional[int] = None) -> torch.Tensor:
    """used_alphas of utils/losses.py:282-286 for this rank's labels: ``rng, noise_rng = split(rng)`` on the first
    output of the 3-way split, then uniform(noise_rng, (B,), minval=alphas_prod'[labels - 1], maxval=alphas_prod'[labels])
    = max(minval, u01 * (maxval - minval) + minval).  labels - 1 = -1 wraps to the last entry (alphas_prod[T]) exactly as
    the jnp indexing does, which is what makes label 0 a real draw in [alphas_prod[T], 1)."""
    first, _label_rng, _sample_rng = split(rng, 3)
    _rng, noise_rng = split(first)
    B = int(labels.shape[0])
    gb = B if global_batch is None else int(global_batch)
    u = uniform(noise_rng, (B,), labels.device, 0.0, 1.0, n_total=gb, offset=sample_offset)
    T1 = int(alphas_prod_ext.shape[0])
    lab = labels.long()
    lo = alphas_prod_ext[(lab - 1) % T1]
    hi = alphas_prod_ext[lab % T1]
    return torch.maximum(lo, u * (hi - lo) + lo).contiguous()


def dsm_loss_draws(rng: ThreefryKey, local_shape: Sequence[int], sigmas: torch.Tensor, *, continuous_noise: bool = False,
                   sample_offset: int = 0, global_batch: Optional[int] = None):
    """(labels, used_sigmas, eps) of denoising_score_matching_loss (utils/losses.py:149-164): labels =
    randint(label_rng, int(continuous_noise), len(sigmas)); used_sigmas = sigmas[labels], or with continuous noise
    uniform(noise_rng, minval=sigmas[labels - 1], maxval=sigmas[labels]) (its minval when the schedule decreases)."""
    _rng, label_rng, sample_rng = split(rng, 3)
    B = int(local_shape[0])
    gb = B if global_batch is None else int(global_batch)
    per = int(np.prod(local_shape[1:]))
    dev = sigmas.device
    labels = randint(label_rng, (B,), int(continuous_noise), int(sigmas.shape[0]), dev, n_total=gb, offset=sample_offset)
    if continuous_noise:
        used = diffusion_loss_used_alphas(rng, labels, sigmas, sample_offset=sample_offset, global_batch=gb)
    else:
        used = sigmas[labels.long()].contiguous()
    eps = normal(sample_rng, tuple(local_shape), dev, n_total=gb * per, offset=sample_offset * per)
    return labels, used, eps


def langevin_key_table(ld_rng: ThreefryKey, iterations: int, consistent: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """Per-update (step_rng, infill_rng) of utils/ebm_utils.py:133 (``rng, step_rng, infill_rng = split(rng, 3)``) or, for
    the consistent sampler, step_rng of :233 (``rng, step_rng = split(rng)``), as uint32 [iterations, 2] tables."""
    step = np.zeros((iterations, 2), dtype=np.uint32)
    infill = np.zeros((iterations, 2), dtype=np.uint32)
    rng = ld_rng
    for i in range(iterations):
        if consistent:
            rng, s = split(rng)
        else:
            rng, s, f = split(rng, 3)
            infill[i] = (f.k0, f.k1)
        step[i] = (s.k0, s.k1)
    return step, infill


def sampler_key_tables(ld_rng: ThreefryKey, iterations: int) -> Tuple[np.ndarray, np.ndarray]:
    """Per-iteration (infill_noise_rng, noise_rng) of utils/ebm_utils.py:329,342,360 as uint32 [iterations, 2]
    tables, row i = i-th call of sample_with_beta (t = T-1-i)."""
    infill = np.zeros((iterations, 2), dtype=np.uint32)
    noise = np.zeros((iterations, 2), dtype=np.uint32)
    rng = ld_rng
    for i in range(iterations):
        rng, _key = split(rng)
        rng, infill_rng = split(rng)
        rng, noise_rng = split(rng)
        infill[i] = (infill_rng.k0, infill_rng.k1)
        noise[i] = (noise_rng.k0, noise_rng.k1)
    return infill, noise


def fill_normal_from_table(out: torch.Tensor, table: torch.Tensor, t_ptr: torch.Tensor, num_sigmas: int, *,
                           n_total: Optional[int] = None, offset: int = 0) -> None:
    """out <- normal(table[(num_sigmas - 1) - *t_ptr]) window; key and t are read on the device (graph-safe)."""
    total, off, count = _window(out.shape, n_total, offset)
    with torch.cuda.device(out.device):
        _lib.check(_lib.get_lib().smd_threefry_normal(out.data_ptr(), total, off, count, 0, 0, table.data_ptr(),
                                                      t_ptr.data_ptr(), -1, num_sigmas - 1, _stream()), "threefry_normal")
