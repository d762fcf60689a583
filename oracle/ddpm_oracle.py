"""CPU restatement ORACLE of the reference DDPM hot path  --  TEST INFRASTRUCTURE ONLY.

This file is the *checker*.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``symbolic-music-diffusion_amd/``) never imports anything from ``oracle/`` and
fails loudly when the HIP library is missing.

PARITY UNPINNED.  The reference (magenta/symbolic-music-diffusion @ v1) ships no
tests, no golden vectors and no fixtures for this path, and its arithmetic lives in
un-vendored third-party packages that are not installable here (flax==0.3.0,
jax==0.2.8, jaxlib==0.1.57, requirements.txt:47,94,95).  This restatement follows
the reference's own call sites line by line and restates the published flax/jax
layer semantics (the ORACLE_ASSUMPTIONS table below); it is pinned only by the
closed-form known-answer tests in tests/test_oracle.py.

Exception -- PINNED: the jax.random restatement at the end of this file (threefry2x32,
PRNGKey / split / random_bits layout, uniform, normal) reproduces published known
answers: the Random123 threefry2x32 vectors and the keys / normal draws printed in
JAX's documentation (tests/golden/jax_random_kat.json).  randint stays unpinned.

Every function cites the reference file:line it follows (paths relative to the
reference repo root).  Arithmetic is torch-CPU; dtype is a parameter (float64 for
parity checks, float32 for the timed CPU baseline).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

# ----------------------------------------------------------------------------------
# Third-party semantics restated from flax 0.3.0 / jax 0.2.8 (not in /root/reference).
# One table so each entry can be flipped if a JAX environment ever contradicts it.
# ----------------------------------------------------------------------------------
ORACLE_ASSUMPTIONS = {
    "dense_kernel_layout": "(in, out); y = x @ kernel + bias",  # flax.nn.Dense
    "dense_init": "lecun_normal: truncated normal, std = sqrt(1/fan_in)/0.87962566",
    "layernorm_eps": 1e-6,  # flax.nn.LayerNorm default epsilon
    "layernorm_var": "biased; E[x^2] - E[x]^2",
    "attention_q_scaling": "query / sqrt(head_dim) before the logits",
    "attention_softmax_axis": "keys",
    "gelu": "tanh approximation",
    "adam": dict(beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0),
    "clip_grads": "g * max_norm / norm  iff  norm >= max_norm (global L2 over leaves)",
    "stepped_lr": "lr0 * gamma**max(0, ceil(step / interval) - 1)",
    "uniform_minval_gt_maxval": "jax 0.2.8 uniform ends with max(minval, .) -> minval",
}

LN_EPS = 1e-6


# ----------------------------------------------------------------------------------
# Configuration of the eps-network
# ----------------------------------------------------------------------------------
@dataclass
class NetConfig:
    """Model kwargs as built at train_ncsn.py:321-326 plus the data shape."""
    architecture: str = "TransformerDDPM"
    data_channels: int = 512          # C  (inputs.shape[-1])
    seq_len: int = 32                 # S  (TransformerDDPM only)
    num_layers: int = 6               # train_ncsn.py:69
    num_heads: int = 8                # train_ncsn.py:70
    num_mlp_layers: int = 2           # train_ncsn.py:71
    mlp_dims: int = 2048              # train_ncsn.py:72
    embed_channels: int = 128         # hard-coded, models/ncsn.py:151
    film_channels: int = 128          # DenseFiLM(t, 128, mlp_dims), models/ncsn.py:174


def param_spec(cfg: NetConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (name, shape) list of every trainable tensor.

    Stable explicit names; shapes are the flax layouts:
    Dense kernel (in,out); attention q/k/v kernels (E,H,d) are kept flattened and
    concatenated as (E, 3*H*d) [query | key | value], out kernel (H,d,E) flattened
    to (H*d, E); FiLM scale/shift Dense kernels concatenated as (4F, 2M)
    [scale | shift].
    """
    C, E, M, F = cfg.data_channels, cfg.embed_channels, cfg.mlp_dims, cfg.film_channels
    spec: List[Tuple[str, Tuple[int, ...]]] = []

    def dense(name, i, o):
        spec.append((name + ".kernel", (i, o)))
        spec.append((name + ".bias", (o,)))

    def ln(name, d):
        spec.append((name + ".scale", (d,)))
        spec.append((name + ".bias", (d,)))

    def film_and_res(prefix_f, prefix_r):
        dense(prefix_f + ".fc1", F, 4 * F)          # models/ncsn.py:53
        dense(prefix_f + ".fc2", 4 * F, 4 * F)      # models/ncsn.py:55
        dense(prefix_f + ".ss", 4 * F, 2 * M)       # models/ncsn.py:60-61 (scale|shift)
        ln(prefix_r + ".ln1", M)                    # models/shared.py:62
        dense(prefix_r + ".fc1", M, M)              # models/shared.py:65
        ln(prefix_r + ".ln2", M)                    # models/shared.py:66
        dense(prefix_r + ".fc2", M, M)              # models/shared.py:69

    if cfg.architecture in ("TransformerDDPM", "TransformerDDPM4"):
        dense("in_proj", C, E)                      # models/ncsn.py:155
        for l in range(cfg.num_layers):
            p = f"enc.{l}"
            ln(p + ".ln1", E)                       # models/ncsn.py:160
            dense(p + ".attn.qkv", E, 3 * E)        # models/ncsn.py:161 (SelfAttention)
            dense(p + ".attn.out", E, E)
            ln(p + ".ln2", E)                       # models/ncsn.py:164
            dense(p + ".mlp.fc1", E, M)             # models/ncsn.py:165
            dense(p + ".mlp.fc2", M, E)             # models/ncsn.py:167
        ln("ln_f", E)                               # models/ncsn.py:170
        dense("up", E, M)                           # models/ncsn.py:171
        for k in range(cfg.num_mlp_layers):
            film_and_res(f"film.{k}", f"res.{k}")   # models/ncsn.py:173-175
        ln("ln_o", M)                               # models/ncsn.py:177
        dense("out_proj", M, C)                     # models/ncsn.py:178
    elif cfg.architecture == "DenseDDPM":
        dense("in_proj", C, M)                      # models/ncsn.py:129
        for k in range(cfg.num_layers):
            film_and_res(f"film.{k}", f"res.{k}")   # models/ncsn.py:130-132
        ln("ln_o", M)                               # models/ncsn.py:133
        dense("out_proj", M, C)                     # models/ncsn.py:134
    else:
        raise ValueError(f"unsupported architecture {cfg.architecture}")
    return spec


def num_params(cfg: NetConfig) -> int:
    return int(sum(int(np.prod(s)) for _, s in param_spec(cfg)))


def init_params(cfg: NetConfig, seed: int = 0, dtype=torch.float64) -> Dict[str, torch.Tensor]:
    """lecun-normal kernels, zero biases, unit LayerNorm scales
    (ORACLE_ASSUMPTIONS 1/2; reference: train_ncsn.py:193-199 init_by_shape).
    NumPy default_rng(seed) so the same file can be produced anywhere."""
    rng = np.random.default_rng(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in param_spec(cfg):
        if name.endswith(".kernel"):
            fan_in = shape[0]
            std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
            # truncated normal on [-2, 2] by rejection (what jax.random.truncated_normal draws)
            w = rng.standard_normal(size=shape)
            bad = np.abs(w) > 2.0
            while bad.any():
                w[bad] = rng.standard_normal(size=int(bad.sum()))
                bad = np.abs(w) > 2.0
            out[name] = torch.from_numpy(w * std).to(dtype)
        elif name.endswith(".scale"):
            out[name] = torch.ones(shape, dtype=dtype)
        else:
            out[name] = torch.zeros(shape, dtype=dtype)
    return out


# ----------------------------------------------------------------------------------
# Layers (L0/L1)
# ----------------------------------------------------------------------------------
def dense(x, p, name):
    """flax.nn.Dense: x @ kernel + bias (call sites models/ncsn.py:53-61,155,165-171,178)."""
    return x @ p[name + ".kernel"] + p[name + ".bias"]


def layer_norm(x, p, name):
    """flax.nn.LayerNorm over the last axis, eps 1e-6, biased variance as E[x^2]-E[x]^2
    (call sites models/ncsn.py:160,164,170,177; models/shared.py:62,66)."""
    mean = x.mean(dim=-1, keepdim=True)
    mean2 = (x * x).mean(dim=-1, keepdim=True)
    var = mean2 - mean * mean
    y = (x - mean) * torch.rsqrt(var + LN_EPS)
    return y * p[name + ".scale"] + p[name + ".bias"]


def gelu(x):
    """flax.nn.gelu (tanh approximation), models/ncsn.py:166."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def swish(x):
    """flax.nn.swish, models/ncsn.py:54, models/shared.py:64,68."""
    return x * torch.sigmoid(x)


def _sinusoid_freqs(half_dim: int, dtype):
    # models/shared.py:41-42 == models/ncsn.py:34-35
    emb = math.log(10000.0) / float(half_dim - 1)
    return torch.exp(torch.arange(half_dim, dtype=dtype) * -emb)


def positional_encoding(seq_len: int, channels: int, dtype=torch.float64):
    """TransformerPositionalEncoding.apply, models/shared.py:36-48: [sin | cos] halves."""
    f = _sinusoid_freqs(channels // 2, dtype)
    emb = torch.arange(seq_len, dtype=dtype)[:, None] * f[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if channels % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb


def noise_encoding(noise, channels: int):
    """NoiseEncoding.apply, models/ncsn.py:28-41. noise: (B,1) noise level sqrt(alpha_bar)."""
    noise = noise.squeeze(-1)
    assert noise.dim() == 1
    f = _sinusoid_freqs(channels // 2, noise.dtype)
    emb = 5000.0 * noise[:, None] * f[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if channels % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb


def self_attention(x, p, name, num_heads: int):
    """flax.nn.SelfAttention(x, num_heads) as called at models/ncsn.py:161: no mask, no
    dropout, qkv_features = out_features = E; q scaled by 1/sqrt(d) before the logits."""
    B, S, E = x.shape
    d = E // num_heads
    qkv = dense(x, p, name + ".qkv")                       # (B,S,3E)  [q | k | v], each (H,d)
    q, k, v = qkv.split(E, dim=-1)
    q = q.reshape(B, S, num_heads, d) / math.sqrt(d)
    k = k.reshape(B, S, num_heads, d)
    v = v.reshape(B, S, num_heads, d)
    logits = torch.einsum("bqhd,bkhd->bhqk", q, k)
    w = torch.softmax(logits, dim=-1)
    o = torch.einsum("bhqk,bkhd->bqhd", w, v).reshape(B, S, E)
    return dense(o, p, name + ".out")


def dense_film(position, p, name, film_channels: int, mlp_dims: int, sequence: bool):
    """DenseFiLM.apply, models/ncsn.py:47-61.  Returns (scale, shift)."""
    assert position.dim() == 2
    e = noise_encoding(position, film_channels)
    e = dense(e, p, name + ".fc1")
    e = swish(e)
    e = dense(e, p, name + ".fc2")
    if sequence:
        e = e[:, None, :]
    ss = dense(e, p, name + ".ss")
    return ss[..., :mlp_dims], ss[..., mlp_dims:]


def dense_res_block(x, p, name, scale, shift):
    """DenseResBlock.apply, models/shared.py:61-75 (FeaturewiseAffine :54-55 = scale*x+shift).
    Input and output widths are equal on this path so the shortcut is the identity (:72-73)."""
    o = layer_norm(x, p, name + ".ln1")
    o = swish(scale * o + shift)
    o = dense(o, p, name + ".fc1")
    o = layer_norm(o, p, name + ".ln2")
    o = swish(scale * o + shift)
    o = dense(o, p, name + ".fc2")
    return o + x


def transformer_ddpm(p, cfg: NetConfig, inputs, t):
    """TransformerDDPM.apply, models/ncsn.py:141-179. inputs (B,S,C); t (B,1,1)."""
    B, S, C = inputs.shape
    E = cfg.embed_channels
    temb = positional_encoding(S, E, inputs.dtype)[None]          # :152-153
    x = dense(inputs, p, "in_proj")                               # :155
    x = x + temb                                                  # :157
    for l in range(cfg.num_layers):                               # :158-168
        pre = f"enc.{l}"
        shortcut = x
        x = layer_norm(x, p, pre + ".ln1")
        x = self_attention(x, p, pre + ".attn", cfg.num_heads)
        x = x + shortcut
        shortcut2 = x
        x = layer_norm(x, p, pre + ".ln2")
        x = dense(x, p, pre + ".mlp.fc1")
        x = gelu(x)
        x = dense(x, p, pre + ".mlp.fc2")
        x = x + shortcut2
    x = layer_norm(x, p, "ln_f")                                  # :170
    x = dense(x, p, "up")                                         # :171
    for k in range(cfg.num_mlp_layers):                           # :173-175
        scale, shift = dense_film(t.squeeze(-1), p, f"film.{k}", cfg.film_channels,
                                  cfg.mlp_dims, sequence=True)
        x = dense_res_block(x, p, f"res.{k}", scale, shift)
    x = layer_norm(x, p, "ln_o")                                  # :177
    return dense(x, p, "out_proj")                                # :178


def dense_ddpm(p, cfg: NetConfig, inputs, t):
    """DenseDDPM.apply, models/ncsn.py:125-135. inputs (B,C); t (B,1).  num_heads /
    num_mlp_layers kwargs passed by train_ncsn.py:321-326 are accepted and ignored."""
    x = dense(inputs, p, "in_proj")
    for k in range(cfg.num_layers):
        scale, shift = dense_film(t, p, f"film.{k}", cfg.film_channels, cfg.mlp_dims,
                                  sequence=False)
        x = dense_res_block(x, p, f"res.{k}", scale, shift)
    x = layer_norm(x, p, "ln_o")
    return dense(x, p, "out_proj")


def make_model(p, cfg: NetConfig) -> Callable:
    """The nn.Model callable of the reference: model(x, cond) -> eps_hat."""
    if cfg.architecture == "DenseDDPM":
        return lambda x, t: dense_ddpm(p, cfg, x, t)
    return lambda x, t: transformer_ddpm(p, cfg, x, t)


# ----------------------------------------------------------------------------------
# Noise schedule (L2)
# ----------------------------------------------------------------------------------
def create_noise_schedule(sigma_begin=1.0, sigma_end=1e-2, L=10, schedule="geometric"):
    """utils/ebm_utils.py:62-86.  jnp defaults to float32, so the table is float32."""
    if schedule == "geometric":
        s = np.exp(np.linspace(np.log(np.float32(sigma_begin)), np.log(np.float32(sigma_end)),
                               L, dtype=np.float32))
    elif schedule == "linear":
        s = np.linspace(np.float32(sigma_begin), np.float32(sigma_end), L, dtype=np.float32)
    elif schedule == "fibonacci":
        v = [1e-6, 2e-6]
        for _ in range(L - 2):
            v.append(v[-1] + v[-2])
        s = np.array(v, dtype=np.float32)
    else:
        raise ValueError(f"Unsupported schedule: {schedule}")
    return s.astype(np.float32)


def alphas_cumprod(betas: np.ndarray) -> np.ndarray:
    """alphas_prod = cumprod(1 - betas) in float32 (utils/ebm_utils.py:315-316, losses.py:277-278)."""
    return np.cumprod((np.float32(1.0) - betas.astype(np.float32)).astype(np.float32),
                      dtype=np.float32)


# ----------------------------------------------------------------------------------
# Objective (L2): diffusion_loss, utils/losses.py:250-308
# ----------------------------------------------------------------------------------
def reduce_fn(x, mode):
    """utils/losses.py:22-30."""
    if mode == "none" or mode is None:
        return x
    if mode == "sum":
        return x.sum()
    if mode == "mean":
        return x.mean()
    raise ValueError("Unsupported reduction option.")


def used_alphas_from_labels(betas: np.ndarray, labels: np.ndarray, u01: Optional[np.ndarray] = None) -> np.ndarray:
    """utils/losses.py:277-286 with the jax-0.2.8 uniform quirk: minval=alphas_prod'[l-1] >
    maxval=alphas_prod'[l] so uniform(...) returns minval exactly (ORACLE_ASSUMPTIONS).  Label 0 (continuous_noise=False
    only) indexes alphas_prod'[-1] = alphas_prod[T] < alphas_prod'[0] = 1: a real draw max(lo, u01 * (hi - lo) + lo) in
    float32, ``u01`` being the [0, 1) floats of jax.random.uniform (required when a label is 0)."""
    ap = np.concatenate([np.ones((1,), np.float32), alphas_cumprod(betas)]).astype(np.float32)
    labels = np.asarray(labels)
    lo, hi = ap[labels - 1], ap[labels % len(ap)]
    if u01 is None:
        if (labels == 0).any():
            raise ValueError("label 0 needs its uniform draw (u01)")
        return lo
    u01 = np.asarray(u01, dtype=np.float32)
    return np.maximum(lo, (u01 * (hi - lo)).astype(np.float32) + lo).astype(np.float32)


def diffusion_loss(batch, model, betas, labels, eps, reduction="mean", u01=None):
    """utils/losses.py:250-308 with the random draws (labels :272-275, eps :294, and for continuous_noise=False the
    uniform floats of :283-286) passed in explicitly.  Conditions on sqrt(alpha)."""
    B = batch.shape[0]
    a = torch.from_numpy(used_alphas_from_labels(betas, labels, u01)).to(batch.dtype)
    a = a.reshape(B, *([1] * (batch.dim() - 1)))
    perturbed = torch.sqrt(a) * batch + torch.sqrt(1 - a) * eps          # :295-296
    pred = model(perturbed, torch.sqrt(a))                                # :299-300
    loss = (eps - pred) ** 2                                              # :304
    loss = loss.mean(dim=tuple(range(1, loss.dim())))                     # :305
    assert loss.shape == batch.shape[:1]
    return reduce_fn(loss, reduction)


# ----------------------------------------------------------------------------------
# Optimisation step (L3): train_ncsn.py:260-288, 340-342; utils/train_utils.py:73-78
# ----------------------------------------------------------------------------------
def stepped_lr(lr0: float, step: int, interval: int, gamma: float) -> float:
    """flax.training.lr_schedule.create_stepped_learning_rate_schedule as used at
    train_ncsn.py:340-342 with lr_step_schedule=[(i, gamma**i)]: boundaries i*interval,
    index = sum(boundaries < step)  ->  lr0 * gamma**max(0, ceil(step/interval)-1)."""
    idx = max(0, int(math.ceil(step / interval)) - 1)
    return lr0 * (gamma ** idx)


@dataclass
class AdamState:
    step: int = 0
    m: Dict[str, torch.Tensor] = field(default_factory=dict)
    v: Dict[str, torch.Tensor] = field(default_factory=dict)


def clip_grads(grads: Dict[str, torch.Tensor], max_norm: float):
    """jax.experimental.optimizers.clip_grads / l2_norm (train_ncsn.py:284-285).
    Returns (clipped grads, norm AFTER clipping as logged by the reference)."""
    norm = torch.sqrt(sum((g * g).sum() for g in grads.values()))
    if float(norm) < max_norm:
        out = grads
    else:
        out = {k: g * (max_norm / norm) for k, g in grads.items()}
    norm_after = torch.sqrt(sum((g * g).sum() for g in out.values()))
    return out, norm_after


def adam_update(p: Dict[str, torch.Tensor], g: Dict[str, torch.Tensor], st: AdamState, lr: float,
                b1=0.9, b2=0.999, eps=1e-8):
    """flax.optim.Adam.apply_param_gradient (train_ncsn.py:287): t = step+1;
    m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps)."""
    t = st.step + 1
    new_p = {}
    for k in p:
        m = st.m.get(k, torch.zeros_like(p[k]))
        v = st.v.get(k, torch.zeros_like(p[k]))
        m = b1 * m + (1 - b1) * g[k]
        v = b2 * v + (1 - b2) * g[k] * g[k]
        mhat = m / (1 - b1 ** t)
        vhat = v / (1 - b2 ** t)
        new_p[k] = p[k] - lr * mhat / (torch.sqrt(vhat) + eps)
        st.m[k], st.v[k] = m, v
    st.step = t
    return new_p


def ema_update(ema: Dict[str, torch.Tensor], p: Dict[str, torch.Tensor], mu: float):
    """EMAHelper.update, utils/train_utils.py:73-78."""
    return {k: ema[k] * mu + p[k] * (1 - mu) for k in ema}


def train_step(p, cfg: NetConfig, st: AdamState, batch, betas, labels, eps, lr: float,
               grad_clip: float = 1.0):
    """train_ncsn.py:260-288: value_and_grad(mean loss) -> clip_grads -> Adam.
    Returns (new params, metrics{'loss','grad','lr'}, raw grads)."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    loss = diffusion_loss(batch, make_model(leaf, cfg), betas, labels, eps, "mean")
    loss.backward()
    grads = {k: v.grad.detach() for k, v in leaf.items()}
    clipped, norm_after = clip_grads(grads, grad_clip)
    new_p = adam_update({k: v.detach() for k, v in leaf.items()}, clipped, st, lr)
    return new_p, {"loss": float(loss), "grad": float(norm_after), "lr": lr}, grads


# ----------------------------------------------------------------------------------
# Reverse sampler (L3): diffusion_dynamics, utils/ebm_utils.py:280-405
# ----------------------------------------------------------------------------------
COLLECTION_STEPS = 40   # utils/ebm_utils.py:320


def collection_index_table(T: int) -> np.ndarray:
    """collection_idx = linspace(1, T, 40).astype(int32), utils/ebm_utils.py:324-325 (float32)."""
    return np.linspace(np.float32(1), np.float32(T), COLLECTION_STEPS,
                       dtype=np.float32).astype(np.int32)


def collection_slot_for_t(T: int, t: int, table: Optional[np.ndarray] = None) -> int:
    """utils/ebm_utils.py:387-394: image_idx = T - t + 1; slot = index in table + 1, or -1."""
    table = collection_index_table(T) if table is None else table
    image_idx = T - t + 1
    hit = np.nonzero(table == image_idx)[0]
    if hit.size == 0:
        return -1
    slot = int(np.sum(hit)) + 1   # sum(arange*mask)+1; T < 40 repeats entries and the sum can pass the last row:
    return slot if slot <= COLLECTION_STEPS else -1   # index_update out of bounds is dropped by XLA's scatter


def reverse_coefficients(betas: np.ndarray) -> Dict[str, np.ndarray]:
    """Per-t constants of sample_with_beta, utils/ebm_utils.py:332-358, in float32 like jnp."""
    f = np.float32
    betas = betas.astype(np.float32)
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
    return dict(alpha_prod=ap, sqrt_alpha_prod=np.sqrt(ap, dtype=np.float32),
                sqrt_one_minus=np.sqrt(f(1) - ap, dtype=np.float32),
                sqrt_recip=sqrt_recip, sqrt_m1=sqrt_m1, mu1=mu1, mu2=mu2, sigma=sigma)


def _norm_metric(v):
    # utils/ebm_utils.py:381-383: sqrt(sum(v^2, axis=1) + 1e-10).mean()  (axis 1 = sequence axis)
    return torch.sqrt((v * v).sum(dim=1) + 1e-10).mean()


def diffusion_dynamics(model, betas: np.ndarray, init, noises, infill=False,
                       infill_samples=None, infill_masks=None, infill_noises=None,
                       t_start: Optional[int] = None, t_stop: int = 0):
    """utils/ebm_utils.py:280-405 with the per-step normal draws passed in explicitly:
    ``noises(t)`` returns z_t of init.shape (the reference draws it at :360-362).
    Returns (state, collection (41,...), ld_metrics (4,T,1)) exactly shaped like the reference.
    ``t_start/t_stop`` allow short teacher-forced rollouts (default: full T-1 .. 0)."""
    T = len(betas)
    dt = init.dtype
    co = {k: torch.from_numpy(v).to(dt) for k, v in reverse_coefficients(betas).items()}
    if not infill:
        infill_samples = torch.zeros_like(init)
        infill_masks = torch.zeros_like(init)
    table = collection_index_table(T)
    start = init * (1 - infill_masks) + infill_samples * infill_masks          # :321
    collection = torch.zeros((COLLECTION_STEPS + 1, *init.shape), dtype=dt)     # :322
    collection[0] = start                                                       # :323
    metrics = torch.zeros((4, T, 1), dtype=dt)
    state = init
    t_hi = T - 1 if t_start is None else t_start
    for i, t in enumerate(range(t_hi, t_stop - 1, -1)):                         # :400-401
        # infill template :342-348
        if infill:
            inz = infill_noises(t)
            noisy_y = co["sqrt_alpha_prod"][t] * infill_samples + co["sqrt_one_minus"][t] * inz
            y = noisy_y if t > 0 else infill_samples
        else:
            y = infill_samples
        z = noises(t) if t > 0 else torch.zeros_like(state)                     # :360-363
        z = z * co["sigma"][t]                                                  # :364
        cond = (co["sqrt_alpha_prod"][t] * torch.ones((state.shape[0], 1), dtype=dt)).reshape(
            state.shape[0], *([1] * (state.dim() - 1)))                         # :367-369
        eps_recon = model(state, cond)                                          # :370
        recon = co["sqrt_recip"][t] * state - co["sqrt_m1"][t] * eps_recon      # :371
        recon = torch.clamp(recon, -1.0, 1.0)                                   # :372
        mu = co["mu1"][t] * recon + co["mu2"][t] * state                        # :373
        nxt = mu + z                                                            # :374
        nxt = nxt * (1 - infill_masks) + y * infill_masks                       # :377
        step = state - nxt                                                      # :380
        row = T - 1 - t
        metrics[0, row, 0] = _norm_metric(eps_recon)                            # grad_norm
        metrics[1, row, 0] = _norm_metric(step)                                 # step_norm
        metrics[2, row, 0] = co["alpha_prod"][t]
        metrics[3, row, 0] = _norm_metric(z)                                    # noise_norm
        slot = collection_slot_for_t(T, t, table)                               # :387-394
        if slot >= 0:
            collection[slot] = nxt
        state = nxt
    return state, collection, metrics


# ----------------------------------------------------------------------------------
# NCSN path ("next" row of SURVEY section 8): denoising score matching + Langevin samplers
# ----------------------------------------------------------------------------------
def dsm_used_sigmas(sigmas: np.ndarray, labels: np.ndarray, continuous_noise: bool, u01: Optional[np.ndarray] = None):
    """utils/losses.py:155-161: sigmas[labels], or uniform(minval=sigmas[labels-1], maxval=sigmas[labels]) =
    max(lo, u01 (hi - lo) + lo) in float32 (on a decreasing schedule lo > hi and the draw degenerates to lo)."""
    sigmas = np.asarray(sigmas, dtype=np.float32)
    labels = np.asarray(labels)
    if not continuous_noise:
        return sigmas[labels]
    lo, hi = sigmas[labels - 1], sigmas[labels]
    if u01 is None:
        if (lo < hi).any():
            raise ValueError("an increasing schedule makes the uniform of :156-159 a real draw: pass u01")
        return lo
    u01 = np.asarray(u01, dtype=np.float32)
    return np.maximum(lo, (u01 * (hi - lo)).astype(np.float32) + lo).astype(np.float32)


def denoising_score_matching_loss(batch, model, sigmas, labels, eps, continuous_noise=False, reduction="mean", u01=None):
    """utils/losses.py:129-179 with the draws (labels :149-152, eps :164) passed in."""
    B = batch.shape[0]
    used = torch.from_numpy(dsm_used_sigmas(sigmas, labels, continuous_noise, u01)).to(batch.dtype)
    used = used.reshape(B, *([1] * (batch.dim() - 1)))                          # :162-163
    noise = eps * used                                                          # :164
    perturbed = batch + noise                                                   # :165
    target = -1 / (used ** 2) * noise                                           # :166
    scores = model(perturbed, used)                                             # :167
    assert target.shape == batch.shape and scores.shape == batch.shape
    target, scores = target.reshape(B, -1), scores.reshape(B, -1)
    loss = 0.5 * ((scores - target) ** 2).sum(dim=-1) * used.reshape(B) ** 2    # :175-177
    return reduce_fn(loss, reduction)


def ald_collection_slot(collection_idx: np.ndarray, image_idx: int) -> int:
    """utils/ebm_utils.py:149-156: idx = sum(arange(n) * in1d(collection_idx, image_idx)) + 1 when any entry matches
    (matching positions ADD UP when linspace repeats a value, i.e. when len(sigmas) * T < 100), else -1."""
    hit = np.nonzero(np.asarray(collection_idx) == image_idx)[0]
    return int(hit.sum()) + 1 if len(hit) else -1


def annealed_langevin_dynamics(model, sigmas, init, epsilon, T, denoise, noises, infill=False, infill_samples=None,
                               infill_masks=None, infill_noises=None):
    """utils/ebm_utils.py:89-198.  ``noises(sigma_i, i)`` / ``infill_noises(sigma_i, i)`` return the N(0,1) draws of
    :141-142 / :137-138.  Returns (state, collection (100 + 1 + denoise, ...), metrics (4, L, T))."""
    dt = init.dtype
    sig = [float(np.float32(s)) for s in np.asarray(sigmas)]
    L = len(sig)
    assert L >= 2
    if not infill:
        infill_samples, infill_masks = torch.zeros_like(init), torch.zeros_like(init)     # :122-124
    n_coll = 100 + 1 + int(bool(denoise))
    collection = torch.zeros((n_coll, *init.shape), dtype=dt)
    collection[0] = init * (1 - infill_masks) + infill_samples * infill_masks              # :127-129
    cidx = np.linspace(1, L * T, 100).astype(np.int32)                                      # :130-132
    metrics = torch.zeros((4, L, T), dtype=dt)
    state = init
    B = init.shape[0]
    for si in range(L):
        sigma = sig[si]
        alpha = float(np.float32(epsilon) * (np.float32(sigma) / np.float32(sig[-1])) ** 2)  # :168
        for i in range(T):
            y = infill_samples + sigma * (infill_noises(si, i) if infill else torch.zeros_like(init))    # :137-138
            cond = torch.full((B, *([1] * (init.dim() - 1))), sigma, dtype=dt)
            grad = model(state, cond)                                                        # :140
            noise = math.sqrt(2 * alpha) * noises(si, i)                                     # :141-142
            nxt = state + alpha * grad + noise                                               # :143
            nxt = nxt * (1 - infill_masks) + y * infill_masks                                # :146
            slot = ald_collection_slot(cidx, si * T + i + 1)                                 # :149-156
            if 0 < slot < n_coll:
                collection[slot] = nxt
            metrics[0, si, i] = _norm_metric(grad)                                           # :159-163
            metrics[1, si, i] = _norm_metric(alpha * grad)
            metrics[2, si, i] = alpha
            metrics[3, si, i] = _norm_metric(noise)
            state = nxt
    if denoise:                                                                              # :189-192
        cond = torch.full((B, *([1] * (init.dim() - 1))), sig[-1], dtype=dt)
        state = state + sig[-1] ** 2 * model(state, cond)
        collection[-1] = state
    return state, collection, metrics


def consistent_langevin_dynamics(model, sigmas, init, epsilon, denoise, noises):
    """utils/ebm_utils.py:201-271.  ``noises(i)`` = the N(0,1) draw of :241-242.  Returns (state, metrics (4, L, 1))."""
    dt = init.dtype
    sig = [float(np.float32(s)) for s in np.asarray(sigmas)]
    L = len(sig)
    assert L >= 2
    beta = math.sqrt(1 - (1 - epsilon / sig[-1] ** 2) ** 2)                                  # :257
    metrics = torch.zeros((4, L, 1), dtype=dt)
    state = init
    B = init.shape[0]
    for i in range(L):
        sigma = sig[i]
        next_sigma = sig[i + 1] if i < L - 1 else 0.0                                        # :236
        alpha = epsilon * (sigma / sig[-1]) ** 2                                             # :238
        cond = torch.full((B, *([1] * (init.dim() - 1))), sigma, dtype=dt)
        grad = model(state, cond)
        noise = beta * next_sigma * noises(i)                                                # :240-241
        metrics[0, i, 0] = _norm_metric(grad)
        metrics[1, i, 0] = _norm_metric(alpha * grad)
        metrics[2, i, 0] = alpha
        metrics[3, i, 0] = _norm_metric(noise)
        state = state + alpha * grad + noise                                                 # :242
    if denoise:                                                                              # :264-265
        cond = torch.full((B, *([1] * (init.dim() - 1))), sig[-1], dtype=dt)
        state = state + sig[-1] ** 2 * model(state, cond)
    return state, metrics


def collate_sampling_metrics(ld_metrics):
    """utils/ebm_utils.py:408-428."""
    _, num_sigmas, num_steps = ld_metrics.shape
    out = [[] for _ in range(num_sigmas)]
    for i in range(num_sigmas):
        g, s, a, n = ld_metrics[:, i, :]
        for j in range(num_steps):
            out[i].append({"slope": g[j], "step": s[j], "alpha": a[j], "noise": n[j]})
    return out


# ----------------------------------------------------------------------------------
# Output-side data transforms (host NumPy), input_pipeline.py:36-48,78-110
# ----------------------------------------------------------------------------------
def normalize_dataset(batch, data_min, data_max):
    """input_pipeline.py:36-40."""
    batch = (batch - data_min) / (data_max - data_min)
    return 2.0 * batch - 1.0


def inverse_data_transform(batch, normalize=True, data_min=0.0, data_max=1.0, slice_idx=None,
                           dim_weights=None, out_channels=512, filler=None):
    """input_pipeline.py:78-110 without the PCA branch (pca_ckpt unused by the DDPM configs).
    The reference fills the non-selected latent dims with an *unseeded* np.random.randn
    (:102-105); ``filler`` lets a test pass that array so outputs are comparable."""
    batch = np.asarray(batch)
    if normalize:
        batch = (batch + 1.0) / 2.0
        batch = (data_max - data_min) * batch + data_min
    if slice_idx is not None:
        transformed = (np.random.randn(*batch.shape[:-1], out_channels)
                       if filler is None else np.array(filler, dtype=np.float64, copy=True))
        transformed[..., slice_idx] = batch
        batch = transformed
    if dim_weights is not None:
        batch = batch / dim_weights
    return batch


# ----------------------------------------------------------------------------------
# Counter-based RNG used by the engine's throughput mode (NOT the reference's threefry;
# JAX-compatible streams are a "next" row).  Philox4x32-10, Random123 constants.
# ----------------------------------------------------------------------------------
_PHILOX_M0, _PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PHILOX_W0, _PHILOX_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(ctr: np.ndarray, key: np.ndarray, rounds: int = 10) -> np.ndarray:
    """ctr (...,4) uint32, key (2,) uint32 -> (...,4) uint32."""
    c = ctr.astype(np.uint32).copy()
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(rounds):
        p0 = _PHILOX_M0 * c[..., 0].astype(np.uint64)
        p1 = _PHILOX_M1 * c[..., 2].astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & mask).astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & mask).astype(np.uint32)
        n0 = hi1 ^ c[..., 1] ^ k0
        n2 = hi0 ^ c[..., 3] ^ k1
        c = np.stack([n0, lo1, n2, lo0], axis=-1)
        with np.errstate(over="ignore"):
            k0 = np.uint32((int(k0) + int(_PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_PHILOX_W1)) & 0xFFFFFFFF)
    return c


def philox_uniform01(bits: np.ndarray) -> np.ndarray:
    """(0,1] float32 from uint32: (bits>>8 + 1) * 2^-24  -- matches csrc/rng.h."""
    return ((bits >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(2.0 ** -24)


def philox_normal4(ctr: np.ndarray, key: np.ndarray) -> np.ndarray:
    """4 standard normals per counter via two Box-Muller pairs -- matches csrc/rng.h."""
    b = philox4x32(ctr, key)
    u = philox_uniform01(b).astype(np.float64)
    r0 = np.sqrt(-2.0 * np.log(u[..., 0]))
    r1 = np.sqrt(-2.0 * np.log(u[..., 2]))
    a0 = 2.0 * np.pi * u[..., 1]
    a1 = 2.0 * np.pi * u[..., 3]
    return np.stack([r0 * np.cos(a0), r0 * np.sin(a0), r1 * np.cos(a1), r1 * np.sin(a1)], axis=-1)


# ----------------------------------------------------------------------------------
# jax.random (jax 0.2.8) restated: threefry2x32 counter PRNG and the split / bits / uniform /
# normal / randint conventions the reference consumes at utils/losses.py:271-294,
# utils/ebm_utils.py:329,342-345,360-362 and train_ncsn.py:318-319,358,536-540.
#
# PINNED against published known answers (tests/golden/jax_random_kat.json): the Random123
# threefry2x32-20 vectors, and the key / normal values printed in JAX's own documentation
# ("JAX - The Sharp Bits", section "JAX PRNG"): PRNGKey(0) -> split -> keys, and
# random.normal(key, (1,)) for six keys.  They fix: the key schedule and rotations, the
# "counter halves" layout of random_bits (incl. the odd-size zero pad), split's reshape, the
# mantissa-fill uniform with its max(minval, .) clamp, and XLA's float32 erf_inv polynomial.
# randint's two-draw multiplier scheme has no published vector: restated from memory (unpinned).
# ----------------------------------------------------------------------------------
_TF_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def threefry2x32(key, x0, x1):
    """Threefry-2x32, 20 rounds (Random123).  key = (k0, k1); x0/x1 uint32 arrays -> (y0, y1)."""
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    ks = (k0, k1, np.uint32(k0 ^ k1 ^ np.uint32(0x1BD11BDA)))
    x0 = np.asarray(x0, dtype=np.uint32).copy()
    x1 = np.asarray(x1, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x0 = x0 + ks[0]
        x1 = x1 + ks[1]
        for i in range(5):
            for r in _TF_ROT[i % 2]:
                x0 = x0 + x1
                x1 = ((x1 << np.uint32(r)) | (x1 >> np.uint32(32 - r))) ^ x0
            x0 = x0 + ks[(i + 1) % 3]
            x1 = x1 + ks[(i + 2) % 3] + np.uint32(i + 1)
    return x0, x1


def jax_prngkey(seed: int):
    """jax.random.PRNGKey with 32-bit ints (x64 disabled): [0, seed mod 2^32]."""
    return (np.uint32(0), np.uint32(int(seed) & 0xFFFFFFFF))


def jax_random_bits(key, n: int) -> np.ndarray:
    """_random_bits(key, 32, shape) flattened: threefry_2x32(key, iota(n)) where the counter vector is split
    into two halves (x0 = first half, x1 = second half, odd n padded with one zero) and the two output words
    are concatenated again."""
    h = (n + 1) // 2
    cnt = np.arange(2 * h, dtype=np.uint32)
    if n % 2:
        cnt[-1] = 0
    y0, y1 = threefry2x32(key, cnt[:h], cnt[h:])
    return np.concatenate([y0, y1])[:n]


def jax_split(key, num: int = 2):
    """jax.random.split: random_bits over 2*num counters reshaped (num, 2)."""
    b = jax_random_bits(key, 2 * num).reshape(num, 2)
    return [(b[j, 0], b[j, 1]) for j in range(num)]


def jax_uniform(key, n: int, minval=0.0, maxval=1.0) -> np.ndarray:
    """jax.random.uniform (float32): mantissa fill -> [1,2) - 1 -> * (maxval - minval) + minval -> max(minval, .).
    minval / maxval may be per-element arrays (utils/losses.py:283-286)."""
    f = np.float32
    bits = jax_random_bits(key, n)
    u01 = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - f(1.0)
    lo = np.asarray(minval, dtype=np.float32)
    hi = np.asarray(maxval, dtype=np.float32)
    return np.maximum(lo, (u01 * (hi - lo)).astype(np.float32) + lo).astype(np.float32)


_ERFINV_LT5 = (2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087,
               -0.00125372503, -0.00417768164, 0.246640727, 1.50140941)
_ERFINV_GE5 = (-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773,
               -0.0076224613, 0.00943887047, 1.00167406, 2.83297682)


def erfinv_f32(x: np.ndarray) -> np.ndarray:
    """XLA's float32 erf_inv (Giles' single-precision polynomial), evaluated in float32 like XLA does."""
    f = np.float32
    x = np.asarray(x, dtype=np.float32)
    w = -np.log(((f(1) - x) * (f(1) + x)).astype(np.float32)).astype(np.float32)
    lt = w < f(5)
    ww = np.where(lt, w - f(2.5), np.sqrt(np.maximum(w, f(0))).astype(np.float32) - f(3)).astype(np.float32)
    p = np.where(lt, f(_ERFINV_LT5[0]), f(_ERFINV_GE5[0])).astype(np.float32)
    for a, b in zip(_ERFINV_LT5[1:], _ERFINV_GE5[1:]):
        p = (np.where(lt, f(a), f(b)) + (p * ww).astype(np.float32)).astype(np.float32)
    return (p * x).astype(np.float32)


def jax_normal(key, n: int) -> np.ndarray:
    """jax.random.normal (float32): sqrt(2) * erf_inv(uniform(key, minval=nextafter(-1, 0), maxval=1))."""
    lo = np.nextafter(np.float32(-1), np.float32(0), dtype=np.float32)
    u = jax_uniform(key, n, lo, np.float32(1))
    return (np.float32(np.sqrt(2)) * erfinv_f32(u)).astype(np.float32)


def jax_randint(key, n: int, minval: int, maxval: int) -> np.ndarray:
    """jax.random.randint (int32), jax 0.2.8: two 32-bit draws from split(key); span = maxval - minval (1 if
    maxval <= minval); multiplier = (2^16 % span)^2 % span; ((hi % span) * multiplier + lo % span) % span."""
    k1, k2 = jax_split(key)
    hi, lo = jax_random_bits(k1, n), jax_random_bits(k2, n)
    maxval = max(minval + 1, maxval)
    span = np.uint32(maxval - minval)
    with np.errstate(over="ignore"):
        mult = np.uint32(np.uint32(1 << 16) % span)
        mult = np.uint32(np.uint32(mult * mult) % span)
        off = ((hi % span) * mult + (lo % span)) % span
    return (np.int32(minval) + off.astype(np.int32)).astype(np.int32)


def jax_diffusion_loss_draws(rng, batch_shape, T: int, continuous_noise: bool = True):
    """The draws of utils/losses.py:271-294 for key ``rng``: (labels int32 (B,), eps float32 batch_shape).
    The uniform of :283-286 is degenerate (minval > maxval -> minval) and needs no bits."""
    _rng, label_rng, sample_rng = jax_split(rng, 3)
    B = int(batch_shape[0])
    labels = jax_randint(label_rng, B, int(continuous_noise), T + int(continuous_noise))
    eps = jax_normal(sample_rng, int(np.prod(batch_shape))).reshape(batch_shape)
    return labels, eps


def jax_diffusion_loss_u01(rng, B: int) -> np.ndarray:
    """[0, 1) floats behind the uniform of utils/losses.py:282-286: noise_rng = split(first output of split(rng, 3))[1]."""
    first, _label_rng, _sample_rng = jax_split(rng, 3)
    _rng, noise_rng = jax_split(first)
    return jax_uniform(noise_rng, B, 0.0, 1.0)


def jax_dsm_loss_draws(rng, batch_shape, sigmas, continuous_noise: bool = False):
    """(labels, used_sigmas, eps) of utils/losses.py:149-164 for key ``rng``."""
    first, label_rng, sample_rng = jax_split(rng, 3)
    B = int(batch_shape[0])
    labels = jax_randint(label_rng, B, int(continuous_noise), len(sigmas))
    u01 = None
    if continuous_noise:
        _rng, noise_rng = jax_split(first)
        u01 = jax_uniform(noise_rng, B, 0.0, 1.0)
    used = dsm_used_sigmas(sigmas, labels, continuous_noise, u01)
    eps = jax_normal(sample_rng, int(np.prod(batch_shape))).reshape(batch_shape)
    return labels, used, eps


def jax_langevin_keys(ld_rng, iterations: int, consistent: bool = False):
    """Per-update keys: annealed ``rng, step_rng, infill_rng = split(rng, 3)`` (utils/ebm_utils.py:133), consistent
    ``rng, step_rng = split(rng)`` (:233).  Returns (step_keys, infill_keys) lists (infill_keys empty when consistent)."""
    rng = ld_rng
    step_keys, infill_keys = [], []
    for _ in range(iterations):
        if consistent:
            rng, s = jax_split(rng)
        else:
            rng, s, f = jax_split(rng, 3)
            infill_keys.append(f)
        step_keys.append(s)
    return step_keys, infill_keys


def jax_sampler_keys(ld_rng, T: int):
    """Per-iteration keys of utils/ebm_utils.py:329,342,360 walking t = T-1 .. 0:
    returns (infill_keys, noise_keys), each a list indexed by ITERATION (0 = first step, t = T-1)."""
    rng = ld_rng
    infill_keys, noise_keys = [], []
    for _ in range(T):
        rng, _key = jax_split(rng)
        rng, infill_rng = jax_split(rng)
        rng, noise_rng = jax_split(rng)
        infill_keys.append(infill_rng)
        noise_keys.append(noise_rng)
    return infill_keys, noise_keys


# ----------------------------------------------------------------------------------
# create_model's initial parameters (train_ncsn.py:193-203 -> flax.nn init_by_shape): the primitives, restated from the
# flax 0.3.0 / jax 0.2.8 sources.  UNPINNED (no published vectors; tests/golden/make_jax_goldens.py dumps the real ones).
# ----------------------------------------------------------------------------------
def jax_fold_in(key, data: int):
    """jax.random.fold_in(key, data) = threefry_2x32(key, PRNGKey(data)): one block on the counter pair (0, data)."""
    y0, y1 = threefry2x32(key, np.array([0], np.uint32), np.array([int(data) & 0xFFFFFFFF], np.uint32))
    return (y0[0], y1[0])


def flax_fold_in_str(key, s: str):
    """flax.nn.base._fold_in_str: fold the first four bytes (big endian) of sha1(s) into the key."""
    import hashlib
    return jax_fold_in(key, int.from_bytes(hashlib.sha1(s.encode("utf-8")).digest()[:4], byteorder="big"))


def jax_truncated_normal(key, n: int, lower=-2.0, upper=2.0) -> np.ndarray:
    """jax.random.truncated_normal (float32): u = uniform(key, minval=erf(lower/sqrt2), maxval=erf(upper/sqrt2));
    sqrt(2) * erf_inv(u), clipped into the open interval (lower, upper)."""
    f = np.float32
    a, b = f(math.erf(lower / math.sqrt(2.0))), f(math.erf(upper / math.sqrt(2.0)))
    u = jax_uniform(key, n, a, b)
    out = (f(np.sqrt(2)) * erfinv_f32(u)).astype(np.float32)
    return np.clip(out, np.nextafter(f(lower), f(np.inf), dtype=np.float32), np.nextafter(f(upper), f(-np.inf), dtype=np.float32))


def jax_lecun_normal(key, shape) -> np.ndarray:
    """jax.nn.initializers.lecun_normal()(key, (fan_in, fan_out)): truncated normal scaled to variance 1 / fan_in."""
    f = np.float32
    std = f(np.sqrt(f(1.0) / f(shape[0]), dtype=np.float32) / f(0.87962566103423978))
    return (jax_truncated_normal(key, int(shape[0]) * int(shape[1])) * std).astype(np.float32).reshape(shape)


def flax_param_key(model_rng, path):
    """Key of the parameter at ``path`` = (child module names ..., parameter name): every level folds its name in."""
    key = model_rng
    for name in path:
        key = flax_fold_in_str(key, name)
    return key


# ----------------------------------------------------------------------------------
# --interpolate (sample_ncsn.py:245-310, 425-435): stochastic encode at the last noise level, 9-point lerp, decode
# ----------------------------------------------------------------------------------
def diffusion_stochastic_encoder(samples: np.ndarray, betas: np.ndarray, rng_seed: int) -> np.ndarray:
    """sample_ncsn.py:245-266.  ``rng, noise_rng = split(PRNGKey(seed))`` and the noise is drawn from ``rng`` (:262-263, not
    from noise_rng); alphas_prod[T] is one past the end and JAX clamps the gather to alphas_prod[T-1]."""
    ap = alphas_cumprod(betas)
    a_T = np.float32(ap[min(len(betas), len(ap) - 1)])
    rng, _noise_rng = jax_split(jax_prngkey(rng_seed))
    noise = jax_normal(rng, int(np.prod(samples.shape))).reshape(samples.shape)
    mu = np.sqrt(a_T, dtype=np.float32) * samples.astype(np.float32)
    sigma = np.sqrt(np.float32(1) - a_T, dtype=np.float32)
    return (mu + sigma * noise).astype(np.float32)


def interpolate_samples(model, betas: np.ndarray, real: np.ndarray, rng_seed: int, points: int = 9):
    """sample_ncsn.py:425-435 + diffusion_decoder (:269-310): goals = roll(starts, 1); both encoded with the SAME seed (the same
    noise); z_alpha = (1 - alpha) z_start + alpha z_goal for alpha in linspace(0, 1, 9); every z decoded by diffusion_dynamics
    with the SAME ld_rng = split(PRNGKey(seed), 3)[1].  Returns (generated (9, N, ...), collection (9, 41, N, ...))."""
    starts = np.asarray(real, dtype=np.float32)
    goals = np.roll(starts, shift=1, axis=0)
    zs, zg = diffusion_stochastic_encoder(starts, betas, rng_seed), diffusion_stochastic_encoder(goals, betas, rng_seed)
    _rng, ld_rng, _model_rng = jax_split(jax_prngkey(rng_seed), 3)
    T = len(betas)
    _infill_keys, noise_keys = jax_sampler_keys(ld_rng, T)
    n = int(np.prod(starts.shape))
    zt = {T - 1 - i: torch.from_numpy(jax_normal(noise_keys[i], n).reshape(starts.shape)) for i in range(T)}
    gens, colls = [], []
    for alpha in np.linspace(0.0, 1.0, points):
        z = ((1 - alpha) * zs + alpha * zg).astype(np.float32)
        g, c, _m = diffusion_dynamics(model, betas, torch.from_numpy(z).double(), lambda t: zt[t].double())
        gens.append(g)
        colls.append(c)
    return torch.stack(gens), torch.stack(colls)
