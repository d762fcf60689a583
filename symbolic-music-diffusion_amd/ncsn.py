"""Host-side mirror of the reference's model / loss / sampler callables for the DDPM path.

Same names, argument meaning and error behaviour as the reference seam:

  create_model          train_ncsn.py:193-203        Model.__call__   models/ncsn.py:141-148
  diffusion_loss        utils/losses.py:250-308      reduce_fn        utils/losses.py:22-30
  diffusion_dynamics    utils/ebm_utils.py:280-405   collate_sampling_metrics  :408-428
  sample                train_ncsn.py:499-551

Every array computation runs in the HIP library (lib.py); this module only allocates tensors,
sequences calls and reproduces the reference's output layouts and quirks.
"""
from __future__ import annotations

import ctypes as C
import logging
import math
import os
import time
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import jax_random as _jr
from . import lib as _lib
from . import schedule as _sched
from .engine import ARCH_IDS, Engine, NetConfig
from .jax_random import ThreefryKey

COLLECTION_STEPS = _sched.COLLECTION_STEPS
_log = logging.getLogger("smd_amd.sampler")


# ------------------------------------------------------------------ rng keys
@dataclass(frozen=True)
class PRNGKey:
    """A 64-bit seed for the engine's own Philox streams (throughput mode: draws happen inside the fused kernels).
    ``jax_random.PRNGKey(seed)`` makes a ``ThreefryKey`` instead; every function below that takes ``rng`` then
    consumes the reference's jax.random streams bit for bit (--rng_impl=threefry on the CLIs)."""
    seed: int

    def __post_init__(self):
        object.__setattr__(self, "seed", int(self.seed) & 0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def make_key(seed: int, impl: str = "philox"):
    """PRNGKey(seed) of the chosen implementation: 'philox' (engine streams) or 'threefry' (jax.random streams)."""
    if impl == "threefry":
        return _jr.PRNGKey(seed)
    if impl != "philox":
        raise ValueError(f"rng_impl must be 'philox' or 'threefry', got {impl!r}")
    return PRNGKey(seed)


def split(key, num: int = 2):
    """jax.random.split (train_ncsn.py:318-319,358): exact for a ThreefryKey, ``num`` independent child seeds for
    the engine's Philox keys."""
    if isinstance(key, ThreefryKey):
        return _jr.split(key, num)
    return tuple(PRNGKey(_splitmix64(key.seed ^ _splitmix64(i + 1))) for i in range(num))


# ------------------------------------------------------------------ model
class Model:
    """nn.Model stand-in: ``model(x, cond) -> eps_hat`` on the HIP engine.

    x: (B, *sample_shape) fp32, cond: (B, 1[, 1]) fp32 noise level sqrt(alpha_bar)
    (models/ncsn.py:141-148 / 125-127)."""

    def __init__(self, cfg: NetConfig, device: str = "cuda:0", seed: Optional[int] = 0,
                 share_params_with: Optional["Model"] = None):
        self.cfg = cfg
        self.engine = Engine(cfg, device, share_params_with=share_params_with.engine if share_params_with else None)
        if share_params_with is None and seed is not None:
            self.engine.init_params(seed)
        self._train_engine: Optional[Engine] = None
        from . import ops as _ops                      # registers torch.ops.smd_amd.*
        self._op_id = _ops.register_engine(self.engine)

    @property
    def params(self) -> torch.Tensor:
        return self.engine.params

    def named_parameters(self):
        return self.engine.named_views()

    def __call__(self, x, cond):
        x = torch.as_tensor(x)
        cond = torch.as_tensor(cond)
        assert x.shape[0] == cond.shape[0], (x.shape, cond.shape)           # models/ncsn.py:32,50
        dev = self.engine.device
        return torch.ops.smd_amd.eps_forward(x.to(dev, torch.float32), cond.to(dev, torch.float32), self._op_id)

    def differentiable(self, ema: bool = False) -> "DifferentiableModel":
        """``model`` for a generic objective under ``torch.autograd`` (train_ncsn.py:279-283 differentiates ANY callable of the
        model): the same network on this model's training handle, with the flat parameter buffer as an autograd leaf."""
        return DifferentiableModel(self, ema)

    def drop_sampler_cache(self) -> None:
        """Forget the cached sampler graphs and their persistent state / collection / metrics buffers."""
        self.__dict__.pop("_sampler_graphs", None)

    def chain_streams(self, n: int):
        """The streams of the concurrent sampling chains, created ONCE per model: HIP maps streams onto a few hardware queues
        round-robin, and a pair created later in a process can land on one queue, where the two chains serialise (measured:
        2.2x per sampling step on the second workload of a bench process)."""
        if getattr(self, "_chain_streams", None) is None or len(self._chain_streams) < n:
            self._chain_streams = [torch.cuda.Stream(device=self.engine.device) for _ in range(n)]
        return self._chain_streams[:n]

    def chain_engines(self, n: int) -> List[Engine]:
        """``n`` inference handles on this model's parameters and operand pack for concurrent sampling chains; their large
        GEMMs take the 256x256 kernel from 128 tiles up (half the CUs each: two chains fill the chip)."""
        if getattr(self, "_chain_engines", None) is None or len(self._chain_engines) != n:
            self._chain_engines = []
            for _ in range(n):
                e = Engine(self.cfg, str(self.engine.device), share_params_with=self.engine)
                e.set_option("nt256_min_tiles", 128)
                self._chain_engines.append(e)
        return self._chain_engines

    def train_engine(self, ema: bool) -> Engine:
        if self._train_engine is None:
            self._train_engine = Engine(self.cfg, str(self.engine.device), share_params_with=self.engine)
            self._train_engine.enable_training(ema)
        return self._train_engine

    def replace(self, params: torch.Tensor) -> "Model":
        """nn.Model.replace(params=...) (train_ncsn.py:405-407): copies into this model's buffer."""
        self.engine.params.copy_(params)
        self.engine.refresh_weights()
        return self

    def num_parameters(self) -> int:
        return self.engine.n_params


class DifferentiableModel:
    """``model(x, cond)`` whose result carries an autograd edge to ``.params`` (a leaf aliasing the engine's flat fp32 parameter
    buffer): ``objective(batch, dm, ...).backward()`` runs the engine's backward pass from d objective / d eps_hat and leaves
    the gradient in ``dm.params.grad`` (flat, the layout of ``named_parameters``) and in the training handle's gradient buffer,
    where ``trainer.train_step`` clips and applies it.  One forward pass may be outstanding per handle."""

    def __init__(self, model: Model, ema: bool = False):
        from . import ops as _ops
        self.model = model
        self.cfg = model.cfg
        self.engine = model.train_engine(ema)
        self._op_id = _ops.register_engine(self.engine)
        self.params = self.engine.params.detach().requires_grad_(True)      # shares storage; its own autograd identity

    def named_parameters(self):
        return self.engine.named_views()

    def __call__(self, x, cond):
        x = torch.as_tensor(x)
        cond = torch.as_tensor(cond)
        assert x.shape[0] == cond.shape[0], (x.shape, cond.shape)           # models/ncsn.py:32,50
        dev = self.engine.device
        return torch.ops.smd_amd.eps_forward_train(x.to(dev, torch.float32), cond.to(dev, torch.float32), self.params, self._op_id)


def config_from_kwargs(architecture: str, input_shape: Sequence[int], model_kwargs: dict,
                       num_timesteps: int = 1000, dtype: str = "bf16") -> NetConfig:
    if architecture not in ARCH_IDS:
        raise ValueError(f"Unsupported architecture {architecture!r} (HIP engine: {sorted(ARCH_IDS)})")
    input_shape = tuple(int(v) for v in input_shape)
    if ARCH_IDS[architecture] == 1:
        if len(input_shape) != 1:
            raise ValueError(f"DenseDDPM takes (B, C) inputs, got per-example shape {input_shape}")
        seq, ch = 1, input_shape[0]
    else:
        if len(input_shape) != 2:
            raise ValueError(f"TransformerDDPM takes (B, S, C) inputs, got per-example shape {input_shape}")
        seq, ch = input_shape
    return NetConfig(architecture=architecture, data_channels=ch, seq_len=seq,
                     num_layers=int(model_kwargs.get("num_layers", 6)),
                     num_heads=int(model_kwargs.get("num_heads", 8)),
                     num_mlp_layers=int(model_kwargs.get("num_mlp_layers", 2)),
                     mlp_dims=int(model_kwargs.get("mlp_dims", 2048)), num_timesteps=num_timesteps, dtype=dtype)


def init_model(model: Model, rng: PRNGKey) -> None:
    """The initial parameters of ``create_model(rng, ...)`` written into an existing model.  ThreefryKey: the reference's
    ``init_by_shape(model_rng)`` draw as far as it can be restated without flax (flax_init.py: every kernel lecun_normal() of its
    folded-in key; UNVERIFIED against flax itself until tests/golden/make_jax_goldens.py has run); engine key: NumPy lecun-normal."""
    if isinstance(rng, ThreefryKey):
        from . import flax_init as _fi
        table = {name: shape for name, _off, shape in model.engine.tensor_table}
        model.engine.load_named(_fi.init_params(model.cfg, rng, table))
    else:
        model.engine.init_params(rng.seed & 0x7FFFFFFF)


def create_model(rng: PRNGKey, input_shape, model_kwargs, batch_size=32, verbose=False, *,
                 architecture: str = "TransformerDDPM", num_timesteps: int = 1000, device: str = "cuda:0",
                 dtype: str = "bf16", init: bool = True) -> Model:
    """train_ncsn.py:193-203.  ``architecture`` replaces the FLAGS.architecture global; DenseDDPM
    accepts-and-ignores num_heads / num_mlp_layers like the reference's kwargs.  ``init=False`` leaves the
    parameters zero for a caller that restores a checkpoint next (the threefry initialiser costs ~5 s of host time) and
    calls ``init_model`` itself when there is none."""
    del batch_size
    cfg = config_from_kwargs(architecture, input_shape, model_kwargs, num_timesteps, dtype)
    model = Model(cfg, device, seed=None)
    if init:
        init_model(model, rng)
    if verbose:
        from .train_utils import report_model
        report_model(model)
    return model


# ------------------------------------------------------------------ objective
def reduce_fn(x, mode):
    """utils/losses.py:22-30."""
    if mode == "none" or mode is None:
        return x
    if mode == "sum":
        return x.sum()
    if mode == "mean":
        return x.mean()
    raise ValueError("Unsupported reduction option.")


def _ensure_schedule(engine: Engine, betas, with_sampler: bool) -> None:
    betas = np.asarray(betas, dtype=np.float32)
    need_film = with_sampler and (engine._sched_tensors is None or engine._sched_tensors["film"] is None)
    if engine.betas is None or need_film or not np.array_equal(engine.betas, betas):
        engine.set_schedule(betas, with_sampler=with_sampler)


def diffusion_loss(batch, model: Model, betas, rng: PRNGKey, continuous_noise=False, reduction="mean", *,
                   labels=None, eps=None, used_alphas=None):
    """utils/losses.py:250-308 (forward only; the training step fuses it with the backward).
    ``labels`` / ``eps`` pass the draws of :272-275 / :294 explicitly (parity mode); otherwise they
    come from the engine's Philox streams keyed by ``rng`` (or, with a ThreefryKey, are the reference's own).
    continuous_noise only moves the label range (:272-275): True -> [1, T], every uniform of :283-286 degenerates to its
    minval; False -> [0, T), where label 0 wraps to minval = alphas_prod[T] < maxval = 1 and is a real uniform draw.
    ``used_alphas`` passes those draws explicitly.  (The discrete-conditioning branch itself is commented out upstream.)"""
    eng = model.train_engine(ema=False)
    batch = torch.as_tensor(batch).to(eng.device, torch.float32).contiguous()
    _ensure_schedule(eng, betas, with_sampler=False)
    eng.bind(batch.shape[0], training=True)
    lab = None if labels is None else torch.as_tensor(labels).to(eng.device, torch.int32).contiguous()
    e = None if eps is None else torch.as_tensor(eps).to(eng.device, torch.float32).contiguous()
    ua = None if used_alphas is None else torch.as_tensor(used_alphas).to(eng.device, torch.float32).contiguous()
    if isinstance(rng, ThreefryKey) and lab is None and e is None:
        lab, e = _jr.diffusion_loss_draws(rng, tuple(batch.shape), len(betas), eng.device,
                                          continuous_noise=bool(continuous_noise))              # :271-294
        if not continuous_noise and ua is None:
            ua = _jr.diffusion_loss_used_alphas(rng, lab, eng._sched_tensors["ape"])            # :282-286
    eng.loss_backward(batch, lab, e, seed=rng.seed, stage=3, used_alphas=ua, continuous_noise=bool(continuous_noise))
    loss = eng.loss_per_sample().clone()
    assert loss.shape == batch.shape[:1]                                          # utils/losses.py:306
    return reduce_fn(loss, reduction)


def _ensure_any_schedule(eng: Engine) -> None:
    """The score-matching loss passes its noise levels per sample; the engine still wants a bound schedule table."""
    if eng.betas is None:
        eng.set_schedule(np.linspace(1e-6, 1e-2, eng.cfg.num_timesteps, dtype=np.float32), with_sampler=False)


def _dsm_draws(rng, batch_shape, sigmas_t: torch.Tensor, continuous_noise: bool, sample_offset: int, global_batch: int):
    """labels / used_sigmas / eps of utils/losses.py:149-164 for this rank's rows.  ThreefryKey: the reference's streams;
    engine key: a torch generator seeded by it draws the GLOBAL batch's labels (so shards agree) and eps comes from Philox."""
    if isinstance(rng, ThreefryKey):
        return _jr.dsm_loss_draws(rng, batch_shape, sigmas_t, continuous_noise=continuous_noise, sample_offset=sample_offset,
                                  global_batch=global_batch)
    g = torch.Generator(device="cpu").manual_seed(rng.seed & 0x7FFFFFFFFFFFFFFF)
    L = int(sigmas_t.shape[0])
    lab = torch.randint(int(continuous_noise), L, (global_batch,), generator=g)[sample_offset:sample_offset + batch_shape[0]]
    lab = lab.to(sigmas_t.device)
    # continuous noise: uniform(minval=sigmas[l-1], maxval=sigmas[l]) returns its minval on a decreasing schedule (:156-159)
    used = sigmas_t[(lab - 1) % L] if continuous_noise else sigmas_t[lab]
    return lab.int(), used.contiguous(), None


def denoising_score_matching_loss(batch, model: Model, sigmas, rng: PRNGKey, continuous_noise=False, reduction="mean", *,
                                  labels=None, eps=None, used_sigmas=None):
    """utils/losses.py:129-179 (forward only; trainer.train_step fuses it with the backward):
    loss_b = 0.5 * sum((model(batch + sigma_b eps, sigma_b) + eps / sigma_b)^2) * sigma_b^2.  ``labels`` / ``eps`` /
    ``used_sigmas`` pass the draws explicitly (parity mode).  The score network is one of the engine's architectures
    conditioned on sigma (the reference's NCSN nets have NameErrors)."""
    eng = model.train_engine(ema=False)
    batch = torch.as_tensor(batch).to(eng.device, torch.float32).contiguous()
    _ensure_any_schedule(eng)
    eng.bind(batch.shape[0], training=True)
    sig = torch.as_tensor(np.asarray(sigmas, dtype=np.float32)).to(eng.device)
    e = None if eps is None else torch.as_tensor(eps).to(eng.device, torch.float32).contiguous()
    if used_sigmas is not None:
        us = torch.as_tensor(used_sigmas).to(eng.device, torch.float32).contiguous()
    elif labels is not None:
        lab = torch.as_tensor(labels).to(eng.device).long()
        us = (sig[(lab - 1) % len(sig)] if continuous_noise else sig[lab]).contiguous()
    else:
        _lab, us, e2 = _dsm_draws(rng, tuple(batch.shape), sig, bool(continuous_noise), 0, batch.shape[0])
        e = e if e is not None else e2
    eng.loss_backward(batch, None, e, seed=rng.seed, stage=3, used_alphas=us, objective="dsm")
    loss = eng.loss_per_sample().clone()
    return reduce_fn(loss, reduction)


# ------------------------------------------------------------------ Langevin samplers (NCSN path)
ALD_COLLECTION_STEPS = 100                                                        # utils/ebm_utils.py:127


def ald_collection_slot(collection_idx: np.ndarray, image_idx: int) -> int:
    """utils/ebm_utils.py:149-156: ``idx = sum(arange(n) * in1d(collection_idx, image_idx)) + 1`` when any entry matches --
    repeated linspace entries (len(sigmas) * T < 100) ADD UP, exactly like upstream -- else -1."""
    hit = np.nonzero(np.asarray(collection_idx) == image_idx)[0]
    return int(hit.sum()) + 1 if len(hit) else -1


def _langevin_io(x, grad, alpha, noise_coef, rng, step, sample_offset, metrics, collect):
    io = _lib.LangevinIO()
    io.x, io.grad = x.data_ptr(), grad.data_ptr()
    io.alpha, io.noise_coef = float(alpha), float(noise_coef)
    io.seed_lo, io.seed_hi = rng.seed & 0xFFFFFFFF, (rng.seed >> 32) & 0xFFFFFFFF
    io.step, io.sample_offset = int(step), int(sample_offset)
    io.metrics_partial = metrics.data_ptr()
    io.collect_out = None if collect is None else collect.data_ptr()
    return io


def _langevin_graph_loop(model: "Model", x: torch.Tensor, io: "_lib.LangevinIO", step_rows: np.ndarray, slots: Optional[np.ndarray],
                         keys: Optional[np.ndarray], metrics: torch.Tensor, collection: Optional[torch.Tensor], use_graph: bool,
                         levels: Optional[np.ndarray] = None, steps_per_level: int = 1) -> None:
    """All ``n = len(step_rows)`` Langevin updates of a run as ONE captured (eps-net forward + fused update) pair replayed
    n - 1 times: the step-dependent arguments (alpha, noise coefficient, infill sigma, the next forward's noise level,
    collection slot, threefry keys) live in device tables indexed by a device-side counter the update kernel advances
    (csrc/diffusion.hip langevin_step_kernel, table mode), exactly as the DDPM sampler keeps t on the device."""
    eng = model.engine
    dev = eng.device
    B, S, Cn = x.shape[0], eng.S, eng.C
    n = int(step_rows.shape[0])
    eng.bind(B, training=False)
    eng._sync_fp8_weights()
    # the noise level is batch-uniform inside both samplers: FiLM scale / shift of every level are tabulated once (as the
    # DDPM sampler does for its T levels) and the forward pass takes the level as a device-side table row
    by_level = levels is not None and eng.set_noise_levels(levels)
    level_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
    tab = torch.from_numpy(np.ascontiguousarray(step_rows[:, :4], dtype=np.float32)).to(dev)
    slot_t = None if slots is None else torch.from_numpy(np.ascontiguousarray(slots, dtype=np.int32)).to(dev)
    key_t = None if keys is None else torch.from_numpy(np.ascontiguousarray(keys).view(np.int32).copy()).to(dev)
    k_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
    arrive = torch.zeros(1, dtype=torch.int32, device=dev)
    sig_vec = torch.full((B,), float(step_rows[0, 4]), dtype=torch.float32, device=dev)       # noise level of update 0's forward
    grad = None if by_level else torch.empty_like(x)          # by level: the update reads the engine's output buffer in place
    io.x = x.data_ptr()
    io.grad = int(_lib.get_lib().smd_engine_pred(eng.h)) if by_level else grad.data_ptr()
    io.step_table, io.n_steps = tab.data_ptr(), n
    io.slot_table = None if slot_t is None else slot_t.data_ptr()
    io.key_table = None if key_t is None else key_t.data_ptr()
    io.k_ptr, io.arrive = k_ptr.data_ptr(), arrive.data_ptr()
    io.collection = None if collection is None else collection.data_ptr()
    io.sigma_out = sig_vec.data_ptr()
    io.metrics_partial = metrics.data_ptr()
    if by_level:
        io.level_out, io.steps_per_level, io.n_levels = level_ptr.data_ptr(), int(steps_per_level), int(len(levels))
    L_ = _lib.get_lib()

    def one():
        st = torch.cuda.current_stream(dev).cuda_stream
        if by_level:
            _lib.check(L_.smd_engine_forward_level(eng.h, x.data_ptr(), level_ptr.data_ptr(), None, st), "forward_level")
        else:
            _lib.check(L_.smd_engine_forward(eng.h, x.data_ptr(), sig_vec.data_ptr(), grad.data_ptr(), st), "forward")
        _lib.check(L_.smd_langevin_step(C.byref(io), B, S, Cn, st), "langevin_step")

    with torch.cuda.device(dev):
        if not use_graph or n < 3:
            for _ in range(n):
                one()
            return
        cur = torch.cuda.current_stream(dev)
        st = torch.cuda.Stream(device=dev)
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            one()                                                  # update 0 (also the warm-up of the capture)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            one()
        with torch.cuda.stream(st):
            for _ in range(n - 1):
                g.replay()
        cur.wait_stream(st)
    torch.cuda.current_stream(dev).synchronize()                   # the tables / counters above go out of scope


def annealed_langevin_dynamics(rng: PRNGKey, model: Model, sigmas, init, epsilon, T, denoise, infill=False,
                               infill_samples=None, infill_masks=None, *, noises: Optional[Callable] = None,
                               infill_noises: Optional[Callable] = None, sample_offset: int = 0,
                               global_num_samples: Optional[int] = None, use_graph: bool = True):
    """utils/ebm_utils.py:89-198.  Returns (state, collection (100 + 1 + int(denoise), ...), ld_metrics (4, L, T)).

    Per update: grad = model(state, sigma) through the engine, then ONE fused kernel (csrc/diffusion.hip
    langevin_step_kernel) applies next = state + alpha grad + sqrt(2 alpha) z, the infill blend, the three norm metrics and
    the collection copy.  ``noises(sigma_i, i)`` / ``infill_noises(sigma_i, i)`` supply the normals explicitly; otherwise a
    ThreefryKey reproduces the reference's ``split(rng, 3)`` stream per update and an engine key uses Philox."""
    eng = model.engine
    dev = eng.device
    sig = np.asarray(sigmas, dtype=np.float32)
    assert len(sig) >= 2                                                            # :181
    L, T = len(sig), int(T)
    init = torch.as_tensor(init).to(dev, torch.float32).contiguous()
    B = init.shape[0]
    S, Cn = eng.S, eng.C
    if infill:
        inf_s = torch.as_tensor(infill_samples).to(dev, torch.float32).contiguous()
        inf_m = torch.as_tensor(infill_masks).to(dev, torch.float32).contiguous()
        start = init * (1 - inf_m) + inf_s * inf_m                                # :128
    else:
        inf_s = inf_m = None
        start = init
    n_coll = ALD_COLLECTION_STEPS + 1 + int(bool(denoise))
    collection = torch.zeros((n_coll, *init.shape), dtype=torch.float32, device=dev)
    collection[0] = start
    cidx = np.linspace(1, L * T, ALD_COLLECTION_STEPS).astype(np.int32)            # :131-133
    x = init.clone()
    metrics = torch.zeros((L * T, B, 3), dtype=torch.float32, device=dev)
    alphas = np.zeros(L, dtype=np.float32)
    jax_mode = isinstance(rng, ThreefryKey) and noises is None
    if jax_mode:
        step_keys, infill_keys = _jr.langevin_key_table(rng, L * T)
        per = int(np.prod(init.shape[1:]))
        n_glob = (B + sample_offset if global_num_samples is None else int(global_num_samples)) * per
    zbuf = torch.zeros_like(x) if noises is not None else None
    izbuf = torch.zeros_like(x) if (infill and infill_noises is not None) else None
    sig_vec = torch.empty((B,), dtype=torch.float32, device=dev)
    if noises is None and infill_noises is None:
        # no explicit draws: every update's arguments go into device tables and the whole run is one replayed graph
        rows = np.zeros((L * T, 5), dtype=np.float32)      # alpha, sqrt(2 alpha), infill sigma, NEXT forward's sigma | own sigma
        slots = np.full((L * T,), -1, dtype=np.int32)
        for si in range(L):
            sigma = np.float32(sig[si])
            alpha = np.float32(epsilon) * (sigma / np.float32(sig[-1])) ** 2     # :168
            alphas[si] = alpha
            for i in range(T):
                k = si * T + i
                nxt = np.float32(sig[min(L - 1, (k + 1) // T)])
                rows[k] = (alpha, np.sqrt(np.float32(2) * alpha), sigma, nxt, sigma)
                slot = ald_collection_slot(cidx, k + 1)                           # :149-156 (duplicates add up, as upstream)
                slots[k] = slot if 0 < slot < n_coll else -1
        io = _lib.LangevinIO()
        io.seed_lo, io.seed_hi = rng.seed & 0xFFFFFFFF, (rng.seed >> 32) & 0xFFFFFFFF
        io.sample_offset = int(sample_offset)
        keys = None
        if jax_mode:
            io.use_threefry, io.tf_n_total = 1, n_glob
            keys = np.concatenate([np.asarray(step_keys, dtype=np.uint32), np.asarray(infill_keys, dtype=np.uint32)], axis=1)
        if inf_m is not None:
            io.infill_samples, io.infill_masks = inf_s.data_ptr(), inf_m.data_ptr()
        _langevin_graph_loop(model, x, io, rows, slots, keys, metrics, collection, use_graph, levels=sig, steps_per_level=T)
    else:
      for si in range(L):
        sigma = np.float32(sig[si])
        alpha = np.float32(epsilon) * (sigma / np.float32(sig[-1])) ** 2         # :168
        alphas[si] = alpha
        sig_vec.fill_(float(sigma))
        for i in range(T):
            grad = model(x, sig_vec)                                                # :140
            image_idx = si * T + i + 1                                              # :149-156 (duplicates add up, as upstream)
            slot = ald_collection_slot(cidx, image_idx)
            k = si * T + i
            io = _langevin_io(x, grad, alpha, np.sqrt(np.float32(2) * alpha), rng, k, sample_offset, metrics[k],
                              collection[slot] if 0 < slot < n_coll else None)
            if jax_mode:
                # explicit infill draws next to the reference's own step-noise stream (noises is None): the update draws
                # jax.random.normal of this iteration's key itself instead of falling back to Philox
                io.use_threefry, io.tf_n_total = 1, n_glob
                io.tf_noise_key[0], io.tf_noise_key[1] = int(step_keys[k][0]), int(step_keys[k][1])
                io.tf_infill_key[0], io.tf_infill_key[1] = int(infill_keys[k][0]), int(infill_keys[k][1])
            if zbuf is not None:
                zbuf.copy_(torch.as_tensor(noises(si, i)).to(dev, torch.float32))
                io.z_in = zbuf.data_ptr()
            if inf_m is not None:
                io.infill_samples, io.infill_masks, io.infill_sigma = inf_s.data_ptr(), inf_m.data_ptr(), float(sigma)
                if izbuf is not None:
                    izbuf.copy_(torch.as_tensor(infill_noises(si, i)).to(dev, torch.float32))
                    io.infill_z_in = izbuf.data_ptr()
            with torch.cuda.device(dev):
                _lib.check(_lib.get_lib().smd_langevin_step(C.byref(io), B, S, Cn, torch.cuda.current_stream().cuda_stream),
                           "langevin_step")
    if denoise:                                                                     # :189-192
        sig_vec.fill_(float(sig[-1]))
        x = x + float(np.float32(sig[-1]) ** 2) * model(x, sig_vec)
        collection[-1] = x
    m = metrics.view(L, T, B, 3)
    denom = float(B) if S == 1 else float(B * Cn)
    per = m.sum(dim=2) / denom                                                      # (L, T, 3)
    ld = torch.zeros((4, L, T), dtype=torch.float32, device=dev)
    ld[0], ld[1], ld[3] = per[..., 0], per[..., 1], per[..., 2]
    ld[2] = torch.from_numpy(alphas).to(dev).view(L, 1).expand(L, T)
    return x, collection, ld


def consistent_langevin_dynamics(rng: PRNGKey, model: Model, sigmas, init, epsilon, T=None, denoise=True, infill=False,
                                 infill_samples=None, infill_masks=None, *, noises: Optional[Callable] = None,
                                 sample_offset: int = 0, global_num_samples: Optional[int] = None, use_graph: bool = True):
    """utils/ebm_utils.py:201-271: one update per noise level, noise = beta * sigma_{i+1} * z.  Returns
    (state, ld_metrics (4, L, 1)) like the reference (two values; T is a null parameter there too)."""
    del T, infill_samples, infill_masks
    if infill:
        raise NotImplementedError                                                   # :228-229
    eng = model.engine
    dev = eng.device
    sig = np.asarray(sigmas, dtype=np.float32)
    assert len(sig) >= 2
    L = len(sig)
    x = torch.as_tensor(init).to(dev, torch.float32).contiguous().clone()
    B, S, Cn = x.shape[0], eng.S, eng.C
    f = np.float32
    beta = np.sqrt(f(1) - (f(1) - f(epsilon) / (sig[-1] ** 2)) ** 2, dtype=np.float32)      # :257
    metrics = torch.zeros((L, B, 3), dtype=torch.float32, device=dev)
    alphas = np.zeros(L, dtype=np.float32)
    jax_mode = isinstance(rng, ThreefryKey) and noises is None
    if jax_mode:
        step_keys, _ = _jr.langevin_key_table(rng, L, consistent=True)
        per = int(np.prod(x.shape[1:]))
        n_glob = (B + sample_offset if global_num_samples is None else int(global_num_samples)) * per
    zbuf = torch.zeros_like(x) if noises is not None else None
    sig_vec = torch.empty((B,), dtype=torch.float32, device=dev)
    if noises is None:                  # one captured (forward + update) pair replayed for every noise level
        rows = np.zeros((L, 5), dtype=np.float32)
        for i in range(L):
            sigma = f(sig[i])
            next_sigma = f(sig[i + 1]) if i < L - 1 else f(0)                      # :236
            alphas[i] = f(epsilon) * (sigma / f(sig[-1])) ** 2                     # :238
            rows[i] = (alphas[i], beta * next_sigma, 0.0, f(sig[min(i + 1, L - 1)]), sigma)
        io = _lib.LangevinIO()
        io.seed_lo, io.seed_hi = rng.seed & 0xFFFFFFFF, (rng.seed >> 32) & 0xFFFFFFFF
        io.sample_offset = int(sample_offset)
        keys = None
        if jax_mode:
            io.use_threefry, io.tf_n_total = 1, n_glob
            keys = np.concatenate([np.asarray(step_keys, dtype=np.uint32), np.zeros((L, 2), np.uint32)], axis=1)
        _langevin_graph_loop(model, x, io, rows, None, keys, metrics, None, use_graph, levels=sig, steps_per_level=1)
    else:
      for i in range(L):
        sigma = f(sig[i])
        next_sigma = f(sig[i + 1]) if i < L - 1 else f(0)                          # :236
        alpha = f(epsilon) * (sigma / f(sig[-1])) ** 2                             # :238
        alphas[i] = alpha
        sig_vec.fill_(float(sigma))
        grad = model(x, sig_vec)
        io = _langevin_io(x, grad, alpha, beta * next_sigma, rng, i, sample_offset, metrics[i], None)
        zbuf.copy_(torch.as_tensor(noises(i)).to(dev, torch.float32))
        io.z_in = zbuf.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().smd_langevin_step(C.byref(io), B, S, Cn, torch.cuda.current_stream().cuda_stream),
                       "langevin_step")
    if denoise:                                                                     # :264-265
        sig_vec.fill_(float(sig[-1]))
        x = x + float(f(sig[-1]) ** 2) * model(x, sig_vec)
    denom = float(B) if S == 1 else float(B * Cn)
    per = metrics.sum(dim=1) / denom                                                # (L, 3)
    ld = torch.zeros((4, L, 1), dtype=torch.float32, device=dev)
    ld[0, :, 0], ld[1, :, 0], ld[3, :, 0] = per[:, 0], per[:, 1], per[:, 2]
    ld[2, :, 0] = torch.from_numpy(alphas).to(dev)
    return x, ld


# ------------------------------------------------------------------ reverse sampler
def collate_sampling_metrics(ld_metrics):
    """utils/ebm_utils.py:408-428."""
    num_metrics, num_sigmas, num_steps = ld_metrics.shape
    del num_metrics
    out = [[] for _ in range(num_sigmas)]
    for i in range(num_sigmas):
        grad_norm, step_norm, alpha, noise_norm = ld_metrics[:, i, :]
        for j in range(num_steps):
            out[i].append({"slope": grad_norm[j], "step": step_norm[j], "alpha": alpha[j], "noise": noise_norm[j]})
    return out


def sampler_chain_sizes(model: "Model", B: int, graphed: bool, allow_pad: bool = True) -> Tuple[List[int], int]:
    """How the graph-replayed walk of B sequences is arranged: ``(sizes, pad)``.  Two concurrent chains (SMD_SAMPLER_CHAINS=1
    turns it off) whenever B >= 128: the batch is split UNEVENLY in multiples of the row granule of the 256-row GEMM tiles
    (8 sequences of 32 tokens), e.g. the reference's default ``sample_size = 1000`` (sample_ncsn.py:54) walks as 504 + 496; a batch
    that is not a multiple of the granule is padded with ``pad`` <= 7 throw-away sequences (samples are independent and every draw
    is keyed by the global sample index, so the padding changes no result; it is sliced off before anything is returned).
    ``allow_pad=False`` (jax.random streams: their counter layout is a function of the true array size): such a batch walks as one chain."""
    eng = model.engine
    if not graphed or os.environ.get("SMD_SAMPLER_CHAINS", "2") != "2" or B < 128 or eng.cfg.mlp_dims % 256:
        return [B], 0
    gran = 256 // math.gcd(256, eng.S)                  # sequences per 256 token rows: 8 for S = 32, 256 for DenseDDPM
    pad = (-B) % gran
    if pad and (not allow_pad or gran > 8):
        return [B], 0
    Bp = B + pad
    h0 = (Bp // 2 + gran - 1) // gran * gran
    if Bp - h0 < gran:
        return [B], 0
    return [h0, Bp - h0], pad


def _sampler_chains(model: "Model", B: int, graphed: bool) -> int:
    """Number of concurrent chains of the graph-replayed walk of B sequences (see sampler_chain_sizes)."""
    return len(sampler_chain_sizes(model, B, graphed)[0])


def _sampler_pipeline_unroll() -> int:
    """Iterations per captured graph of the pipelined two-chain walk (0: the two chains free-running as one-step graphs, the
    round-4 arrangement; SMD_SAMPLER_PIPELINE=0 / SMD_SAMPLER_UNROLL=U in the environment).  Default 8: the join and the launch
    gap of both chains coincide once per replay -- 1000-step walk 0.552 / 0.535 / 0.5325 / 0.531 s at 1 / 4 / 8 / 16 (profiles/r5m_sampler_walk.txt)."""
    if os.environ.get("SMD_SAMPLER_PIPELINE", "1") == "0":
        return 0
    return max(1, int(os.environ.get("SMD_SAMPLER_UNROLL", "8")))


def diffusion_dynamics(rng: PRNGKey, model: Model, betas, init, epsilon=None, T=None, denoise=None, infill=False,
                       infill_samples=None, infill_masks=None, *, noises: Optional[Callable] = None,
                       infill_noises: Optional[Callable] = None, t_start: Optional[int] = None, t_stop: int = 0,
                       use_graph: bool = True, sample_offset: int = 0, global_num_samples: Optional[int] = None):
    """utils/ebm_utils.py:280-405.  Returns (state, collection (41, ...), ld_metrics (4, T, 1)).

    epsilon / T / denoise are null parameters as in the reference.  ``noises(t)`` /
    ``infill_noises(t)`` supply the normal draws of :360-362 / :342-345 explicitly (parity mode);
    otherwise the fused reverse-step kernel draws them from Philox keyed by (rng, global sample
    index, t) and the step is replayed from a captured hipGraph.  With a ``ThreefryKey`` the draws are the
    reference's own: the three splits per iteration (:329,342,360) are unrolled on the host into per-iteration
    key tables and a threefry kernel inside the captured step writes normal(noise_rng) (and normal(infill_rng))
    for this rank's rows of the (global_num_samples, ...) state.  ``t_start/t_stop`` bound the walk
    (default T-1 .. 0); the key sequence always starts at the first iteration, like the reference's scan."""
    del epsilon, T, denoise
    eng = model.engine
    dev = eng.device
    init = torch.as_tensor(init).to(dev, torch.float32).contiguous()
    B = init.shape[0]
    if tuple(init.shape[1:]) != eng.cfg.sample_shape:
        raise ValueError(f"init shape {tuple(init.shape)} != (B, {eng.cfg.sample_shape})")
    nT = len(betas)
    if infill:
        inf_s = torch.as_tensor(infill_samples).to(dev, torch.float32).contiguous()
        inf_m = torch.as_tensor(infill_masks).to(dev, torch.float32).contiguous()
        start = init * (1 - inf_m) + inf_s * inf_m                                # :321
    else:
        inf_s = inf_m = None
        start = init
    x = init.clone()
    t_hi = nT - 1 if t_start is None else int(t_start)
    steps = list(range(t_hi, t_stop - 1, -1))
    explicit = noises is not None or infill_noises is not None
    jax_mode = isinstance(rng, ThreefryKey) and not explicit
    graphed = use_graph and not explicit and len(steps) > 1
    per = int(np.prod(init.shape[1:]))
    n_glob = (B + sample_offset if global_num_samples is None else int(global_num_samples)) * per
    nk_d = ik_d = None
    if jax_mode:
        # the reference's own draws: the three splits per iteration (:329,342,360) are unrolled on the host into key
        # tables; the fused reverse step reads row (t_hi - t) on the device and evaluates jax.random.normal for its
        # elements of the global (N, S, C) array in place of its Philox draw
        ik, nk = _jr.sampler_key_tables(rng, len(steps))
        nk_d = torch.from_numpy(nk.view(np.int32).copy()).to(dev)
        ik_d = torch.from_numpy(ik.view(np.int32).copy()).to(dev) if infill else None

    # Two concurrent half-batch chains (graph replay only): samples are independent, so the batch is walked as two chains
    # of B/2 on two streams, each its own engine handle; one chain's launch-latency-bound encoder kernels then overlap the
    # other's MFMA phases.  Draws are keyed by the GLOBAL sample index, so the result does not depend on the split.
    # PIPELINED (default): two FREE-running chains drift into phase within ~100 iterations -- both in their encoder, then both
    # in their 2048-wide GEMMs -- which is the slow mode (1650-1740 steps/s against 1900-2000 half a period apart: the
    # round-4 "bimodality", profiles/r5a_chain_phase_*.txt).  So chain A is captured as (output stage + reverse update of
    # iteration k, stem of iteration k + 1) and chain B as (stem, output stage) of iteration k, `unroll` iterations per graph,
    # and every replay of one chain waits for the previous replay of the other: the phase is re-locked every `unroll`
    # iterations and the two halves always complement each other (include/smd_hip.h smd_engine_sample_step_part).
    # The split is uneven in multiples of 8 sequences (B = 1000, the reference's default sample_size: 504 + 496) and a batch that
    # is not a multiple of 8 is padded with throw-away sequences, so every B >= 128 takes this path (sampler_chain_sizes).
    sizes, pad = sampler_chain_sizes(model, B, graphed, allow_pad=not jax_mode)
    nchains = len(sizes)
    offs = [0] + list(np.cumsum(sizes)[:-1])
    unroll = _sampler_pipeline_unroll() if nchains == 2 else 0
    engines = model.chain_engines(nchains) if nchains > 1 else [eng]
    model.sampler_arrangement = dict(batch=B, chains=nchains, chain_sizes=list(sizes), padded=pad, graphed=bool(graphed),
                                     pipelined_unroll=unroll, rng="threefry" if jax_mode else ("explicit" if explicit else "philox"))
    _log.info("diffusion_dynamics: B=%d as %d chain(s) %s (+%d padding), %s", B, nchains, list(sizes), pad,
              f"pipelined, {unroll} steps per graph" if unroll else ("graph replay" if graphed else "eager launches"))
    if pad:
        zp = lambda t: None if t is None else torch.cat([t, torch.zeros((pad, *t.shape[1:]), dtype=t.dtype, device=dev)])
        x, start, inf_s, inf_m = zp(x), zp(start), zp(inf_s), zp(inf_m)
    for c, e in enumerate(engines):
        _ensure_schedule(e, betas, with_sampler=True)
        e.bind(sizes[c], training=False)
    # A captured step is valid for LATER runs too as long as every pointer it holds is: the state / collection / metrics
    # buffers, the device-resident timestep, Philox key (smd_sample_io.key_ptr) and jax.random key tables are kept with the
    # graphs (one set per model) and refilled per run; the weights are read through the shared operand pack.  A second
    # sample() call then pays no warm-up step, capture or instantiation (7 % of a 1000-step walk).
    # ... and as long as nothing the captured kernels bake in has changed under it: every handle counts its binds, schedule
    # changes and option changes (Engine.generation); process-wide tuning knobs are part of the key through lib.tuning_epoch()
    sig = tuple((id(e), e.generation) for e in engines) + (_lib.tuning_epoch(),)
    ckey = (B, tuple(sizes), pad, nchains, unroll, bool(infill), jax_mode, nT, t_hi, len(steps), int(sample_offset), int(n_glob), tuple(init.shape[1:]))
    cache = model.__dict__.setdefault("_sampler_graphs", {})
    entry = cache.get("entry") if graphed else None
    reuse = entry is not None and entry["key"] == ckey and entry["sig"] == sig
    if reuse:
        x = entry["x"]
        x[:B].copy_(init)
        if pad:
            x[B:].zero_()
        chains = entry["chains"]
        if jax_mode:
            entry["nk_d"].copy_(nk_d)
            if ik_d is not None:
                entry["ik_d"].copy_(ik_d)
        if infill:
            entry["inf_s"].copy_(inf_s)
            entry["inf_m"].copy_(inf_m)
    else:
        if graphed:
            cache.pop("entry", None)                  # one set of persistent buffers per model
            entry = dict(key=ckey, sig=sig, x=x, nk_d=nk_d, ik_d=ik_d, inf_s=inf_s, inf_m=inf_m)
        chains = []
    key_words = torch.tensor([rng.seed & 0xFFFFFFFF, (rng.seed >> 32) & 0xFFFFFFFF], dtype=torch.int64).to(torch.int32)
    for c, e in enumerate(engines):
        if c == 0:
            e.refresh_weights()    # the fp32 master may have been trained since the last call (the operand pack is shared)
        e.prepare_sampler()        # FiLM scale/shift tables for all T noise levels (3 small GEMMs per block)
        h = sizes[c]
        lo, hi = int(offs[c]), int(offs[c]) + h
        if reuse:
            ch = chains[c]
            ch["coll"].zero_()
            ch["metrics"].zero_()
            ch["t_ptr"].fill_(t_hi)
        else:
            ch = dict(eng=e, x=x[lo:hi], coll=torch.zeros((COLLECTION_STEPS + 1, h, *init.shape[1:]), dtype=torch.float32, device=dev),
                      metrics=torch.zeros((nT, h, 3), dtype=torch.float32, device=dev),
                      t_ptr=torch.tensor([t_hi], dtype=torch.int32, device=dev),
                      key=torch.zeros(2, dtype=torch.int32, device=dev))
            io = _lib.SampleIO()
            io.x = ch["x"].data_ptr(); io.t_ptr = ch["t_ptr"].data_ptr()
            io.seed_lo = rng.seed & 0xFFFFFFFF; io.seed_hi = (rng.seed >> 32) & 0xFFFFFFFF
            io.key_ptr = ch["key"].data_ptr()
            io.sample_offset = sample_offset + lo
            io.infill_samples = None if inf_s is None else inf_s[lo:hi].data_ptr()
            io.infill_masks = None if inf_m is None else inf_m[lo:hi].data_ptr()
            io.metrics_partial = ch["metrics"].data_ptr()
            io.collection = ch["coll"].data_ptr()
            io.slot_table = e.slot_table.data_ptr()
            if jax_mode:
                io.tf_noise_keys = nk_d.data_ptr()
                io.tf_infill_keys = None if ik_d is None else ik_d.data_ptr()
                io.tf_n_total = n_glob
                io.tf_t0 = t_hi
            ch["io"] = io
            chains.append(ch)
        ch["key"].copy_(key_words)
        ch["coll"][0] = start[lo:hi]                                              # :322-323
        e.load_state(ch["x"])

    if explicit:
        ch = chains[0]
        zbuf = torch.zeros_like(x)
        izbuf = torch.zeros_like(x) if infill else None
        ch["io"].z_in = zbuf.data_ptr()
        ch["io"].infill_z_in = None if izbuf is None else izbuf.data_ptr()
        for t in steps:
            if t > 0 and noises is not None:
                zbuf.copy_(torch.as_tensor(noises(t)).to(dev, torch.float32))
            if t > 0 and izbuf is not None and infill_noises is not None:
                izbuf.copy_(torch.as_tensor(infill_noises(t)).to(dev, torch.float32))
            ch["eng"].sample_step(ch["io"])
    elif graphed:
        cur = torch.cuda.current_stream(dev)
        replays = len(steps) - 1

        def capture(ch, body):
            ch["stream"].synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=ch["stream"]):
                body()
            return g

        # the first iteration as plain launches on the chain's stream: the warm-up of a capture, and whatever a handle does
        # lazily in front of a forward pass (fp8 mode: e4m3 copies of refreshed weights) happens here, outside the graphs
        streams = model.chain_streams(len(chains))
        for ch, st in zip(chains, streams):
            ch["stream"] = st
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                ch["eng"].sample_step(ch["io"])
        if not reuse:
            for c, ch in enumerate(chains):
                e, io = ch["eng"], ch["io"]
                if unroll:          # chain 0: (output stage, next stem) x unroll; chain 1: (stem, output stage) x unroll
                    order = (2, 1) if c == 0 else (1, 2)
                    ch["graph"] = capture(ch, lambda e=e, io=io, order=order: [e.sample_step(io, part) for _ in range(unroll) for part in order])
                else:
                    ch["graph"] = capture(ch, lambda e=e, io=io: e.sample_step(io))
            entry["chains"] = chains
            cache["entry"] = entry
        timing = os.environ.get("SMD_SAMPLER_TIMING") == "1"          # diagnostics (tools/sampler_walk_time.py): blocks the host twice
        if timing:
            torch.cuda.synchronize()
            t_loop = time.perf_counter()
        if unroll:
            A, Bc = chains
            with torch.cuda.stream(A["stream"]):
                A["eng"].sample_step(A["io"], 1)                      # the pipeline's prologue: chain A's stem of the next iteration
            ev = [[torch.cuda.Event() for _ in range(2)] for _ in range(2)]   # [chain][replay parity]
            q, r = divmod(replays, unroll)
            for i in range(q):
                for c, ch in enumerate(chains):
                    with torch.cuda.stream(ch["stream"]):
                        if i > 0:
                            ch["stream"].wait_event(ev[1 - c][(i - 1) & 1])     # the other chain's previous replay
                        ch["graph"].replay()
                        ev[c][i & 1].record(ch["stream"])
            for _ in range(r):                                        # the iterations that do not fill a graph, as plain launches
                with torch.cuda.stream(A["stream"]):
                    A["eng"].sample_step(A["io"], 2)
                    A["eng"].sample_step(A["io"], 1)
                with torch.cuda.stream(Bc["stream"]):
                    Bc["eng"].sample_step(Bc["io"], 1)
                    Bc["eng"].sample_step(Bc["io"], 2)
            # (chain A ends one stem ahead: a pass over the final state that nothing reads)
        else:
            for _ in range(replays):
                for ch in chains:
                    with torch.cuda.stream(ch["stream"]):
                        ch["graph"].replay()
        if timing:
            t_issued = time.perf_counter()
            torch.cuda.synchronize()
            model._sampler_timing = dict(replays=replays, host_issue_s=t_issued - t_loop, loop_s=time.perf_counter() - t_loop, reused=reuse)
        for ch in chains:
            cur.wait_stream(ch["stream"])
    else:
        for _ in steps:
            chains[0]["eng"].sample_step(chains[0]["io"])

    if graphed:
        x = x[:B].clone()                  # the persistent state buffer belongs to the cached graphs
    collection = (chains[0]["coll"].clone() if graphed else chains[0]["coll"]) if nchains == 1 else torch.cat([ch["coll"] for ch in chains], dim=1)[:, :B]
    metrics_partial = chains[0]["metrics"] if nchains == 1 else torch.cat([ch["metrics"] for ch in chains], dim=1)[:, :B]
    # ld_metrics rows (grad_norm, step_norm, alpha_prod, noise_norm), one column per iteration (:380-405)
    denom = float(B) if eng.S == 1 else float(B * eng.C)
    per_t = metrics_partial.sum(dim=1) / denom                                      # (T, 3) indexed by t
    ap = chains[0]["eng"]._sched_tensors["coef"][:, 5]
    ld = torch.zeros((4, nT, 1), dtype=torch.float32, device=dev)
    idx = torch.tensor(steps, device=dev, dtype=torch.long)
    rows = torch.tensor([nT - 1 - t for t in steps], device=dev, dtype=torch.long)
    ld[0, rows, 0] = per_t[idx, 0]
    ld[1, rows, 0] = per_t[idx, 1]
    ld[2, rows, 0] = ap[idx]
    ld[3, rows, 0] = per_t[idx, 2]
    return x, collection, ld


def sample(scorenet: Model, sigmas, rng: PRNGKey, sample_shape, num_samples=2400, sampling="ald", epsilon=1e-3,
           steps=100, denoise=True, *, sample_offset: int = 0, use_graph: bool = True,
           global_num_samples: Optional[int] = None):
    """train_ncsn.py:499-551.  'ddpm': N(0,1) init + diffusion_dynamics.  'ald' / 'cas': uniform(-sqrt(12)/2, sqrt(12)/2)
    init (:542-547) + annealed / consistent Langevin dynamics with the score network ``scorenet``.  The reference unpacks
    three values from every sampler although consistent_langevin_dynamics returns two (a ValueError upstream); here 'cas'
    returns a two-entry collection [init, final state]."""
    if sampling not in ("ddpm", "ald", "cas"):
        raise ValueError(f"Unknown sampling algorithm: {sampling}")
    init_rng, ld_rng = split(rng)                                                    # :536
    eng = scorenet.engine
    if tuple(sample_shape) != eng.cfg.sample_shape:
        raise ValueError(f"sample_shape {tuple(sample_shape)} != model shape {eng.cfg.sample_shape}")
    if sampling != "ddpm":
        rho = float(np.sqrt(np.float32(12)) / 2)                                     # :543
        n_all = num_samples + sample_offset if global_num_samples is None else int(global_num_samples)
        per = int(np.prod(sample_shape))
        if isinstance(init_rng, ThreefryKey):
            init = _jr.uniform(init_rng, (num_samples, *sample_shape), eng.device, -rho, rho, n_total=n_all * per,
                               offset=sample_offset * per)
        else:      # engine key: one seeded draw of the GLOBAL array, this rank keeps its rows
            g = torch.Generator(device="cpu").manual_seed(init_rng.seed & 0x7FFFFFFFFFFFFFFF)
            init = ((torch.rand((n_all, *sample_shape), generator=g) * 2 - 1) * rho)[sample_offset:sample_offset + num_samples]
            init = init.to(eng.device)
        if sampling == "ald":
            generated, collection, ld = annealed_langevin_dynamics(ld_rng, scorenet, sigmas, init, epsilon, steps, denoise,
                                                                   False, sample_offset=sample_offset,
                                                                   global_num_samples=global_num_samples, use_graph=use_graph)
        else:
            generated, ld = consistent_langevin_dynamics(ld_rng, scorenet, sigmas, init, epsilon, steps, denoise, False,
                                                         sample_offset=sample_offset, global_num_samples=global_num_samples,
                                                         use_graph=use_graph)
            collection = torch.stack([init, generated])
        return generated, collection, collate_sampling_metrics(ld.cpu().numpy())
    eng.bind(num_samples, training=False)
    init = torch.empty((num_samples, *sample_shape), dtype=torch.float32, device=eng.device)
    if isinstance(init_rng, ThreefryKey):                                            # :539-540 N(0,1), jax stream
        per = int(np.prod(sample_shape))
        n_glob = (num_samples + sample_offset if global_num_samples is None else int(global_num_samples)) * per
        _jr.normal(init_rng, init.shape, eng.device, n_total=n_glob, offset=sample_offset * per, out=init)
    else:
        eng.init_state(init, init_rng.seed, sample_offset)                           # :539-540 N(0,1)
    generated, collection, ld_metrics = diffusion_dynamics(ld_rng, scorenet, sigmas, init, epsilon, steps, denoise,
                                                           False, sample_offset=sample_offset, use_graph=use_graph,
                                                           global_num_samples=global_num_samples)
    return generated, collection, collate_sampling_metrics(ld_metrics.cpu().numpy())
