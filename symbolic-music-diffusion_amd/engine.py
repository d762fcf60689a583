"""Python owner of an smd_engine handle: torch allocates every device buffer, the C-ABI library
(include/smd_hip.h) launches the HIP kernels on torch's current stream.

The eps-network state is one flat fp32 parameter buffer (+ Adam m/v, EMA, gradients of the same
size) with named views in the reference's flax layout, see ``tensor_table``.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import os

import numpy as np
import torch

from . import lib as _lib
from . import schedule as _sched

ARCH_IDS = {"TransformerDDPM": 0, "TransformerDDPM4": 0, "DenseDDPM": 1}


@dataclass
class NetConfig:
    """model kwargs of train_ncsn.py:321-326 + data shape.  ``TransformerDDPM4`` (named by
    configs/ddpm-multi-32seq-512.cfg:1 but absent upstream) maps to TransformerDDPM."""
    architecture: str = "TransformerDDPM"
    data_channels: int = 512
    seq_len: int = 32
    num_layers: int = 6
    num_heads: int = 8
    num_mlp_layers: int = 2
    mlp_dims: int = 2048
    num_timesteps: int = 1000
    dtype: str = "bf16"          # "fp8": e4m3 DenseResBlock forward GEMMs (--dtype=fp8, BASELINE config 5)

    @property
    def sample_shape(self) -> Tuple[int, ...]:
        if ARCH_IDS[self.architecture] == 1:
            return (self.data_channels,)
        return (self.seq_len, self.data_channels)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _joined(name: str) -> property:
    """Buffer attribute whose READ first orders the current stream behind a deferred output-stage optimiser update (any
    handle on the same parameters): ``opt_overlap`` lets the 86 % of the update that the next forward pass does not need
    until its `up` projection run on the engine's side stream, so everything that looks at the parameters, the Adam state,
    the EMA or the operand pack through Python -- another handle, a checkpoint, a test -- joins here."""
    def get(self):
        self._join_pending()
        return self.__dict__.get("_buf_" + name)

    def set(self, value):
        self.__dict__["_buf_" + name] = value
    return property(get, set)


class Engine:
    """One engine handle + its parameter buffers.  ``bind(batch, training)`` (re)allocates the
    activation workspace for a batch size / mode."""
    params = _joined("params")
    wpack = _joined("wpack")
    grads = _joined("grads")
    m = _joined("m")
    v = _joined("v")
    ema = _joined("ema")

    def __init__(self, cfg: NetConfig, device: str = "cuda:0", share_params_with: Optional["Engine"] = None):
        if cfg.architecture not in ARCH_IDS:
            raise ValueError(f"Unsupported architecture {cfg.architecture!r}; the HIP engine implements "
                             f"{sorted(ARCH_IDS)} (DenseNCSN/ConvNCSN are broken upstream, models/ncsn.py:92,111)")
        if not torch.cuda.is_available():
            raise RuntimeError("smd_amd.Engine needs a ROCm GPU: the HIP path has no CPU fallback")
        self.cfg = cfg
        self.device = torch.device(device)
        self.L = _lib.get_lib()
        d = _lib.ModelDesc(ARCH_IDS[cfg.architecture], cfg.data_channels,
                           1 if ARCH_IDS[cfg.architecture] == 1 else cfg.seq_len, cfg.num_layers, cfg.num_heads,
                           cfg.num_mlp_layers, cfg.mlp_dims, 128, 128, cfg.num_timesteps)
        h = C.c_void_p()
        _lib.check(self.L.smd_engine_create(C.byref(d), C.byref(h)), "smd_engine_create")
        self.h = h
        if cfg.dtype not in ("bf16", "fp8"):
            raise ValueError(f"dtype must be 'bf16' or 'fp8', got {cfg.dtype!r}")
        if cfg.dtype == "fp8":
            _lib.check(self.L.smd_engine_set_option(h, b"fp8", 1), "set_option fp8")
        self.S = d.seq_len
        self.C = cfg.data_channels
        self.n_params = int(self.L.smd_engine_param_count(h))
        self.head_offset = int(self.L.smd_engine_head_param_offset(h))
        self.tensor_table: List[Tuple[str, int, Tuple[int, ...]]] = []
        for i in range(self.L.smd_engine_num_tensors(h)):
            name, off, r, c = C.c_char_p(), C.c_int64(), C.c_int32(), C.c_int32()
            _lib.check(self.L.smd_engine_tensor_info(h, i, C.byref(name), C.byref(off), C.byref(r), C.byref(c)))
            shape = (r.value, c.value) if c.value > 0 else (r.value,)
            self.tensor_table.append((name.value.decode(), off.value, shape))
        with torch.cuda.device(self.device):
            if share_params_with is not None:
                self.params, self.wpack = share_params_with.params, share_params_with.wpack
            else:
                self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
                self.wpack = torch.zeros(int(self.L.smd_engine_wpack_elems(h)), dtype=torch.int16, device=self.device)
            _lib.check(self.L.smd_engine_bind_params(h, _ptr(self.params), _ptr(self.wpack)), "bind_params")
        # engines sharing one operand pack: a weight refresh through any handle invalidates every handle's e4m3 copies
        self._wstate = share_params_with._wstate if share_params_with is not None else {"ver": 0, "pending": None}
        self._wseen = -1
        self._opt_overlap = 0
        for kv in filter(None, os.environ.get("SMD_ENGINE_OPTS", "").split(",")):      # A/B runs: "key=value,key=value"
            k, _, v = kv.partition("=")
            _lib.check(self.L.smd_engine_set_option(h, k.strip().encode(), int(v)), f"SMD_ENGINE_OPTS {kv}")
        self._label_min = 1
        self._loss_kind = 0
        self.generation = 0          # bumped by whatever invalidates a captured step of this handle: bind, schedule, options
        self.grads = self.m = self.v = self.ema = None
        self.step_counter = None
        self.metrics = None
        self.workspace = None
        self.batch = 0
        self.training = False
        self._sched_tensors = None
        self.betas = None

    def _join_pending(self) -> None:
        """Order the current stream behind a deferred output-stage update of whichever handle shares these buffers."""
        ws = self.__dict__.get("_wstate")
        if not ws:
            return
        p = ws.get("pending")
        if p is not None and getattr(p, "h", None):
            ws["pending"] = None
            with torch.cuda.device(p.device):
                _lib.check(p.L.smd_engine_join_update(p.h, _stream()), "join_update")

    def set_opt_overlap(self, value: int) -> None:
        """Engine option "opt_overlap" (include/smd_hip.h): bit 0 defers the output-stage update to the side stream, bit 1
        reduces that slice's norm partials early.  SMD_OPT_OVERLAP in the environment overrides (A/B runs; the trainer's default
        is 0: DESIGN.md section 6)."""
        value = int(os.environ.get("SMD_OPT_OVERLAP", value))
        if value != self._opt_overlap:
            self.set_option("opt_overlap", value)
            self._opt_overlap = value

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.smd_engine_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def named_views(self, flat: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        flat = self.params if flat is None else flat
        out = {}
        for name, off, shape in self.tensor_table:
            out[name] = flat[off:off + int(np.prod(shape))].view(*shape)
        return out

    def load_named(self, tensors: Dict[str, "torch.Tensor | np.ndarray"]) -> None:
        views = self.named_views()
        missing = set(views) - set(tensors)
        if missing:
            raise KeyError(f"missing parameters: {sorted(missing)[:5]} ...")
        for k, v in views.items():
            src = torch.as_tensor(np.asarray(tensors[k].cpu() if torch.is_tensor(tensors[k]) else tensors[k]),
                                  dtype=torch.float32)
            if tuple(src.shape) != tuple(v.shape):
                raise ValueError(f"{k}: shape {tuple(src.shape)} != {tuple(v.shape)}")
            v.copy_(src.to(self.device))
        self.refresh_weights()

    def init_params(self, seed: int = 0) -> None:
        """lecun-normal kernels / zero biases / unit LN scales (flax defaults used by
        train_ncsn.py:193-199 create_model); host NumPy so the init is device independent."""
        rng = np.random.default_rng(seed)
        flat = np.zeros(self.n_params, dtype=np.float32)
        for name, off, shape in self.tensor_table:
            n = int(np.prod(shape))
            if name.endswith(".kernel"):
                std = math.sqrt(1.0 / shape[0]) / 0.87962566103423978
                w = rng.standard_normal(size=shape)
                bad = np.abs(w) > 2.0
                while bad.any():
                    w[bad] = rng.standard_normal(size=int(bad.sum()))
                    bad = np.abs(w) > 2.0
                flat[off:off + n] = (w * std).astype(np.float32).ravel()
            elif name.endswith(".scale"):
                flat[off:off + n] = 1.0
        self.params.copy_(torch.from_numpy(flat).to(self.device))
        self.refresh_weights()

    def refresh_weights(self) -> None:
        self._join_pending()
        with torch.cuda.device(self.device):
            _lib.check(self.L.smd_engine_refresh_weights(self.h, _stream()), "refresh_weights")
        self._wstate["ver"] += 1
        self._wseen = self._wstate["ver"]

    def _sync_fp8_weights(self) -> None:
        """fp8 mode: tell this handle when the shared bf16 operand pack was refreshed through another handle."""
        if self.cfg.dtype == "fp8" and self._wseen != self._wstate["ver"]:
            self._join_foreign_pending()
            _lib.check(self.L.smd_engine_set_option(self.h, b"w8_dirty", 1), "set_option w8_dirty")
            self._wseen = self._wstate["ver"]

    def set_option(self, key: str, value: int) -> None:
        _lib.check(self.L.smd_engine_set_option(self.h, key.encode(), int(value)), "set_option")
        if key != "w8_dirty":
            self.generation += 1

    def _join_foreign_pending(self) -> None:
        """A deferred output-stage update of ANOTHER handle on these parameters (opt_overlap bit 0) must be complete before
        this handle reads them; this handle's own deferred update is joined inside the library."""
        p = self._wstate.get("pending")
        if p is not None and p is not self:
            self._join_pending()

    # ------------------------------------------------------------------ binding
    def set_schedule(self, betas: np.ndarray, with_sampler: bool = True) -> None:
        betas = np.asarray(betas, dtype=np.float32)
        if len(betas) != self.cfg.num_timesteps:
            raise ValueError(f"schedule has {len(betas)} steps, model built for {self.cfg.num_timesteps}")
        self.betas = betas
        self.generation += 1
        coef = _sched.reverse_coefficient_table(betas)
        ape = np.concatenate([np.ones(1, np.float32), _sched.alphas_cumprod(betas)])
        dev = self.device
        t = dict(coef=torch.from_numpy(coef).to(dev), sqrt_ap=torch.from_numpy(np.ascontiguousarray(coef[:, 6])).to(dev),
                 ape=torch.from_numpy(ape).to(dev),
                 slot=torch.from_numpy(_sched.collection_slot_table(len(betas))).to(dev))
        t["film"] = (torch.zeros(int(self.L.smd_engine_film_table_floats(self.h)), dtype=torch.float32, device=dev)
                     if with_sampler else None)
        self._sched_tensors = t
        _lib.check(self.L.smd_engine_bind_schedule(self.h, _ptr(t["coef"]), _ptr(t["sqrt_ap"]), _ptr(t["ape"]),
                                                   _ptr(t["film"])), "bind_schedule")

    def set_noise_levels(self, levels: np.ndarray) -> bool:
        """FiLM tables for an arbitrary list of batch-uniform noise levels (the sigma schedule of the Langevin samplers):
        row r of the tables = level r.  Returns False when the list does not fit the engine's table (num_timesteps rows).
        Replaces the bound DDPM schedule (``betas`` becomes None, so the DDPM sampler rebinds its own)."""
        levels = np.asarray(levels, dtype=np.float32).reshape(-1)
        T = self.cfg.num_timesteps
        if len(levels) > T:
            return False
        lv = np.full((T,), levels[-1], dtype=np.float32)
        lv[:len(levels)] = levels
        dev = self.device
        t = dict(coef=torch.zeros(T, 8, dtype=torch.float32, device=dev), sqrt_ap=torch.from_numpy(lv).to(dev),
                 ape=torch.ones(T + 1, dtype=torch.float32, device=dev),
                 slot=torch.full((T,), -1, dtype=torch.int32, device=dev),
                 film=torch.zeros(int(self.L.smd_engine_film_table_floats(self.h)), dtype=torch.float32, device=dev))
        self._sched_tensors, self.betas = t, None
        self.generation += 1
        _lib.check(self.L.smd_engine_bind_schedule(self.h, _ptr(t["coef"]), _ptr(t["sqrt_ap"]), _ptr(t["ape"]), _ptr(t["film"])),
                   "bind_schedule")
        self.prepare_sampler()
        return True

    def bind(self, batch: int, training: bool) -> None:
        if self.workspace is not None and self.batch == batch and self.training == training:
            return
        with torch.cuda.device(self.device):
            nbytes = int(self.L.smd_engine_workspace_bytes(self.h, batch, int(training)))
            self.workspace = None
            self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(self.L.smd_engine_bind_workspace(self.h, _ptr(self.workspace), nbytes, batch, int(training),
                                                        _stream()), "bind_workspace")
        self.batch, self.training = batch, training
        self.generation += 1

    def enable_training(self, ema: bool) -> None:
        if self.grads is not None:
            return
        z = lambda: torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.grads, self.m, self.v = z(), z(), z()
        self.ema = self.params.clone() if ema else None
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.metrics = torch.zeros(4, dtype=torch.float32, device=self.device)
        _lib.check(self.L.smd_engine_bind_train(self.h, _ptr(self.grads), _ptr(self.m), _ptr(self.v), _ptr(self.ema),
                                                _ptr(self.step_counter), _ptr(self.metrics)), "bind_train")
        # weight-gradient GEMMs overlap the dgrad/LayerNorm chain on the engine's low-priority side stream
        with torch.cuda.device(self.device):
            self.set_option("side_wgrad", int(os.environ.get("SMD_SIDE_WGRAD", "1")))

    # ------------------------------------------------------------------ model(x, cond)
    def forward(self, x: torch.Tensor, noise_level: torch.Tensor) -> torch.Tensor:
        """nn.Model.__call__ of the reference (models/ncsn.py:141-148): x (B,*shape) fp32,
        cond (B,1[,1]) fp32 noise level sqrt(alpha_bar) -> eps_hat (B,*shape) fp32."""
        x = x.to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        if tuple(x.shape[1:]) != self.cfg.sample_shape:
            raise ValueError(f"input shape {tuple(x.shape)} != (B, {self.cfg.sample_shape})")
        s = noise_level.to(self.device, torch.float32).reshape(-1).contiguous()
        if s.numel() != B:
            raise ValueError(f"noise level has {s.numel()} entries for batch {B}")
        self.bind(B, training=False)
        self._sync_fp8_weights()
        self._join_pending()
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _lib.check(self.L.smd_engine_forward(self.h, _ptr(x), _ptr(s), _ptr(out), _stream()), "forward")
        return out

    # ------------------------------------------------------------------ training
    def loss_backward(self, x0: torch.Tensor, labels: Optional[torch.Tensor] = None,
                      eps: Optional[torch.Tensor] = None, seed: int = 0, sample_offset: int = 0,
                      global_batch: Optional[int] = None, stage: int = 0, *, used_alphas: Optional[torch.Tensor] = None,
                      continuous_noise: bool = True, objective: str = "ddpm") -> None:
        """``continuous_noise=False``: labels in [0, T) (utils/losses.py:272-275) and a real uniform used_alpha for label 0;
        ``used_alphas`` ([B] floats) passes the draws of :283-286 explicitly.  ``objective="dsm"``: the denoising
        score-matching loss (:129-179) on the same network; ``used_alphas`` then carries the per-sample used_sigmas."""
        B = x0.shape[0] if x0 is not None else self.batch
        kind = {"ddpm": 0, "dsm": 1}[objective]
        if kind != self._loss_kind:
            _lib.check(self.L.smd_engine_set_option(self.h, b"loss_kind", kind), "set_option loss_kind")
            self._loss_kind = kind
        lm = 1 if continuous_noise else 0
        if lm != self._label_min:
            _lib.check(self.L.smd_engine_set_option(self.h, b"label_min", lm), "set_option label_min")
            self._label_min = lm
        _lib.check(self.L.smd_engine_set_used_alphas(self.h, _ptr(used_alphas)), "set_used_alphas")
        gb = B if global_batch is None else global_batch
        inv = 1.0 / (gb * float(np.prod(self.cfg.sample_shape)))
        self._join_foreign_pending()
        self._sync_fp8_weights()
        with torch.cuda.device(self.device):
            _lib.check(self.L.smd_engine_loss_backward(self.h, _ptr(x0), _ptr(labels), _ptr(eps), seed & 0xFFFFFFFF,
                                                       (seed >> 32) & 0xFFFFFFFF, sample_offset, inv, stage,
                                                       _stream()), "loss_backward")

    def forward_train(self, x: torch.Tensor, noise_level: torch.Tensor) -> torch.Tensor:
        """model(x, cond) through the TRAINING workspace (activations saved): the forward half of value_and_grad over an
        arbitrary objective (train_ncsn.py:279-283; include/smd_hip.h smd_engine_forward_train)."""
        if self.grads is None:
            raise RuntimeError("forward_train: a training handle (Model.train_engine / create_optimizer) is needed")
        x = x.to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        if tuple(x.shape[1:]) != self.cfg.sample_shape:
            raise ValueError(f"input shape {tuple(x.shape)} != (B, {self.cfg.sample_shape})")
        s = noise_level.to(self.device, torch.float32).reshape(-1).contiguous()
        if s.numel() != B:
            raise ValueError(f"noise level has {s.numel()} entries for batch {B}")
        self.bind(B, training=True)
        self._sync_fp8_weights()
        self._join_pending()
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _lib.check(self.L.smd_engine_forward_train(self.h, _ptr(x), _ptr(s), _ptr(out), _stream()), "forward_train")
        self._fwd_generation = getattr(self, "_fwd_generation", 0) + 1
        return out

    def backward_from(self, dpred: torch.Tensor, stage: int = 0) -> None:
        """Parameter gradients of the last forward_train from d objective / d eps_hat, written to ``self.grads``."""
        if dpred is not None:
            dpred = dpred.to(self.device, torch.float32).contiguous()
            if tuple(dpred.shape) != (self.batch, *self.cfg.sample_shape):
                raise ValueError(f"dpred shape {tuple(dpred.shape)} != {(self.batch, *self.cfg.sample_shape)}")
        with torch.cuda.device(self.device):
            _lib.check(self.L.smd_engine_backward_from(self.h, _ptr(dpred), int(stage), _stream()), "backward_from")

    def optimizer_step(self, lr0: float, lr_gamma: float = 0.98, lr_interval: int = 10000, grad_clip: float = 1.0,
                       mu: float = 0.999, grad_scale: float = 1.0, beta1: float = 0.9, beta2: float = 0.999,
                       eps: float = 1e-8) -> None:
        h = _lib.TrainHyper(lr0, lr_gamma, lr_interval, beta1, beta2, eps, grad_clip, mu, grad_scale)
        p = self._wstate.get("pending")
        if p is not None and p is not self:
            self._join_pending()
        with torch.cuda.device(self.device):
            _lib.check(self.L.smd_engine_optimizer_step(self.h, C.byref(h), _stream()), "optimizer_step")
        if self._opt_overlap & 1:
            self._wstate["pending"] = self      # output-stage slice still in flight on this handle's side stream
        self._wstate["ver"] += 1            # the step re-casts the operand pack (this handle re-quantises itself)
        self._wseen = self._wstate["ver"]

    # ------------------------------------------------------------------ data-parallel gradient buckets (stem slice)
    def grad_buckets(self) -> List[Tuple[int, int]]:
        """(offset, length) of the stem-slice gradient buckets in backward order (include/smd_hip.h)."""
        out = []
        for b in range(int(self.L.smd_engine_num_grad_buckets(self.h))):
            off, ln = C.c_int64(), C.c_int64()
            _lib.check(self.L.smd_engine_grad_bucket(self.h, b, C.byref(off), C.byref(ln)), "grad_bucket")
            out.append((off.value, ln.value))
        return out

    def wait_grad_bucket(self, bucket: int, stream: "torch.cuda.Stream") -> None:
        """``stream`` waits for the event the stem backward recorded behind bucket ``bucket`` (option dp_layer_events)."""
        _lib.check(self.L.smd_engine_wait_grad_bucket(self.h, bucket, stream.cuda_stream), "wait_grad_bucket")

    def _borrow(self, ptr: int, shape) -> torch.Tensor:
        """View of an engine-internal fp32 buffer inside our workspace tensor."""
        base = self.workspace.data_ptr()
        off = ptr - base
        n = int(np.prod(shape))
        assert 0 <= off and off + 4 * n <= self.workspace.numel()
        return self.workspace[off:off + 4 * n].view(torch.float32).view(*shape)

    def debug_tensor(self, name: str, index: int = 0) -> torch.Tensor:
        """A saved activation of the last training-mode forward pass (include/smd_hip.h smd_engine_debug_tensor): a view into
        the workspace, fp32 or bf16."""
        ptr, r, c, dt = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int32()
        _lib.check(self.L.smd_engine_debug_tensor(self.h, name.encode(), int(index), C.byref(ptr), C.byref(r), C.byref(c), C.byref(dt)),
                   "debug_tensor")
        size = 2 if dt.value == 1 else 4
        off = ptr.value - self.workspace.data_ptr()
        n = r.value * c.value
        assert 0 <= off and off + size * n <= self.workspace.numel()
        raw = self.workspace[off:off + size * n]
        return raw.view(torch.bfloat16 if dt.value == 1 else torch.float32).view(r.value, c.value)

    def loss_per_sample(self) -> torch.Tensor:
        return self._borrow(self.L.smd_engine_loss_per_sample(self.h), (self.batch,))

    def last_pred(self) -> torch.Tensor:
        return self._borrow(self.L.smd_engine_pred(self.h), (self.batch, *self.cfg.sample_shape))

    # ------------------------------------------------------------------ sampling
    def prepare_sampler(self) -> None:
        if self._sched_tensors is None or self._sched_tensors["film"] is None:
            raise RuntimeError("set_schedule(betas, with_sampler=True) first")
        self._join_pending()
        with torch.cuda.device(self.device):
            _lib.check(self.L.smd_engine_prepare_sampler(self.h, _stream()), "prepare_sampler")

    def init_state(self, x: torch.Tensor, seed: int, sample_offset: int = 0) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self.L.smd_engine_init_state(self.h, _ptr(x), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF,
                                                    sample_offset, _stream()), "init_state")

    def load_state(self, x: torch.Tensor) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self.L.smd_engine_load_state(self.h, _ptr(x), _stream()), "load_state")

    def sample_step(self, io: "_lib.SampleIO", part: int = 0) -> None:
        """One reverse iteration; ``part`` 1 / 2: its stem / its output stage + reverse update only (include/smd_hip.h
        smd_engine_sample_step_part: the pipelined two-chain walk)."""
        self._sync_fp8_weights()
        self._join_pending()
        with torch.cuda.device(self.device):
            if part == 0:
                _lib.check(self.L.smd_engine_sample_step(self.h, C.byref(io), _stream()), "sample_step")
            else:
                _lib.check(self.L.smd_engine_sample_step_part(self.h, C.byref(io), int(part), _stream()), "sample_step_part")

    @property
    def slot_table(self) -> torch.Tensor:
        return self._sched_tensors["slot"]
