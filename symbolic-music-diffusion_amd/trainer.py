"""Optimisation step and data-parallel training for the DDPM path.

Mirrors train_ncsn.py:187-190 (create_optimizer), :206-257 (eval_step / evaluate), :260-288
(train_step) on top of the HIP engine, and adds what the reference does not have:
data-parallel training, one process per GPU, gradients summed with RCCL all-reduce over xGMI
(torch.distributed backend "nccl") in two buckets overlapped with the stem backward.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from .engine import Engine
from .jax_random import ThreefryKey, diffusion_loss_draws, diffusion_loss_used_alphas
from .ncsn import (Model, PRNGKey, _dsm_draws, _ensure_any_schedule, _ensure_schedule, denoising_score_matching_loss,
                   diffusion_loss)


@dataclass
class Hyper:
    """Flags that shape the update (train_ncsn.py:53,61-63,94-96)."""
    learning_rate: float = 3e-4
    grad_clip: float = 1.0
    lr_gamma: float = 0.98
    lr_schedule_interval: int = 10000
    ema: bool = True
    mu: float = 0.999


class Optimizer:
    """flax.optim.Optimizer stand-in (train_ncsn.py:187-190): ``target`` is the model, the Adam
    moments / step counter / EMA live in the training engine's flat buffers."""

    def __init__(self, model: Model, learning_rate: float, ema: bool = False):
        self.target = model
        self.learning_rate = learning_rate
        self.engine: Engine = model.train_engine(ema)
        if ema and self.engine.ema is None:
            self.engine.ema = self.engine.params.clone()      # engine was created without EMA: rebind
            from . import lib as _lib
            e = self.engine
            _lib.check(e.L.smd_engine_bind_train(e.h, e.grads.data_ptr(), e.m.data_ptr(), e.v.data_ptr(),
                                                 e.ema.data_ptr(), e.step_counter.data_ptr(), e.metrics.data_ptr()))

    @property
    def step(self) -> int:
        return int(self.engine.step_counter.item())

    def state_dict(self) -> Dict[str, torch.Tensor]:
        e = self.engine
        return {"params": e.params, "adam_m": e.m, "adam_v": e.v, "step": e.step_counter,
                **({"ema": e.ema} if e.ema is not None else {})}


def create_optimizer(model: Model, learning_rate: float, ema: bool = False) -> Optimizer:
    """train_ncsn.py:187-190 (optim.Adam(learning_rate).create(model))."""
    return Optimizer(model, learning_rate, ema)


class LazyMetrics(dict):
    """Metrics stay on the device until somebody formats them (no per-step host sync)."""

    def resolve(self) -> Dict[str, float]:
        return {k: (float(v) if torch.is_tensor(v) else v) for k, v in self.items()}


def train_step(objective, batch, optimizer: Optimizer, sigmas, rng: PRNGKey, learning_rate: float, *,
               grad_clip: float = 1.0, mu: float = 0.999, labels=None, eps=None, comm: "Optional[GradComm]" = None,
               lr_gamma: float = 1.0, lr_interval: int = 1, sample_offset: int = 0,
               global_batch: Optional[int] = None, continuous_noise: bool = True, used_alphas=None):
    """train_ncsn.py:260-288: value_and_grad(mean diffusion_loss) -> clip_grads -> Adam
    (+ EMA, fused).  ``learning_rate`` is the step's LR as in the reference; pass ``lr_gamma`` /
    ``lr_interval`` instead to let the kernel evaluate the stepped schedule from its own step
    counter (learning_rate is then lr0).  ``continuous_noise`` is FLAGS.continuous_noise (:278): False draws the labels
    in [0, T) and gives label 0 its real uniform noise level (utils/losses.py:272-286).  Returns (optimizer, metrics{'loss','grad','lr'}).

    ``objective``: ``diffusion_loss`` / ``denoising_score_matching_loss`` of this package run fused (q-sample, forward, loss and
    backward as one engine call).  ANY other callable ``objective(batch, model, sigmas, rng, continuous_noise, 'mean')`` written in
    torch is differentiated as the reference differentiates it (jax.value_and_grad, :279-283): ``model`` is then a
    ``DifferentiableModel`` whose ``model(x, cond)`` runs the engine's forward and whose autograd backward is the engine's backward
    pass from d objective / d eps_hat (one model call per objective; second-order objectives like 'ssm' are not supported)."""
    eng = optimizer.engine
    batch = torch.as_tensor(batch).to(eng.device, torch.float32).contiguous()
    if objective is not diffusion_loss and objective is not denoising_score_matching_loss:
        return _train_step_generic(objective, batch, optimizer, sigmas, rng, learning_rate, grad_clip=grad_clip, mu=mu, comm=comm,
                                   lr_gamma=lr_gamma, lr_interval=lr_interval, continuous_noise=continuous_noise)
    dsm = objective is denoising_score_matching_loss
    if dsm:
        _ensure_any_schedule(eng)
    else:
        _ensure_schedule(eng, sigmas, with_sampler=False)
    eng.bind(batch.shape[0], training=True)
    lab = None if labels is None else torch.as_tensor(labels).to(eng.device, torch.int32).contiguous()
    e = None if eps is None else torch.as_tensor(eps).to(eng.device, torch.float32).contiguous()
    world = 1 if comm is None else comm.world_size
    gb = batch.shape[0] * world if global_batch is None else global_batch
    ua = None if used_alphas is None else torch.as_tensor(used_alphas).to(eng.device, torch.float32).contiguous()
    if dsm:
        # utils/losses.py:149-164: the noise level travels per sample (used_alphas carries the used_sigmas)
        sig = torch.as_tensor(np.asarray(sigmas, dtype=np.float32)).to(eng.device)
        if ua is None and lab is not None:
            ua = (sig[(lab.long() - 1) % len(sig)] if continuous_noise else sig[lab.long()]).contiguous()
        elif ua is None:
            _lab, ua, e2 = _dsm_draws(rng, tuple(batch.shape), sig, bool(continuous_noise), sample_offset, gb)
            e = e if e is not None else e2
        lab = None
    elif isinstance(rng, ThreefryKey) and lab is None and e is None:
        # the reference's own draws (utils/losses.py:271-294) for this rank's rows of the global batch
        lab, e = diffusion_loss_draws(rng, tuple(batch.shape), len(sigmas), eng.device, sample_offset=sample_offset,
                                      global_batch=gb, continuous_noise=continuous_noise)
        if not continuous_noise and ua is None:
            ua = diffusion_loss_used_alphas(rng, lab, eng._sched_tensors["ape"], sample_offset=sample_offset,
                                            global_batch=gb)
    kw = dict(seed=rng.seed, sample_offset=sample_offset, global_batch=gb, used_alphas=ua, continuous_noise=continuous_noise,
              objective="dsm" if dsm else "ddpm")
    # optimiser placement (engine option "opt_overlap"): 0 = the whole sweep on this stream.  Deferring the
    # output-stage slice to the side stream (bit 0) measured +1 % at best on a process's FIRST training engine and HALVED the
    # step rate of every second later engine of the same process (profiles/r4q_extras_opt_overlap.txt: the per-step
    # main-waits-on-side / side-waits-on-main pattern on whatever hardware queue the new side stream is mapped to), so it is
    # opt-in: SMD_OPT_OVERLAP=3 (1 with a communicator).
    eng.set_opt_overlap(0)
    if comm is None:
        eng.loss_backward(batch, lab, e, stage=0, **kw)
    else:
        layered = comm.layer_buckets and batch.is_cuda and eng.head_offset > 0
        if layered != getattr(eng, "_dp_layer_events", False):
            eng.set_option("dp_layer_events", int(layered))
            eng._dp_layer_events = layered
        eng.loss_backward(batch, lab, e, stage=1, **kw)
        comm.reduce_async(eng.grads[eng.head_offset:])          # output-stage gradients are final
        eng.loss_backward(None, None, None, stage=2, **kw)
        if layered:
            # the stem slice in backward order, one collective per encoder layer: layer l's starts behind the event the engine
            # recorded when ITS gradients became final, not behind the whole stem backward; the last bucket (in_proj + layer 0)
            # waits for the stream as before
            g = eng.grads
            buckets = eng.grad_buckets()
            for b, (off, ln) in enumerate(buckets):
                early = b + 1 < len(buckets)
                comm.reduce_async(g[off:off + ln], after=(lambda s, b=b: eng.wait_grad_bucket(b, s)) if early else None)
        else:
            comm.reduce_async(eng.grads[:eng.head_offset])
        comm.wait()
    # gradients were scaled by 1/(global_batch*S*C) at the loss, so the all-reduce SUM is the global mean
    eng.optimizer_step(learning_rate, lr_gamma, lr_interval, grad_clip, mu, 1.0)
    loss = eng.loss_per_sample().mean()
    metrics = LazyMetrics(loss=loss, grad=eng.metrics[1], lr=eng.metrics[2])
    return optimizer, metrics


def _train_step_generic(objective, batch, optimizer: Optimizer, sigmas, rng, learning_rate, *, grad_clip, mu, comm, lr_gamma,
                        lr_interval, continuous_noise):
    """train_ncsn.py:279-287 for an arbitrary objective: torch autograd over ``objective`` with the engine's forward / backward
    underneath (ops.py smd_amd::eps_forward_train), then the same fused clip + Adam (+ EMA) sweep as the fused objectives."""
    eng = optimizer.engine
    dm = getattr(optimizer, "_differentiable", None)
    if dm is None:
        dm = optimizer._differentiable = optimizer.target.differentiable(ema=eng.ema is not None)
    eng.set_opt_overlap(0)
    dm.params.grad = None
    with torch.enable_grad():
        loss = objective(batch, dm, sigmas, rng, continuous_noise, "mean")
    if not (torch.is_tensor(loss) and loss.requires_grad and loss.numel() == 1):
        raise ValueError("train_step: objective(batch, model, sigmas, rng, continuous_noise, 'mean') must return a scalar tensor "
                         "computed from model(x, cond)")
    loss.backward()
    if dm.params.grad is None:
        raise ValueError("train_step: the objective did not reach the model's parameters")
    # the engine's gradient buffer holds the last backward pass; autograd's accumulated leaf gradient is the authoritative one
    # (it may BE the engine's buffer: eps_backward returns an alias that autograd can adopt)
    if dm.params.grad.data_ptr() != eng.grads.data_ptr():
        eng.grads.copy_(dm.params.grad)
    dm.params.grad = None
    world = 1
    if comm is not None:
        world = comm.world_size
        comm.reduce_async(eng.grads)
        comm.wait()
    eng.optimizer_step(learning_rate, lr_gamma, lr_interval, grad_clip, mu, 1.0 / world)     # the objective's mean is per rank
    return optimizer, LazyMetrics(loss=loss.detach(), grad=eng.metrics[1], lr=eng.metrics[2])


def eval_step(objective, batch, model: Model, sigmas, rng: PRNGKey, continuous_noise: bool = True):
    """train_ncsn.py:206-221: summed loss of one batch."""
    return objective(batch, model, sigmas, rng, continuous_noise, "sum")


def evaluate(dataset, model: Model, sigmas, rng: PRNGKey, continuous_noise: bool = True, objective=diffusion_loss):
    """train_ncsn.py:224-257: mean per-example loss over the dataset (iterable of batches)."""
    from .ncsn import split
    count, total = 0, 0.0
    for inputs in dataset:
        count += inputs.shape[0]
        rng, eval_rng = split(rng)
        total += float(eval_step(objective, inputs, model, sigmas, eval_rng, continuous_noise))
    return {"loss": total / max(count, 1)}


class GradComm:
    """Gradient reduction on a side stream (RCCL when the tensors are on GPUs, gloo in the CPU tests); SUM (the loss
    already carries 1/global_count).  Every rank ends with bitwise the same reduced gradient in every mode.

    ``buckets``: every reduce_async(flat) is cut into this many contiguous chunks, each its own collective, so the ring
    starts moving the first chunk while later ones are still queued (xGMI is point-to-point: a ring all-reduce is
    per-link bound, ~153 GB/s, and a 100 MB fp32 gradient is ~1.3 ms of wire time at 8 ranks however it is cut; smaller
    chunks only shorten the pipeline fill).  ``payload="bf16"``: the chunk is rounded to bf16 into a persistent staging
    buffer, reduced in bf16 and widened back in place -- half the bytes on the links for ~2^-9 relative rounding per
    addend; off by default because the reference reduces fp32 (jax.lax.pmean, train_ncsn.py:282).
    ``algorithm="rs_ag"``: reduce_scatter + all_gather instead of all_reduce -- on the 8-GPU xGMI mesh every GPU has a direct
    link to every other, so each of the two phases is ONE hop of n/8 per peer over 7 links in parallel
    where a ring makes 2 x 7 dependent hops; which one RCCL's own all_reduce picks is its tuning, this makes it a choice.
    ``layer_buckets``: the stem slice is reduced per encoder layer in backward order, each collective gated by the engine's
    per-layer gradient event (train_step; engine option dp_layer_events) instead of by the end of the stem backward.  OFF by
    default: six extra 2.4 MB collectives per step are latency-bound on a real communicator (a launch + >= 20 us each on a
    stream that competes with the stem backward for CUs) and have not been measured to win on RCCL.
    ``measure_exposed``: HIP events around the point where the compute stream waits for the communication stream;
    ``exposed_comm_us()`` returns the mean stall per step (what the overlap did NOT hide).
    ``emulate_load`` (dry runs on one GPU over gloo): every reduction first runs a copy + add of the chunk's size on the
    communication stream -- kernels of a collective's shape sharing the CUs with the backward pass, which gloo's host-side
    reduction would not provide."""

    def __init__(self, group=None, buckets: int = 1, payload: str = "fp32", algorithm: str = "all_reduce",
                 layer_buckets: bool = False, measure_exposed: bool = False, emulate_load: bool = False):
        import torch.distributed as dist
        if payload not in ("fp32", "bf16"):
            raise ValueError(f"payload must be 'fp32' or 'bf16', got {payload!r}")
        if algorithm not in ("all_reduce", "rs_ag"):
            raise ValueError(f"algorithm must be 'all_reduce' or 'rs_ag', got {algorithm!r}")
        self.dist = dist
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.buckets = max(1, int(buckets))
        self.payload = payload
        self.algorithm = algorithm
        self.layer_buckets = bool(layer_buckets)
        self.measure_exposed = bool(measure_exposed)
        self.emulate_load = bool(emulate_load)
        self.collectives = 0                # issued since construction (the bench line reports collectives per step)
        self._exposed = []                  # (event before the stall, event after it) per wait()
        self._load = {}                     # numel -> two scratch buffers of the emulated collective
        self._works = []
        self._stream = None
        self._stage = {}                    # (data_ptr, numel) -> bf16 staging buffer
        self._shard = {}                    # (data_ptr, numel, dtype) -> reduce_scatter output shard
        self._pending = []                  # (fp32 chunk, bf16 buffer) to widen after the collective

    def _chunks(self, flat: torch.Tensor):
        n = flat.numel()
        step = -(-n // self.buckets)
        step = -(-step // 1024) * 1024      # 4 KiB-aligned chunk starts
        return [flat[i:min(i + step, n)] for i in range(0, n, step)]

    def _collective(self, buf: torch.Tensor) -> None:
        """SUM over the ranks of ``buf`` in place, asynchronously (the works are collected in self._works)."""
        d, W = self.dist, self.world_size
        self.collectives += 1
        if self.emulate_load and buf.is_cuda:
            sc = self._load.get(buf.numel())
            if sc is None:
                sc = self._load[buf.numel()] = (torch.empty_like(buf), torch.ones_like(buf))
            sc[0].copy_(buf)                # the receive copy ...
            sc[0].add_(sc[1])               # ... and the reduction of a ring step, on the communication stream
        if self.algorithm == "all_reduce" or buf.numel() < 2 * W:
            self._works.append(d.all_reduce(buf, op=d.ReduceOp.SUM, group=self.group, async_op=True))
            return
        main = buf.numel() - buf.numel() % W
        key = (buf.data_ptr(), buf.numel(), buf.dtype)
        shard = self._shard.get(key)
        if shard is None:
            shard = self._shard[key] = torch.empty(main // W, dtype=buf.dtype, device=buf.device)
        # phase 1: rank r ends with the sum of slice r; phase 2: every rank collects the W reduced slices (identical bytes
        # everywhere by construction).  The all_gather reads what the reduce_scatter wrote: wait() orders the two (on
        # GPUs it orders the communication stream, it does not block the host).
        d.reduce_scatter_tensor(shard, buf[:main], op=d.ReduceOp.SUM, group=self.group, async_op=True).wait()
        self._works.append(d.all_gather_into_tensor(buf[:main], shard, group=self.group, async_op=True))
        if main < buf.numel():
            self._works.append(d.all_reduce(buf[main:], op=d.ReduceOp.SUM, group=self.group, async_op=True))

    def _reduce(self, chunk: torch.Tensor) -> None:
        if self.payload == "bf16":
            key = (chunk.data_ptr(), chunk.numel())
            buf = self._stage.get(key)
            if buf is None:
                buf = self._stage[key] = torch.empty(chunk.numel(), dtype=torch.bfloat16, device=chunk.device)
            buf.copy_(chunk)
            self._collective(buf)
            self._pending.append((chunk, buf))
        else:
            self._collective(chunk)

    def reduce_async(self, flat: torch.Tensor, after=None) -> None:
        """Start the reduction of ``flat`` on the communication stream.  ``after(stream)``: a callable that makes that stream
        wait for exactly what ``flat`` depends on (an engine gradient-bucket event); default: everything enqueued on the
        current stream so far."""
        if self.world_size == 1:
            return
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            if after is not None:
                after(self._stream)
            else:
                self._stream.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._stream):
                for c in self._chunks(flat):
                    self._reduce(c)
        else:
            for c in self._chunks(flat):
                self._reduce(c)

    def wait(self) -> None:
        import contextlib
        cuda = self._stream is not None
        with (torch.cuda.stream(self._stream) if cuda else contextlib.nullcontext()):
            for w in self._works:
                w.wait()                    # on GPUs: orders the comm stream after the collective, no host block
            for chunk, buf in self._pending:
                chunk.copy_(buf)            # widen bf16 -> fp32 in place
        self._works.clear()
        self._pending.clear()
        if cuda:
            cur = torch.cuda.current_stream()
            if self.measure_exposed:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(cur)
            cur.wait_stream(self._stream)
            if self.measure_exposed:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(cur)
                self._exposed.append((e0, e1))

    def exposed_comm_us(self) -> Optional[float]:
        """Mean time per step the compute stream stood waiting for the communication stream since the last call (needs
        ``measure_exposed``); blocks until the recorded events have completed."""
        if not self._exposed:
            return None
        self._exposed[-1][1].synchronize()
        v = sum(a.elapsed_time(b) for a, b in self._exposed) * 1e3 / len(self._exposed)
        self._exposed.clear()
        return v

    def describe(self) -> Dict[str, object]:
        """What ran: backend, library version, world size and the shape of the exchange (for the bench line)."""
        d = self.dist
        backend = d.get_backend(self.group)
        ver = None
        if backend == "nccl":
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                ver = "unknown"
        return {"backend": backend, "library": "RCCL" if backend == "nccl" else backend, "version": ver,
                "world_size": self.world_size, "algorithm": self.algorithm, "buckets_per_stage": self.buckets,
                "payload": self.payload, "layer_buckets": self.layer_buckets, "emulated_load": self.emulate_load}

    def broadcast_params(self, flat: torch.Tensor, src: int = 0) -> None:
        if self.world_size > 1:
            self.dist.broadcast(flat, src=src, group=self.group)


def device_identity(index: Optional[int] = None) -> str:
    """A string that names the PHYSICAL device behind ``cuda:index`` of this process: its UUID and its PCI domain:bus:device as far as
    the runtime reports them, else the ordinal -- what the bench line lists per rank so that N ranks can be seen to sit on N distinct
    GPUs (a CPU-only process reports ``cpu:<hostname>:<pid>``)."""
    if not torch.cuda.is_available():
        import socket
        return f"cpu:{socket.gethostname()}:{os.getpid()}"
    i = torch.cuda.current_device() if index is None else int(index)
    pr = torch.cuda.get_device_properties(i)
    parts = []
    uuid = getattr(pr, "uuid", None)
    if uuid is not None and str(uuid).strip("0-") != "":
        parts.append(f"uuid:{uuid}")
    bus = getattr(pr, "pci_bus_id", None)
    if bus is not None:
        parts.append(f"pci:{getattr(pr, 'pci_domain_id', 0):04x}:{bus:02x}:{getattr(pr, 'pci_device_id', 0):02x}")
    # (both when both exist: two ranks count as sharing a device only if EVERY identifier the runtime gives agrees)
    return "|".join(parts) if parts else f"ordinal:{i}"


def gather_device_identities(dist=None, group=None, mine: Optional[str] = None) -> List[str]:
    """Every rank's ``device_identity()`` in rank order (all_gather_object over the job's process group; one entry without
    torch.distributed).  ``assert_distinct_devices`` is the check on top of it."""
    mine = device_identity() if mine is None else mine
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [mine]
    out: List[Optional[str]] = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, mine, group=group)
    return [str(o) for o in out]


def assert_distinct_devices(ids: List[str], allow_shared: bool = False) -> None:
    """A data-parallel job whose ranks share a GPU measures nothing: refuse (``allow_shared``: the one-GPU test hook)."""
    if len(set(ids)) != len(ids) and not allow_shared:
        raise RuntimeError(f"data-parallel ranks share a device: {ids}")


def shard_bounds(num_items: int, world_size: int, rank: int):
    """Contiguous shard [lo, hi) of ``num_items`` independent units (samples) for ``rank``."""
    base, rem = divmod(num_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
