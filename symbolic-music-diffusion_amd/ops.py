"""PyTorch custom ops (``torch.ops.smd_amd.*``) over the C-ABI -- the "PyTorch-ROCm custom ops" seam of the north star.

Each op is a thin, shape-checked registration of one C entry point of ``include/smd_hip.h``: device tensors in, device
tensors out, launched on torch's current HIP stream; fake-tensor (meta) rules are registered so the ops trace under
``torch.compile`` / ``torch.export`` without running.  There is no CPU implementation: a CPU tensor raises.

  smd_amd::eps_forward(x, noise_level, engine)          model(x, cond), models/ncsn.py:141-179 / 125-135
  smd_amd::eps_forward_train(x, noise_level, params, engine)   the same, DIFFERENTIABLE with respect to the flat parameter
                                                        buffer: forward in the training workspace, backward =
                                                        the engine's own backward pass from d/d eps_hat (autograd formula
                                                        registered with torch.library.register_autograd)
  smd_amd::gemm_bf16_nt(a, bt, bias)                    nn.Dense, bf16 operands, fp32 accumulate -> bf16
  smd_amd::ddpm_reverse_step_(x, eps_hat, coef, t, ...) utils/ebm_utils.py:327-394, in place
  smd_amd::q_sample(x0, alphas_prod_ext, labels, eps)   utils/losses.py:271-296

``engine`` is the integer id of a live ``smd_amd.engine.Engine`` (``register_engine``); custom-op schemas carry tensors
and scalars only.
"""
from __future__ import annotations

import weakref
from typing import Tuple

import torch

from . import lib as _lib

_ENGINES: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()


def register_engine(engine) -> int:
    """Id under which ``engine`` is reachable from ``torch.ops.smd_amd.eps_forward`` (weakly held)."""
    _ENGINES[id(engine)] = engine
    return id(engine)


def _need_gpu(*ts: torch.Tensor) -> None:
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("smd_amd ops run on the GPU only (no CPU fallback): got a tensor on " + str(t.device))


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


@torch.library.custom_op("smd_amd::eps_forward", mutates_args=())
def eps_forward(x: torch.Tensor, noise_level: torch.Tensor, engine: int) -> torch.Tensor:
    _need_gpu(x, noise_level)
    eng = _ENGINES.get(engine)
    if eng is None:
        raise RuntimeError(f"smd_amd::eps_forward: no live engine with id {engine} (ops.register_engine)")
    return eng.forward(x, noise_level)


@eps_forward.register_fake
def _(x, noise_level, engine):
    return torch.empty_like(x, dtype=torch.float32)


# ---- model(x, cond), differentiable w.r.t. the parameters: jax.value_and_grad(loss_fn)(optimizer.target), train_ncsn.py:279-283
@torch.library.custom_op("smd_amd::eps_forward_train", mutates_args=())
def eps_forward_train(x: torch.Tensor, noise_level: torch.Tensor, params: torch.Tensor, engine: int) -> torch.Tensor:
    """``engine`` is a TRAINING handle; ``params`` must be (an alias of) its flat fp32 parameter buffer -- it is an argument so
    that autograd has an edge to hang the parameter gradient on, the kernels read the engine's own operand pack."""
    _need_gpu(x, noise_level, params)
    eng = _ENGINES.get(engine)
    if eng is None:
        raise RuntimeError(f"smd_amd::eps_forward_train: no live engine with id {engine} (ops.register_engine)")
    if params.data_ptr() != eng.params.data_ptr() or params.numel() != eng.params.numel():
        raise ValueError("smd_amd::eps_forward_train: `params` is not the engine's parameter buffer")
    return eng.forward_train(x, noise_level)


@eps_forward_train.register_fake
def _(x, noise_level, params, engine):
    return torch.empty_like(x, dtype=torch.float32)


@torch.library.custom_op("smd_amd::eps_backward", mutates_args=())
def eps_backward(dpred: torch.Tensor, engine: int, generation: int) -> torch.Tensor:
    """The engine's backward pass from d objective / d eps_hat; returns the flat parameter gradient: a fresh tensor object that
    ALIASES the engine's gradient buffer (no 100 MB copy per step).  The buffer is overwritten by the handle's next backward
    pass -- one outstanding forward / backward pair per handle, as eps_forward_train already requires; autograd either adopts
    the alias as ``params.grad`` (trainer._train_step_generic then skips its copy-back) or clones it while accumulating."""
    _need_gpu(dpred)
    eng = _ENGINES.get(engine)
    if eng is None:
        raise RuntimeError(f"smd_amd::eps_backward: no live engine with id {engine}")
    if getattr(eng, "_fwd_generation", 0) != generation:
        raise RuntimeError("smd_amd::eps_backward: the engine has run another training forward pass since this one (one "
                           "workspace per handle: call backward() before the next model(x, cond), or use a second handle)")
    eng.backward_from(dpred)
    return eng.grads.detach()


@eps_backward.register_fake
def _(dpred, engine, generation):
    eng = _ENGINES.get(engine)
    return dpred.new_empty((eng.n_params if eng is not None else 0,), dtype=torch.float32)


def _eps_train_setup(ctx, inputs, output):
    _x, _s, _params, engine = inputs
    eng = _ENGINES.get(engine)
    ctx.engine = engine
    ctx.generation = getattr(eng, "_fwd_generation", 0)


def _eps_train_backward(ctx, grad_out):
    # d/dx is not produced (the reference never differentiates the objective w.r.t. the data either)
    return None, None, torch.ops.smd_amd.eps_backward(grad_out.contiguous(), ctx.engine, ctx.generation), None


torch.library.register_autograd("smd_amd::eps_forward_train", _eps_train_backward, setup_context=_eps_train_setup)


@torch.library.custom_op("smd_amd::gemm_bf16_nt", mutates_args=())
def gemm_bf16_nt(a: torch.Tensor, bt: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """C[M,N] = A[M,K] Bt[N,K]^T + bias (bf16 operands with K % 64 == 0, fp32 bias, bf16 result)."""
    _need_gpu(a, bt, bias)
    if a.dtype != torch.bfloat16 or bt.dtype != torch.bfloat16 or bias.dtype != torch.float32:
        raise TypeError("gemm_bf16_nt: a, bt bf16 and bias fp32")
    if a.dim() != 2 or bt.dim() != 2 or a.shape[1] != bt.shape[1] or bias.numel() != bt.shape[0]:
        raise ValueError(f"gemm_bf16_nt: shapes {tuple(a.shape)} x {tuple(bt.shape)}^T + {tuple(bias.shape)}")
    a, bt, bias = a.contiguous(), bt.contiguous(), bias.contiguous()
    M, K = a.shape
    N = bt.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.get_lib().smd_gemm_bf16_nt(a.data_ptr(), K, bt.data_ptr(), K, M, N, K, bias.data_ptr(), 0, None, 0,
                                                   None, 0, out.data_ptr(), N, _stream()), "gemm_bf16_nt")
    return out


@gemm_bf16_nt.register_fake
def _(a, bt, bias):
    return a.new_empty((a.shape[0], bt.shape[0]), dtype=torch.bfloat16)


@torch.library.custom_op("smd_amd::ddpm_reverse_step_", mutates_args=("x",))
def ddpm_reverse_step_(x: torch.Tensor, eps_hat: torch.Tensor, coef: torch.Tensor, t: torch.Tensor, z: torch.Tensor) -> None:
    """x <- one sample_with_beta iteration at the device timestep ``t`` (int32[1]) with the explicit draw ``z``."""
    _need_gpu(x, eps_hat, coef, t, z)
    if x.dtype != torch.float32 or not x.is_contiguous() or x.shape != eps_hat.shape or x.shape != z.shape:
        raise ValueError("ddpm_reverse_step_: x, eps_hat, z are contiguous fp32 tensors of one shape")
    B = x.shape[0]
    S, Cn = (1, x.shape[1]) if x.dim() == 2 else (x.shape[1], x.shape[2])
    with torch.cuda.device(x.device):
        _lib.check(_lib.get_lib().smd_ddpm_reverse_step(x.data_ptr(), eps_hat.contiguous().data_ptr(), B, S, Cn, coef.data_ptr(),
                                                        coef.shape[0], t.data_ptr(), z.contiguous().data_ptr(), 0, 0, 0, None,
                                                        None, None, _stream()), "ddpm_reverse_step")


@torch.library.custom_op("smd_amd::q_sample", mutates_args=())
def q_sample(x0: torch.Tensor, alphas_prod_ext: torch.Tensor, labels: torch.Tensor,
             eps: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(x_t as bf16 [B*S][C], noise level sqrt(alpha) [B]) for explicit labels in [1, T] and eps."""
    _need_gpu(x0, alphas_prod_ext, labels, eps)
    if x0.dtype != torch.float32 or eps.shape != x0.shape or labels.dtype != torch.int32:
        raise ValueError("q_sample: x0, eps fp32 of one shape, labels int32")
    x0, eps = x0.contiguous(), eps.contiguous()
    B = x0.shape[0]
    S, Cn = (1, x0.shape[1]) if x0.dim() == 2 else (x0.shape[1], x0.shape[2])
    xt = torch.empty((B * S, Cn), dtype=torch.bfloat16, device=x0.device)
    eo = torch.empty_like(x0)
    s = torch.empty((B,), dtype=torch.float32, device=x0.device)
    with torch.cuda.device(x0.device):
        _lib.check(_lib.get_lib().smd_q_sample(x0.data_ptr(), B, S, Cn, Cn, alphas_prod_ext.numel() - 1, alphas_prod_ext.data_ptr(),
                                               labels.data_ptr(), 1, None, eps.data_ptr(), 0, 0, None, 0, xt.data_ptr(),
                                               eo.data_ptr(), s.data_ptr(), _stream()), "q_sample")
    return xt, s


@q_sample.register_fake
def _(x0, alphas_prod_ext, labels, eps):
    B = x0.shape[0]
    rows = B if x0.dim() == 2 else B * x0.shape[1]
    return x0.new_empty((rows, x0.shape[-1]), dtype=torch.bfloat16), x0.new_empty((B,), dtype=torch.float32)
