"""TEST INFRASTRUCTURE (like everything under oracle/): the eps-network of ddpm_oracle.py evaluated in float64 WITH THE ENGINE'S
fp8 (--dtype=fp8, BASELINE config 5) ROUNDING POINTS, beside bf16_emulation.py (whose bf16 rounding points stay in force for
everything that is not a DenseResBlock GEMM).  It restates the ENGINE's storage formats, not the reference.

What the engine does in fp8 mode (csrc/engine.hip::run_network / backward_head, csrc/norm.hip, csrc/smd_common.h):

  forward, both Dense layers of every DenseResBlock (models/shared.py:61-75; 77 % of the forward flops)
      A operand  the FiLM-LayerNorm kernel writes swish(scale LN(x) + shift) -- its fp32 value, NOT the bf16 copy -- as OCP e4m3
                 (4 exponent bits, bias 7, 3 mantissa bits, max 448, round-to-nearest-even, saturating) with ONE E8M0 (power of
                 two) scale per token row: e = floor(log2(amax)) - 8, plus one when amax's mantissa exceeds 1.75 (2^8 x 1.75 =
                 448); q = e4m3(clamp(x 2^-e, +-448)); the bf16 copy of the row is kept for the weight gradient only
      B operand  the bf16 operand pack of the kernel, quantised the same way per OUTPUT feature (rows of Wt [N][K])
      product    v_mfma_scale_f32_32x32x64_f8f6f4: fp32 accumulation of q_a q_b, scaled by 2^(e_a + e_b); bias, (bf16
                 residual,) bf16 result as in bf16 mode
  backward, the dgrad of those layers (option fp8_dgrad, default on): dX = dY W^T with dY (already bf16) quantised per token row and
      the DGRAD layout of the pack (W [K][N]) quantised per INPUT feature; the result is stored bf16 as in bf16 mode
  everything else -- the weight gradients (bf16 X^T dY), the encoder, FiLM generators, in/out projections -- is bf16_emulation.py.

tests/test_gpu_fp8.py / tests/test_gpu_full_walk.py use it to attribute the fp8 engine's distance from the exact oracle (VERDICT r5
weak #1b): what is left against THIS model is the kernels' own error (accumulation order, hardware transcendentals)."""
import numpy as np
import torch

import bf16_emulation as E

E4M3_MAX = 448.0


def e4m3_row_exponent(amax: torch.Tensor) -> torch.Tensor:
    """csrc/smd_common.h e4m3_row_exponent, elementwise on a float32 tensor of row maxima: int32 exponents (0 for an all-zero row)."""
    a = amax.detach().to(torch.float32).contiguous()
    bits = a.view(torch.int32)
    e = ((bits >> 23) & 0xFF) - 127 - 8
    e = e + ((bits & 0x7FFFFF) > 0x600000).to(torch.int32)
    e = torch.clamp(e, min=-126)
    return torch.where(a > 0, e, torch.zeros_like(e))


def round_e4m3(x: torch.Tensor) -> torch.Tensor:
    """round-to-nearest-even onto the OCP e4m3 grid (values already clamped to +-448), result in x's dtype"""
    return x.to(torch.float32).to(torch.float8_e4m3fn).to(x.dtype)


def q8_rows(x: torch.Tensor) -> torch.Tensor:
    """pack4_e4m3 + the row scale applied back: the values the scaled MFMA multiplies, per row of the last axis"""
    xd = x.detach()
    e = e4m3_row_exponent(xd.to(torch.float32).abs().amax(dim=-1, keepdim=True))
    s = torch.ldexp(torch.ones_like(xd[..., :1]), e)       # 2^e, exact
    q = round_e4m3(torch.clamp(xd / s, -E4M3_MAX, E4M3_MAX))
    return q * s


class F8Dense(torch.autograd.Function):
    """a W + b of one DenseResBlock layer on e4m3 operands, with the engine's backward: e4m3 dgrad, bf16 wgrad"""

    @staticmethod
    def forward(ctx, a, Wb, bias):
        ctx.save_for_backward(a, Wb)
        aq = q8_rows(a)                       # per token row over the contraction axis
        wq = q8_rows(Wb.t()).t()              # forward pack Wt [N][K]: per output feature over the contraction axis
        return aq @ wq + bias

    @staticmethod
    def backward(ctx, dY):
        a, Wb = ctx.saved_tensors
        K, N = Wb.shape
        dY2 = dY.reshape(-1, N)               # (bf16 already: the hook on the layer's output rounded it)
        dA = (q8_rows(dY2) @ q8_rows(Wb).t()).reshape(a.shape)        # dgrad pack W [K][N]: per input feature over the outputs
        dW = E._round(a.detach().reshape(-1, K)).t() @ dY2            # the weight gradient contracts the bf16 copies
        return dA, dW, dY2.sum(dim=0)


def make_model(p, cfg, backward=False, noise_embedding=None, fp8_dgrad=True):
    """model(x, cond) -> eps_hat as the fp8 engine computes it (float64 arithmetic).  ``fp8_dgrad=False``: engine option fp8_dgrad = 0
    (the dgrad GEMMs stay bf16).  Other arguments: bf16_emulation.make_model."""
    inner = E.make_model(p, cfg, backward=backward, noise_embedding=noise_embedding)

    class _F8FwdOnly(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a, Wb, bias):
            ctx.save_for_backward(a, Wb)
            return q8_rows(a) @ q8_rows(Wb.t()).t() + bias

        @staticmethod
        def backward(ctx, dY):
            a, Wb = ctx.saved_tensors
            K, N = Wb.shape
            dY2 = dY.reshape(-1, N)
            return (dY2 @ Wb.t()).reshape(a.shape), E._round(a.detach().reshape(-1, K)).t() @ dY2, dY2.sum(dim=0)

    fn = F8Dense.apply if fp8_dgrad else _F8FwdOnly.apply

    def model(x, t):
        E._RES_DENSE = fn
        try:
            return inner(x, t)
        finally:
            E._RES_DENSE = None
    return model
