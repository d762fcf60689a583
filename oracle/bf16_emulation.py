"""TEST INFRASTRUCTURE (like everything under oracle/): the eps-network of ddpm_oracle.py evaluated in float64 WITH THE
ENGINE'S bf16 ROUNDING POINTS, so that the HIP path can be checked at network level far below its own rounding noise.

Against the plain fp64 oracle the bf16 engine sits at 6e-3 (rel-L2 on eps_hat): any systematic error below that -- a
missing epilogue term, a wrong FiLM broadcast -- is invisible there and rests on the single-kernel tests.  Here every
value the engine rounds to bf16 is rounded at the same place (and nothing else is), so what is left between the two is fp32
accumulation order, the hardware exp / rcp / rsq approximations and rounding flips next to bf16 ties: ~1e-4.  A term that is
wrong by 1e-3 of the output shows.  This file restates the ENGINE's storage formats (symbolic-music-diffusion_amd/csrc), not
the reference; the layer semantics come from ddpm_oracle.py (which cites the reference lines).

Rounding points of the default inference / training forward path (csrc/engine.hip::run_network, encoder_fused.hip, norm.hip):
  every Dense kernel ............ bf16 operand pack of the fp32 master (biases, LayerNorm scale / bias stay fp32)
  network input x_t ............. bf16 (q_sample / cast_pad_bf16)
  residual stream h [R][128] .... fp32 (in_proj output + positional encoding, attention and MLP outputs)
  LayerNorm outputs a1, a2, af .. bf16 (GEMM A operands)
  q, k, v ....................... bf16 after the bias; q then scaled by 1/sqrt(d) and rounded again (attn_block_fwd)
  softmax probabilities ......... bf16 (B fragments of the p.v product); logits, max, sum, 1/sum in fp32
  attention output o ............ bf16 (A operand of out_proj)
  MLP hidden u = gelu(z) ........ bf16; z = a2 W1 + b1 stays in the fp32 accumulators
  trunk y [R][2048] ............. bf16 (`up` output, every DenseResBlock output: trunk_bf16)
  FiLM generator ................ noise embedding bf16 (its angle (5000 s) f evaluated in fp32), swish(fc1) bf16, fc2 output
                                  bf16, scale | shift fp32
  DenseResBlock ................. swish(scale LN(x) + shift) bf16 twice, fc1 output o1 bf16, fc2 + residual -> trunk bf16
  output stage .................. LN_o(y) bf16, eps_hat fp32

The BACKWARD pass is emulated too (make_model(..., backward=True), then loss.backward()): every rounding above is a
straight-through estimator (the engine's backward kernels differentiate the un-rounded function at the rounded values), and
every GRADIENT the engine stores in bf16 is rounded by a tensor hook at the same place:
  d eps_hat (2 (pred - eps) / N), dY of every Dense (both its dgrad and its wgrad read the bf16 copy), every dgrad output
  (gradients of the GEMM A operands: a1, a2-free, o, af, the two ResBlock activations, ao, the FiLM chain), the trunk gradient
  dy_k (LayerNorm path + residual, summed in fp32, stored bf16: resgrad_bf16), d o1, d scale | shift (cast before the FiLM
  dgrad), attention: dO, dS (probabilities and dP stay fp32), dq, dk, dv; MLP: dz = du gelu'(z) (du stays in the accumulators,
  da2 goes to the LayerNorm backward as fp32 partial tiles).  The residual stream's gradient dh is fp32; the branches read its
  bf16 copy (hooks on the out_proj / fc2 / in_proj outputs).  Weight gradients are fp32 accumulations of bf16 x bf16 products.
"""
import math

import numpy as np
import torch

import ddpm_oracle as O


def _round(x):
    return x.to(torch.float32).to(torch.bfloat16).to(x.dtype)


def rb(x):
    """round-to-nearest-even to bf16 of the float32 value (the engine converts fp32 registers), returned in x's dtype;
    straight-through for autograd (the engine's backward kernels treat a stored bf16 activation as the value itself)"""
    if x.requires_grad:
        return x + (_round(x) - x).detach()
    return _round(x)


_BACKWARD = False


def H(x):
    """the gradient wrt x is stored in bf16 by the engine: round it where autograd hands it on (backward emulation only)"""
    if _BACKWARD and x.requires_grad:
        x.register_hook(_round)
    return x


def f32(x):
    return x.to(torch.float32).to(x.dtype)


class _P:
    """parameter access with the engine's storage formats: kernels bf16(fp32(p)), everything else fp32(p)"""

    def __init__(self, p):
        self.p = p
        self._w = {}

    def w(self, name):
        if name not in self._w:
            self._w[name] = rb(self.p[name + ".kernel"])
        return self._w[name]

    def b(self, name):
        return f32(self.p[name + ".bias"])

    def ln(self, x, name):
        q = {name + ".scale": f32(self.p[name + ".scale"]), name + ".bias": f32(self.p[name + ".bias"])}
        return O.layer_norm(x, q, name)


def _noise_embedding_bf16(noise, channels):
    """csrc/diffusion.hip noise_embed_kernel: f = expf(i * -(ln 1e4 / (half - 1))), angle = (5000 s) f, both in fp32"""
    half = channels // 2
    i = np.arange(half, dtype=np.float32)
    f = np.exp(i * np.float32(-(9.210340371976184 / float(half - 1))), dtype=np.float32)
    s = noise.reshape(-1).to(torch.float32).numpy()
    arg = (np.float32(5000.0) * s)[:, None] * f[None, :]
    arg = torch.from_numpy(arg.astype(np.float64)).to(noise.dtype)
    return rb(torch.cat([torch.sin(arg), torch.cos(arg)], dim=1))


_EMB_OVERRIDE = None      # (B, film_channels) tensor: the embedding the device kernel produced (see make_model)


def _film(P, t, name, film_channels, mlp_dims, sequence):
    e = _noise_embedding_bf16(t, film_channels) if _EMB_OVERRIDE is None else _EMB_OVERRIDE.to(t.dtype)
    zf = H(e @ P.w(name + ".fc1") + P.b(name + ".fc1"))          # d zf1 = (dp W2^T) swish'(zf1) -> bf16
    e = rb(O.swish(zf))
    e = H(rb(e @ P.w(name + ".fc2") + P.b(name + ".fc2")))       # dp -> bf16
    if sequence:
        e = e[:, None, :]
    ss = H(e @ P.w(name + ".ss") + P.b(name + ".ss"))            # d scale | shift: fp32 sums of both LayerNorms, cast to bf16
    return ss[..., :mlp_dims], ss[..., mlp_dims:]


_RES_DENSE = None      # fp8 mode (oracle/e4m3_emulation.py): callable(a_unrounded, W_bf16, bias) -> a W + b on e4m3 operands


def _res_dense(P, act, name):
    """one Dense of a DenseResBlock on the FiLM-LayerNorm output `act` (un-rounded): bf16 operands by default; in fp8 mode the
    LayerNorm kernel writes the row as e4m3 + an E8M0 row scale (and a bf16 copy for the weight gradient), see e4m3_emulation.py"""
    if _RES_DENSE is not None:
        return _RES_DENSE(H(act), P.w(name), P.b(name))
    return H(rb(act)) @ P.w(name) + P.b(name)


def _res_block(P, y, name, scale, shift):
    a = O.swish(scale * P.ln(y, name + ".ln1") + shift)          # (H: dgrad output of fc1 -> bf16)
    o1 = H(rb(_res_dense(P, a, name + ".fc1")))                  # d o1 (LayerNorm 2 backward output) -> bf16
    a = O.swish(scale * P.ln(o1, name + ".ln2") + shift)         # (H: dgrad output of fc2 -> bf16)
    return H(rb(_res_dense(P, a, name + ".fc2") + y))            # trunk gradient dy -> bf16


def _attention(P, a1, name, num_heads):
    B, S, E = a1.shape
    d = E // num_heads
    qkv = a1 @ P.w(name + ".qkv") + P.b(name + ".qkv")
    q, k, v = qkv.split(E, dim=-1)
    q, k, v = H(rb(q)), H(rb(k)), H(rb(v))                        # dq (x 1/sqrt d), dk, dv -> bf16 (dqkv)
    q = rb(q * (1.0 / math.sqrt(d)))
    q, k, v = (z.reshape(B, S, num_heads, d) for z in (q, k, v))
    logits = H(torch.einsum("bqhd,bkhd->bhqk", q, k))            # dS = p (dP - rowdot) -> bf16
    w = rb(torch.softmax(logits, dim=-1))
    o = H(rb(torch.einsum("bhqk,bkhd->bqhd", w, v).reshape(B, S, E)))    # dO = dh_mid Wo^T -> bf16
    return H(o @ P.w(name + ".out") + P.b(name + ".out"))        # the branch reads the bf16 copy of dh


def transformer_ddpm(p, cfg, inputs, t):
    P = _P(p)
    B, S, C = inputs.shape
    temb = O.positional_encoding(S, cfg.embed_channels, inputs.dtype)[None]
    x = H(rb(inputs) @ P.w("in_proj") + P.b("in_proj")) + f32(temb)
    for l in range(cfg.num_layers):
        pre = f"enc.{l}"
        x = _attention(P, H(rb(P.ln(x, pre + ".ln1"))), pre + ".attn", cfg.num_heads) + x     # da1 -> bf16
        a2 = rb(P.ln(x, pre + ".ln2"))                                                        # da2: fp32 partial tiles
        z = H(a2 @ P.w(pre + ".mlp.fc1") + P.b(pre + ".mlp.fc1"))                             # dz = du gelu'(z) -> bf16
        u = rb(O.gelu(z))
        x = H(u @ P.w(pre + ".mlp.fc2") + P.b(pre + ".mlp.fc2")) + x
    af = H(rb(P.ln(x, "ln_f")))
    y = H(rb(af @ P.w("up") + P.b("up")))
    for k in range(cfg.num_mlp_layers):
        scale, shift = _film(P, t.squeeze(-1), f"film.{k}", cfg.film_channels, cfg.mlp_dims, True)
        y = _res_block(P, y, f"res.{k}", scale, shift)
    ao = H(rb(P.ln(y, "ln_o")))
    return H(ao @ P.w("out_proj") + P.b("out_proj"))             # d eps_hat -> bf16


def dense_ddpm(p, cfg, inputs, t):
    P = _P(p)
    y = H(rb(rb(inputs) @ P.w("in_proj") + P.b("in_proj")))
    for k in range(cfg.num_layers):
        scale, shift = _film(P, t, f"film.{k}", cfg.film_channels, cfg.mlp_dims, False)
        y = _res_block(P, y, f"res.{k}", scale, shift)
    ao = H(rb(P.ln(y, "ln_o")))
    return H(ao @ P.w("out_proj") + P.b("out_proj"))


def make_model(p, cfg, backward=False, noise_embedding=None):
    """model(x, cond) -> eps_hat as the bf16 engine computes it, up to fp32 accumulation order (float64 arithmetic).
    backward=True: parameters with requires_grad get the engine's gradient (bf16 gradient storage emulated by hooks).
    noise_embedding: the (B, 128) sinusoidal embedding as the device kernel produced it (smd_noise_embed).  Its angles
    reach 5000 rad in float32 (models/ncsn.py:36-38 computes them in float32 too), where ONE ulp of the frequency moves the
    sine by 3e-4: whichever float32 exp / sincos evaluates it, a few per cent of the bf16 entries land on the other side of
    a rounding boundary, and every FiLM scale / shift moves with them.  Passing the device's own embedding takes that
    library-level ambiguity out of the comparison."""
    fn = dense_ddpm if cfg.architecture == "DenseDDPM" else transformer_ddpm

    def model(x, t):
        global _BACKWARD, _EMB_OVERRIDE
        _BACKWARD, _EMB_OVERRIDE = bool(backward), noise_embedding
        try:
            return fn(p, cfg, x, t)
        finally:
            _BACKWARD, _EMB_OVERRIDE = False, None
    return model
