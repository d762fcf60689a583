"""The oracle's layer formulas against INDEPENDENT LIBRARY implementations of the same published semantics.

The reference's arithmetic lives in flax 0.3.0 / jax 0.2.8, which cannot be installed here, so the oracle restates it
(oracle/ddpm_oracle.py, ORACLE_ASSUMPTIONS).  These tests guard that restatement against transcription errors by computing
the same quantities with torch's own library kernels -- code written by other people from the same public definitions:

  layer_norm            F.layer_norm(eps=1e-6)                       (biased variance, affine)
  gelu (tanh form)      F.gelu(approximate="tanh")
  swish                 F.silu
  self_attention        nn.MultiheadAttention (packed in_proj [q;k;v], q scaled by 1/sqrt(d) before the logits, heads = contiguous
                        d-wide slices of E) and F.scaled_dot_product_attention on pre-scaled q with scale = 1
  encoder layer         nn.TransformerEncoderLayer(norm_first=True, activation=tanh-gelu, layer_norm_eps=1e-6)
  whole TransformerDDPM a torch.nn module graph assembled from nn.Linear / nn.LayerNorm / nn.MultiheadAttention
  Adam                  torch.optim.Adam (bias-corrected, eps outside the square root: the flax.optim.Adam placement)
  clip_grads            torch.nn.utils.clip_grad_norm_ (its 1e-6 in the denominator is the only difference)
  sinusoidal encodings  closed forms
All in float64 on the CPU.  These do not replace golden vectors of the reference itself (parity stays "unpinned" until
tests/golden/make_jax_goldens.py can run), they make a transcription error in the oracle visible.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import ddpm_oracle as O

DT = torch.float64


def rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a - b).norm() / (b.norm() + 1e-300))


def jittered(cfg, seed=0):
    p = O.init_params(cfg, seed, DT)
    g = torch.Generator().manual_seed(seed + 1)
    for k in p:
        if k.endswith(".bias"):
            p[k] = 0.1 * torch.randn(p[k].shape, generator=g, dtype=DT)
        elif k.endswith(".scale"):
            p[k] = 1 + 0.1 * torch.randn(p[k].shape, generator=g, dtype=DT)
    return p


def test_layernorm_gelu_swish_against_torch_functional():
    g = torch.Generator().manual_seed(0)
    x = 3 * torch.randn(5, 7, 128, generator=g, dtype=DT) + 0.5
    p = {"n.scale": 1 + 0.2 * torch.randn(128, generator=g, dtype=DT), "n.bias": 0.3 * torch.randn(128, generator=g, dtype=DT)}
    assert rel(O.layer_norm(x, p, "n"), F.layer_norm(x, (128,), p["n.scale"], p["n.bias"], eps=1e-6)) < 1e-13
    # torch's default eps (1e-5) is NOT the reference's: the test must be able to tell them apart on small-variance rows
    xs = 1e-3 * torch.randn(4, 128, generator=g, dtype=DT)
    assert rel(O.layer_norm(xs, p, "n"), F.layer_norm(xs, (128,), p["n.scale"], p["n.bias"], eps=1e-5)) > 1e-2
    assert rel(O.layer_norm(xs, p, "n"), F.layer_norm(xs, (128,), p["n.scale"], p["n.bias"], eps=1e-6)) < 1e-10
    z = torch.linspace(-6, 6, 1001, dtype=DT)
    assert rel(O.gelu(z), F.gelu(z, approximate="tanh")) < 1e-14
    assert rel(O.gelu(z), F.gelu(z)) > 1e-5                      # the erf form is a different function
    assert rel(O.swish(z), F.silu(z)) < 1e-15


@pytest.mark.parametrize("H", [8, 16, 4])
def test_self_attention_against_nn_multihead_attention_and_sdpa(H):
    E, B, S = 128, 3, 32
    g = torch.Generator().manual_seed(H)
    p = {"a.qkv.kernel": 0.2 * torch.randn(E, 3 * E, generator=g, dtype=DT), "a.qkv.bias": 0.1 * torch.randn(3 * E, generator=g, dtype=DT),
         "a.out.kernel": 0.2 * torch.randn(E, E, generator=g, dtype=DT), "a.out.bias": 0.1 * torch.randn(E, generator=g, dtype=DT)}
    x = torch.randn(B, S, E, generator=g, dtype=DT)
    got = O.self_attention(x, p, "a", H)
    mha = nn.MultiheadAttention(E, H, bias=True, batch_first=True, dtype=DT)
    with torch.no_grad():
        mha.in_proj_weight.copy_(p["a.qkv.kernel"].T)           # rows [q; k; v], each (out, in): flax kernel (in, out) transposed
        mha.in_proj_bias.copy_(p["a.qkv.bias"])
        mha.out_proj.weight.copy_(p["a.out.kernel"].T)
        mha.out_proj.bias.copy_(p["a.out.bias"])
        ref, _ = mha(x, x, x, need_weights=False)
    assert rel(got, ref) < 1e-12
    # the same through F.scaled_dot_product_attention with q scaled BEFORE the logits (the flax order) and scale = 1
    d = E // H
    qkv = x @ p["a.qkv.kernel"] + p["a.qkv.bias"]
    q, k, v = (t.reshape(B, S, H, d).transpose(1, 2) for t in qkv.split(E, dim=-1))
    o = F.scaled_dot_product_attention(q / math.sqrt(d), k, v, scale=1.0).transpose(1, 2).reshape(B, S, E)
    assert rel(got, o @ p["a.out.kernel"] + p["a.out.bias"]) < 1e-12


class TorchDenseFiLM(nn.Module):
    """models/ncsn.py:47-61 from torch library layers."""

    def __init__(self, F_, M):
        super().__init__()
        self.fc1, self.fc2, self.ss = nn.Linear(F_, 4 * F_, dtype=DT), nn.Linear(4 * F_, 4 * F_, dtype=DT), nn.Linear(4 * F_, 2 * M, dtype=DT)
        self.F_, self.M = F_, M

    def forward(self, s):                                         # s: (B,) noise levels
        half = self.F_ // 2
        f = torch.exp(torch.arange(half, dtype=DT) * -(math.log(10000.0) / (half - 1)))
        a = 5000.0 * s[:, None] * f[None]
        e = torch.cat([a.sin(), a.cos()], 1)
        ss = self.ss(self.fc2(F.silu(self.fc1(e))))
        return ss[:, :self.M], ss[:, self.M:]


class TorchResBlock(nn.Module):
    """models/shared.py:61-75."""

    def __init__(self, M):
        super().__init__()
        self.ln1, self.fc1, self.ln2, self.fc2 = nn.LayerNorm(M, eps=1e-6, dtype=DT), nn.Linear(M, M, dtype=DT), nn.LayerNorm(M, eps=1e-6, dtype=DT), nn.Linear(M, M, dtype=DT)

    def forward(self, x, scale, shift):
        o = self.fc1(F.silu(scale * self.ln1(x) + shift))
        o = self.fc2(F.silu(scale * self.ln2(o) + shift))
        return o + x


class TorchTransformerDDPM(nn.Module):
    """models/ncsn.py:141-179 assembled from torch.nn library modules only."""

    def __init__(self, cfg):
        super().__init__()
        E, M = cfg.embed_channels, cfg.mlp_dims
        self.cfg = cfg
        self.in_proj = nn.Linear(cfg.data_channels, E, dtype=DT)
        self.enc = nn.ModuleList([nn.TransformerEncoderLayer(E, cfg.num_heads, dim_feedforward=M, dropout=0.0,
                                                             activation=lambda t: F.gelu(t, approximate="tanh"), layer_norm_eps=1e-6,
                                                             batch_first=True, norm_first=True, dtype=DT) for _ in range(cfg.num_layers)])
        self.ln_f, self.up = nn.LayerNorm(E, eps=1e-6, dtype=DT), nn.Linear(E, M, dtype=DT)
        self.film = nn.ModuleList([TorchDenseFiLM(cfg.film_channels, M) for _ in range(cfg.num_mlp_layers)])
        self.res = nn.ModuleList([TorchResBlock(M) for _ in range(cfg.num_mlp_layers)])
        self.ln_o, self.out_proj = nn.LayerNorm(M, eps=1e-6, dtype=DT), nn.Linear(M, cfg.data_channels, dtype=DT)

    def load_oracle(self, p):
        def lin(m, name):
            m.weight.data.copy_(p[name + ".kernel"].T); m.bias.data.copy_(p[name + ".bias"])

        def ln(m, name):
            m.weight.data.copy_(p[name + ".scale"]); m.bias.data.copy_(p[name + ".bias"])
        lin(self.in_proj, "in_proj")
        for l, e in enumerate(self.enc):
            pre = f"enc.{l}"
            ln(e.norm1, pre + ".ln1"); ln(e.norm2, pre + ".ln2")
            e.self_attn.in_proj_weight.data.copy_(p[pre + ".attn.qkv.kernel"].T); e.self_attn.in_proj_bias.data.copy_(p[pre + ".attn.qkv.bias"])
            lin(e.self_attn.out_proj, pre + ".attn.out"); lin(e.linear1, pre + ".mlp.fc1"); lin(e.linear2, pre + ".mlp.fc2")
        ln(self.ln_f, "ln_f"); lin(self.up, "up")
        for k in range(self.cfg.num_mlp_layers):
            lin(self.film[k].fc1, f"film.{k}.fc1"); lin(self.film[k].fc2, f"film.{k}.fc2"); lin(self.film[k].ss, f"film.{k}.ss")
            ln(self.res[k].ln1, f"res.{k}.ln1"); lin(self.res[k].fc1, f"res.{k}.fc1"); ln(self.res[k].ln2, f"res.{k}.ln2"); lin(self.res[k].fc2, f"res.{k}.fc2")
        ln(self.ln_o, "ln_o"); lin(self.out_proj, "out_proj")

    def forward(self, x, t):                                       # x (B,S,C), t (B,1,1)
        B, S, _ = x.shape
        E = self.cfg.embed_channels
        half = E // 2
        f = torch.exp(torch.arange(half, dtype=DT) * -(math.log(10000.0) / (half - 1)))
        a = torch.arange(S, dtype=DT)[:, None] * f[None]
        pe = torch.cat([a.sin(), a.cos()], 1)                      # [sin | cos] halves (models/shared.py:44)
        h = self.in_proj(x) + pe[None]
        for e in self.enc:
            h = e(h)
        y = self.up(self.ln_f(h))
        for k in range(self.cfg.num_mlp_layers):
            sc, sh = self.film[k](t.reshape(B))
            y = self.res[k](y, sc[:, None, :], sh[:, None, :])
        return self.out_proj(self.ln_o(y))


@pytest.mark.parametrize("C,L,H,K,M", [(42, 2, 8, 1, 256), (64, 3, 16, 3, 128)])
def test_whole_network_and_its_gradient_against_torch_nn_modules(C, L, H, K, M):
    cfg = O.NetConfig(data_channels=C, num_layers=L, num_heads=H, num_mlp_layers=K, mlp_dims=M)
    p = jittered(cfg, seed=3)
    net = TorchTransformerDDPM(cfg)
    net.load_oracle(p)
    net.train()                                                    # (the fused inference fast path of TransformerEncoderLayer is not wanted)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 32, C, generator=g, dtype=DT)
    t = (0.05 + 0.95 * torch.rand(3, generator=g, dtype=DT)).view(3, 1, 1)
    leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    out_o = O.make_model(leaf, cfg)(x, t)
    out_t = net(x, t)
    assert rel(out_o, out_t) < 1e-11
    w = torch.randn(out_o.shape, generator=g, dtype=DT)
    (out_o * w).sum().backward()
    (out_t * w).sum().backward()
    assert rel(leaf["enc.0.attn.qkv.kernel"].grad, net.enc[0].self_attn.in_proj_weight.grad.T) < 1e-10
    assert rel(leaf["res.0.fc1.kernel"].grad, net.res[0].fc1.weight.grad.T) < 1e-10
    assert rel(leaf["film.0.ss.kernel"].grad, net.film[0].ss.weight.grad.T) < 1e-10
    assert rel(leaf["in_proj.kernel"].grad, net.in_proj.weight.grad.T) < 1e-10
    assert rel(leaf["enc.1.ln2.scale"].grad, net.enc[1].norm2.weight.grad) < 1e-10


def test_encodings_closed_forms():
    pe = O.positional_encoding(32, 128)
    s = torch.arange(32, dtype=DT)[:, None]
    i = torch.arange(64, dtype=DT)[None]
    fr = 10000.0 ** (-i / 63.0)
    assert rel(pe[:, :64], torch.sin(s * fr)) < 1e-12 and rel(pe[:, 64:], torch.cos(s * fr)) < 1e-12
    lv = torch.tensor([[0.3], [0.999]], dtype=DT)
    ne = O.noise_encoding(lv, 128)
    assert rel(ne[:, :64], torch.sin(5000.0 * lv * fr)) < 1e-10 and rel(ne[:, 64:], torch.cos(5000.0 * lv * fr)) < 1e-10


def test_adam_and_clip_against_torch_optim():
    g = torch.Generator().manual_seed(9)
    shapes = {"a": (7, 5), "b": (11,), "c": (3, 2)}
    p0 = {k: torch.randn(s, generator=g, dtype=DT) for k, s in shapes.items()}
    params = [nn.Parameter(p0[k].clone()) for k in shapes]
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    st = O.AdamState()
    p = {k: v.clone() for k, v in p0.items()}
    for step in range(4):
        grads = {k: (3.0 if step == 1 else 0.05) * torch.randn(s, generator=g, dtype=DT) for k, s in shapes.items()}
        clipped, norm_after = O.clip_grads(grads, 1.0)
        for prm, k in zip(params, shapes):
            prm.grad = grads[k].clone()
        total = torch.nn.utils.clip_grad_norm_(params, 1.0)          # coefficient 1 / (norm + 1e-6), clamped to 1
        ref_norm = math.sqrt(sum(float((g_ * g_).sum()) for g_ in grads.values()))
        assert abs(float(total) - ref_norm) < 1e-12
        for prm, k in zip(params, shapes):
            assert rel(clipped[k], prm.grad) < 3e-6                  # the library's +1e-6 is the whole difference
            prm.grad = clipped[k].clone()                            # continue from the oracle's exact clip
        assert float(norm_after) <= 1.0 + 1e-12
        lr = O.stepped_lr(1e-3, step, 2, 0.5)
        for grp in opt.param_groups:
            grp["lr"] = lr
        opt.step()
        p = O.adam_update(p, clipped, st, lr)
        for prm, k in zip(params, shapes):
            assert rel(p[k], prm.data) < 1e-13, (step, k)
    assert st.step == 4
