"""GPU parity of the engine (eps-net forward, loss+backward, optimiser step, reverse sampler) against
the CPU restatement oracle on identical weights / inputs / noise draws.

Tolerances: bf16-MFMA path rel-L2 <= 1e-2 on eps_hat and on the gradient,
loss scalar <= 5e-3 relative; fp32 elementwise pieces <= 1e-5.
"""
import numpy as np
import pytest
import torch

import ddpm_oracle as O

pytestmark = pytest.mark.gpu

BETAS = O.create_noise_schedule(1e-6, 0.01, 1000, "linear")


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def make(arch="TransformerDDPM", C=512, L=6, H=8, K=2, M=2048, seed=0, jitter=True):
    """Oracle params (fp64) + an smd_amd Model loaded with the same values."""
    import smd_amd.ncsn as N
    from smd_amd.engine import NetConfig
    ocfg = O.NetConfig(architecture=arch, data_channels=C, num_layers=L, num_heads=H, num_mlp_layers=K, mlp_dims=M)
    p = O.init_params(ocfg, seed, torch.float64)
    if jitter:   # non-trivial biases / LayerNorm affine so every term is exercised
        g = torch.Generator().manual_seed(seed + 1)
        for k in p:
            if k.endswith(".bias"):
                p[k] = 0.1 * torch.randn(p[k].shape, generator=g, dtype=torch.float64)
            elif k.endswith(".scale"):
                p[k] = 1 + 0.1 * torch.randn(p[k].shape, generator=g, dtype=torch.float64)
    cfg = NetConfig(architecture=arch, data_channels=C, seq_len=32, num_layers=L, num_heads=H, num_mlp_layers=K,
                    mlp_dims=M, num_timesteps=1000)
    model = N.Model(cfg, "cuda:0", seed=None)
    model.engine.load_named(p)
    return ocfg, p, model


def device_noise_embedding(s):
    """NoiseEncoding (models/ncsn.py:28-41) as the device kernel evaluates it: (B, 128) bf16 values, returned as float64"""
    import smd_amd.lib as lib
    sd = torch.as_tensor(s, dtype=torch.float32).reshape(-1).cuda().contiguous()
    out = torch.zeros(sd.numel(), 128, dtype=torch.bfloat16, device="cuda")
    lib.check(lib.get_lib().smd_noise_embed(sd.data_ptr(), sd.numel(), 128, out.data_ptr(), 128, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out.double().cpu()


def data(B, shape, seed=1234):
    g = torch.Generator().manual_seed(seed)
    x0 = torch.clamp(0.25 * torch.randn(B, *shape, generator=g), -1, 1)
    return x0, g


@pytest.mark.parametrize("arch,C,L,H,K", [("TransformerDDPM", 512, 6, 8, 2), ("TransformerDDPM", 42, 2, 8, 1),
                                           ("TransformerDDPM", 146, 2, 16, 3), ("DenseDDPM", 512, 3, 8, 2)])
def test_forward_parity(arch, C, L, H, K):
    ocfg, p, model = make(arch, C, L, H, K)
    B = 6
    shape = (C,) if arch == "DenseDDPM" else (32, C)
    x, g = data(B, shape)
    s = 0.05 + 0.95 * torch.rand(B, generator=g)
    cond = s.view(B, *([1] * len(shape)))
    ref = O.make_model(p, ocfg)(x.double(), cond.double())
    out = model(x, cond)
    e = rel(out, ref)
    print(f"forward {arch} C={C} L={L} H={H} K={K}: rel-L2 {e:.3e}")
    assert e < 1e-2


def test_forward_batch_invariance_and_determinism():
    _, _, model = make(C=512, L=2, K=1)
    x, g = data(8, (32, 512))
    s = torch.rand(8, generator=g).view(8, 1, 1)
    a = model(x, s).clone()
    b = model(x, s)
    assert torch.equal(a, b)                                  # same inputs twice -> bitwise equal
    c = model(x[:3], s[:3])
    assert rel(c, a[:3]) < 1e-6                               # rows are independent of the batch they ride in


@pytest.mark.parametrize("arch,C,L,H,K,tr", [("TransformerDDPM", 512, 2, 8, 1, 1), ("TransformerDDPM", 512, 2, 8, 1, 0),
                                              ("TransformerDDPM", 42, 6, 16, 2, 1), ("DenseDDPM", 512, 2, 8, 2, 1),
                                              ("TransformerDDPM", 512, 2, 8, 1, 3), ("DenseDDPM", 512, 2, 8, 2, 3)])
def test_loss_and_gradient_parity(arch, C, L, H, K, tr):
    """tr: 1 LDS-transpose wgrad kernels (default path), 0 explicit-transpose fallback, 3 = 1 + the optional paths
    (grouped 128-wide wgrads, single stream, unfused encoder forward, fused training MLP)."""
    ocfg, p, model = make(arch, C, L, H, K)
    B = 64 if arch == "DenseDDPM" else 4
    shape = (C,) if arch == "DenseDDPM" else (32, C)
    x0, g = data(B, shape)
    labels = torch.randint(1, 1001, (B,), generator=g)
    labels[0] = 1                                             # alpha = 1 (zero-noise) corner of the quirk
    eps = torch.randn(B, *shape, generator=g)
    leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss_ref = O.diffusion_loss(x0.double(), O.make_model(leaf, ocfg), BETAS, labels.numpy(), eps.double(), "none")
    loss_ref.mean().backward()

    eng = model.train_engine(ema=True)
    eng.set_option("tr_path", 1 if tr == 3 else tr)
    if tr == 3:
        eng.set_option("group_wgrad", 1 if arch == "TransformerDDPM" else 0)
        eng.set_option("fused_attn_bwd", 0)
        eng.set_option("side_wgrad", 0)
        eng.set_option("fused_encoder", 2 if arch == "TransformerDDPM" else 0)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(B, training=True)
    eng.loss_backward(x0.cuda(), labels.int().cuda(), eps.cuda(), stage=0)
    torch.cuda.synchronize()
    le = rel(eng.loss_per_sample(), loss_ref)
    print(f"loss per-sample rel {le:.3e}; mean {float(eng.loss_per_sample().mean()):.6f} vs {float(loss_ref.mean()):.6f}")
    assert abs(float(eng.loss_per_sample().mean()) - float(loss_ref.mean())) / float(loss_ref.mean()) < 5e-3
    gv = eng.named_views(eng.grads)
    worst, num, den = 0.0, 0.0, 0.0
    for k, v in leaf.items():
        r = rel(gv[k], v.grad)
        worst = max(worst, r)
        num += float((gv[k].double().cpu() - v.grad).pow(2).sum())
        den += float(v.grad.pow(2).sum())
        if r > 2e-2:
            print(f"  grad {k:28s} rel {r:.3e} |g| {float(v.grad.norm()):.3e}")
    total = (num / den) ** 0.5
    print(f"gradient {arch} C={C} tr={tr}: whole-vector rel-L2 {total:.3e}, worst tensor {worst:.3e}")
    assert total < 1e-2
    assert worst < 6e-2                                        # small bias tensors carry the most bf16 noise


@pytest.mark.parametrize("B", [5, 7, 12])
def test_loss_and_gradient_parity_at_odd_batch_sizes(B):
    """Batch sizes that leave the shapes the fused encoder paths are built for: B = 5 / 7 (160 / 224 token rows: the hidden-split
    MLP forward runs one sample per group, the backward falls back to the unfused kernels), B = 12 (384 rows = three groups of 128:
    hidden-split forward and recompute backward).  Same tolerances as the regular sizes."""
    ocfg, p, model = make("TransformerDDPM", 42, 2, 8, 1)
    x0, g = data(B, (32, 42))
    labels = torch.randint(1, 1001, (B,), generator=g)
    eps = torch.randn(B, 32, 42, generator=g)
    leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss_ref = O.diffusion_loss(x0.double(), O.make_model(leaf, ocfg), BETAS, labels.numpy(), eps.double(), "none")
    loss_ref.mean().backward()
    eng = model.train_engine(ema=False)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(B, training=True)
    outs = []
    for rep in range(2):
        eng.loss_backward(x0.cuda(), labels.int().cuda(), eps.cuda(), stage=0)
        torch.cuda.synchronize()
        outs.append(eng.grads.clone())
    assert torch.equal(outs[0], outs[1])                       # repeatable
    m_ref = float(loss_ref.detach().mean())
    assert abs(float(eng.loss_per_sample().mean()) - m_ref) / m_ref < 5e-3
    gv = eng.named_views(eng.grads)
    num = sum(float((gv[k].double().cpu() - v.grad).pow(2).sum()) for k, v in leaf.items())
    den = sum(float(v.grad.pow(2).sum()) for v in leaf.values())
    total = (num / den) ** 0.5
    print(f"gradient at B={B}: whole-vector rel-L2 {total:.3e}")
    assert total < 1e-2
    # and the sampler's forward at the same batch size
    s = 0.05 + 0.95 * torch.rand(B, generator=g)
    ref = O.make_model(p, ocfg)(x0.double(), s.view(B, 1, 1).double())
    assert rel(model(x0, s.view(B, 1, 1)), ref) < 1e-2


@pytest.mark.parametrize("arch,C,L,H,K", [("TransformerDDPM", 512, 6, 8, 2), ("TransformerDDPM", 512, 8, 16, 3), ("TransformerDDPM", 42, 2, 8, 1),
                                           ("TransformerDDPM", 146, 2, 16, 3), ("DenseDDPM", 512, 3, 8, 2)])
def test_forward_against_the_bf16_emulating_oracle(arch, C, L, H, K):
    """oracle/bf16_emulation.py evaluates the same network in float64 with every bf16 rounding of the engine at the same
    place.  What this CAN show at network level is bounded by a property of bf16 storage itself, measured here: two
    evaluations that differ by fp32 accumulation order only -- the engine's own inference and training paths -- are already
    2e-3 apart at L = 6 (a 1e-7 difference flips a rounding in ~1 element in 10^4, the flip is a whole ulp, the next
    LayerNorm spreads it over the row, a few per cent of the next layer's roundings flip ...: after four or five rounded
    layers two runs are decorrelated to a constant fraction of the rounding noise).  So eps_hat sits at 1e-3 (DenseDDPM, three
    blocks) to 5e-3 (eight encoder layers) from the emulation against 6.3e-3 ... 6.7e-3 from the exact oracle; the sharp checks
    are the gradient test below (1.2e-3: the gradient depends smoothly on the activations) and the layer-by-layer
    teacher-forced test (test_saved_activations_layer_by_layer, 1e-4: no cascade)."""
    import bf16_emulation as E
    ocfg, p, model = make(arch, C, L, H, K)
    B = 8
    shape = (C,) if arch == "DenseDDPM" else (32, C)
    x, g = data(B, shape)
    s = 0.05 + 0.95 * torch.rand(B, generator=g)
    cond = s.view(B, *([1] * len(shape)))
    exact = O.make_model(p, ocfg)(x.double(), cond.double())
    emu_own = E.make_model(p, ocfg)(x.double(), cond.double())
    emu = E.make_model(p, ocfg, noise_embedding=device_noise_embedding(s))(x.double(), cond.double())
    out = model(x, cond)
    e_exact, e_emu = rel(out, exact), rel(out, emu)
    print(f"{arch} C={C} L={L} H={H} K={K}: eps_hat vs exact fp64 oracle {e_exact:.3e}; vs bf16-emulating oracle {e_emu:.3e} "
          f"(with the emulation's own float32 noise embedding: {rel(out, emu_own):.3e}; the emulation itself vs exact: {rel(emu, exact):.3e})")
    assert e_exact < 1e-2
    assert e_emu < (2e-3 if arch == "DenseDDPM" else 0.9 * e_exact)


@pytest.mark.parametrize("arch,C,L,H,K", [("TransformerDDPM", 512, 6, 8, 2), ("TransformerDDPM", 146, 2, 16, 3), ("TransformerDDPM", 42, 2, 8, 1),
                                           ("DenseDDPM", 512, 3, 8, 2)])
def test_gradient_against_the_bf16_emulating_oracle(arch, C, L, H, K):
    """The backward pass below its rounding noise: autograd through oracle/bf16_emulation.py (every forward rounding a
    straight-through estimator, every gradient the engine stores in bf16 rounded by a hook at the same place) against the
    engine's loss_backward on the same labels / eps.  Tolerances 2.5e-3 (4e-3 DenseDDPM) whole gradient / 2e-2 worst tensor
    instead of 1e-2 / 6e-2 against the exact oracle (measured 1.2e-3 ... 1.4e-3, 3.0e-3); the printed line also shows how much of
    the distance to the exact oracle is explained by the forward roundings alone (hooks off)."""
    import bf16_emulation as E
    ocfg, p, model = make(arch, C, L, H, K)
    B = 8
    shape = (C,) if arch == "DenseDDPM" else (32, C)
    x0, g = data(B, shape)
    labels = torch.randint(1, 1001, (B,), generator=g)
    eps = torch.randn(B, *shape, generator=g)
    eng = model.train_engine(ema=False)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(B, training=True)
    eng.loss_backward(x0.cuda(), labels.int().cuda(), eps.cuda(), stage=0)
    torch.cuda.synchronize()
    gv = {k: v.double().cpu().clone() for k, v in eng.named_views(eng.grads).items()}
    loss_eng = eng.loss_per_sample().double().cpu().clone()

    def oracle(mk):
        leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        per = O.diffusion_loss(x0.double(), mk(leaf), BETAS, labels.numpy(), eps.double(), "none")
        per.mean().backward()
        return {k: v.grad for k, v in leaf.items()}, per.detach()

    def dist(ref):
        num = sum(float((gv[k] - ref[k]).pow(2).sum()) for k in ref)
        den = sum(float(ref[k].pow(2).sum()) for k in ref)
        worst = max((rel(gv[k], ref[k]), k) for k in ref if float(ref[k].norm()) > 0)
        return (num / den) ** 0.5, worst

    g_exact, l_exact = oracle(lambda q: O.make_model(q, ocfg))
    lv = torch.from_numpy(O.used_alphas_from_labels(BETAS, labels.numpy())).sqrt()          # float32 sqrt(alpha'[l - 1]): the cond of this batch
    emb = device_noise_embedding(lv)
    g_fwd, _ = oracle(lambda q: E.make_model(q, ocfg, backward=False, noise_embedding=emb))
    g_emu, l_emu = oracle(lambda q: E.make_model(q, ocfg, backward=True, noise_embedding=emb))
    (e_exact, _), (e_fwd, _), (e_emu, (w_emu, w_name)) = dist(g_exact), dist(g_fwd), dist(g_emu)
    print(f"{arch} C={C} L={L} K={K}: gradient vs exact fp64 oracle {e_exact:.3e}; vs forward-rounding emulation {e_fwd:.3e}; vs forward + "
          f"backward emulation {e_emu:.3e} (worst tensor {w_name} {w_emu:.3e}); per-sample loss vs exact {rel(loss_eng, l_exact):.3e}, vs "
          f"emulation {rel(loss_eng, l_emu):.3e}")
    assert e_exact < 1e-2
    assert e_emu < (4e-3 if arch == "DenseDDPM" else 2.5e-3) and w_emu < 2e-2
    assert e_emu < 0.5 * e_exact
    assert rel(loss_eng, l_emu) < 5e-4


@pytest.mark.parametrize("arch,C,L,H,K,B", [("TransformerDDPM", 512, 6, 8, 2, 8), ("TransformerDDPM", 146, 2, 16, 3, 8), ("DenseDDPM", 42, 3, 8, 2, 8),
                                             ("TransformerDDPM", 512, 6, 8, 2, 256)])
def test_saved_activations_layer_by_layer(arch, C, L, H, K, B):
    """The semantic check below the rounding noise (VERDICT r3 missing #3), without the cascade that limits every network-level
    comparison of two bf16 evaluations: each kernel's SAVED output (training-mode forward, smd_engine_debug_tensor) against the
    float64 evaluation of that one layer on the engine's OWN saved inputs, rounded where the kernel rounds
    (oracle/bf16_emulation.py building blocks).  One rounding stage per comparison: what is left is fp32 accumulation order, the
    hardware exp / rcp / rsq, and isolated flips next to bf16 ties -- measured: fp32 outputs agree to 1.4e-5 (the MLP output,
    through flips of its bf16 hidden activations), bf16 outputs to 4.4e-5 rel-L2 with <= 0.02 % of the elements one ulp off;
    asserted: 5e-5 / 2e-4 / 0.2 %.  An epilogue term, a FiLM broadcast or a scale on the wrong side of a rounding that
    is wrong by 1e-3 of a layer's output fails here, at B = 8 and at the benchmark's B = 256 (in situ: 8192-row GEMMs)."""
    import bf16_emulation as E
    ocfg, p, model = make(arch, C, L, H, K)
    shape = (C,) if arch == "DenseDDPM" else (32, C)
    x0, g = data(B, shape)
    labels = torch.randint(1, 1001, (B,), generator=g)
    eps = torch.randn(B, *shape, generator=g)
    eng = model.train_engine(ema=False)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(B, training=True)
    eng.loss_backward(x0.cuda(), labels.int().cuda(), eps.cuda(), stage=3)          # forward + loss only: the saved activations stay
    torch.cuda.synchronize()
    T = lambda name, i=0: eng.debug_tensor(name, i).double().cpu()
    P = E._P(p)
    rb = E.rb
    S = 1 if arch == "DenseDDPM" else 32
    E_, M = 128, ocfg.mlp_dims
    worst = {}

    def check(name, got, want, is_bf16):
        e = rel(got, want)
        frac = float((got != want).double().mean()) if is_bf16 else 0.0
        worst[name.split("[")[0]] = max(worst.get(name.split("[")[0], (0.0, 0.0)), (e, frac))
        assert e < (2e-4 if is_bf16 else 5e-5), f"{name}: rel {e:.3e} ({frac * 100:.2f} % of the elements differ)"
        assert frac < 0.002, f"{name}: {frac * 100:.2f} % of the bf16 elements differ"

    x_in = T("x_bf16")[:, :C]
    if arch == "TransformerDDPM":
        pe = O.positional_encoding(32, E_, torch.float64).float().double().repeat(B, 1)
        check("h[0]", T("h", 0), x_in @ P.w("in_proj") + P.b("in_proj") + pe, False)
        for l in range(L):
            pre = f"enc.{l}"
            h, a1, qkv, o, h_mid, a2 = T("h", l), T("a1", l), T("qkv", l), T("o", l), T("h_mid", l), T("a2", l)
            check(f"a1[{l}]", a1, rb(P.ln(h, pre + ".ln1")), True)
            check(f"qkv[{l}]", qkv, rb(a1 @ P.w(pre + ".attn.qkv") + P.b(pre + ".attn.qkv")), True)
            d = E_ // H
            q, k, v = (z.reshape(B, 32, H, d) for z in qkv.split(E_, dim=-1))
            w = rb(torch.softmax(torch.einsum("bqhd,bkhd->bhqk", rb(q * (1.0 / d ** 0.5)), k), dim=-1))
            check(f"o[{l}]", o, rb(torch.einsum("bhqk,bkhd->bqhd", w, v).reshape(B * 32, E_)), True)
            check(f"h_mid[{l}]", h_mid, o @ P.w(pre + ".attn.out") + P.b(pre + ".attn.out") + h, False)
            check(f"a2[{l}]", a2, rb(P.ln(h_mid, pre + ".ln2")), True)
            u = rb(O.gelu(a2 @ P.w(pre + ".mlp.fc1") + P.b(pre + ".mlp.fc1")))
            h_next = T("h", l + 1) if l + 1 < L else T("h_last")
            check(f"h[{l + 1}]", h_next, u @ P.w(pre + ".mlp.fc2") + P.b(pre + ".mlp.fc2") + h_mid, False)
        af = T("af")
        check("af", af, rb(P.ln(T("h_last"), "ln_f")), True)
        check("y[0]", T("y", 0), rb(af @ P.w("up") + P.b("up")), True)
        nblk = K
    else:
        check("y[0]", T("y", 0), rb(x_in @ P.w("in_proj") + P.b("in_proj")), True)
        nblk = L
    emb = T("emb")
    for k in range(nblk):
        f1, pp, ss = T("f1", k), T("p", k), T("ss", k)
        check(f"f1[{k}]", f1, rb(O.swish(emb @ P.w(f"film.{k}.fc1") + P.b(f"film.{k}.fc1"))), True)
        check(f"p[{k}]", pp, rb(f1 @ P.w(f"film.{k}.fc2") + P.b(f"film.{k}.fc2")), True)
        check(f"ss[{k}]", ss, pp @ P.w(f"film.{k}.ss") + P.b(f"film.{k}.ss"), False)
        scale, shift = (z.repeat_interleave(S, 0) for z in (ss[:, :M], ss[:, M:]))
        y, ya1, o1, ya2 = T("y", k), T("ya1", k), T("o1", k), T("ya2", k)
        check(f"ya1[{k}]", ya1, rb(O.swish(scale * P.ln(y, f"res.{k}.ln1") + shift)), True)
        check(f"o1[{k}]", o1, rb(ya1 @ P.w(f"res.{k}.fc1") + P.b(f"res.{k}.fc1")), True)
        check(f"ya2[{k}]", ya2, rb(O.swish(scale * P.ln(o1, f"res.{k}.ln2") + shift)), True)
        check(f"y[{k + 1}]", T("y", k + 1), rb(ya2 @ P.w(f"res.{k}.fc2") + P.b(f"res.{k}.fc2") + y), True)
    ao = T("ao")
    check("ao", ao, rb(P.ln(T("y", nblk), "ln_o")), True)
    check("pred", T("pred"), ao @ P.w("out_proj") + P.b("out_proj"), False)
    print(f"{arch} C={C} L={L} K={K} B={B}: worst rel (fraction of bf16 elements one ulp off) per saved tensor: "
          + ", ".join(f"{n} {e:.1e} ({f * 100:.2f} %)" for n, (e, f) in worst.items()))


@pytest.mark.parametrize("C,L,H,K,B,dtype", [(512, 6, 8, 2, 256, "bf16"), (146, 2, 16, 3, 8, "bf16"), (512, 6, 8, 2, 256, "fp8")])
def test_every_weight_gradient_from_its_saved_operands(C, L, H, K, B, dtype):
    """train_ncsn.py:282-283 "gradients must be the gradients", teacher-forced and in situ: every Dense kernel / bias gradient
    of a training step against float64 X^T dY / colsum(dY) of exactly the operands the engine's weight-gradient GEMM read (the
    saved activation X and the gradient activation dY, both bf16, each in its own workspace slot: smd_engine_debug_tensor).
    No rounding between the operands and the result but the fp32 accumulation over 8192 token rows: 1e-5 -- for the grouped
    128-wide launches with their slab reduces, the four-problem 256 x 256 launch, the bias rows of the ones-MFMA, the ragged
    in_proj / out_proj of C = 146, and the e4m3 mode (whose weight gradients stay bf16)."""
    import smd_amd.ncsn as N
    from smd_amd.engine import NetConfig
    cfg = NetConfig(architecture="TransformerDDPM", data_channels=C, seq_len=32, num_layers=L, num_heads=H, num_mlp_layers=K,
                    num_timesteps=1000, dtype=dtype)
    model = N.Model(cfg, "cuda:0", seed=2)
    eng = model.train_engine(ema=False)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(B, training=True)
    x0, g = data(B, (32, C))
    eng.grads.fill_(float("nan"))
    eng.loss_backward(x0.cuda(), None, None, seed=17, stage=0)
    torch.cuda.synchronize()
    gv = eng.named_views(eng.grads)
    T = lambda name, i=0: eng.debug_tensor(name, i).double()          # on the device: 8192 x 2048 x 2048 products in float64
    pairs = [("out_proj", T("ao"), T("dpred")[:, :C]), ("up", T("af"), T("dyb", 0)), ("in_proj", T("x_bf16")[:, :C], T("dhb", 0))]
    for k in range(K):
        pairs += [(f"res.{k}.fc2", T("ya2", k), T("dyb", k + 1)), (f"res.{k}.fc1", T("ya1", k), T("do1", k)),
                  (f"film.{k}.ss", T("p", k), T("dss_bf16", k)), (f"film.{k}.fc2", T("f1", k), T("dp", k)), (f"film.{k}.fc1", T("emb"), T("df1", k))]
    for l in range(L):
        pairs += [(f"enc.{l}.mlp.fc2", T("u", l), T("dhb", 2 * l + 2)), (f"enc.{l}.mlp.fc1", T("a2", l), T("dz1", l)),
                  (f"enc.{l}.attn.out", T("o", l), T("dhb", 2 * l + 1)), (f"enc.{l}.attn.qkv", T("a1", l), T("dqkv", l))]
    worst = (0.0, "")
    for name, X, dY in pairs:
        eW = rel(gv[name + ".kernel"], (X.t() @ dY).cpu())
        eb = rel(gv[name + ".bias"], dY.sum(0).cpu())
        worst = max(worst, (eW, name + ".kernel"), (eb, name + ".bias"))
        assert eW < 1e-5 and eb < 1e-5, f"{name}: dW rel {eW:.2e}, db rel {eb:.2e}"
    assert len(pairs) == 3 + 5 * K + 4 * L
    print(f"C={C} L={L} K={K} B={B} {dtype}: {2 * len(pairs)} weight / bias gradients vs float64 of their saved operands, worst {worst[1]} {worst[0]:.2e}")
    # ---- the two fused encoder backward kernels whose inputs AND outputs all persist, teacher-forced the same way (bf16
    # outputs: one rounding stage, flips next to ties): mlp_hs_bwd (recomputed u = gelu(z), dz = (dh W2^T) gelu'(z)) and
    # attn_block_bwd (dq | dk | dv from the saved q k v and dh_mid)
    import bf16_emulation as E
    rb = E.rb
    pv = {k: v.double() for k, v in eng.named_views(eng.params).items()}
    Wb = lambda name: rb(pv[name + ".kernel"])
    d = 128 // H
    worst_e = (0.0, "")
    for l in range(L):
        pre = f"enc.{l}"
        a2, dh_in = T("a2", l), T("dhb", 2 * l + 2)
        z = (a2 @ Wb(pre + ".mlp.fc1") + pv[pre + ".mlp.fc1.bias"]).requires_grad_(True)
        u_ref = O.gelu(z)
        (gp,) = torch.autograd.grad(u_ref.sum(), z)
        dz_ref = rb((dh_in @ Wb(pre + ".mlp.fc2").t()) * gp)
        for name, got, want in (("u", T("u", l), rb(u_ref.detach())), ("dz1", T("dz1", l), dz_ref)):
            e = rel(got, want)
            worst_e = max(worst_e, (e, f"{name}[{l}]"))
            assert e < 2e-4, f"{name}[{l}]: rel {e:.2e}"
        qkv, dh_mid = T("qkv", l), T("dhb", 2 * l + 1)
        q, k, v = (t_.reshape(B, 32, H, d) for t_ in qkv.split(128, dim=-1))
        dO = rb(dh_mid @ Wb(pre + ".attn.out").t()).reshape(B, 32, H, d)
        sc = 1.0 / d ** 0.5
        p_ = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", q, k) * sc, dim=-1)
        dP = torch.einsum("bqhd,bkhd->bhqk", dO, v)
        dS = rb(p_ * (dP - (p_ * dP).sum(-1, keepdim=True)))
        dq = rb(torch.einsum("bhqk,bkhd->bqhd", dS, k) * sc)
        dk = rb(torch.einsum("bhqk,bqhd->bkhd", dS, q) * sc)
        dv = rb(torch.einsum("bhqk,bqhd->bkhd", rb(p_), dO))
        want = torch.cat([t_.reshape(B * 32, 128) for t_ in (dq, dk, dv)], dim=1)
        e = rel(T("dqkv", l), want)
        worst_e = max(worst_e, (e, f"dqkv[{l}]"))
        assert e < 2e-4, f"dqkv[{l}]: rel {e:.2e}"
    print(f"   fused encoder backward kernels vs float64 of their saved inputs (bf16 outputs): worst {worst_e[1]} {worst_e[0]:.2e}")


@pytest.mark.parametrize("C,L,H,B", [(512, 6, 8, 256), (146, 2, 16, 8)])
def test_encoder_backward_chain_from_snapshots(C, L, H, B):
    """The rest of the encoder backward, teacher-forced: the shared per-layer gradient buffers (da2 partial tiles, dh behind
    each LayerNorm backward, dA_E) are copied out behind every kernel (smd_engine_debug_snapshots), so each kernel's output can
    be re-derived in float64 from exactly its inputs: da2 = dz W1^T (sum of the four partial tiles), the LayerNorm-2 backward
    on those tiles + the residual gradient (ln128_bwd_parts: fp32 dh and its bf16 copy), da1 = dqkv Wqkv^T (the tail of
    attn_block_bwd), the LayerNorm-1 backward (layernorm_bwd_narrow128), and the four LayerNorm parameter gradients of the layer.
    fp32 outputs 1e-5, bf16 outputs 2e-4.  Together with test_every_weight_gradient_from_its_saved_operands every kernel of the
    encoder backward is pinned in situ, at B = 256."""
    import bf16_emulation as E
    import smd_amd.lib as lib
    import smd_amd.ncsn as N
    from smd_amd.engine import NetConfig
    cfg = NetConfig(architecture="TransformerDDPM", data_channels=C, seq_len=32, num_layers=L, num_heads=H, num_mlp_layers=2, num_timesteps=1000)
    model = N.Model(cfg, "cuda:0", seed=4)
    g = torch.Generator().manual_seed(3)
    for k_, v_ in model.engine.named_views().items():           # non-trivial LayerNorm affine
        if k_.endswith(".scale"):
            v_.copy_((1 + 0.1 * torch.randn(v_.shape, generator=g)).cuda())
    model.engine.refresh_weights()
    eng = model.train_engine(ema=False)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(B, training=True)
    Lc = lib.get_lib()
    nbytes = int(Lc.smd_engine_debug_snapshot_bytes(eng.h))
    snap = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    lib.check(Lc.smd_engine_debug_snapshots(eng.h, snap.data_ptr(), nbytes))
    x0, _ = data(B, (32, C))
    try:
        eng.loss_backward(x0.cuda(), None, None, seed=23, stage=0)
        torch.cuda.synchronize()
    finally:
        lib.check(Lc.smd_engine_debug_snapshots(eng.h, None, 0))
    R, RE = B * 32, B * 32 * 128
    SEG = {"parts": (0, 16, torch.float32), "dh_ln2": (16, 4, torch.float32), "dA_E": (20, 2, torch.bfloat16), "dh_ln1": (22, 4, torch.float32),
           "dh_in": (26, 4, torch.float32), "h_mid": (30, 4, torch.float32)}

    def seg(l, name):
        off, ln, dt = SEG[name]
        slot = L - 1 - l                                          # slots are filled in execution order: last layer first
        return snap[(slot * 34 + off) * RE:(slot * 34 + off + ln) * RE].view(dt).double()

    T = lambda name, i=0: eng.debug_tensor(name, i).double()
    pv = {k: v.double() for k, v in eng.named_views(eng.params).items()}
    gv = eng.named_views(eng.grads)
    rb = E.rb

    def ln_bwd(x, dy, gamma):
        mean = x.mean(-1, keepdim=True)
        rstd = torch.rsqrt((x * x).mean(-1, keepdim=True) - mean * mean + 1e-6)
        xh = (x - mean) * rstd
        dxh = dy * gamma
        dx = rstd * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))
        return dx, (dy * xh).sum(0), dy.sum(0)

    worst = {}

    def check(name, got, want, tol):
        e = rel(got, want)
        worst[name] = max(worst.get(name, 0.0), e)
        assert e < tol, f"{name}: rel {e:.2e}"

    for l in range(L):
        pre = f"enc.{l}"
        parts = seg(l, "parts").view(4, R, 128)
        da2 = (parts[0] + parts[1]) + (parts[2] + parts[3])
        check("da2 = dz W1^T", da2, T("dz1", l) @ rb(pv[pre + ".mlp.fc1.kernel"]).t(), 1e-5)
        h_mid = T("h_mid", l)
        assert torch.equal(seg(l, "h_mid").view(R, 128), h_mid)
        dx, dg, db = ln_bwd(h_mid, da2, pv[pre + ".ln2.scale"])
        dh2 = dx + seg(l, "dh_in").view(R, 128)
        check("ln2 backward -> dh (fp32)", seg(l, "dh_ln2").view(R, 128), dh2, 1e-5)
        check("ln2 backward -> dh (bf16 copy)", T("dhb", 2 * l + 1), rb(dh2), 2e-4)
        check("ln2 dgamma", gv[pre + ".ln2.scale"], dg.cpu(), 1e-5)
        check("ln2 dbeta", gv[pre + ".ln2.bias"], db.cpu(), 1e-5)
        dA = seg(l, "dA_E").view(R, 128)
        check("da1 = dqkv Wqkv^T (bf16)", dA, rb(T("dqkv", l) @ rb(pv[pre + ".attn.qkv.kernel"]).t()), 2e-4)
        dx, dg, db = ln_bwd(T("h", l), dA, pv[pre + ".ln1.scale"])
        dh1 = dx + seg(l, "dh_ln2").view(R, 128)
        check("ln1 backward -> dh (fp32)", seg(l, "dh_ln1").view(R, 128), dh1, 1e-5)
        check("ln1 backward -> dh (bf16 copy)", T("dhb", 2 * l), rb(dh1), 2e-4)
        check("ln1 dgamma", gv[pre + ".ln1.scale"], dg.cpu(), 1e-5)
        check("ln1 dbeta", gv[pre + ".ln1.bias"], db.cpu(), 1e-5)
        if l + 1 < L:                                             # the dh that enters layer l is what layer l + 1's ln1 backward left
            assert torch.equal(seg(l, "dh_in"), seg(l + 1, "dh_ln1"))
    print(f"C={C} L={L} H={H} B={B}: encoder backward chain vs float64 of each kernel's own inputs: "
          + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))


@pytest.mark.parametrize("arch,C,L,K,B", [("TransformerDDPM", 512, 6, 2, 256), ("DenseDDPM", 512, 3, 3, 512), ("TransformerDDPM", 146, 2, 3, 8)])
def test_output_stage_backward_chain_from_snapshots(arch, C, L, K, B):
    """The output stage's backward, teacher-forced like the encoder's: the shared dX buffer is copied out behind every dgrad
    (smd_engine_debug_snapshots), so each kernel is checked in float64 on exactly its inputs -- the dgrads of out_proj and of
    both Dense of every block (bf16), LayerNorm_o backward, the two FiLM + swish LayerNorm backwards of every block with their
    residual-gradient add (bf16 dX), the per-sample dscale | dshift sums they accumulate (fp32), the LayerNorm parameter
    gradients, the FiLM generator's two dgrads (bf16, the second through swish'), and `up`'s dgrad chained into LayerNorm_f's
    backward (its dX lives in the buffer the encoder then overwrites, so the pair is checked as one).  With the encoder chain
    and the weight-gradient tests, every kernel of loss_backward is pinned in situ."""
    import bf16_emulation as E
    import smd_amd.lib as lib
    import smd_amd.ncsn as N
    from smd_amd.engine import NetConfig
    dense = arch == "DenseDDPM"
    cfg = NetConfig(architecture=arch, data_channels=C, seq_len=32, num_layers=L, num_heads=8 if C == 512 else 16,
                    num_mlp_layers=K, num_timesteps=1000)
    model = N.Model(cfg, "cuda:0", seed=6)
    g = torch.Generator().manual_seed(8)
    for k_, v_ in model.engine.named_views().items():           # non-trivial LayerNorm affine
        if k_.endswith(".scale"):
            v_.copy_((1 + 0.1 * torch.randn(v_.shape, generator=g)).cuda())
    model.engine.refresh_weights()
    eng = model.train_engine(ema=False)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(B, training=True)
    Lc = lib.get_lib()
    nbytes = int(Lc.smd_engine_debug_snapshot_bytes(eng.h))
    snap = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    lib.check(Lc.smd_engine_debug_snapshots(eng.h, snap.data_ptr(), nbytes))
    x0, _ = data(B, (C,) if dense else (32, C))
    try:
        eng.loss_backward(x0.cuda(), None, None, seed=29, stage=0)
        torch.cuda.synchronize()
    finally:
        lib.check(Lc.smd_engine_debug_snapshots(eng.h, None, 0))
    S = 1 if dense else 32
    R, M, Eh = B * S, 2048, 128
    KB = L if dense else K                                        # blocks of the output stage
    stem_bytes = L * 34 * R * Eh
    assert nbytes == stem_bytes + (2 * KB + 1) * R * M * 2

    def dA(i):
        return snap[stem_bytes + i * R * M * 2: stem_bytes + (i + 1) * R * M * 2].view(torch.bfloat16).view(R, M).double()

    T = lambda name, i=0: eng.debug_tensor(name, i).double()
    pv = {k: v.double() for k, v in eng.named_views(eng.params).items()}
    gv = eng.named_views(eng.grads)
    rb = E.rb
    worst = {}

    def check(name, got, want, tol):
        e = rel(got, want)
        worst[name] = max(worst.get(name, 0.0), e)
        assert e < tol, f"{name}: rel {e:.2e}"

    def dswish(z):
        sg = torch.sigmoid(z)
        return sg * (1 + z * (1 - sg))

    def film_ln_bwd(x, dout, gamma, beta, ss):
        """backward of swish(scale * LN(x) + shift): dx, dgamma, dbeta, dscale | dshift per sample"""
        mean = x.mean(-1, keepdim=True)
        rstd = torch.rsqrt((x * x).mean(-1, keepdim=True) - mean * mean + 1e-6)
        xh = (x - mean) * rstd
        n = xh * gamma + beta
        sc = ss[:, :M].repeat_interleave(S, 0)
        sh = ss[:, M:].repeat_interleave(S, 0)
        dpre = dout * dswish(sc * n + sh)
        dsc = (dpre * n).view(B, S, M).sum(1)
        dsh = dpre.view(B, S, M).sum(1)
        dn = dpre * sc
        dxh = dn * gamma
        dx = rstd * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))
        return dx, (dn * xh).sum(0), dn.sum(0), torch.cat([dsc, dsh], 1)

    # out_proj's dgrad and LayerNorm_o's backward
    Cp = T("dpred").shape[1]
    Wout = torch.zeros(M, Cp, dtype=torch.float64, device="cuda")
    Wout[:, :C] = rb(pv["out_proj.kernel"])
    check("d ao = d eps_hat Wout^T (bf16)", dA(0), rb(T("dpred") @ Wout.t()), 2e-4)
    x = T("y", KB)
    mean = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt((x * x).mean(-1, keepdim=True) - mean * mean + 1e-6)
    xh = (x - mean) * rstd
    dxh = dA(0) * pv["ln_o.scale"]
    dx = rstd * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))
    check("ln_o backward (bf16)", T("dyb", KB), rb(dx), 2e-4)
    check("ln_o dgamma", gv["ln_o.scale"], (dA(0) * xh).sum(0).cpu(), 1e-5)
    check("ln_o dbeta", gv["ln_o.bias"], dA(0).sum(0).cpu(), 1e-5)
    for k in range(KB - 1, -1, -1):
        pre, j = f"res.{k}", KB - 1 - k
        d2 = dA(1 + 2 * j)
        check("d ya2 = dy W2^T (bf16)", d2, rb(T("dyb", k + 1) @ rb(pv[pre + ".fc2.kernel"]).t()), 2e-4)
        dx, dg, db, dss2 = film_ln_bwd(T("o1", k), d2, pv[pre + ".ln2.scale"], pv[pre + ".ln2.bias"], T("ss", k))
        check("FiLM ln2 backward (bf16)", T("do1", k), rb(dx), 2e-4)
        check("ln2 dgamma", gv[pre + ".ln2.scale"], dg.cpu(), 1e-5)
        check("ln2 dbeta", gv[pre + ".ln2.bias"], db.cpu(), 1e-5)
        d1 = dA(2 + 2 * j)
        check("d ya1 = do1 W1^T (bf16)", d1, rb(T("do1", k) @ rb(pv[pre + ".fc1.kernel"]).t()), 2e-4)
        dx, dg, db, dss1 = film_ln_bwd(T("y", k), d1, pv[pre + ".ln1.scale"], pv[pre + ".ln1.bias"], T("ss", k))
        check("FiLM ln1 backward + residual (bf16)", T("dyb", k), rb(dx + T("dyb", k + 1)), 2e-4)
        check("ln1 dgamma", gv[pre + ".ln1.scale"], dg.cpu(), 1e-5)
        check("ln1 dbeta", gv[pre + ".ln1.bias"], db.cpu(), 1e-5)
        check("dscale | dshift (fp32, both LayerNorms)", T("dss", k), dss2 + dss1, 1e-5)
        assert torch.equal(eng.debug_tensor("dss", k).to(torch.bfloat16), eng.debug_tensor("dss_bf16", k))
        fp = f"film.{k}"
        check("dp = dss Wss^T (bf16)", T("dp", k), rb(T("dss_bf16", k) @ rb(pv[fp + ".ss.kernel"]).t()), 2e-4)
        check("df1 = (dp W2^T) swish'(zf1) (bf16)", T("df1", k), rb((T("dp", k) @ rb(pv[fp + ".fc2.kernel"]).t()) * dswish(T("zf1", k))), 2e-4)
    if not dense:
        # `up`'s dgrad and LayerNorm_f's backward as one (the dX between them is overwritten by the encoder)
        da = rb(T("dyb", 0) @ rb(pv["up.kernel"]).t())
        x = T("h_last")
        mean = x.mean(-1, keepdim=True)
        rstd = torch.rsqrt((x * x).mean(-1, keepdim=True) - mean * mean + 1e-6)
        xh = (x - mean) * rstd
        dxh = da * pv["ln_f.scale"]
        dx = rstd * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))
        dh_in = snap[26 * R * Eh: 30 * R * Eh].view(torch.float32).view(R, Eh).double()       # slot 0 = layer L-1, segment dh_in
        check("up dgrad + ln_f backward (fp32)", dh_in, dx, 1e-4)
        check("up dgrad + ln_f backward (bf16 copy)", T("dhb", 2 * L), rb(dx), 3e-4)
        check("ln_f dgamma", gv["ln_f.scale"], (da * xh).sum(0).cpu(), 1e-4)
        check("ln_f dbeta", gv["ln_f.bias"], da.sum(0).cpu(), 1e-4)
    print(f"{arch} C={C} L={L} K={K} B={B}: output-stage backward chain vs float64 of each kernel's own inputs:\n     "
          + "\n     ".join(f"{k_} {v_:.1e}" for k_, v_ in worst.items()))


def test_optimizer_step_matches_oracle():
    ocfg, p, model = make(C=42, L=2, K=1)
    B = 4
    x0, g = data(B, (32, 42))
    labels = torch.randint(1, 1001, (B,), generator=g)
    eps = torch.randn(B, 32, 42, generator=g)
    eng = model.train_engine(ema=True)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(B, training=True)
    st = O.AdamState()
    ema = {k: v.clone() for k, v in p.items()}
    cur = {k: v.clone() for k, v in p.items()}
    for step in range(3):
        eng.loss_backward(x0.cuda(), labels.int().cuda(), eps.cuda(), stage=0)
        torch.cuda.synchronize()
        grads = {k: v.double().cpu().clone() * (25.0 if step == 1 else 1.0) for k, v in eng.named_views(eng.grads).items()}
        if step == 1:
            eng.grads.mul_(25.0)                               # force the clip branch once
        lr = O.stepped_lr(1e-3, step, 2, 0.5)                  # interval 2, gamma .5 -> lr changes at step 3
        clipped, norm_after = O.clip_grads(grads, 1.0)
        cur = O.adam_update(cur, clipped, st, lr)
        ema = O.ema_update(ema, cur, 0.999)
        eng.optimizer_step(1e-3, 0.5, 2, 1.0, 0.999)
        torch.cuda.synchronize()
        m = eng.metrics.cpu()
        assert abs(float(m[1]) - float(norm_after)) / float(norm_after) < 1e-5
        assert abs(float(m[2]) - lr) / lr < 1e-6
        pv, ev = eng.named_views(eng.params), eng.named_views(eng.ema)
        num = sum(float((pv[k].double().cpu() - cur[k]).pow(2).sum()) for k in cur)
        den = sum(float((cur[k] - p[k]).pow(2).sum()) for k in cur)
        print(f"step {step}: update rel err {(num / den) ** 0.5:.3e}, |g| {float(m[0]):.4f} -> {float(m[1]):.4f}, lr {float(m[2]):.2e}")
        assert (num / den) ** 0.5 < 1e-4                       # fp32 Adam vs fp64 oracle, relative to the update size
        assert max(rel(ev[k], ema[k]) for k in ema) < 1e-6
    assert int(eng.step_counter.item()) == 3
    # the refreshed bf16 operand pack must reflect the new weights: forward parity with updated params
    x, _ = data(3, (32, 42), seed=9)
    s = torch.tensor([0.3, 0.6, 0.9]).view(3, 1, 1)
    ref = O.make_model(cur, ocfg)(x.double(), s.double())
    assert rel(model(x, s), ref) < 1e-2


@pytest.mark.parametrize("arch,C,L,K,B", [("TransformerDDPM", 512, 6, 2, 256), ("TransformerDDPM", 42, 2, 1, 4), ("TransformerDDPM", 146, 2, 3, 4),
                                          ("DenseDDPM", 42, 3, 2, 8), ("DenseDDPM", 512, 2, 2, 64)])
def test_one_sweep_optimizer_recast_and_overlap_modes(arch, C, L, K, B, monkeypatch):
    """train_ncsn.py:284-287 as ONE sweep (csrc/optim.hip adam_recast_kernel): clip + Adam + EMA and the bf16 re-cast of every
    Dense kernel in both operand layouts per 64 x 64 tile.
      (a) the operand pack the sweep leaves equals a fresh re-cast of the updated fp32 master, bit for bit (aligned tiles, the
          ragged in_proj / out_proj kernels of C = 42 / 146, DenseDDPM);
      (b) engine option "opt_overlap": 0 (everything on the caller's stream), 1 (output-stage slice of the update on the side
          stream under the next forward pass), 3 (+ that slice's norm partials reduced early on the side stream) give bitwise
          the same 4-step trajectory -- parameters, Adam moments, EMA, operand pack, metrics, losses;
      (c) another handle on the same buffers (the inference engine) called straight after a step with a deferred update sees
          the finished update (Python-side join)."""
    import smd_amd.ncsn as N
    from smd_amd.trainer import create_optimizer, train_step
    _, p, model = make(arch, C, L, 8, K)
    shape = (C,) if arch == "DenseDDPM" else (32, C)
    x0, g = data(B, shape)
    xd = x0.cuda()
    opt = create_optimizer(model, 1e-3, ema=True)
    eng = opt.engine
    start = eng.params.clone()
    xs = torch.clamp(0.25 * torch.randn(3, *shape, generator=g), -1, 1)
    ss = torch.tensor([0.3, 0.6, 0.9]).view(3, *([1] * len(shape)))

    def trajectory(mode):
        monkeypatch.setenv("SMD_OPT_OVERLAP", str(mode))
        eng.params.copy_(start)
        eng.m.zero_(); eng.v.zero_(); eng.ema.copy_(start); eng.step_counter.zero_()
        eng.refresh_weights()
        losses, mets = [], []
        for i in range(4):
            # interval 2 / gamma .5: the stepped LR changes inside the run; clip 0.5 (the norm is ~1): the clip branch is taken
            _, m = train_step(N.diffusion_loss, xd, opt, BETAS, N.PRNGKey(11), 1e-3, grad_clip=0.5, lr_gamma=0.5, lr_interval=2)
            losses.append(m["loss"].clone()); mets.append(eng.metrics.clone())
        out = model(xs, ss)                                   # (c): no synchronize between the deferred update and this call
        torch.cuda.synchronize()
        assert torch.equal(out, model(xs, ss))
        return dict(params=eng.params.clone(), m=eng.m.clone(), v=eng.v.clone(), ema=eng.ema.clone(), wpack=eng.wpack.clone(),
                    losses=torch.stack(losses), metrics=torch.stack(mets), step=int(eng.step_counter.item()))

    ref = trajectory(0)
    assert ref["step"] == 4 and not torch.equal(ref["params"], start) and bool(torch.isfinite(ref["params"]).all())
    assert float(ref["metrics"][0, 0]) > 0.5 and abs(float(ref["metrics"][0, 1]) - 0.5) < 1e-6          # clipped to 0.5
    assert abs(float(ref["metrics"][3, 2]) / float(ref["metrics"][0, 2]) - 0.5) < 1e-6                  # lr halved by step 3
    # (a)
    pack = ref["wpack"]
    eng.refresh_weights()
    torch.cuda.synchronize()
    assert torch.equal(pack, eng.wpack), "the fused re-cast differs from recast_all of the same parameters"
    for mode in (1, 3):
        got = trajectory(mode)
        for k in ref:
            same = got[k] == ref[k] if k == "step" else torch.equal(got[k], ref[k])
            assert same, f"opt_overlap={mode}: {k} differs from the single-stream trajectory"


def test_first_adam_step_is_lr_sign():
    _, p, model = make(C=42, L=2, K=1)
    eng = model.train_engine(ema=False)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(4, training=True)
    x0, g = data(4, (32, 42))
    before = eng.params.clone()
    eng.loss_backward(x0.cuda(), None, None, seed=3, stage=0)
    eng.optimizer_step(1e-3, 1.0, 1, 1e9, 0.999)
    torch.cuda.synchronize()
    delta = (eng.params - before)
    gsel = eng.grads.abs() > 1e-6
    assert torch.allclose(delta[gsel], -1e-3 * torch.sign(eng.grads[gsel]), rtol=2e-2, atol=1e-6)


@pytest.mark.parametrize("arch,C", [("TransformerDDPM", 42), ("DenseDDPM", 512)])
def test_sampler_teacher_forced_and_rollout(arch, C):
    import smd_amd.ncsn as N
    ocfg, p, model = make(arch, C, 2, 8, 1)
    B = 4
    shape = (C,) if arch == "DenseDDPM" else (32, C)
    g = torch.Generator().manual_seed(4321)
    init = torch.randn(B, *shape, generator=g)
    zs = {t: torch.randn(B, *shape, generator=g) for t in range(1000)}
    omodel = O.make_model(p, ocfg)
    # teacher-forced single steps
    for t in (999, 500, 1, 0):
        ref, _, mref = O.diffusion_dynamics(omodel, BETAS, init.double(), lambda tt: zs[tt].double(), t_start=t, t_stop=t)
        got, _, mgot = N.diffusion_dynamics(N.PRNGKey(0), model, BETAS, init, noises=lambda tt: zs[tt], t_start=t, t_stop=t)
        e = rel(got, ref)
        row = 999 - t
        print(f"teacher-forced t={t}: state rel {e:.3e}; metrics got {mgot[:, row, 0].tolist()} ref {mref[:, row, 0].tolist()}")
        assert e < 1e-2
        assert rel(mgot[:, row, 0], mref[:, row, 0]) < 1e-2
    # 30-step free-running rollout (covers the first collection hit at t=975 -> slot 2)
    ref, cref, mref = O.diffusion_dynamics(omodel, BETAS, init.double(), lambda tt: zs[tt].double(), t_stop=970)
    got, cgot, mgot = N.diffusion_dynamics(N.PRNGKey(0), model, BETAS, init, noises=lambda tt: zs[tt], t_stop=970)
    print(f"rollout 30 steps: state rel {rel(got, ref):.3e} collection[2] rel {rel(cgot[2], cref[2]):.3e}")
    assert rel(got, ref) < 2e-2
    assert torch.equal(cgot[0].cpu(), init)                    # collection[0] = init
    assert float(cgot[1].abs().max()) == 0.0                   # slot 1 is never written (reference quirk)
    assert rel(cgot[2], cref[2]) < 2e-2
    assert float(cgot[3:].abs().max()) == 0.0
    assert rel(mgot[:, :30, 0], mref[:, :30, 0]) < 1e-2
    assert tuple(cgot.shape) == (41, B, *shape) and tuple(mgot.shape) == (4, 1000, 1)


def test_sampler_graph_matches_eager_and_is_shard_invariant():
    import smd_amd.ncsn as N
    _, _, model = make(C=42, L=2, K=1)
    key = N.PRNGKey(7)
    eng = model.engine
    eng.set_schedule(BETAS)
    B = 6
    eng.bind(B, training=False)
    init = torch.empty(B, 32, 42, device="cuda")
    eng.init_state(init, 99, 0)
    a, ca, ma = N.diffusion_dynamics(key, model, BETAS, init, t_stop=960, use_graph=True)
    b, cb, mb = N.diffusion_dynamics(key, model, BETAS, init, t_stop=960, use_graph=False)
    assert torch.equal(a, b) and torch.equal(ca, cb)           # hipGraph replay == eager launches, bitwise
    # shard invariance: samples 2..5 generated alone with sample_offset=2 equal the joint run
    c, _, _ = N.diffusion_dynamics(key, model, BETAS, init[2:], t_stop=960, use_graph=False, sample_offset=2)
    assert rel(c, a[2:]) < 1e-5


@pytest.mark.parametrize("rng_impl,infill,dtype", [("philox", False, "bf16"), ("threefry", True, "bf16"), ("philox", True, "fp8")])
def test_cached_sampler_graphs_serve_later_runs(rng_impl, infill, dtype):
    """diffusion_dynamics keeps its captured step (per model: state / collection / metrics buffers, device-side timestep, Philox
    key through smd_sample_io.key_ptr, jax.random key tables) and replays it for LATER runs: another rng, another initial
    state, other infill data, and -- through the shared operand pack -- other weights must give exactly what a fresh eager
    walk gives; the first run's result must come back bit for bit when its arguments come back."""
    import smd_amd.ncsn as N
    from smd_amd.engine import NetConfig
    C, B, T_STOP = 512, 8, 985
    cfg = NetConfig(architecture="TransformerDDPM", data_channels=C, seq_len=32, num_layers=2, num_heads=8, num_mlp_layers=1,
                    num_timesteps=1000, dtype=dtype)
    model = N.Model(cfg, "cuda:0", seed=3)
    g = torch.Generator().manual_seed(5)

    def args(seed):
        init = torch.randn(B, 32, C, generator=g)
        kw = {}
        if infill:
            mask = torch.zeros(B, 32, C)
            mask[:, 8:24] = 1.0
            kw = dict(infill=True, infill_samples=torch.clamp(0.25 * torch.randn(B, 32, C, generator=g), -1, 1), infill_masks=mask)
        return N.make_key(seed, rng_impl), init, kw

    def run(a, graph):
        key, init, kw = a
        return N.diffusion_dynamics(key, model, BETAS, init, t_stop=T_STOP, use_graph=graph, **kw)

    a1, a2 = args(11), args(12)
    r1 = run(a1, True)
    graphs = [id(ch["graph"]) for ch in model._sampler_graphs["entry"]["chains"]]
    r2 = run(a2, True)                                      # replays the graphs run 1 captured
    assert [id(ch["graph"]) for ch in model._sampler_graphs["entry"]["chains"]] == graphs
    e2 = run(a2, False)
    for u, v in zip(r2, e2):
        assert torch.equal(u, v), "a later run through the cached graphs differs from the eager walk"
    assert not torch.equal(r1[0], r2[0])
    # other weights: the graphs read the shared operand pack, the FiLM tables are rebuilt per run
    model.engine.params.mul_(1.01)
    model.engine.refresh_weights()
    r3, e3 = run(a1, True), run(a1, False)
    for u, v in zip(r3, e3):
        assert torch.equal(u, v)
    assert not torch.equal(r3[0], r1[0])
    model.engine.params.div_(1.01)
    # (mul / div by 1.01 is not an exact round trip in fp32: compare against a fresh eager walk, and the cached graphs against it)
    model.engine.refresh_weights()
    r4, e4 = run(a1, True), run(a1, False)
    for u, v in zip(r4, e4):
        assert torch.equal(u, v)
    assert [id(ch["graph"]) for ch in model._sampler_graphs["entry"]["chains"]] == graphs
    # another batch size: new graphs, and the old result buffers handed out earlier are untouched (they were clones)
    keep = r4[0].clone()
    N.diffusion_dynamics(a1[0], model, BETAS, a1[1][:4], t_stop=T_STOP, use_graph=True,
                         **({k: (v[:4] if torch.is_tensor(v) else v) for k, v in a1[2].items()}))
    assert torch.equal(keep, r4[0])


def test_sample_api_full_walk_small():
    import smd_amd.ncsn as N
    _, _, model = make(C=42, L=2, K=1)
    gen, coll, met = N.sample(model, BETAS, N.PRNGKey(1), (32, 42), num_samples=4, sampling="ddpm")
    assert tuple(gen.shape) == (4, 32, 42) and tuple(coll.shape) == (41, 4, 32, 42)
    assert torch.isfinite(gen).all() and float(gen.abs().max()) <= 1.0 + 1e-5     # t=0 step clips x0 to [-1,1]
    assert len(met) == 1000 and set(met[0][0]) == {"slope", "step", "alpha", "noise"}
    hit = [k for k in range(41) if float(coll[k].abs().max()) > 0]
    assert hit == [0] + list(range(2, 41))
    with pytest.raises(ValueError):
        N.sample(model, BETAS, N.PRNGKey(1), (32, 42), num_samples=4, sampling="hmc")


@pytest.mark.parametrize("rng_impl", ["philox", "threefry"])
@pytest.mark.parametrize("infill", [False, True])
def test_two_chain_sampler_equals_one_chain_and_eager(rng_impl, infill, monkeypatch):
    """The production default walks a graphed batch of >= 128 sequences as TWO concurrent half-batch chains (own engine
    handle, stream, hipGraph, nt256_min_tiles = 128 each).  Draws are keyed by the global sample index, infill slices and
    the collection / metrics are concatenated per chain, every handle advances its own device-side t: the result must be
    the one-chain walk's and the eager walk's (bitwise where the kernel selection is the same, else within the
    GEMM-variant rounding), for Philox and jax.random streams, with and without infill."""
    import smd_amd.ncsn as N
    _, _, model = make(C=512, L=2, K=1)
    B = 128
    key = N.make_key(5, rng_impl)
    g = torch.Generator().manual_seed(77)
    init = torch.randn(B, 32, 512, generator=g)
    kw = {}
    if infill:
        mask = torch.zeros(B, 32, 512)
        mask[:, 8:24] = 1.0                                        # the middle 16 of 32 latents (sample_ncsn.py:189-243)
        kw = dict(infill=True, infill_samples=torch.clamp(0.25 * torch.randn(B, 32, 512, generator=g), -1, 1), infill_masks=mask)
    T_STOP = 972                                                    # 28 iterations: the t = 975 snapshot (slot 2) is taken

    def walk(chains, graph):
        monkeypatch.setenv("SMD_SAMPLER_CHAINS", str(chains))
        assert N._sampler_chains(model, B, graph) == (2 if (chains == 2 and graph) else 1)
        return N.diffusion_dynamics(key, model, BETAS, init, t_stop=T_STOP, use_graph=graph, **kw)

    x2, c2, m2 = walk(2, True)
    x1, c1, m1 = walk(1, True)
    xe, ce, me = walk(1, False)
    assert torch.equal(x1, xe) and torch.equal(c1, ce) and torch.equal(m1, me)      # one chain: replay == eager, bitwise
    assert float(c2[2].abs().max()) > 0 and float(c2[1].abs().max()) == 0 and float(c2[3:].abs().max()) == 0
    assert torch.equal(c2[0], c1[0])
    # the half-batch handles may select other GEMM variants (64 tiles per chain vs 128): bf16 rounding differences only
    assert rel(x2, x1) < 5e-3, rel(x2, x1)
    assert rel(c2[2], c1[2]) < 5e-3
    rows = slice(0, 1000 - T_STOP)
    assert rel(m2[:, rows, 0], m1[:, rows, 0]) < 5e-3
    assert torch.equal(m2[2], m1[2])                                                # the alpha_prod row is table data
    # every sample of the second chain really advanced (its own t counter and its own slice of the noise stream)
    assert float((x2[B // 2:] - init[B // 2:].cuda()).abs().max()) > 0
    if infill:
        m = kw["infill_masks"].cuda().bool()
        assert rel(x2[m], x1[m]) < 5e-3


def test_train_step_arbitrary_objective_through_autograd():
    """train_ncsn.py:279-283 differentiates ANY objective callable with jax.value_and_grad.  Here: a Huber and an L1 denoising
    objective written in torch against ``model(x, cond)``; ``train_step`` differentiates them through smd_amd::eps_forward_train
    (forward in the training workspace, backward = the engine's backward pass from d objective / d eps_hat).  Oracle: the same
    objective on the fp64 restatement under torch autograd.  Tolerances of SURVEY 8c for the smooth objective (loss 5e-3,
    gradient 1e-2); the L1 objective's d/d eps_hat is sign(residual), so every residual smaller than the bf16 forward error
    flips a whole +-1/N entry: measured 1.6e-2, bound 3e-2."""
    import smd_amd.ncsn as N
    from smd_amd.trainer import create_optimizer, train_step
    ocfg, p, model = make(C=42, L=2, K=1)
    B = 8
    x0, g = data(B, (32, 42))
    eps = torch.randn(B, 32, 42, generator=g)
    a = 0.05 + 0.9 * torch.rand(B, generator=g)

    def objective_of(kind):
        def objective(batch, mdl, sigmas, rng, continuous_noise, reduction):
            del sigmas, rng, continuous_noise
            dt, dev = batch.dtype, batch.device
            aa = a.to(dev, dt).view(B, 1, 1)
            xt = aa.sqrt() * batch + (1 - aa).sqrt() * eps.to(dev, dt)
            r = eps.to(dev, dt) - mdl(xt, aa.sqrt())
            per = (r.abs() if kind == "l1" else torch.nn.functional.huber_loss(r, torch.zeros_like(r), reduction="none", delta=0.5)).mean(dim=(1, 2))
            return per.mean() if reduction == "mean" else per.sum()
        return objective

    opt = create_optimizer(model, 1e-3, ema=False)
    for kind, tol in (("huber", 1e-2), ("l1", 3e-2)):
        model.engine.load_named(p)                                                 # both objectives from the same parameters
        obj = objective_of(kind)
        leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        want = obj(x0.double(), O.make_model(leaf, ocfg), None, None, True, "mean")
        want.backward()
        before = model.engine.params.clone()
        _, m = train_step(obj, x0, opt, BETAS, N.PRNGKey(0), 1e-3, grad_clip=1e9)
        m = m.resolve()
        gv = opt.engine.named_views(opt.engine.grads)
        num = sum(float((gv[k].double().cpu() - leaf[k].grad).pow(2).sum()) for k in leaf)
        den = sum(float(leaf[k].grad.pow(2).sum()) for k in leaf)
        print(f"{kind} objective through autograd: loss {m['loss']:.6f} vs {float(want):.6f}; gradient rel {(num / den) ** 0.5:.3e}; |g| {m['grad']:.4f}")
        assert abs(m["loss"] - float(want)) / float(want) < 5e-3
        assert (num / den) ** 0.5 < tol
        assert abs(m["grad"] - den ** 0.5) / den ** 0.5 < tol                     # metric 'grad' = norm after (no) clipping
        assert not torch.equal(before, model.engine.params)                        # the Adam step was applied
    # the fused objective still takes the fused path, and a second model call per objective is refused loudly
    _, m2 = train_step(N.diffusion_loss, x0, opt, BETAS, N.PRNGKey(0), 1e-3)
    assert np.isfinite(m2.resolve()["loss"])

    def twice(batch, mdl, sigmas, rng, continuous_noise, reduction):
        s = torch.ones(B, 1, 1, device=batch.device)
        return (mdl(batch, s) - mdl(batch * 0.5, s)).pow(2).mean()

    with pytest.raises(RuntimeError, match="another training forward"):
        train_step(twice, x0, opt, BETAS, N.PRNGKey(0), 1e-3)


@pytest.mark.parametrize("arch,unroll", [("TransformerDDPM", 1), ("TransformerDDPM", 4), ("TransformerDDPM", 8), ("DenseDDPM", 4)])
def test_pipelined_two_chain_walk_is_the_free_running_walk_bitwise(arch, unroll, monkeypatch):
    """The default two-chain walk is software-pipelined (chain A: output stage + reverse update of iteration k, stem of k + 1;
    chain B: stem, output stage of k; `unroll` iterations per captured graph, a cross-chain event per replay; remainder
    iterations as plain launches: smd_engine_sample_step_part).  Per chain it launches exactly the kernels of the one-step
    graphs in the same order, so state, collection and metrics must be BITWISE those of the free-running arrangement
    (SMD_SAMPLER_PIPELINE=0), for walk lengths that do and do not fill the last graph."""
    import smd_amd.ncsn as N
    _, _, model = make(arch, C=512, L=2, K=1)
    B = 128 if arch == "TransformerDDPM" else 512                  # two chains need 256-row multiples per chain (DenseDDPM: S = 1)
    shape = (32, 512) if arch == "TransformerDDPM" else (512,)
    init = torch.randn(B, *shape, generator=torch.Generator().manual_seed(3))
    key = N.PRNGKey(21)
    assert N._sampler_chains(model, B, True) == 2
    for t_stop in (1000 - 1 - 2 * unroll, 1000 - 27):             # 1 + 2 * unroll iterations (fills the graphs), 27 (a remainder)
        monkeypatch.setenv("SMD_SAMPLER_PIPELINE", "0")
        ref = N.diffusion_dynamics(key, model, BETAS, init, t_stop=t_stop)
        monkeypatch.setenv("SMD_SAMPLER_PIPELINE", "1")
        monkeypatch.setenv("SMD_SAMPLER_UNROLL", str(unroll))
        assert N._sampler_pipeline_unroll() == unroll
        got = N.diffusion_dynamics(key, model, BETAS, init, t_stop=t_stop)
        again = N.diffusion_dynamics(key, model, BETAS, init, t_stop=t_stop)      # through the cached graphs
        for u, v, w_ in zip(ref, got, again):
            assert torch.equal(u, v) and torch.equal(u, w_)
        assert model._sampler_graphs["entry"]["key"][4] == unroll


@pytest.mark.parametrize("arch,C,K,dtype", [("TransformerDDPM", 512, 2, "bf16"), ("TransformerDDPM", 146, 3, "bf16"), ("DenseDDPM", 512, 2, "bf16"),
                                             ("TransformerDDPM", 512, 2, "fp8")])
def test_sample_step_parts_compose_to_the_whole_step_bitwise(arch, C, K, dtype):
    """smd_engine_sample_step_part: part 1 (the stem, plus the first `sample_split` LayerNorm + Dense half-blocks of the output stage)
    followed by part 2 (the rest + the fused reverse update) IS smd_engine_sample_step, for every split point, on every
    architecture and in fp8 mode: same kernels, same order, same buffers."""
    import smd_amd.lib as lib
    import smd_amd.ncsn as N
    from smd_amd.engine import NetConfig
    B = 8 if arch == "TransformerDDPM" else 256
    cfg = NetConfig(architecture=arch, data_channels=C, seq_len=32, num_layers=2, num_mlp_layers=K, num_timesteps=1000, dtype=dtype)
    if dtype == "fp8":
        B = 8 * 8                                                  # the e4m3 GEMMs want 256-row multiples
    model = N.Model(cfg, "cuda:0", seed=5)
    eng = model.engine
    eng.set_schedule(BETAS, with_sampler=True)
    eng.bind(B, training=False)
    eng.prepare_sampler()
    shape = (B, C) if arch == "DenseDDPM" else (B, 32, C)
    x = torch.empty(*shape, device="cuda")
    t_ptr = torch.zeros(1, dtype=torch.int32, device="cuda")
    met = torch.zeros(1000, B, 3, device="cuda")
    io = lib.SampleIO()
    io.x, io.t_ptr, io.metrics_partial = x.data_ptr(), t_ptr.data_ptr(), met.data_ptr()
    io.seed_lo, io.seed_hi, io.sample_offset = 11, 0, 0

    def walk(split, steps=3):
        eng.set_option("sample_split", max(split, 0))
        eng.init_state(x, 77, 0)
        t_ptr.fill_(999)
        met.zero_()
        for _ in range(steps):
            if split < 0:
                eng.sample_step(io)
            else:
                eng.sample_step(io, 1)
                eng.sample_step(io, 2)
        torch.cuda.synchronize()
        return x.clone(), met.clone(), int(t_ptr.item())

    ref = walk(-1)
    assert ref[2] == 996 and float(ref[0].abs().max()) > 0
    for split in range(0, 2 * K + 1):                              # 2 K = the whole output stage but its final norm + Dense in part 1
        got = walk(split)
        assert got[2] == 996 and torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), f"split {split}"
    with pytest.raises(ValueError):
        eng.sample_step(io, 3)


@pytest.mark.parametrize("C,L,H,K,B", [(512, 6, 8, 2, 256), (146, 2, 16, 3, 8), (512, 8, 16, 3, 128)])
def test_layernorm_backwards_inside_the_attention_backward_launch(C, L, H, K, B):
    """Engine option fused_attn_bwd = 2 (default since round 6): per encoder layer ONE launch runs LayerNorm-2 backward, the
    attention backward and LayerNorm-1 backward (csrc/encoder_fused.hip attn_block_bwd_kernel<d, true>) instead of three.  Against
    the three-launch path (= 1; what test_encoder_backward_chain_from_snapshots checks kernel by kernel against fp64): the loss is
    bitwise the same and the two gradients differ by fp32 summation order in the LayerNorm-1 backward (the stand-alone kernel adds its
    rows in another order) AMPLIFIED by bf16 storage: a 1e-7 difference in dh flips the rounding of about one element in 10^4 of its
    bf16 copy, a whole bf16 step, and every layer below inherits it -- measured 3.4e-5 on the whole gradient, 1.3e-3 on in_proj's (the end
    of the chain) at L = 6, the same mechanism and size as the inference-vs-training-path difference documented at
    test_forward_against_the_bf16_emulating_oracle.  Bounds: 1e-4 / 5e-3.  The fused path is bitwise repeatable."""
    ocfg, p, model = make("TransformerDDPM", C, L, H, K)
    x0, g = data(B, (32, C))
    labels = torch.randint(1, 1001, (B,), generator=g)
    eps = torch.randn(B, 32, C, generator=g)
    eng = model.train_engine(ema=False)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(B, training=True)
    out = {}
    for mode in (2, 1, 2):
        eng.set_option("fused_attn_bwd", mode)
        eng.grads.fill_(float("nan"))
        eng.loss_backward(x0.cuda(), labels.int().cuda(), eps.cuda(), stage=0)
        torch.cuda.synchronize()
        out.setdefault(mode, []).append((eng.grads.clone(), eng.loss_per_sample().clone()))
    (g2, l2), (g2b, l2b) = out[2]
    (g1, l1), = out[1]
    assert bool(torch.isfinite(g2).all())
    assert torch.equal(g2, g2b) and torch.equal(l2, l2b)                       # repeatable
    assert torch.equal(l2, l1)
    v2, v1 = eng.named_views(g2), eng.named_views(g1)
    worst = max((rel(v2[k], v1[k]), k) for k in v1 if float(v1[k].norm()) > 0)
    print(f"fused LayerNorm backwards C={C} L={L} B={B}: whole gradient rel {rel(g2, g1):.2e}, worst tensor {worst[1]} {worst[0]:.2e}")
    assert rel(g2, g1) < 1e-4 and worst[0] < 5e-3
    assert not torch.equal(g2, g1) or L == 0                                     # the other path really ran
    eng.set_option("fused_attn_bwd", 2)
