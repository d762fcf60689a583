"""The loss-side inputs of diffusion_loss (utils/losses.py:271-296) through the C-ABI: q-sample on shapes whose S*C is not a
multiple of 4 (DenseDDPM on sliced latents), --nocontinuous_noise (labels in [0, T), a real uniform noise level for label 0),
and the stand-alone smd_q_sample / smd_mse_fwd_bwd / smd_adam_clip_ema entries against the oracle."""
import numpy as np
import pytest
import torch

import ddpm_oracle as O

pytestmark = pytest.mark.gpu

T = 1000
BETAS = O.create_noise_schedule(1e-6, 0.01, T, "linear")
APE = np.concatenate([np.ones(1, np.float32), O.alphas_cumprod(BETAS)]).astype(np.float32)


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def L():
    import smd_amd.lib as lib
    return lib.get_lib()


def ck(rc):
    import smd_amd.lib as lib
    lib.check(rc)


P = lambda t: None if t is None else t.data_ptr()
st = lambda: torch.cuda.current_stream().cuda_stream


def q_sample(L, x0, labels=None, label_min=1, used_alphas=None, eps=None, seed=(7, 9), offset=0, Cp=None):
    B = x0.shape[0]
    S, C = (1, x0.shape[1]) if x0.dim() == 2 else x0.shape[1:]
    Cp = Cp or -(-C // 64) * 64
    ape = torch.from_numpy(APE).cuda()
    xt = torch.full((B * S, Cp), 768.0, dtype=torch.bfloat16, device="cuda")
    eo = torch.zeros(B, S, C, device="cuda")
    so = torch.zeros(B, device="cuda")
    keep = [x0.cuda(), None if labels is None else labels.int().cuda(), None if used_alphas is None else used_alphas.cuda(),
            None if eps is None else eps.cuda()]
    ck(L.smd_q_sample(P(keep[0]), B, S, C, Cp, T, P(ape), P(keep[1]), label_min, P(keep[2]), P(keep[3]), seed[0], seed[1], None,
                      offset, P(xt), P(eo), P(so), st()))
    torch.cuda.synchronize()
    return xt.float().cpu().view(B, S, Cp), eo.cpu(), so.cpu()


@pytest.mark.parametrize("shape", [(42,), (146,), (3, 42), (32, 42), (1,), (32, 512)])
def test_q_sample_matches_the_oracle_on_any_row_length(L, shape):
    g = torch.Generator().manual_seed(sum(shape))
    B = 9
    x0 = torch.clamp(0.25 * torch.randn(B, *shape, generator=g), -1, 1)
    eps = torch.randn(B, *shape, generator=g)
    labels = torch.randint(1, T + 1, (B,), generator=g)
    xt, eo, so = q_sample(L, x0, labels, eps=eps)
    S, C = (1, shape[0]) if len(shape) == 1 else shape
    a = torch.from_numpy(O.used_alphas_from_labels(BETAS, labels.numpy())).view(B, *([1] * len(shape)))
    want = torch.sqrt(a) * x0 + torch.sqrt(1 - a) * eps                                    # utils/losses.py:295-296
    assert torch.equal(eo.view(B, *shape), eps)
    assert torch.allclose(so, torch.sqrt(a).flatten(), rtol=1e-6)
    got = xt[..., :C].reshape(B, *shape)
    assert torch.equal(got, want.to(torch.bfloat16).float()) or rel(got, want) < 3e-3      # bf16 network input
    assert (xt[..., C:] == 768.0).all()                                                     # padding is the caller's


def test_philox_draws_do_not_depend_on_the_row_length(L):
    """The scalar-tail variants use the same groups of four normals: sample b of a 42-wide draw is the first 42 elements
    of sample b of a 44-wide draw (smd_rng_normal and the eps of smd_q_sample)."""
    B = 5
    outs = {}
    for n in (42, 44):
        o = torch.zeros(B, n, device="cuda")
        ck(L.smd_rng_normal(P(o), B, n, 11, 13, 3, 2, st()))
        torch.cuda.synchronize()
        outs[n] = o.cpu()
        assert abs(float(o.mean())) < 0.3 and 0.7 < float(o.std()) < 1.3
    assert torch.equal(outs[42], outs[44][:, :42])
    x42, x44 = torch.zeros(B, 42), torch.zeros(B, 44)
    _, e42, s42 = q_sample(L, x42, Cp=64)
    _, e44, s44 = q_sample(L, x44, Cp=64)
    assert torch.equal(e42[:, 0], e44[:, 0, :42]) and torch.equal(s42, s44)
    assert float(e42.abs().min()) > 0                                                       # every element was written


def test_philox_labels_cover_the_flag_dependent_range(L):
    """continuous_noise: randint(1, T+1) -> noise level sqrt(alphas_prod'[l-1]) in the table; --nocontinuous_noise:
    randint(0, T), label 0 -> sqrt of a uniform draw in [alphas_prod[T], 1) that is (almost surely) not in the table."""
    B = 200_000
    x0 = torch.zeros(B, 4)
    table = torch.from_numpy(np.sqrt(APE.astype(np.float32)))
    for label_min in (1, 0):
        _, _, s = q_sample(L, x0, label_min=label_min, Cp=64)
        idx = torch.bucketize(-s, -table).clamp(0, T)                                      # the table is decreasing
        near = torch.minimum((table[idx] - s).abs(), (table[(idx - 1).clamp(0)] - s).abs())
        on_table = near < 1e-7
        if label_min == 1:
            assert on_table.all()
            assert float(s.max()) == 1.0 and abs(float(s.min()) - float(table[T - 1])) < 1e-6    # labels 1 and T both occur
        else:
            off = ~on_table
            frac = float(off.float().mean())
            print(f"label_min=0: {int(off.sum())} of {B} samples drew label 0 ({frac * T:.2f} x the expected 1/T)")
            assert 0.5 / T < frac < 2.0 / T
            assert float(s[off].min()) >= float(table[T]) - 1e-6 and float(s[off].max()) < 1.0
            # label T is not drawn any more: the smallest table level present is alphas_prod'[T-2]
            assert abs(float(s[on_table].min()) - float(table[T - 2])) < 1e-6


def _model(C=42, arch="DenseDDPM", Lyr=2):
    import smd_amd.ncsn as N
    from smd_amd.engine import NetConfig
    ocfg = O.NetConfig(architecture=arch, data_channels=C, num_layers=Lyr, num_heads=8, num_mlp_layers=1, mlp_dims=2048)
    p = O.init_params(ocfg, 0, torch.float64)
    g = torch.Generator().manual_seed(1)
    for k in p:
        if k.endswith(".bias"):
            p[k] = 0.1 * torch.randn(p[k].shape, generator=g, dtype=torch.float64)
    cfg = NetConfig(architecture=arch, data_channels=C, seq_len=32, num_layers=Lyr, num_heads=8, num_mlp_layers=1,
                    mlp_dims=2048, num_timesteps=T)
    model = N.Model(cfg, "cuda:0", seed=None)
    model.engine.load_named(p)
    return ocfg, p, model


@pytest.mark.parametrize("C", [42, 146])
def test_dense_ddpm_on_sliced_latents_trains_and_samples(C):
    """configs/ddpm-mel-1seq-512.cfg with --slice_ckpt: DenseDDPM on (C,) vectors, C = 42 / 146 (S*C % 4 != 0)."""
    import smd_amd.ncsn as N
    ocfg, p, model = _model(C)
    B = 64
    g = torch.Generator().manual_seed(2)
    x0 = torch.clamp(0.25 * torch.randn(B, C, generator=g), -1, 1)
    labels = torch.randint(1, T + 1, (B,), generator=g)
    eps = torch.randn(B, C, generator=g)
    leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss_ref = O.diffusion_loss(x0.double(), O.make_model(leaf, ocfg), BETAS, labels.numpy(), eps.double(), "none")
    loss_ref.mean().backward()
    eng = model.train_engine(ema=False)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(B, training=True)
    eng.loss_backward(x0.cuda(), labels.int().cuda(), eps.cuda(), stage=0)
    torch.cuda.synchronize()
    assert abs(float(eng.loss_per_sample().mean()) - float(loss_ref.mean())) / float(loss_ref.mean()) < 5e-3
    gv = eng.named_views(eng.grads)
    num = sum(float((gv[k].double().cpu() - v.grad).pow(2).sum()) for k, v in leaf.items())
    den = sum(float(v.grad.pow(2).sum()) for v in leaf.values())
    print(f"DenseDDPM C={C}: loss {float(eng.loss_per_sample().mean()):.5f} vs {float(loss_ref.mean()):.5f}, "
          f"gradient rel {(num / den) ** 0.5:.3e}")
    assert (num / den) ** 0.5 < 1e-2
    eng.loss_backward(x0.cuda(), None, None, seed=5, stage=0)                                # on-device draws
    torch.cuda.synchronize()
    assert torch.isfinite(eng.loss_per_sample()).all()
    # Philox-initialised reverse walk (the last 5 steps) runs on this row length too
    out, coll, _ = N.diffusion_dynamics(N.PRNGKey(3), model, BETAS, torch.zeros(4, C), t_start=4, use_graph=False)
    init = torch.zeros(4, C, device="cuda")
    model.engine.bind(4, training=False)
    model.engine.init_state(init, seed=3)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and float(init.abs().min()) > 0 and 0.7 < float(init.std()) < 1.3


def test_nocontinuous_noise_label_zero_parity():
    """--nocontinuous_noise: explicit labels (with zeros) + explicit uniform draws against the oracle, then the reference's
    own jax.random streams (labels, eps and the label-0 uniforms) from a key whose batch really contains a label 0."""
    import smd_amd.ncsn as N
    from smd_amd.jax_random import ThreefryKey
    ocfg, p, model = _model(42, "TransformerDDPM")
    B = 8
    g = torch.Generator().manual_seed(4)
    x0 = torch.clamp(0.25 * torch.randn(B, 32, 42, generator=g), -1, 1)
    labels = torch.randint(0, T, (B,), generator=g)
    labels[1] = labels[5] = 0
    eps = torch.randn(B, 32, 42, generator=g)
    u = torch.rand(B, generator=g)
    om = O.make_model(p, ocfg)
    with torch.no_grad():
        ref = O.diffusion_loss(x0.double(), om, BETAS, labels.numpy(), eps.double(), "none", u01=u.numpy())
    ua = torch.from_numpy(O.used_alphas_from_labels(BETAS, labels.numpy(), u.numpy()))
    got = N.diffusion_loss(x0, model, BETAS, N.PRNGKey(0), False, "none", labels=labels, eps=eps, used_alphas=ua)
    print(f"explicit draws: per-sample loss rel {rel(got, ref):.3e}; label-0 alphas {ua[1]:.4f} {ua[5]:.4f}")
    assert rel(got, ref) < 5e-3
    # without the explicit alphas, a label 0 still takes a noise level in [alphas_prod[T], 1) (Philox uniform)
    eng = model.train_engine(ema=False)
    eng.loss_backward(x0.cuda(), labels.int().cuda(), eps.cuda(), stage=3, continuous_noise=False)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.loss_per_sample()).all()

    seed = next(s for s in range(5000) if (O.jax_diffusion_loss_draws(O.jax_prngkey(s), (B, 1), T, False)[0] == 0).any())
    key = O.jax_prngkey(seed)
    lab_j, eps_j = O.jax_diffusion_loss_draws(key, (B, 32, 42), T, continuous_noise=False)
    u_j = O.jax_diffusion_loss_u01(key, B)
    with torch.no_grad():
        ref_j = O.diffusion_loss(x0.double(), om, BETAS, lab_j, torch.from_numpy(eps_j).double(), "none", u01=u_j)
    got_j = N.diffusion_loss(x0, model, BETAS, ThreefryKey(int(key[0]), int(key[1])), False, "none")
    print(f"threefry key PRNGKey({seed}): labels {lab_j.tolist()}, per-sample loss rel {rel(got_j, ref_j):.3e}")
    assert (lab_j == 0).any() and rel(got_j, ref_j) < 5e-3
    # and the flag's default keeps the old stream: labels in [1, T]
    lab_c, _ = O.jax_diffusion_loss_draws(key, (B, 32, 42), T, continuous_noise=True)
    assert lab_c.min() >= 1


def test_mse_and_adam_entries_match_the_oracle(L):
    import smd_amd.lib as lib
    g = torch.Generator().manual_seed(6)
    B, S, C, Cp = 5, 3, 42, 64
    pred, eps = torch.randn(B, S, C, generator=g), torch.randn(B, S, C, generator=g)
    pd, ed = pred.cuda(), eps.cuda()
    loss = torch.zeros(B, device="cuda")
    dp = torch.zeros(B * S, Cp, dtype=torch.bfloat16, device="cuda")
    inv = 1.0 / (B * S * C)
    ck(L.smd_mse_fwd_bwd(P(pd), P(ed), B, S, C, Cp, inv, P(loss), P(dp), st()))
    torch.cuda.synchronize()
    assert torch.allclose(loss.cpu(), ((eps - pred) ** 2).mean(dim=(1, 2)), rtol=1e-5)      # utils/losses.py:304-305
    assert rel(dp.float().cpu().view(B, S, Cp)[..., :C], 2 * (pred - eps) * inv) < 3e-3

    n = 100_003
    w = torch.randn(n, generator=g, dtype=torch.float64)
    gr = 0.01 * torch.randn(n, generator=g, dtype=torch.float64)
    state = O.AdamState()
    ema = {"w": w.clone()}
    wd, gd = w.float().cuda(), gr.float().cuda()
    m, v, e = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), w.float().cuda()
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    scratch, metrics = torch.zeros(1024, device="cuda"), torch.zeros(4, device="cuda")
    h = lib.TrainHyper(1e-3, 0.98, 10000, 0.9, 0.999, 1e-8, 1.0, 0.999, 1.0)
    import ctypes
    params = {"w": w.clone()}
    for it in range(3):
        ck(L.smd_adam_clip_ema(P(wd), P(gd), P(m), P(v), P(e), n, ctypes.byref(h), P(step), P(scratch), P(metrics), st()))
        clipped, norm = O.clip_grads({"w": gr}, 1.0)
        params = O.adam_update(params, clipped, state, O.stepped_lr(1e-3, it, 10000, 0.98))
        ema = O.ema_update(ema, params, 0.999)
    torch.cuda.synchronize()
    print(f"adam_clip_ema x3: params rel {rel(wd, params['w']):.2e}, ema rel {rel(e, ema['w']):.2e}, "
          f"norm {float(metrics[0]):.5f} vs {float(torch.sqrt((gr * gr).sum())):.5f}")
    assert rel(wd, params["w"]) < 1e-6 and rel(e, ema["w"]) < 1e-6 and int(step) == 3
    assert abs(float(metrics[0]) - float(torch.sqrt((gr * gr).sum()))) < 1e-4


def test_torch_custom_ops_run_the_c_abi(L):
    """torch.ops.smd_amd.*: the same kernels reached as PyTorch custom ops (device tensors, current stream)."""
    import smd_amd.ops as ops
    ocfg, p, model = _model(42, "TransformerDDPM")
    g = torch.Generator().manual_seed(12)
    x = torch.clamp(0.25 * torch.randn(4, 32, 42, generator=g), -1, 1).cuda()
    s = (0.1 + 0.9 * torch.rand(4, 1, 1, generator=g)).cuda()
    with torch.no_grad():
        ref = O.make_model(p, ocfg)(x.double().cpu(), s.double().cpu())
    out = torch.ops.smd_amd.eps_forward(x, s, ops.register_engine(model.engine))
    assert rel(out, ref) < 1e-2 and torch.equal(out, model(x, s))            # Model.__call__ is this op
    a = torch.randn(256, 128, generator=g).to(torch.bfloat16).cuda()
    bt = (0.05 * torch.randn(384, 128, generator=g)).to(torch.bfloat16).cuda()
    bias = torch.randn(384, generator=g).cuda()
    c = torch.ops.smd_amd.gemm_bf16_nt(a, bt, bias)
    assert rel(c.float(), a.double() @ bt.double().t() + bias.double()) < 4e-3
    labels = torch.randint(1, T + 1, (4,), generator=g).int().cuda()
    eps = torch.randn(4, 32, 42, generator=g).cuda()
    xt, lvl = torch.ops.smd_amd.q_sample(x, torch.from_numpy(APE).cuda(), labels, eps)
    al = torch.from_numpy(O.used_alphas_from_labels(BETAS, labels.cpu().numpy())).cuda().view(4, 1, 1)
    assert rel(xt.float().view(4, 32, 42), torch.sqrt(al) * x + torch.sqrt(1 - al) * eps) < 3e-3
    assert torch.allclose(lvl, torch.sqrt(al).flatten(), rtol=1e-6)
    coef = torch.from_numpy(O_coef_table()).cuda()
    t = torch.tensor([500], dtype=torch.int32, device="cuda")
    z = torch.randn(4, 32, 42, generator=g).cuda()
    x2 = x.clone()
    torch.ops.smd_amd.ddpm_reverse_step_(x2, out, coef, t, z)
    torch.cuda.synchronize()
    co = O.reverse_coefficients(BETAS)
    recon = torch.clamp(float(co["sqrt_recip"][500]) * x - float(co["sqrt_m1"][500]) * out, -1, 1)
    want = float(co["mu1"][500]) * recon + float(co["mu2"][500]) * x + float(co["sigma"][500]) * z
    assert rel(x2, want) < 1e-5
    with pytest.raises(RuntimeError, match="GPU only"):
        torch.ops.smd_amd.eps_forward(x.cpu(), s.cpu(), ops.register_engine(model.engine))


def O_coef_table():
    import smd_amd.schedule as S
    return S.reverse_coefficient_table(BETAS)
