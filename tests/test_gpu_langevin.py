"""The NCSN path around the same network (SURVEY section 8 "next" row): denoising score matching
(utils/losses.py:129-179) and the annealed / consistent Langevin samplers (utils/ebm_utils.py:89-271) on the HIP engine
against the oracle restatements, with explicit draws and with the reference's own jax.random streams."""
import numpy as np
import pytest
import torch

import ddpm_oracle as O

pytestmark = pytest.mark.gpu

SIGMAS = O.create_noise_schedule(1.0, 0.01, 10, "geometric").astype(np.float32)


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def make(arch="TransformerDDPM", C=42, L=2, T=10, seed=0):
    import smd_amd.ncsn as N
    from smd_amd.engine import NetConfig
    ocfg = O.NetConfig(architecture=arch, data_channels=C, num_layers=L, num_heads=8, num_mlp_layers=1, mlp_dims=2048)
    p = O.init_params(ocfg, seed, torch.float64)
    g = torch.Generator().manual_seed(seed + 1)
    for k in p:
        if k.endswith(".bias"):
            p[k] = 0.1 * torch.randn(p[k].shape, generator=g, dtype=torch.float64)
    cfg = NetConfig(architecture=arch, data_channels=C, seq_len=32, num_layers=L, num_heads=8, num_mlp_layers=1,
                    mlp_dims=2048, num_timesteps=T)
    model = N.Model(cfg, "cuda:0", seed=None)
    model.engine.load_named(p)
    return ocfg, p, model


@pytest.mark.parametrize("arch,cn", [("TransformerDDPM", False), ("TransformerDDPM", True), ("DenseDDPM", False)])
def test_dsm_loss_and_gradient_parity(arch, cn):
    import smd_amd.ncsn as N
    from smd_amd.trainer import create_optimizer, train_step
    ocfg, p, model = make(arch)
    B = 64 if arch == "DenseDDPM" else 4
    shape = (42,) if arch == "DenseDDPM" else (32, 42)
    g = torch.Generator().manual_seed(3)
    x0 = torch.clamp(0.25 * torch.randn(B, *shape, generator=g), -1, 1)
    labels = torch.randint(int(cn), 10, (B,), generator=g)
    labels[0], labels[1] = 9, int(cn)                                        # smallest sigma (largest 1/sigma) and the first
    eps = torch.randn(B, *shape, generator=g)
    leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref = O.denoising_score_matching_loss(x0.double(), O.make_model(leaf, ocfg), SIGMAS, labels.numpy(), eps.double(), cn, "none")
    ref.mean().backward()
    got = N.denoising_score_matching_loss(x0, model, SIGMAS, N.PRNGKey(0), cn, "none", labels=labels, eps=eps)
    print(f"dsm {arch} continuous_noise={cn}: per-sample loss rel {rel(got, ref):.3e}  (mean {float(got.mean()):.4f} vs {float(ref.mean()):.4f})")
    assert rel(got, ref) < 1e-2
    opt = create_optimizer(model, 1e-3, ema=False)
    eng = opt.engine
    before = eng.params.clone()
    _, metrics = train_step(N.denoising_score_matching_loss, x0, opt, SIGMAS, N.PRNGKey(0), 1e-3, labels=labels, eps=eps,
                            continuous_noise=cn, grad_clip=1e9)
    torch.cuda.synchronize()
    gv = eng.named_views(eng.grads)
    num = sum(float((gv[k].double().cpu() - v.grad).pow(2).sum()) for k, v in leaf.items())
    den = sum(float(v.grad.pow(2).sum()) for v in leaf.values())
    print(f"   gradient whole-vector rel {(num / den) ** 0.5:.3e}; train_step loss {float(metrics['loss']):.4f}")
    assert (num / den) ** 0.5 < 1e-2
    assert abs(float(metrics["loss"]) - float(ref.mean())) / float(ref.mean()) < 5e-3
    assert not torch.equal(eng.params, before)                              # Adam moved the weights


@pytest.mark.parametrize("cn", [False, True])
def test_dsm_draws_are_the_reference_streams(cn):
    import smd_amd.ncsn as N
    from smd_amd.jax_random import ThreefryKey
    ocfg, p, model = make()
    B = 6
    g = torch.Generator().manual_seed(5)
    x0 = torch.clamp(0.25 * torch.randn(B, 32, 42, generator=g), -1, 1)
    key = O.jax_prngkey(21)
    labels, used, eps = O.jax_dsm_loss_draws(key, (B, 32, 42), SIGMAS, cn)
    assert labels.min() >= int(cn) and labels.max() <= 9
    with torch.no_grad():
        ref = O.denoising_score_matching_loss(x0.double(), O.make_model(p, ocfg), SIGMAS, labels, torch.from_numpy(eps).double(),
                                              cn, "none")
    got = N.denoising_score_matching_loss(x0, model, SIGMAS, ThreefryKey(int(key[0]), int(key[1])), cn, "none")
    print(f"dsm threefry continuous_noise={cn}: labels {labels.tolist()} loss rel {rel(got, ref):.3e}")
    assert rel(got, ref) < 1e-2
    a = N.denoising_score_matching_loss(x0, model, SIGMAS, N.PRNGKey(4), cn, "none")          # engine key: repeatable
    b = N.denoising_score_matching_loss(x0, model, SIGMAS, N.PRNGKey(4), cn, "none")
    assert torch.equal(a, b) and torch.isfinite(a).all()


def _normals(shape, n, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*shape, generator=g) for _ in range(n)]


@pytest.mark.parametrize("arch,L,T,infill,denoise", [("TransformerDDPM", 3, 4, False, True), ("TransformerDDPM", 3, 4, True, False),
                                                     ("DenseDDPM", 2, 50, False, True)])
def test_annealed_langevin_teacher_forced(arch, L, T, infill, denoise):
    """Same explicit normals into the oracle and the engine: state, the 100-slot collection (with upstream's index
    arithmetic: repeated linspace entries add up when L*T < 100) and the four metric rows."""
    import smd_amd.ncsn as N
    ocfg, p, model = make(arch)
    sig = O.create_noise_schedule(1.0, 0.05, L, "geometric").astype(np.float32)
    B = 3
    shape = (B, 42) if arch == "DenseDDPM" else (B, 32, 42)
    g = torch.Generator().manual_seed(8)
    init = (torch.rand(*shape, generator=g) * 2 - 1) * (12 ** 0.5 / 2)
    zs, izs = _normals(shape, L * T, 1), _normals(shape, L * T, 2)
    inf_s = torch.clamp(0.25 * torch.randn(*shape, generator=g), -1, 1)
    inf_m = torch.zeros(*shape)
    if infill:
        inf_m[:, 8:24] = 1.0
    eps = 2e-5 if L * T < 100 else 1e-4
    with torch.no_grad():
        rs, rc, rm = O.annealed_langevin_dynamics(O.make_model(p, ocfg), sig, init.double(), eps, T, denoise,
                                                  lambda si, i: zs[si * T + i].double(), infill, inf_s.double(), inf_m.double(),
                                                  lambda si, i: izs[si * T + i].double())
    gs, gc, gm = N.annealed_langevin_dynamics(N.PRNGKey(0), model, sig, init, eps, T, denoise, infill, inf_s, inf_m,
                                              noises=lambda si, i: zs[si * T + i], infill_noises=lambda si, i: izs[si * T + i])
    torch.cuda.synchronize()
    print(f"ald {arch} L={L} T={T} infill={infill} denoise={denoise}: state rel {rel(gs, rs):.3e} collection rel {rel(gc, rc):.3e} "
          f"metrics rel {rel(gm, rm):.3e}")
    assert gc.shape == rc.shape == (101 + int(denoise), *shape) and gm.shape == rm.shape == (4, L, T)
    assert rel(gs, rs) < 1e-2 and rel(gc, rc) < 1e-2 and rel(gm, rm) < 1e-2
    filled = (rc.flatten(1).abs().sum(1) > 0)
    assert torch.equal(filled, (gc.flatten(1).abs().sum(1) > 0).cpu())       # the same collection slots were written
    if infill:
        # masked region: template + sigma * infill noise of the LAST update, untouched by the score
        want = inf_s + float(sig[-1]) * izs[-1]
        assert torch.allclose(gs.cpu()[:, 8:24], want[:, 8:24], atol=1e-5)


def test_consistent_langevin_teacher_forced_and_streams():
    import smd_amd.ncsn as N
    from smd_amd.jax_random import ThreefryKey
    ocfg, p, model = make()
    B, shape = 2, (2, 32, 42)
    g = torch.Generator().manual_seed(9)
    init = (torch.rand(*shape, generator=g) * 2 - 1) * (12 ** 0.5 / 2)
    zs = _normals(shape, 10, 3)
    om = O.make_model(p, ocfg)
    with torch.no_grad():
        rs, rm = O.consistent_langevin_dynamics(om, SIGMAS, init.double(), 5e-6, True, lambda i: zs[i].double())
    gs, gm = N.consistent_langevin_dynamics(N.PRNGKey(0), model, SIGMAS, init, 5e-6, None, True, noises=lambda i: zs[i])
    print(f"cas explicit: state rel {rel(gs, rs):.3e} metrics rel {rel(gm, rm):.3e}")
    assert gm.shape == (4, 10, 1) and rel(gs, rs) < 1e-2 and rel(gm, rm) < 1e-2
    with pytest.raises(NotImplementedError):
        N.consistent_langevin_dynamics(N.PRNGKey(0), model, SIGMAS, init, 5e-6, None, True, True)
    # the reference's stream: rng, step_rng = split(rng) per level, z = normal(step_rng, state.shape)
    key = O.jax_prngkey(13)
    sk, _ = O.jax_langevin_keys(key, 10, consistent=True)
    n = int(np.prod(shape))
    with torch.no_grad():
        rs2, rm2 = O.consistent_langevin_dynamics(om, SIGMAS, init.double(), 5e-6, False,
                                                  lambda i: torch.from_numpy(O.jax_normal(sk[i], n).reshape(shape)).double())
    gs2, gm2 = N.consistent_langevin_dynamics(ThreefryKey(int(key[0]), int(key[1])), model, SIGMAS, init, 5e-6, None, False)
    print(f"cas threefry: state rel {rel(gs2, rs2):.3e} metrics rel {rel(gm2, rm2):.3e}")
    assert rel(gs2, rs2) < 1e-2 and rel(gm2, rm2) < 1e-2


def test_annealed_langevin_reference_streams_and_sample_api():
    import smd_amd.ncsn as N
    from smd_amd.jax_random import ThreefryKey
    ocfg, p, model = make()
    L, T, shape = 3, 3, (2, 32, 42)
    sig = O.create_noise_schedule(1.0, 0.05, L, "geometric").astype(np.float32)
    g = torch.Generator().manual_seed(10)
    init = (torch.rand(*shape, generator=g) * 2 - 1) * (12 ** 0.5 / 2)
    inf_s = torch.clamp(0.25 * torch.randn(*shape, generator=g), -1, 1)
    inf_m = torch.zeros(*shape)
    inf_m[:, :16] = 1.0
    key = O.jax_prngkey(17)
    sk, fk = O.jax_langevin_keys(key, L * T)
    n = int(np.prod(shape))
    draw = lambda k: torch.from_numpy(O.jax_normal(k, n).reshape(shape)).double()
    with torch.no_grad():
        rs, rc, rm = O.annealed_langevin_dynamics(O.make_model(p, ocfg), sig, init.double(), 2e-5, T, True,
                                                  lambda si, i: draw(sk[si * T + i]), True, inf_s.double(), inf_m.double(),
                                                  lambda si, i: draw(fk[si * T + i]))
    tk = ThreefryKey(int(key[0]), int(key[1]))
    gs, gc, gm = N.annealed_langevin_dynamics(tk, model, sig, init, 2e-5, T, True, True, inf_s, inf_m)
    print(f"ald threefry + infill: state rel {rel(gs, rs):.3e} collection rel {rel(gc, rc):.3e} metrics rel {rel(gm, rm):.3e}")
    assert rel(gs, rs) < 1e-2 and rel(gc, rc) < 1e-2 and rel(gm, rm) < 1e-2
    # mixed arguments: a ThreefryKey with EXPLICIT infill draws and no explicit step noise goes through the eager
    # loop; the step noise must still be the reference's threefry stream (not Philox of key.seed).  With the infill draws
    # set to what the infill keys would have produced, the run equals the all-threefry one.
    gm_s, gm_c, _ = N.annealed_langevin_dynamics(tk, model, sig, init, 2e-5, T, True, True, inf_s, inf_m,
                                                 infill_noises=lambda si, i: draw(fk[si * T + i]).float())
    print(f"ald threefry step noise + explicit infill draws vs all-threefry: state rel {rel(gm_s, gs):.3e}")
    assert rel(gm_s, gs) < 1e-4 and rel(gm_c, gc) < 1e-4
    # a 1-row shard of the same global draw reproduces row 1 (the key window, not the shard, indexes the stream)
    g1, _, _ = N.annealed_langevin_dynamics(tk, model, sig, init[1:], 2e-5, T, True, True, inf_s[1:], inf_m[1:],
                                            sample_offset=1, global_num_samples=2)
    assert rel(g1, gs[1:]) < 1e-4
    # train_ncsn.sample (:499-551): uniform(-sqrt(12)/2, sqrt(12)/2) init, 102 / 2 collection rows, collated metrics
    for sampling, rows in (("ald", 102), ("cas", 2)):
        for k in (N.PRNGKey(2), ThreefryKey(0, 2)):
            gen, coll, met = N.sample(model, sig, k, (32, 42), num_samples=3, sampling=sampling, epsilon=2e-5, steps=T, denoise=True)
            assert tuple(gen.shape) == (3, 32, 42) and coll.shape[0] == rows and torch.isfinite(gen).all()
            assert len(met) == L and len(met[0]) == (T if sampling == "ald" else 1)
            assert set(met[0][0]) == {"slope", "step", "alpha", "noise"}
            assert float(coll[0].abs().max()) <= 12 ** 0.5 / 2 + 1e-6 and float(coll[0].std()) > 0.8     # uniform, unit variance
    a, _, _ = N.sample(model, sig, N.PRNGKey(2), (32, 42), num_samples=3, sampling="ald", epsilon=2e-5, steps=T)
    b, _, _ = N.sample(model, sig, N.PRNGKey(2), (32, 42), num_samples=2, sampling="ald", epsilon=2e-5, steps=T, sample_offset=1,
                       global_num_samples=3)
    assert rel(b, a[1:]) < 1e-4                                              # Philox mode is shard-invariant too
    with pytest.raises(ValueError):
        N.sample(model, sig, N.PRNGKey(1), (32, 42), num_samples=2, sampling="hmc")


@pytest.mark.parametrize("rng_impl", ["philox", "threefry"])
def test_langevin_graph_replay_equals_eager_launches(rng_impl):
    """The annealed / consistent Langevin loops run as ONE captured (forward + update) pair whose step-dependent arguments
    (alpha, noise coefficient, infill sigma, next noise level, collection slot, threefry keys) come from device tables indexed
    by a device-side counter: the replayed graph, the same launches issued eagerly, and -- for the collection arithmetic --
    the reference's slot rule must agree bit for bit."""
    import smd_amd.ncsn as N
    from smd_amd.engine import NetConfig
    model = N.Model(NetConfig(data_channels=42, num_layers=2, num_heads=8, num_mlp_layers=1), "cuda:0", seed=3)
    sig = O.create_noise_schedule(1.0, 0.05, 6, "geometric")
    g = torch.Generator().manual_seed(2)
    init = (torch.rand(4, 32, 42, generator=g) * 2 - 1) * 1.7
    mask = torch.zeros(4, 32, 42)
    mask[:, :8] = 1.0
    kw = dict(infill=True, infill_samples=0.3 * torch.randn(4, 32, 42, generator=g), infill_masks=mask)
    key = N.make_key(9, rng_impl)
    for extra in ({}, kw):
        a = N.annealed_langevin_dynamics(key, model, sig, init, 2e-5, 7, True, use_graph=True, **extra)
        b = N.annealed_langevin_dynamics(key, model, sig, init, 2e-5, 7, True, use_graph=False, **extra)
        for u, v in zip(a, b):
            assert torch.equal(u, v)
        x, coll, ld = a
        assert tuple(coll.shape) == (102, 4, 32, 42) and tuple(ld.shape) == (4, 6, 7)
        cidx = np.linspace(1, 6 * 7, 100).astype(np.int32)
        hit = sorted({N.ald_collection_slot(cidx, k + 1) for k in range(42)} - {-1})
        written = [r for r in range(1, 101) if float(coll[r].abs().max()) > 0]
        assert written == [s for s in hit if 0 < s < 102]
        assert torch.isfinite(x).all() and float(ld[2, 0, 0]) > float(ld[2, -1, 0]) > 0      # alpha shrinks with sigma
    c1 = N.consistent_langevin_dynamics(key, model, sig, init, 2e-5, None, True, use_graph=True)
    c2 = N.consistent_langevin_dynamics(key, model, sig, init, 2e-5, None, True, use_graph=False)
    assert torch.equal(c1[0], c2[0]) and torch.equal(c1[1], c2[1])
