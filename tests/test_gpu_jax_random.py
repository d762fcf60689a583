"""GPU parity of the jax.random-compatible streams: the device kernels against the numpy
restatement (itself pinned by tests/golden/jax_random_kat.json), and the engine's train step / reverse sampler
consuming them exactly where the reference draws (utils/losses.py:271-294, utils/ebm_utils.py:329-362,
train_ncsn.py:536-540).

Bit-exact: bits, randint, uniform.  normal goes through logf / sqrtf on the device: |delta| <= 4e-7 * (1 + |z|).
"""
import json
import os

import numpy as np
import pytest
import torch

import ddpm_oracle as O

pytestmark = pytest.mark.gpu

BETAS = O.create_noise_schedule(1e-6, 0.01, 1000, "linear")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def okey(k):
    return (np.uint32(k.k0), np.uint32(k.k1))


def close_normal(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.max(np.abs(got - want) / (1 + np.abs(want))) <= 4e-7


@pytest.mark.parametrize("n", [1, 2, 7, 10, 4097, 1_000_001])
def test_device_bits_uniform_randint_bit_exact(n, dev):
    import smd_amd.jax_random as J
    key = J.PRNGKey(20 + n)
    got = J.bits(key, n, dev).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, O.jax_random_bits(okey(key), n))
    u = J.uniform(key, (n,), dev, -0.25, 3.5).cpu().numpy()
    assert np.array_equal(u, O.jax_uniform(okey(key), n, -0.25, 3.5))
    r = J.randint(key, (n,), 1, 1001, dev).cpu().numpy()
    assert np.array_equal(r, O.jax_randint(okey(key), n, 1, 1001))
    z = J.normal(key, (n,), dev).cpu().numpy()
    assert close_normal(z, O.jax_normal(okey(key), n))


def test_device_windows_reproduce_the_global_array(dev):
    """A rank that owns elements [offset, offset+count) of an n_total array draws exactly that slice."""
    import smd_amd.jax_random as J
    key = J.PRNGKey(5)
    n = 12_345                                   # odd: the zero-pad counter case
    full = J.normal(key, (n,), dev)
    fb = J.bits(key, n, dev)
    for off, cnt in [(0, 100), (6000, 300), (6172, 2), (6173, 6172), (n - 1, 1)]:
        w = J.normal(key, (cnt,), dev, n_total=n, offset=off)
        assert torch.equal(w, full[off:off + cnt])
        r = J.randint(key, (cnt,), 0, 1000, dev, n_total=n, offset=off).cpu().numpy()
        assert np.array_equal(r, O.jax_randint(okey(key), n, 0, 1000)[off:off + cnt])
    assert np.array_equal(fb.cpu().numpy().view(np.uint32), O.jax_random_bits(okey(key), n))


def test_device_normal_reproduces_jax_documentation_values(dev):
    import smd_amd.jax_random as J
    d = json.load(open(os.path.join(GOLDEN, "jax_random_kat.json")))["jax_docs"]
    key = J.PRNGKey(0)
    n1 = lambda k: float(J.normal(k, (1,), dev)[0])
    assert abs(n1(key) - d["normal_key0"]) < 5e-7
    for gen in ("split1", "split2"):
        key, sub = J.split(key)
        assert [key.k0, key.k1] == d[gen]["key"] and [sub.k0, sub.k1] == d[gen]["subkey"]
        assert abs(n1(sub) - d[gen]["normal_subkey"]) < 5e-7
    key, *subs = J.split(key, 4)
    for s, want in zip(subs, d["split3_of_4"]["normals_subkeys"]):
        assert abs(n1(s) - want) < 5e-7


def test_device_key_table_indexed_by_a_device_counter(dev):
    """smd_threefry_normal with key_table / idx_ptr: the key is picked on the device (row idx_add + idx_mul * *idx_ptr)."""
    import smd_amd.jax_random as J
    ik, nk = J.sampler_key_tables(J.PRNGKey(9), 5)
    table = torch.from_numpy(nk.view(np.int32).copy()).to(dev)
    t_ptr = torch.tensor([0], dtype=torch.int32, device=dev)
    out = torch.empty(3, 7, device=dev)
    for it in range(5):
        t_ptr.fill_(4 - it)                                   # the sampler's t runs 4, 3, ... ; row = 4 - t = iteration
        J.fill_normal_from_table(out, table, t_ptr, 5)
        want = J.normal(J.ThreefryKey(int(nk[it, 0]), int(nk[it, 1])), (3, 7), dev)
        assert torch.equal(out, want)


def _model(C=42, L=2, K=1):
    from test_gpu_engine import make
    return make(C=C, L=L, K=K)


def test_train_step_consumes_the_reference_draws():
    """train_step(rng=ThreefryKey) == train_step with the oracle's jax draws passed explicitly; a 2-way sharded
    batch reproduces the rows of the global draw."""
    import smd_amd.jax_random as J
    import smd_amd.ncsn as N
    ocfg, p, model = _model()
    B, shape = 8, (32, 42)
    g = torch.Generator().manual_seed(3)
    x0 = torch.clamp(0.25 * torch.randn(B, *shape, generator=g), -1, 1)
    rng = J.PRNGKey(11)
    labels, eps = O.jax_diffusion_loss_draws(okey(rng), (B, *shape), 1000)
    assert labels.min() >= 1 and labels.max() <= 1000
    a = N.diffusion_loss(x0, model, BETAS, rng, True, "none").cpu()                 # FLAGS.continuous_noise default (:52)
    b = N.diffusion_loss(x0, model, BETAS, N.PRNGKey(0), True, "none", labels=labels, eps=eps).cpu()
    assert torch.allclose(a, b, rtol=2e-5, atol=1e-7)
    # the draws themselves, incl. the sharded window
    lab_d, eps_d = J.diffusion_loss_draws(rng, (B, *shape), 1000, "cuda:0")
    assert np.array_equal(lab_d.cpu().numpy(), labels) and close_normal(eps_d.cpu().numpy(), eps)
    lab_h, eps_h = J.diffusion_loss_draws(rng, (B // 2, *shape), 1000, "cuda:0", sample_offset=B // 2, global_batch=B)
    assert torch.equal(lab_h, lab_d[B // 2:]) and torch.equal(eps_h, eps_d[B // 2:])
    # oracle loss on the same draws (bf16 path tolerance)
    ref = O.diffusion_loss(x0.double(), O.make_model(p, ocfg), BETAS, labels, torch.from_numpy(eps).double(), "none")
    assert float((a.double() - ref).abs().max() / ref.abs().max()) < 1e-2


@pytest.mark.parametrize("infill", [False, True])
def test_sampler_consumes_the_reference_draws(infill):
    """diffusion_dynamics(rng=ThreefryKey): per-iteration keys from the three splits of utils/ebm_utils.py:329,342,360
    (host), normals from the device kernel inside the captured step == the same walk with the oracle's normals passed
    explicitly; hipGraph == eager; a shard reproduces its rows."""
    import smd_amd.jax_random as J
    import smd_amd.ncsn as N
    _, _, model = _model()
    B, shape, steps = 6, (32, 42), 12
    t_stop = 1000 - steps
    rng = J.PRNGKey(77)
    per = int(np.prod(shape))
    init = J.normal(J.PRNGKey(5), (B, *shape), "cuda:0")
    ik, nk = O.jax_sampler_keys(okey(rng), steps)
    zs = {999 - i: torch.from_numpy(O.jax_normal(nk[i], B * per).reshape(B, *shape)) for i in range(steps)}
    izs = {999 - i: torch.from_numpy(O.jax_normal(ik[i], B * per).reshape(B, *shape)) for i in range(steps)}
    kw = {}
    if infill:
        g = torch.Generator().manual_seed(9)
        samples = torch.clamp(0.25 * torch.randn(B, *shape, generator=g), -1, 1)
        masks = torch.zeros(B, *shape)
        masks[:, :8] = 1
        masks[:, -8:] = 1
        kw = dict(infill_samples=samples * masks, infill_masks=masks)
    a, ca, ma = N.diffusion_dynamics(rng, model, BETAS, init, None, None, None, infill, t_stop=t_stop, use_graph=True, **kw)
    b, cb, mb = N.diffusion_dynamics(rng, model, BETAS, init, None, None, None, infill, t_stop=t_stop, use_graph=False, **kw)
    assert torch.equal(a, b) and torch.equal(ca, cb)
    e, ce, me = N.diffusion_dynamics(N.PRNGKey(0), model, BETAS, init, None, None, None, infill, t_stop=t_stop,
                                     noises=lambda t: zs[t], infill_noises=(lambda t: izs[t]) if infill else None, **kw)
    err = float((a - e).abs().max())
    print(f"jax-stream sampler vs explicit oracle normals (infill={infill}): max abs {err:.2e}")
    # 1e-7 differences of the device normals (logf) flip a few bf16 roundings inside the network; a different
    # stream would differ by O(1)
    assert err < 2e-3
    assert torch.allclose(ma, me, rtol=2e-3, atol=1e-5)
    c, _, _ = N.diffusion_dynamics(rng, model, BETAS, init[2:5], None, None, None, infill, t_stop=t_stop, use_graph=False,
                                   sample_offset=2, global_num_samples=B,
                                   **({k: v[2:5] for k, v in kw.items()}))
    assert float((c - a[2:5]).abs().max()) < 1e-4


def test_sample_api_threefry_init_and_sharding():
    import smd_amd.jax_random as J
    import smd_amd.ncsn as N
    _, _, model = _model()
    rng = J.PRNGKey(1)
    init_rng, _ld = J.split(rng)
    want_init = O.jax_normal(okey(init_rng), 4 * 32 * 42).reshape(4, 32, 42)
    gen, coll, met = N.sample(model, BETAS, rng, (32, 42), num_samples=4, sampling="ddpm")
    assert close_normal(coll[0].cpu().numpy(), want_init)                   # collection[0] = init (:323, :539-540)
    assert torch.isfinite(gen).all() and float(gen.abs().max()) <= 1.0 + 1e-5
    half, _, _ = N.sample(model, BETAS, rng, (32, 42), num_samples=2, sampling="ddpm", sample_offset=2, global_num_samples=4)
    # 1000 steps; the rows ride in a different batch size, i.e. through different GEMM tilings (K-split order):
    # 1e-7 differences per step, amplified by the walk; a different stream would differ by O(1)
    assert float((half - gen[2:]).abs().max()) < 1e-2


def test_create_model_with_a_threefry_key_draws_the_flax_initialiser():
    """create_model(model_rng) with a jax.random key: the engine's step-0 weights are flax_init.init_params (the reference's
    init_by_shape restated, train_ncsn.py:193-203), not the engine's own NumPy initialiser."""
    import smd_amd.flax_init as FI
    import smd_amd.jax_random as J
    import smd_amd.ncsn as N
    rng = J.PRNGKey(0)
    _rng, model_rng, _s = N.split(rng, num=3)                                       # train_ncsn.py:318-319
    kw = dict(num_layers=2, num_heads=8, num_mlp_layers=1, mlp_dims=2048)
    model = N.create_model(model_rng, (32, 42), kw, architecture="TransformerDDPM")
    table = {name: shape for name, _off, shape in model.engine.tensor_table}
    want = FI.init_params(model.cfg, model_rng, table)
    got = model.named_parameters()
    for k in table:
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    other = N.create_model(N.PRNGKey(0), (32, 42), kw, architecture="TransformerDDPM")     # engine key: the old initialiser
    assert not np.array_equal(other.named_parameters()["in_proj.kernel"].cpu().numpy(), want["in_proj.kernel"])


def test_interpolate_matches_the_oracle_restatement():
    """--interpolate (sample_ncsn.py:245-310,425-435): stochastic encode with the reference's key reuse (noise from ``rng``,
    alphas_prod[T] clamped), 9-point lerp, every point decoded with the same ld_rng; 50-level schedule, jax.random streams."""
    import ddpm_oracle as O
    import sample_ncsn as S
    import smd_amd.ncsn as N
    from smd_amd.engine import NetConfig
    T = 50
    betas = O.create_noise_schedule(1e-6, 0.01, T, "linear")
    ocfg = O.NetConfig(data_channels=42, num_layers=2, num_heads=8, num_mlp_layers=1)
    p = O.init_params(ocfg, 0, torch.float64)
    model = N.Model(NetConfig(data_channels=42, num_layers=2, num_heads=8, num_mlp_layers=1, num_timesteps=T), "cuda:0", seed=None)
    model.engine.load_named(p)
    g = torch.Generator().manual_seed(3)
    real = torch.clamp(0.25 * torch.randn(3, 32, 42, generator=g), -1, 1).numpy()
    seed = 7
    with torch.no_grad():
        gref, cref = O.interpolate_samples(O.make_model(p, ocfg), betas, real, seed)
    root = N.make_key(seed, "threefry")
    rng, _model_rng = N.split(root)                                              # load_model(): what main() passes as ``rng``
    gen, coll, met = S.interpolate_samples(model, betas, real, 0, 3, rng, seed, "threefry", "cuda:0", use_graph=True)
    assert gen.shape == (9, 3, 32, 42) and coll.shape == (9, 41, 3, 32, 42)
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))
    # the encoder alone is elementwise fp32: exact up to the normal's 1e-7
    z = S.diffusion_stochastic_encoder(real, betas, rng, "cuda:0", 0, 3)
    assert rel(z, O.diffusion_stochastic_encoder(real, betas, seed)) < 1e-6
    assert rel(gen, gref.numpy()) < 2e-2, rel(gen, gref.numpy())
    assert rel(coll[:, 0], cref[:, 0].numpy()) < 1e-6                             # slot 0 = the interpolated latent itself
    assert rel(gen[0], gref[0].numpy()) < 2e-2 and rel(gen[8], gref[8].numpy()) < 2e-2
    # both ends share the encoder noise: the end points' latents differ by sqrt(alpha_bar_T) (goals - starts) exactly
    a_T = float(O.alphas_cumprod(betas)[-1])
    assert rel(coll[8, 0] - coll[0, 0], np.sqrt(a_T) * (np.roll(real, 1, axis=0) - real)) < 1e-5
    assert len(met) == T
