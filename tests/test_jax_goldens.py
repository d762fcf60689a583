"""Parity against golden vectors produced BY THE REFERENCE ITSELF (tests/golden/make_jax_goldens.py).

Skipped while tests/golden/jax/ does not exist: the generator needs jax 0.2.8 / flax 0.3.0, which cannot be installed in the
build container.  Once the files are there these tests ARE the pin: the oracle restatement (layers, loss, optimiser step,
sampler, jax.random draws, flax initialiser and auto-names, checkpoint layout) against the reference's own float32 numbers.
Tolerance 2e-5 relative (float32 reference vs float64 restatement; the sampler 2e-4 after 20 steps)."""
import os

import numpy as np
import pytest
import torch

import ddpm_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jax")
pytestmark = pytest.mark.skipif(not os.path.isdir(GOLD), reason="tests/golden/jax absent: run tests/golden/make_jax_goldens.py "
                                                                 "where jax==0.2.8 / flax==0.3.0 are installed")
NETS = {"transformer_small": dict(architecture="TransformerDDPM", data_channels=42, num_layers=2, num_heads=8, num_mlp_layers=1, mlp_dims=256),
        "dense_small": dict(architecture="DenseDDPM", data_channels=42, num_layers=2, mlp_dims=256)}


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def unflatten(z, prefix):
    tree = {}
    for k in z.files:
        if k.startswith(prefix + "/"):
            node = tree
            parts = k[len(prefix) + 1:].split("/")
            for p in parts[:-1]:
                node = node.setdefault(p, {})
            node[parts[-1]] = z[k]
    return tree


def load(name):
    import smd_amd.flax_io as FIO
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = O.NetConfig(**NETS[name])
    template = dict(O.param_spec(cfg))
    named = lambda prefix: {k: torch.from_numpy(v).double() for k, v in FIO.params_from_flax(unflatten(z, prefix), cfg, template).items()}
    return z, cfg, template, named


@pytest.mark.parametrize("name", list(NETS))
def test_flax_auto_names_and_initialiser(name):
    import smd_amd.flax_init as FI
    import smd_amd.flax_io as FIO
    import smd_amd.jax_random as J
    z, cfg, template, named = load(name)
    rule, ac = FIO.detect_naming(unflatten(z, "init"), cfg)                # raises if no known rule matches the real tree
    mk = z["model_rng"]
    mine = FI.init_params(cfg, J.ThreefryKey(int(mk[0]), int(mk[1])), template, rule=rule, attention_class=ac)
    ref = named("init")
    for k in template:
        assert np.allclose(mine[k], ref[k].numpy(), rtol=0, atol=2e-7 * (1 + np.abs(ref[k].numpy()).max())), k


@pytest.mark.parametrize("name", list(NETS))
def test_forward_loss_gradient_step(name):
    z, cfg, template, named = load(name)
    p = named("init")
    model = O.make_model(p, cfg)
    assert rel(model(torch.from_numpy(z["x"]).double(), torch.from_numpy(z["t"]).double()), z["eps_hat"]) < 2e-5
    betas = z["betas"].astype(np.float32)
    assert np.array_equal(betas, O.create_noise_schedule(1e-6, 0.01, len(betas), "linear"))
    key = (np.uint32(z["loss_key"][0]), np.uint32(z["loss_key"][1]))
    labels, eps = O.jax_diffusion_loss_draws(key, z["batch"].shape, len(betas))
    batch = torch.from_numpy(z["batch"]).double()
    loss = O.diffusion_loss(batch, model, betas, labels, torch.from_numpy(eps).double(), "none")
    assert rel(loss, z["loss_none"]) < 2e-5 and abs(float(loss.mean()) - float(z["loss_mean"])) < 2e-5 * float(z["loss_mean"])
    st = O.AdamState()
    newp, metrics, grads = O.train_step(p, cfg, st, batch, betas, labels, torch.from_numpy(eps).double(), 1e-3, 1.0)
    gref, sref = named("grad"), named("step")
    for k in template:
        assert rel(grads[k], gref[k]) < 5e-5 or float(gref[k].norm()) < 1e-12, k
        assert rel(newp[k] - p[k], sref[k] - p[k]) < 1e-3 or float((sref[k] - p[k]).norm()) < 1e-12, k      # the update itself
    assert abs(metrics["loss"] - float(z["step_metrics"][0])) < 2e-5 * abs(float(z["step_metrics"][0]))
    assert abs(metrics["grad"] - float(z["step_metrics"][1])) < 5e-5 * abs(float(z["step_metrics"][1]))


@pytest.mark.parametrize("name", list(NETS))
def test_reverse_sampler(name):
    z, cfg, template, named = load(name)
    model = O.make_model(named("init"), cfg)
    betas = z["betas"].astype(np.float32)
    T = len(betas)
    key = (np.uint32(z["smp_key"][0]), np.uint32(z["smp_key"][1]))
    _infill, noise_keys = O.jax_sampler_keys(key, T)
    shape = z["smp_init"].shape
    zs = {T - 1 - i: torch.from_numpy(O.jax_normal(noise_keys[i], int(np.prod(shape))).reshape(shape)).double() for i in range(T)}
    with torch.no_grad():
        state, coll, met = O.diffusion_dynamics(model, betas, torch.from_numpy(z["smp_init"]).double(), lambda t: zs[t])
    assert rel(state, z["smp_state"]) < 2e-4
    assert tuple(coll.shape) == tuple(z["smp_collection"].shape) and rel(coll, z["smp_collection"]) < 2e-4
    assert rel(met, z["smp_metrics"]) < 2e-4


@pytest.mark.parametrize("name", list(NETS))
def test_checkpoint_file_of_the_reference_is_read(name):
    import smd_amd.flax_io as FIO
    z, cfg, template, named = load(name)
    sd = FIO.read_file(os.path.join(GOLD, name + "_checkpoint", "checkpoint_0"))
    params, grad_ema, grad_sq_ema, step, ema_params, mu, early = FIO.split_state_dict(sd, cfg, template)
    sref = named("step")
    for k in template:
        assert np.allclose(params[k], sref[k].numpy(), rtol=0, atol=1e-7), k
    assert int(step) == 1 and abs(float(mu) - 0.999) < 1e-7
