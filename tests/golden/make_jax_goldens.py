"""Golden vectors FROM THE REFERENCE ITSELF -- the recipe that pins parity the day a JAX environment exists.

The reference (magenta/symbolic-music-diffusion) ships no tests or golden vectors, and its arithmetic lives in
flax==0.3.0 / jax==0.2.8 / jaxlib==0.1.57 (requirements.txt:47,94,95), which cannot be installed in the build container
(no network).  Everything in oracle/ddpm_oracle.py is therefore a restatement ("parity unpinned").
This script closes that gap wherever those packages ARE importable:

    pip install "jax==0.2.8" "jaxlib==0.1.57" "flax==0.3.0" absl-py numpy      # any machine, CPU is enough
    python tests/golden/make_jax_goldens.py --reference /path/to/symbolic-music-diffusion
    git add tests/golden/jax && python -m pytest tests/test_jax_goldens.py

It imports the reference's own modules (models/ncsn.py, utils/losses.py, utils/ebm_utils.py, utils/train_utils.py -- none of
them needs TensorFlow) and calls the reference's own functions on fixed seeds; nothing is re-implemented here except the ten
lines of train_ncsn.train_step / create_model that live in a module importing TensorFlow (they are restated from the same
library calls, train_ncsn.py:193-203,279-287, and used only if ``import train_ncsn`` fails).  Outputs, all float32 as the
reference computes them, under tests/golden/jax/:

    <net>.npz           net in {transformer_small, dense_small}:
        init/<flax path>          create_model(PRNGKey(SEED)) initial parameters  (pins lecun-normal init + flax auto-names)
        x, t, eps_hat             one forward pass                                 (models/ncsn.py:125-135 / 141-179)
        loss_key, batch, betas    inputs of diffusion_loss
        loss_none, loss_mean      utils/losses.py:250-308 with continuous_noise=True, reduction 'none' / 'mean'
        grad/<flax path>          jax.grad of the mean loss
        step/<flax path>, step_metrics (loss, grad, lr)   one train_step with grad_clip 1.0, lr 1e-3
        smp_key, smp_init, smp_state, smp_collection, smp_metrics   diffusion_dynamics over a 20-level schedule
    <net>_checkpoint/checkpoint_0   flax.training.checkpoints.save_checkpoint((optimizer, ema, early_stop), 0)
    versions.json                   jax / jaxlib / flax versions that produced the files

tests/test_jax_goldens.py consumes them (skipped while the directory is absent) and is what flips parity to "pinned".
"""
import argparse
import json
import os
import sys

SEED = 0
T_SMALL = 20


def flatten(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        key = f"{prefix}/{k}" if prefix else str(k)
        if isinstance(v, dict):
            out.update(flatten(v, key))
        else:
            out[key] = v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("SMD_REFERENCE", "/root/reference"))
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "jax"))
    a = ap.parse_args()
    try:
        import jax
        import jax.numpy as jnp
        import jax.experimental.optimizers as jopt
        import jaxlib
        import flax
        from flax import nn, optim, serialization
        from flax.training import checkpoints
    except Exception as e:                                        # the build container: say so, change nothing
        print(f"make_jax_goldens: jax / flax not importable here ({e!r}); nothing written. See the module docstring.")
        return 2
    import numpy as np
    sys.path.insert(0, a.reference)
    import models.ncsn as ncsn                                    # noqa: E402  (the reference's own modules)
    import utils.ebm_utils as ebm_utils                           # noqa: E402
    import utils.losses as losses                                 # noqa: E402
    import utils.train_utils as train_utils                       # noqa: E402

    os.makedirs(a.out, exist_ok=True)
    json.dump({"jax": jax.__version__, "jaxlib": jaxlib.__version__, "flax": flax.__version__, "reference": os.path.abspath(a.reference)},
              open(os.path.join(a.out, "versions.json"), "w"), indent=1)

    def create_model(rng, clazz, input_shape, model_kwargs, batch_size):
        """train_ncsn.py:193-203 (that module imports TensorFlow; the same three library calls)."""
        module = clazz.partial(**model_kwargs)
        _, initial_params = module.init_by_shape(rng, [((batch_size, *input_shape), jnp.float32),
                                                       ((batch_size, *([1] * len(input_shape))), jnp.float32)])
        return nn.Model(module, initial_params)

    def train_step(batch, optimizer, betas, rng, lr, grad_clip):
        """train_ncsn.py:279-287 with objective = diffusion_loss, continuous_noise = True."""
        def loss_fn(model):
            return losses.diffusion_loss(batch, model, betas, rng, True, "mean")
        loss, grad = jax.value_and_grad(loss_fn)(optimizer.target)
        grad = jopt.clip_grads(grad, grad_clip)
        metrics = np.array([loss, jopt.l2_norm(grad), lr], np.float32)
        return optimizer.apply_gradient(grad, learning_rate=lr), metrics

    nets = {
        "transformer_small": (ncsn.TransformerDDPM, (32, 42), dict(num_layers=2, num_heads=8, num_mlp_layers=1, mlp_dims=256), 3),
        "dense_small": (ncsn.DenseDDPM, (42,), dict(num_layers=2, mlp_dims=256), 4),
    }
    for name, (clazz, shape, kwargs, B) in nets.items():
        rng = jax.random.PRNGKey(SEED)
        rng, model_rng, data_rng, loss_rng, smp_rng = jax.random.split(rng, 5)
        model = create_model(model_rng, clazz, shape, kwargs, B)
        out = {"model_rng": np.asarray(model_rng), "kwargs": json.dumps(kwargs), "shape": np.asarray(shape)}
        for k, v in flatten(serialization.to_state_dict(model.params)).items():
            out["init/" + k] = np.asarray(v)
        k1, k2, k3 = jax.random.split(data_rng, 3)
        x = jnp.clip(0.25 * jax.random.normal(k1, (B, *shape)), -1, 1)
        t = jax.random.uniform(k2, (B, *([1] * len(shape))), minval=0.05, maxval=1.0)
        out.update(x=np.asarray(x), t=np.asarray(t), eps_hat=np.asarray(model(x, t)))
        betas = ebm_utils.create_noise_schedule(1e-6, 0.01, T_SMALL, schedule="linear")
        batch = jnp.clip(0.25 * jax.random.normal(k3, (B, *shape)), -1, 1)
        out.update(loss_key=np.asarray(loss_rng), batch=np.asarray(batch), betas=np.asarray(betas),
                   loss_none=np.asarray(losses.diffusion_loss(batch, model, betas, loss_rng, True, "none")),
                   loss_mean=np.asarray(losses.diffusion_loss(batch, model, betas, loss_rng, True, "mean")))
        grad = jax.grad(lambda m: losses.diffusion_loss(batch, m, betas, loss_rng, True, "mean"))(model)
        for k, v in flatten(serialization.to_state_dict(grad.params)).items():
            out["grad/" + k] = np.asarray(v)
        optimizer = optim.Adam(learning_rate=1e-3).create(model)
        new_opt, metrics = train_step(batch, optimizer, betas, loss_rng, 1e-3, 1.0)
        for k, v in flatten(serialization.to_state_dict(new_opt.target.params)).items():
            out["step/" + k] = np.asarray(v)
        out["step_metrics"] = metrics
        init = jax.random.normal(jax.random.fold_in(smp_rng, 1), (B, *shape))
        state, collection, ld_metrics = ebm_utils.diffusion_dynamics(smp_rng, model, betas, init, 0.0, 0, False, False)
        out.update(smp_key=np.asarray(smp_rng), smp_init=np.asarray(init), smp_state=np.asarray(state),
                   smp_collection=np.asarray(collection), smp_metrics=np.asarray(ld_metrics))
        np.savez_compressed(os.path.join(a.out, name + ".npz"), **out)
        # the checkpoint triple of train_ncsn.py:395-399 after that one step
        ema = train_utils.EMAHelper(mu=0.999, params=model.params).update(new_opt.target)
        early_stop = train_utils.EarlyStopping()
        ck = os.path.join(a.out, name + "_checkpoint")
        os.makedirs(ck, exist_ok=True)
        checkpoints.save_checkpoint(ck, (new_opt, ema, early_stop), 0, keep=1)
        print(f"make_jax_goldens: wrote {name}.npz ({len(out)} arrays) and {ck}/checkpoint_0")
    return 0


if __name__ == "__main__":
    sys.exit(main())
