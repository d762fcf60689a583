"""Drop-in for the reference's ``python sample_ncsn.py --flagfile=... --sample_seed --sample_size
--sampling_dir`` on the MI355X HIP engine: unconditional generation, infilling (--infill) and
interpolation (--interpolate) for DDPM checkpoints, writing {sampling_dir}/ncsn/{generated,
collection,real}.pkl in the reference's layout (sample_ncsn.py:368-471).

Multi-GPU sampling is embarrassingly parallel: under torch.distributed.run each rank generates a
contiguous shard of the samples (Philox counters are keyed by the GLOBAL sample index, so the
result does not depend on the number of GPUs) and rank 0 gathers and writes the files.
"""
from __future__ import annotations

import logging
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import smd_amd  # noqa: E402,F401
from smd_amd import flags as F  # noqa: E402
from smd_amd import schedule  # noqa: E402

log = logging.getLogger("smd_amd")


def load_model(FLAGS, shape, device):
    """sample_ncsn.py:331-342: dummy-parameter model + restore_checkpoint.  The reference samples
    from optimizer.target (raw weights) even when an EMA was trained (:353-354); --sample_ema opts
    into the EMA weights."""
    from smd_amd import checkpoint, ncsn
    model_kwargs = dict(num_layers=FLAGS.num_layers, num_heads=FLAGS.num_heads,
                        num_mlp_layers=FLAGS.num_mlp_layers, mlp_dims=FLAGS.mlp_dims)
    rng = ncsn.make_key(FLAGS.sample_seed, FLAGS.rng_impl)
    rng, model_rng = ncsn.split(rng)
    model = ncsn.create_model(model_rng, shape, model_kwargs, batch_size=1, verbose=True,
                              architecture=FLAGS.architecture, num_timesteps=FLAGS.num_sigmas, device=device, dtype=FLAGS.dtype,
                              init=False)                       # the weights come from the checkpoint (sample_ncsn.py:331-342)
    found = (checkpoint.load_ema_params if FLAGS.sample_ema else
             lambda d, e: checkpoint.restore_checkpoint(d, e, load_optimizer_state=False)[0])(FLAGS.model_dir, model.engine)
    if not found:
        log.warning("no checkpoint under %s: sampling from freshly initialised weights", FLAGS.model_dir)
        ncsn.init_model(model, model_rng)
    return model, rng


def generate_samples(FLAGS, model, rng, sample_shape, num_samples, sigmas, sample_offset=0, global_num_samples=None):
    """sample_ncsn.py:313-365."""
    from smd_amd import ncsn
    rng, sample_rng = ncsn.split(rng)
    t0 = time.time()
    generated, collection, ld_metrics = ncsn.sample(
        model, sigmas, sample_rng, sample_shape, num_samples=num_samples, sampling=FLAGS.sampling,
        epsilon=FLAGS.ld_epsilon, steps=FLAGS.ld_steps, denoise=FLAGS.denoise, sample_offset=sample_offset,
        use_graph=FLAGS.graph, global_num_samples=global_num_samples)
    torch.cuda.synchronize()
    log.info("Generated samples in %f seconds", time.time() - t0)
    return generated, collection, ld_metrics


def infill_samples(FLAGS, model, rng, samples, masks, sigmas, sample_offset=0, global_num_samples=None):
    """sample_ncsn.py:189-242 (init ~ U[0,1) like :228: jax.random.uniform with --rng_impl=threefry, torch's device
    generator otherwise)."""
    from smd_amd import jax_random, ncsn
    init_rng, ld_rng = ncsn.split(rng)
    if isinstance(init_rng, jax_random.ThreefryKey):
        per = int(np.prod(samples.shape[1:]))
        n_glob = (global_num_samples or (len(samples) + sample_offset)) * per
        init = jax_random.uniform(init_rng, samples.shape, model.engine.device, n_total=n_glob, offset=sample_offset * per)
    else:
        g = torch.Generator(device=model.engine.device).manual_seed(init_rng.seed & 0x7FFFFFFFFFFFFFFF)
        init = torch.rand(samples.shape, generator=g, device=model.engine.device)
    if FLAGS.sampling == "ald":                                             # :219-221
        generated, collection, ld_metrics = ncsn.annealed_langevin_dynamics(
            ld_rng, model, sigmas, init, FLAGS.ld_epsilon, FLAGS.ld_steps, FLAGS.denoise, True,
            infill_samples=samples, infill_masks=masks, sample_offset=sample_offset, global_num_samples=global_num_samples)
        return generated, collection, ncsn.collate_sampling_metrics(ld_metrics.cpu().numpy())
    if FLAGS.sampling == "cas":                                             # :222-223 -> NotImplementedError (:228-229)
        ncsn.consistent_langevin_dynamics(ld_rng, model, sigmas, init, FLAGS.ld_epsilon, FLAGS.ld_steps, FLAGS.denoise, True)
    generated, collection, ld_metrics = ncsn.diffusion_dynamics(
        ld_rng, model, sigmas, init, FLAGS.ld_epsilon, FLAGS.ld_steps, FLAGS.denoise, True,
        infill_samples=samples, infill_masks=masks, use_graph=FLAGS.graph, sample_offset=sample_offset,
        global_num_samples=global_num_samples)
    return generated, collection, ncsn.collate_sampling_metrics(ld_metrics.cpu().numpy())


def diffusion_stochastic_encoder(samples, sigmas, rng, device="cuda:0", sample_offset=0, global_num_samples=None):
    """sample_ncsn.py:245-266: z = sqrt(ap[T]) x + sqrt(1-ap[T]) noise.  alphas_prod[T] is one past
    the end upstream; JAX clamps the gather, i.e. it uses ap[T-1] (SURVEY quirks ledger).  ``rng`` is the first
    child of split(PRNGKey(seed)) -- the reference draws the noise from it, not from noise_rng (:262-263)."""
    from smd_amd import jax_random
    ap = schedule.alphas_cumprod(sigmas)
    a_T = float(ap[-1])
    if isinstance(rng, jax_random.ThreefryKey):
        per = int(np.prod(samples.shape[1:]))
        n_glob = (global_num_samples or (len(samples) + sample_offset)) * per
        noise = jax_random.normal(rng, samples.shape, device, n_total=n_glob, offset=sample_offset * per).cpu().numpy()
    else:
        g = torch.Generator().manual_seed(rng.seed & 0x7FFFFFFFFFFFFFFF)
        noise = torch.randn(samples.shape, generator=g).numpy()
    return np.sqrt(a_T) * samples + np.sqrt(1 - a_T) * noise


def interpolate_samples(model, sigmas, real, lo, hi, rng, sample_seed, rng_impl, dev, use_graph=True, points=9):
    """sample_ncsn.py:425-435 + diffusion_decoder (:269-310) for this rank's rows [lo, hi): goals = roll(starts, 1); both are
    encoded with the same key (the same noise); every one of the 9 interpolated latents is decoded with the SAME
    ld_rng = split(PRNGKey(sample_seed), 3)[1].  Returns (generated (9, n, ...), collection (9, 41, n, ...), collated metrics)."""
    from smd_amd import ncsn
    num = len(real)
    starts = real[lo:hi]
    goals = np.roll(real, shift=1, axis=0)[lo:hi]
    zs, zg = (diffusion_stochastic_encoder(v, sigmas, rng, dev, lo, num) for v in (starts, goals))
    _, ld_rng, _ = ncsn.split(ncsn.make_key(sample_seed, rng_impl), num=3)                 # :271-272 (the root key)
    gens, colls = [], []
    for i, alpha in enumerate(np.linspace(0.0, 1.0, points)):
        g, c, ld = ncsn.diffusion_dynamics(ld_rng, model, sigmas, ((1 - alpha) * zs + alpha * zg).astype(np.float32),
                                           use_graph=use_graph, sample_offset=lo, global_num_samples=num)
        gens.append(g.cpu().numpy())
        colls.append(c.cpu().numpy())
        log.info("Generated samples %i out of %i", i, points)
    return np.stack(gens), np.stack(colls), ncsn.collate_sampling_metrics(ld.cpu().numpy())


def main(argv):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    FLAGS = F.make_flags(include_sample=True)
    FLAGS.parse(argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0:
        log.info(FLAGS.flags_into_string())
    if FLAGS.sampling not in ("ddpm", "ald", "cas"):
        raise SystemExit(f"Unknown sampling algorithm: {FLAGS.sampling}")
    if FLAGS.interpolate and FLAGS.sampling != "ddpm":
        raise SystemExit("--interpolate is a DDPM mode (sample_ncsn.py:250,270 assert it)")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(dev))
    from smd_amd import data, ncsn
    from smd_amd.trainer import shard_bounds
    from train_ncsn import build_datasets, log_langevin_dynamics, model_shape

    slice_idx = data.load(os.path.expanduser(FLAGS.slice_ckpt)) if FLAGS.slice_ckpt else None
    dim_weights = data.load(os.path.expanduser(FLAGS.dim_weights_ckpt)) if FLAGS.dim_weights_ckpt else None
    shape = model_shape(FLAGS, slice_idx)
    sigmas = schedule.create_noise_schedule(FLAGS.sigma_begin, FLAGS.sigma_end, FLAGS.num_sigmas,
                                            schedule=FLAGS.schedule_type)
    # "real" examples: the reference needs the eval set even for unconditional sampling (:387-402)
    if FLAGS.synthetic:
        n = FLAGS.sample_size
        real = data.SyntheticLatents(shape, max(n, 1), max(n, 1), 4321).take_examples(n)
        tmin, tmax, emin, emax = -1.0, 1.0, -1.0, 1.0
    else:
        train_ds, eval_ds, _, _ = build_datasets(FLAGS, shape, None, 0, 1)
        real = eval_ds.take_examples(FLAGS.sample_size)
        tmin, tmax, emin, emax = train_ds.min, train_ds.max, eval_ds.min, eval_ds.max
    num = len(real)
    lo, hi = shard_bounds(num, world, rank)
    model, rng = load_model(FLAGS, shape, dev)

    if FLAGS.infill:                                                        # :405-423
        samples = np.copy(real[lo:hi])
        masks = np.zeros(samples.shape, np.float32)
        if len(shape) == 1:
            samples[:, shape[0] // 2:] = 0
            masks[:, :shape[0] // 2] = 1
        else:
            idx = list(range(shape[0]))
            fixed_idx, infilled_idx = idx[:8] + idx[-8:], idx[8:-8]
            samples[:, infilled_idx, :] = 0
            masks[:, fixed_idx, :] = 1
        generated, collection, ld_metrics = infill_samples(FLAGS, model, rng, samples, masks, sigmas, sample_offset=lo,
                                                           global_num_samples=num)
    elif FLAGS.interpolate:                                                 # :425-435
        generated, collection, ld_metrics = interpolate_samples(model, sigmas, real, lo, hi, rng, FLAGS.sample_seed, FLAGS.rng_impl,
                                                                dev, FLAGS.graph)
    else:                                                                   # :437-439
        generated, collection, ld_metrics = generate_samples(FLAGS, model, rng, shape, hi - lo, sigmas, sample_offset=lo,
                                                             global_num_samples=num)

    generated = generated.cpu().numpy() if torch.is_tensor(generated) else generated
    collection = collection.cpu().numpy() if torch.is_tensor(collection) else collection
    if world > 1:                                                           # host-side gather of the shards
        import torch.distributed as dist
        gathered = [None] * world
        dist.gather_object((generated, collection), gathered if rank == 0 else None, dst=0)
        if rank == 0:
            axis = 1 if FLAGS.interpolate else 0
            generated = np.concatenate([g for g, _ in gathered], axis=axis)
            collection = np.concatenate([c for _, c in gathered], axis=axis + 1)

    if rank == 0 and FLAGS.flush:                                           # :452-471
        log_dir = FLAGS.sampling_dir
        pca = data.load(os.path.expanduser(FLAGS.pca_ckpt)) if FLAGS.pca_ckpt else None       # :380-381
        inv = lambda b, mn, mx: data.inverse_data_transform(b, FLAGS.normalize, pca, mn, mx, slice_idx, dim_weights)
        if not FLAGS.interpolate:
            data.save(inv(collection, tmin, tmax), os.path.join(log_dir, "ncsn/collection.pkl"))
        data.save(inv(real, emin, emax), os.path.join(log_dir, "ncsn/real.pkl"))
        data.save(inv(generated, tmin, tmax), os.path.join(log_dir, "ncsn/generated.pkl"))
        if FLAGS.compute_metrics:
            log_langevin_dynamics(ld_metrics, 0, log_dir)
            log.warning("--compute_metrics: the reference's evaluate() calls functions that do not exist in "
                        "utils/metrics.py (SURVEY section 2); only the sampler scalars were written")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv)
