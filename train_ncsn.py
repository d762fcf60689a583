"""Drop-in for the reference's ``python train_ncsn.py --flagfile=configs/ddpm-mel-32seq-512.cfg``
on the MI355X HIP engine (DDPM path: --loss=ddpm --sampling=ddpm, TransformerDDPM / DenseDDPM).

Keeps the flag surface (train_ncsn.py:48-128), the loop order and quirks of train()
(train_ncsn.py:291-496) and the files written under --model_dir.  Data-parallel:
launch with ``python -m torch.distributed.run --nproc-per-node N train_ncsn.py ...`` (one process
per GPU, RCCL gradient all-reduce); rank 0 logs / evaluates / saves.
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


def log_langevin_dynamics(ld_metrics, step, output_dir):
    """train_ncsn.py:131-160: per-sampling-run scalars (slope, step, alpha, noise)."""
    from smd_amd.train_utils import JsonlWriter
    w = JsonlWriter(os.path.join(output_dir, f"sampling_epoch{step}"))
    for i, per_sigma in enumerate(ld_metrics):
        for j, m in enumerate(per_sigma):
            for k, v in m.items():
                w.scalar(k, float(v), i * len(per_sigma) + j)
    w.close()


def build_datasets(FLAGS, sample_shape, device, rank, world):
    from smd_amd import data
    slice_idx = data.load(os.path.expanduser(FLAGS.slice_ckpt)) if FLAGS.slice_ckpt else None
    dim_weights = data.load(os.path.expanduser(FLAGS.dim_weights_ckpt)) if FLAGS.dim_weights_ckpt else None
    if FLAGS.synthetic:
        train = data.SyntheticLatents(sample_shape, FLAGS.synthetic_examples, FLAGS.batch_size, 1234, device, rank, world)
        valid = data.SyntheticLatents(sample_shape, max(FLAGS.batch_size * 2, 512), FLAGS.batch_size, 4321, device)
    else:
        train, valid = data.open_dataset(FLAGS.dataset, FLAGS.batch_size, sample_shape, device, rank, world,
                                         FLAGS.normalize, slice_idx, dim_weights, data_shape=[int(v) for v in FLAGS.data_shape],
                                         pca_ckpt=FLAGS.pca_ckpt, slice_ckpt=FLAGS.slice_ckpt,
                                         dim_weights_ckpt=FLAGS.dim_weights_ckpt, shuffle=True, seed=FLAGS.seed)
    return train, valid, slice_idx, dim_weights


def model_shape(FLAGS, slice_idx):
    """Per-example shape the network sees: --data_shape with the last dim replaced by the slice size
    (input_pipeline.py:43-48) -- (32,42) for configs/ddpm-mel-32seq-512.cfg, (32,512) when synthetic."""
    shape = [int(v) for v in FLAGS.data_shape]
    if slice_idx is not None and not FLAGS.synthetic:
        shape[-1] = len(slice_idx)
    return tuple(shape)


def train(FLAGS, train_batches, valid_batches, sigmas, output_dir, rank=0, world=1, verbose=True):
    """train_ncsn.py:291-496."""
    from smd_amd import checkpoint, ncsn, train_utils
    from smd_amd.trainer import GradComm, create_optimizer, evaluate, train_step

    train_writer = train_utils.JsonlWriter(os.path.join(output_dir, "train")) if rank == 0 else None
    eval_writer = train_utils.JsonlWriter(os.path.join(output_dir, "eval")) if rank == 0 else None
    input_shape = train_batches.sample_shape
    rng = ncsn.make_key(FLAGS.seed, FLAGS.rng_impl)
    rng, model_rng, sample_rng = ncsn.split(rng, num=3)                          # :318-319
    model_kwargs = dict(num_layers=FLAGS.num_layers, num_heads=FLAGS.num_heads,
                        num_mlp_layers=FLAGS.num_mlp_layers, mlp_dims=FLAGS.mlp_dims)
    dev = f"cuda:{torch.cuda.current_device()}"
    model = ncsn.create_model(model_rng, input_shape, model_kwargs, FLAGS.batch_size, verbose=verbose and rank == 0,
                              architecture=FLAGS.architecture, num_timesteps=len(sigmas), device=dev, dtype=FLAGS.dtype)
    optimizer = create_optimizer(model, FLAGS.learning_rate, ema=FLAGS.ema)       # :332
    optimizer.engine.set_option("trunk_bf16", 2 if FLAGS.trunk_dtype == "bf16" else 1)
    optimizer.engine.trunk_dtype = FLAGS.trunk_dtype                               # recorded in the checkpoint metadata
    optimizer.engine.fp8_dgrad = int(bool(FLAGS.fp8_dgrad))                        # likewise (only meaningful with --dtype=fp8)
    optimizer.engine.set_option("fp8_dgrad", optimizer.engine.fp8_dgrad)
    if rank == 0:
        log.info("training trunk dtype: %s (GEMM operands: %s%s)", FLAGS.trunk_dtype, FLAGS.dtype,
                 "" if FLAGS.dtype != "fp8" else (", e4m3 dgrads" if FLAGS.fp8_dgrad else ", bf16 dgrads"))
    comm = GradComm(algorithm=FLAGS.dp_algorithm, layer_buckets=FLAGS.dp_layer_buckets) if world > 1 else None
    if comm is not None:
        comm.broadcast_params(model.params)
        model.engine.refresh_weights()
        if optimizer.engine.ema is not None:
            optimizer.engine.ema.copy_(model.params)
    early_stop = train_utils.EarlyStopping(patience=1)                            # :333
    ema = train_utils.EMAHelper(FLAGS.mu, optimizer.engine.ema if FLAGS.ema else model.params.clone(),
                                fused=FLAGS.ema)                                  # :336 (untouched init when --ema=False)

    # :345-352 (ssm needs a second backward through the network and is a toy-only objective upstream)
    objective = {"ddpm": ncsn.diffusion_loss, "dsm": ncsn.denoising_score_matching_loss}[FLAGS.loss]
    sampling_step = -1
    for epoch in range(FLAGS.epochs):
        start_time = time.time()
        if hasattr(train_batches, "set_epoch"):
            train_batches.set_epoch(epoch)          # the shuffle of epoch e is a function of (seed, e), also after a resume
        for step, batch in enumerate(train_batches):
            rng, train_rng = ncsn.split(rng)                                      # :358
            global_step = step + epoch * train_batches.examples                   # :359
            # the stepped LR schedule (:340-342) is evaluated on the device from the update counter
            optimizer, train_metrics = train_step(
                objective, batch, optimizer, sigmas, train_rng, FLAGS.learning_rate,
                grad_clip=FLAGS.grad_clip, mu=FLAGS.mu, comm=comm, lr_gamma=FLAGS.lr_gamma,
                lr_interval=FLAGS.lr_schedule_interval, sample_offset=rank * FLAGS.batch_size,
                global_batch=FLAGS.batch_size * world, continuous_noise=FLAGS.continuous_noise)
            if FLAGS.ema:
                ema = ema.update(optimizer.target)                                # :364-365 (fused: no-op)

            if step % FLAGS.logging_freq == 0 and rank == 0:                      # :367-378
                m = train_metrics.resolve()
                elapsed = time.time() - start_time
                m["batch/s"] = (step + 1) / elapsed
                m["ms/batch"] = elapsed * 1000 / (step + 1)
                train_utils.log_metrics(m, step, train_batches.examples, epoch=epoch, summary_writer=train_writer,
                                        verbose=verbose)

            if (step % FLAGS.snapshot_freq == 0 and step > 0) or step == train_batches.examples - 1:   # :380-381
                sampling_step += 1
                rng, eval_rng = ncsn.split(rng)
                if rank == 0:
                    eval_metrics = evaluate(valid_batches, optimizer.target, sigmas, eval_rng, FLAGS.continuous_noise, objective)
                    train_utils.log_metrics(eval_metrics, global_step, train_batches.examples * FLAGS.epochs,
                                            summary_writer=eval_writer, verbose=verbose)
                    improved, early_stop = early_stop.update(eval_metrics["loss"])            # :393
                    if (not FLAGS.early_stopping and FLAGS.save_ckpt) or \
                            (FLAGS.early_stopping and improved and FLAGS.save_ckpt):          # :395-399
                        checkpoint.save_checkpoint(output_dir, (optimizer, ema, early_stop), sampling_step,
                                                   keep=FLAGS.checkpoints_to_keep, fmt=FLAGS.ckpt_format)
                # early stop is decided BEFORE snapshot sampling, as the reference returns first (:401-403)
                if comm is not None:                                                           # keep ranks in step
                    flag = torch.tensor([1.0 if (FLAGS.early_stopping and early_stop.should_stop) else 0.0],
                                        device=dev)
                    comm.dist.broadcast(flag, src=0)
                    stop = bool(flag.item())
                else:
                    stop = FLAGS.early_stopping and early_stop.should_stop
                if stop:                                                                       # :401-403
                    if rank == 0:
                        train_writer.flush()
                        eval_writer.flush()
                    log.info("EARLY STOP: Ended training after %s epochs.", epoch + 1)
                    return optimizer
                if rank == 0:
                    if FLAGS.snapshot_sampling:                                               # :404-414
                        scorenet = ncsn.Model(model.cfg, dev, seed=None)
                        scorenet.replace(ema.params if FLAGS.ema else optimizer.target.params)
                        rng, _unused = ncsn.split(rng)
                        generated, collection, ld_metrics = ncsn.sample(
                            scorenet, sigmas, rng, input_shape, num_samples=FLAGS.eval_samples,
                            sampling=FLAGS.sampling, epsilon=FLAGS.ld_epsilon, steps=FLAGS.ld_steps,
                            denoise=FLAGS.denoise)
                        log_langevin_dynamics(ld_metrics, sampling_step, output_dir)
                        del scorenet
                    train_writer.flush()
                    eval_writer.flush()
            if FLAGS.max_steps is not None and global_step >= FLAGS.max_steps:                # :492-494
                return optimizer
    return optimizer


def main(argv):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    FLAGS = F.make_flags(include_sample=False)
    FLAGS.parse(argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0:
        log.info(FLAGS.flags_into_string())
        log.info("Platform: gfx950 HIP engine (smd_amd %s)", smd_amd.__version__)
    if FLAGS.loss not in ("ddpm", "dsm") or FLAGS.sampling not in ("ddpm", "ald", "cas"):
        raise SystemExit("this engine covers --loss=ddpm|dsm and --sampling=ddpm|ald|cas (ssm needs a double backward)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    from smd_amd import data
    slice_idx = data.load(os.path.expanduser(FLAGS.slice_ckpt)) if FLAGS.slice_ckpt else None
    shape = model_shape(FLAGS, slice_idx)
    train_ds, eval_ds, _, _ = build_datasets(FLAGS, shape, f"cuda:{local_rank}", rank, world)
    noise_schedule = schedule.create_noise_schedule(FLAGS.sigma_begin, FLAGS.sigma_end, FLAGS.num_sigmas,
                                                    schedule=FLAGS.schedule_type)               # :574-577
    train(FLAGS, train_ds, eval_ds, noise_schedule, FLAGS.model_dir, rank, world, verbose=FLAGS.verbose)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv)
