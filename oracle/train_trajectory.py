"""TEST INFRASTRUCTURE (like everything under oracle/): N FREE-RUNNING train steps of the CPU restatement oracle
(ddpm_oracle.train_step: utils/losses.py:250-308 -> train_ncsn.py:260-288 loss, value_and_grad, clip_grads, Adam) on a small fixed
problem with explicit per-step labels / eps -- the reference side of tests/test_gpu_trajectory.py, which runs the same loop on the
HIP engine.  A command line so that the test can run several variants side by side as processes:

  python oracle/train_trajectory.py --out run.npz [--dtype float32|float64] [--steps N] [--grad-noise SIGMA] [--emulate bf16|fp8]

--grad-noise SIGMA: the CONTROL run -- every gradient tensor g gets unbiased Gaussian noise of relative size SIGMA
(g + SIGMA * ||g|| / sqrt(numel) * N(0, 1)) before clip_grads: what a training loop does when its gradients carry rounding noise of
the size the bf16 / fp8 engine's have against the exact ones (6e-3 / 2e-2), and nothing else is different.

--emulate bf16|fp8: the FORMAT run -- the loss is evaluated through oracle/bf16_emulation.py / e4m3_emulation.py (float64 arithmetic
with the engine's rounding points in the forward AND the backward pass: bf16 operand pack of the current master weights, bf16 /
e4m3 activations and gradients where the engine stores them so); clip_grads and Adam as in every other run.  What a training run
in the engine's number formats does when every kernel is exact."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ddpm_oracle as O  # noqa: E402

C, L, H, K, B = 42, 2, 8, 1, 16
LR, CLIP = 1e-3, 1.0                      # configs/ddpm-base.cfg: --learning_rate=1e-3; train_ncsn.py:61 grad_clip 1
BETAS = O.create_noise_schedule(1e-6, 0.01, 1000, "linear")
SNAP_EVERY = 50


def net_config():
    return O.NetConfig(data_channels=C, num_layers=L, num_heads=H, num_mlp_layers=K)


def initial_params():
    return O.init_params(net_config(), 3, torch.float32)


def dataset():
    """128 structured latents (a rank-6 pattern + noise: there is something to fit) as 8 batches of 16 that the loop cycles
    through, and 64 held-out latents of the same distribution."""
    g = torch.Generator().manual_seed(4242)
    basis = torch.randn(6, 32, C, generator=g)

    def draw(n):
        coef = torch.randn(n, 6, generator=g)
        return torch.clamp(0.35 * torch.einsum("bk,ksc->bsc", coef, basis) / 6 ** 0.5 + 0.05 * torch.randn(n, 32, C, generator=g), -1, 1)
    train = draw(8 * B).view(8, B, 32, C)
    return train, draw(4 * B)


def draws(step, n=B):
    """labels (utils/losses.py:272-275, continuous_noise: [1, T]) and eps (:294) of train step `step`"""
    g = torch.Generator().manual_seed(900_000 + step)
    return torch.randint(1, 1001, (n,), generator=g), torch.randn(n, 32, C, generator=g)


def run(dtype, steps, grad_noise=0.0, noise_seed=7, emulate=None):
    cfg = net_config()
    p = {k: v.to(dtype) for k, v in initial_params().items()}
    train, _held = dataset()
    st = O.AdamState()
    gn = torch.Generator().manual_seed(noise_seed)
    losses, snaps = [], {}
    for it in range(steps):
        lab, eps = draws(it)
        if emulate:
            import bf16_emulation as E
            import e4m3_emulation as F8
            mk = (E if emulate == "bf16" else F8).make_model
            leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
            loss = O.diffusion_loss(train[it % 8].to(dtype), mk(leaf, cfg, backward=True), BETAS, lab.numpy(), eps.to(dtype), "mean")
            loss.backward()
            clipped, _ = O.clip_grads({k: v.grad.detach() for k, v in leaf.items()}, CLIP)
            p = O.adam_update({k: v.detach() for k, v in leaf.items()}, clipped, st, LR)
            losses.append(float(loss.detach()))
        elif grad_noise == 0.0:
            p, m, _ = O.train_step(p, cfg, st, train[it % 8].to(dtype), BETAS, lab.numpy(), eps.to(dtype), LR, CLIP)
            losses.append(m["loss"])
        else:                               # train_step with the noise between value_and_grad and clip_grads
            leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
            loss = O.diffusion_loss(train[it % 8].to(dtype), O.make_model(leaf, cfg), BETAS, lab.numpy(), eps.to(dtype), "mean")
            loss.backward()
            grads = {}
            for k, v in leaf.items():
                g = v.grad.detach()
                grads[k] = g + (grad_noise * float(g.norm()) / max(g.numel(), 1) ** 0.5) * torch.randn(g.shape, generator=gn).to(dtype)
            clipped, _ = O.clip_grads(grads, CLIP)
            p = O.adam_update({k: v.detach() for k, v in leaf.items()}, clipped, st, LR)
            losses.append(float(loss.detach()))
        if (it + 1) % SNAP_EVERY == 0:
            snaps[it + 1] = np.concatenate([v.double().numpy().ravel() for _k, v in sorted(p.items())])
    return p, np.array(losses), snaps


def flat(p):
    return np.concatenate([v.detach().double().cpu().numpy().ravel() for _k, v in sorted(p.items())])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--grad-noise", type=float, default=0.0)
    ap.add_argument("--emulate", choices=["bf16", "fp8"], default=None)
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    p, losses, snaps = run(torch.float64 if a.emulate else getattr(torch, a.dtype), a.steps, a.grad_noise, emulate=a.emulate)
    np.savez(a.out, losses=losses, final=flat(p), names=np.array(sorted(p)), **{f"snap_{k}": v for k, v in snaps.items()})


if __name__ == "__main__":
    main()
