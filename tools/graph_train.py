"""Experiment: the whole train step (loss_backward + optimizer_step, two streams inside the engine) captured in one hipGraph
and replayed, against eager launches.  Philox draws are keyed by (seed, sample, DEVICE step counter), so a replay with a
fixed key still draws fresh labels / noise every step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.ncsn as N
import smd_amd.schedule as S
from smd_amd.engine import NetConfig
from smd_amd.trainer import create_optimizer, train_step

dev = "cuda:0"
betas = S.create_noise_schedule(1e-6, 0.01, 1000, "linear")
cfg = NetConfig(architecture="TransformerDDPM", data_channels=512, seq_len=32, num_timesteps=1000)
B = 256
g = torch.Generator().manual_seed(1234)
x0 = torch.clamp(0.25 * torch.randn(B, 32, 512, generator=g), -1, 1).to(dev)
key = N.PRNGKey(0)


def build():
    model = N.Model(cfg, dev, seed=0)
    opt = create_optimizer(model, 1e-3, ema=False)
    return model, opt


def step(opt):
    train_step(N.diffusion_loss, x0, opt, betas, key, 1e-3, grad_clip=1.0, lr_gamma=0.98, lr_interval=10000)


def timed(f, n=100):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


m1, o1 = build()
eager = timed(lambda: step(o1))
m2, o2 = build()
for _ in range(3):
    step(o2)                                   # warm-up: workspaces bound, side stream created, events pooled
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step(o2)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, capture_error_mode="relaxed"):
    step(o2)
torch.cuda.synchronize()
replay = timed(graph.replay)
print(f"graph_train: eager {eager * 1e6:.1f} us/step ({1 / eager:.1f} steps/s) | graph replay {replay * 1e6:.1f} us/step ({1 / replay:.1f} steps/s)")
# same number of optimiser steps on both models from the same init -> same weights?
torch.cuda.synchronize()
n1, n2 = int(o1.engine.step_counter), int(o2.engine.step_counter)
print(f"graph_train: optimiser steps eager {n1}, graphed {n2}; loss eager {float(o1.engine.loss_per_sample().mean()):.6f} graphed {float(o2.engine.loss_per_sample().mean()):.6f}")
