"""Where does the gradient error on TRAINED weights come from?  (tests/test_gpu_full_walk.py: after 1500 train steps the
forward still matches the fp32 oracle to 7e-3, the whole-vector gradient only to 4e-2; random-init weights: 6e-3.)
Trains the bench network for `--steps` steps, then compares one loss_backward at B = 256 against the oracle's autograd gradient
for several engine option sets (bf16 trunk / bf16 residual-gradient chain / fused kernels on and off), with the per-tensor
contributions to the squared error.
    python tools/trained_grad_diag.py [--steps 1500]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import ddpm_oracle as O
import smd_amd.ncsn as N
from smd_amd.engine import Engine, NetConfig
from test_gpu_full_walk import BETAS, _train, make

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=1500)
ap.add_argument("--dtype", default="bf16")
args = ap.parse_args()
B, C = 256, 512
ocfg, p0, model = make(C, 6, 8, 2, dtype=args.dtype)
opt, losses, x_last = _train(model, args.steps)
print("loss every 100 steps:", " ".join(f"{v:.3f}" for v in losses))
p = {k: v.detach().float().cpu().clone() for k, v in model.engine.named_views().items()}
g = torch.Generator().manual_seed(31)
x0 = x_last
labels = torch.randint(1, 1001, (B,), generator=g)
eps = torch.randn(B, 32, C, generator=g)


def oracle_grads(params, dtype):
    leaf = {k: v.to(dtype).clone().requires_grad_(True) for k, v in params.items()}
    loss = O.diffusion_loss(x0.to(dtype), O.make_model(leaf, ocfg), BETAS, labels.numpy(), eps.to(dtype), "none")
    loss.mean().backward()
    return {k: v.grad.double() for k, v in leaf.items()}, float(loss.mean())


torch.set_num_threads(min(os.cpu_count() or 1, 64))
ref, lref = oracle_grads(p, torch.float32)
den = sum(float(v.pow(2).sum()) for v in ref.values())
print(f"oracle fp32 loss {lref:.6f}; |g| {den ** 0.5:.4f}")
if os.environ.get("ORACLE64") == "1":                      # how far is the fp32 oracle itself from fp64 on these weights?
    ref64, _ = oracle_grads(p, torch.float64)
    num = sum(float((ref[k] - ref64[k]).pow(2).sum()) for k in ref)
    print(f"oracle fp32 vs fp64 gradient rel {(num / den) ** 0.5:.3e}")
    ref = ref64


def run(tag, opts):
    eng = Engine(model.cfg, "cuda:0", share_params_with=model.engine)
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.enable_training(False)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(B, training=True)
    eng.loss_backward(x0.cuda(), labels.int().cuda(), eps.cuda(), stage=0)
    torch.cuda.synchronize()
    gv = eng.named_views(eng.grads)
    per = {k: float((gv[k].double().cpu() - ref[k]).pow(2).sum()) for k in ref}
    num = sum(per.values())
    top = sorted(per, key=lambda k: -per[k])[:6]
    print(f"{tag:34s} whole-vector rel {(num / den) ** 0.5:.3e}; loss {float(eng.loss_per_sample().mean()):.6f}; largest contributions: "
          + ", ".join(f"{k} {per[k] / num * 100:.0f}% (own rel {(per[k] / float(ref[k].pow(2).sum())) ** 0.5:.2e})" for k in top))
    del eng


run("default", {})
run("trunk_bf16=1 (fp32 trunk in training)", {"trunk_bf16": 1})
run("resgrad_bf16=0", {"resgrad_bf16": 0})
run("trunk_bf16=1 + resgrad_bf16=0", {"trunk_bf16": 1, "resgrad_bf16": 0})
run("+ mlp_hs=0", {"trunk_bf16": 1, "resgrad_bf16": 0, "mlp_hs": 0})
run("+ fused_encoder=0, fused_attn_bwd=0", {"trunk_bf16": 1, "resgrad_bf16": 0, "mlp_hs": 0, "fused_encoder": 0, "fused_attn_bwd": 0})
# random-init weights through the same code, for scale
ocfg2, p2, model2 = make(C, 6, 8, 2, dtype=args.dtype)
pr = {k: v.float() for k, v in p2.items()}
ref, _ = oracle_grads(pr, torch.float32)
den = sum(float(v.pow(2).sum()) for v in ref.values())
model = model2
run("random-init weights, default", {})
