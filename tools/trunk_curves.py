"""Loss curves of the same training run with the 2048-wide trunk stored in bf16 (default) and in fp32 (the bf16
trunk was justified by a single-step parity number only).  Same seed, same synthetic batches (B = 256, ddpm-mel-32seq-512
network), STEPS optimisation steps each; prints the mean loss over windows of 20 steps and the final relative difference."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.ncsn as N
import smd_amd.schedule as S
from smd_amd.engine import NetConfig
from smd_amd.trainer import create_optimizer, train_step

STEPS = int(os.environ.get("STEPS", "400"))
B = 256
betas = S.create_noise_schedule(1e-6, 0.01, 1000, "linear")
g = torch.Generator().manual_seed(1234)
data = torch.clamp(0.25 * torch.randn(16 * B, 32, 512, generator=g), -1, 1).cuda()
curves = {}
for name, opt_val in (("bf16 trunk", 2), ("fp32 trunk", 1)):
    cfg = NetConfig(architecture="TransformerDDPM", data_channels=512, seq_len=32, num_timesteps=1000)
    model = N.Model(cfg, "cuda:0", seed=0)
    opt = create_optimizer(model, 1e-3, ema=False)
    opt.engine.set_option("trunk_bf16", opt_val)
    key = N.make_key(0, "philox")
    losses = []
    for step in range(STEPS):
        key, sub = N.split(key)
        x0 = data[(step % 16) * B:(step % 16 + 1) * B]
        _, m = train_step(N.diffusion_loss, x0, opt, betas, sub, 1e-3, grad_clip=1.0, lr_gamma=0.98, lr_interval=10000)
        losses.append(m)
    curves[name] = [float(m.resolve()["loss"]) for m in losses]
print(f"# {STEPS} train steps, B = {B}, lr 1e-3, synthetic latents; mean loss per window of 20 steps")
print("# step      " + "   ".join(f"{k:>12s}" for k in curves) + "    rel diff")
for i in range(0, STEPS, 20):
    a, b = (sum(c[i:i + 20]) / len(c[i:i + 20]) for c in curves.values())
    print(f"{i:5d}-{i + 19:<5d} {a:12.5f}   {b:12.5f}   {abs(a - b) / b:9.2e}")
a, b = (sum(c[-50:]) / 50 for c in curves.values())
print(f"last 50 steps: bf16 trunk {a:.5f}, fp32 trunk {b:.5f}, relative difference {abs(a - b) / b:.2e}")
