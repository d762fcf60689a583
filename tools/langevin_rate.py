"""Updates/s of the graph-replayed annealed Langevin sampler vs the DDPM reverse sampler (B = 256, ddpm-mel-32seq-512 network)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.ncsn as N
import smd_amd.schedule as S
from smd_amd.engine import NetConfig
B = 256
model = N.Model(NetConfig(architecture="TransformerDDPM", data_channels=512, seq_len=32, num_timesteps=1000), "cuda:0", seed=0)
betas = S.create_noise_schedule(1e-6, 0.01, 1000, "linear")
sig = S.create_noise_schedule(1.0, 0.01, 10, "geometric")
init = torch.randn(B, 32, 512, device="cuda")
for name, fn, n in (("ald graph", lambda: N.annealed_langevin_dynamics(N.PRNGKey(1), model, sig, init, 2e-5, 40, False, use_graph=True), 400),
                    ("ald eager", lambda: N.annealed_langevin_dynamics(N.PRNGKey(1), model, sig, init, 2e-5, 40, False, use_graph=False), 400),
                    ("ddpm graph (400 steps)", lambda: N.diffusion_dynamics(N.PRNGKey(1), model, betas, init, t_start=999, t_stop=600), 400)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:24s} {n / dt:8.1f} updates/s ({dt * 1e3 / n:.3f} ms each)")
