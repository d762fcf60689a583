"""Wall time of consecutive real ncsn.sample() calls (1000 reverse steps of 256 sequences): first call (allocations, capture), later
calls (cached graphs).  python tools/sampler_walk_time.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smd_amd.ncsn as N
import smd_amd.schedule as S
from smd_amd.engine import NetConfig

model = N.Model(NetConfig(architecture="TransformerDDPM", data_channels=512, seq_len=32, num_timesteps=1000), "cuda:0", seed=0)
betas = S.create_noise_schedule(1e-6, 0.01, 1000, "linear")
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gen, coll, _ = N.sample(model, betas, N.PRNGKey(11 + i), (32, 512), num_samples=256, sampling="ddpm")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tm = getattr(model, "_sampler_timing", None)
    extra = "" if not tm else (f"; replay loop {tm['loop_s']:.4f} s for {tm['replays']} replays = {tm['loop_s'] / tm['replays'] * 1e6:.0f} us/step "
                               f"(host issued them in {tm['host_issue_s']:.4f} s), everything else {dt - tm['loop_s']:.4f} s, cached graphs {tm['reused']}")
    print(f"sampler_walk call {i}: {dt:.4f} s = {1000 / dt:.0f} steps/s, finite {bool(torch.isfinite(gen).all())}{extra}")
    del gen, coll

# the walk under the other chain arrangements (graphs are re-captured when the arrangement changes): free-running one-step graphs
# (round 4), pipelined with U steps per graph
for env in ({"SMD_SAMPLER_PIPELINE": "0"}, {"SMD_SAMPLER_UNROLL": "1"}, {"SMD_SAMPLER_UNROLL": "8"}, {"SMD_SAMPLER_UNROLL": "16"}, {}):
    for k in ("SMD_SAMPLER_PIPELINE", "SMD_SAMPLER_UNROLL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ts = []
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gen, coll, _ = N.sample(model, betas, N.PRNGKey(21 + i), (32, 512), num_samples=256, sampling="ddpm")
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        del gen, coll
    print(f"sampler_walk {env or 'default (pipelined, 4 steps per graph)'}: " + " ".join(f"{t:.4f}" for t in ts) + f" s -> {1000 / min(ts):.0f} steps/s")
