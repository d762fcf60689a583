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

# the same cached graphs replayed in blocks of 100 (what bench.py's sample loop does), t restarted before each block
import smd_amd.lib as lib
ent = model._sampler_graphs["entry"]
for blk in range(6):
    for ch in ent["chains"]:
        lib.check(lib.get_lib().smd_set_timestep(ch["t_ptr"].data_ptr(), 999 if blk % 2 == 0 else 500, ch["stream"].cuda_stream))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        for ch in ent["chains"]:
            with torch.cuda.stream(ch["stream"]):
                ch["graph"].replay()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"sampler_walk block {blk} (t from {999 if blk % 2 == 0 else 500}): {dt / 100 * 1e6:.0f} us/step")
