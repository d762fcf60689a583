"""Wall time of a real 1000-step ncsn.sample() at several batch sizes, two pipelined chains (default) against one chain
(SMD_SAMPLER_CHAINS=1): where does the two-chain arrangement stop paying?  python tools/sampler_batch_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.ncsn as N
import smd_amd.schedule as S
from smd_amd.engine import NetConfig
betas = S.create_noise_schedule(1e-6, 0.01, 1000, "linear")
cfg = NetConfig(architecture="TransformerDDPM", data_channels=512, seq_len=32, num_timesteps=1000)
model = N.Model(cfg, "cuda:0", seed=0)
for B in (128, 256, 512, 1000, 2048):
    res = {}
    for chains in ("2", "1"):
        os.environ["SMD_SAMPLER_CHAINS"] = chains
        model.drop_sampler_cache()
        ts = []
        for i in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gen, coll, _m = N.sample(model, betas, N.PRNGKey(11 + i), (32, 512), num_samples=B, sampling="ddpm")
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            del gen, coll
        res[chains] = (ts[-1], dict(model.sampler_arrangement)["chain_sizes"])
    print(f"B={B:5d}: two chains {res['2'][1]} {res['2'][0]:.3f} s = {B * 1000 / res['2'][0] / 1e3:7.1f} k sequence-steps/s | one chain {res['1'][0]:.3f} s = "
          f"{B * 1000 / res['1'][0] / 1e3:7.1f} k   ({(res['1'][0] / res['2'][0] - 1) * 100:+.1f} % for two chains)")
