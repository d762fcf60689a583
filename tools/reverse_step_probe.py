"""Where the fused reverse step's time goes (round 6): smd_ddpm_reverse_step at (B, 32, 512) with in-kernel Philox noise vs explicit z,
with / without the metrics, at B = 256 and 128, each timed as back-to-back launches (HIP events) and reported against the algorithmic
bytes 3 x B x S x C x 4 -- and smd_q_sample / smd_mse_loss_grad the same way.  python tools/reverse_step_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import smd_amd.lib as lib
import smd_amd.schedule as S
L = lib.get_lib()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
P = lambda t: None if t is None else t.data_ptr()
T = 1000
betas = S.create_noise_schedule(1e-6, 0.01, T, "linear")
coef = torch.from_numpy(S.reverse_coefficient_table(np.asarray(betas, dtype=np.float32))).to(dev)
ape = torch.from_numpy(np.concatenate([np.ones(1, np.float32), S.alphas_cumprod(betas)])).to(dev)


def timeit(fn, reps=100):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


for B in (256, 128):
    Sq, C = 32, 512
    x = torch.randn(B, Sq, C, device=dev)
    eh = torch.randn(B, Sq, C, device=dev)
    z = torch.randn(B, Sq, C, device=dev)
    t = torch.full((1,), 500, dtype=torch.int32, device=dev)
    met = torch.zeros(T, B, 3, device=dev)
    mb = B * Sq * C * 4 * 3 / 1e6
    for name, zz, mm in (("philox+metrics", None, met), ("philox", None, None), ("explicit z+metrics", z, met), ("explicit z", z, None)):
        us = timeit(lambda: lib.check(L.smd_ddpm_reverse_step(P(x), P(eh), B, Sq, C, P(coef), T, P(t), P(zz), 11, 22, 0, P(mm), None, None, st)))
        print(f"reverse_step B={B} {name:20s}: {us:6.2f} us  {mb / us:5.2f} TB/s of the algorithmic {mb:.0f} MB")
    x0 = torch.clamp(0.25 * torch.randn(B, Sq, C, device=dev), -1, 1)
    xt = torch.empty(B * Sq, C, dtype=torch.bfloat16, device=dev)
    eo = torch.empty(B, Sq, C, device=dev)
    s = torch.empty(B, device=dev)
    lab = torch.randint(1, 1001, (B,), dtype=torch.int32, device=dev)
    for name, ee in (("philox", None), ("explicit eps", z)):
        us = timeit(lambda: lib.check(L.smd_q_sample(P(x0), B, Sq, C, C, T, P(ape), P(lab), 1, None, P(ee), 7, 0, None, 0, P(xt), P(eo), P(s), st)))
        print(f"q_sample     B={B} {name:20s}: {us:6.2f} us  {mb / us:5.2f} TB/s of the algorithmic {mb:.0f} MB")
