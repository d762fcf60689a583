"""Experiment: one sampling chain of B sequences vs two concurrent chains of B/2 on two streams (same GPU).
Does overlapping one chain's epilogues / HBM-bound kernels with the other's MFMA phases pay?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.lib as lib
import smd_amd.ncsn as N
import smd_amd.schedule as S
from smd_amd.engine import NetConfig

dev = "cuda:0"
betas = S.create_noise_schedule(1e-6, 0.01, 1000, "linear")
cfg = NetConfig(architecture="TransformerDDPM", data_channels=512, seq_len=32, num_timesteps=1000)


def chain(B, stream):
    model = N.Model(cfg, dev, seed=0)
    eng = model.engine
    eng.set_schedule(betas, with_sampler=True)
    eng.bind(B, training=False)
    eng.prepare_sampler()
    x = torch.empty(B, 32, 512, device=dev)
    eng.init_state(x, 4321, 0)
    t_ptr = torch.tensor([999], dtype=torch.int32, device=dev)
    mp = torch.zeros(1000, B, 3, device=dev)
    io = lib.SampleIO()
    io.x, io.t_ptr = x.data_ptr(), t_ptr.data_ptr()
    io.seed_lo, io.seed_hi, io.sample_offset = 7, 0, 0
    io.metrics_partial, io.slot_table = mp.data_ptr(), eng.slot_table.data_ptr()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        eng.sample_step(io)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        eng.sample_step(io)
    return g, (model, x, t_ptr, mp, io), t_ptr


def run(graphs_streams, tptrs, steps=60):
    for t in tptrs:
        t.fill_(999)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for g, s in graphs_streams:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


for force in (1, 2):
    lib.check(lib.get_lib().smd_set_tuning(b"gemm_nt256", force))
    s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    g256, keep0, t0p = chain(256, s0)
    ga, keep1, t1p = chain(128, s1)
    gb, keep2, t2p = chain(128, s2)
    for _ in range(2):
        one = run([(g256, s0)], [t0p])
        two = run([(ga, s1), (gb, s2)], [t1p, t2p])
        seq = run([(ga, s1), (gb, s1)], [t1p, t2p])
        print(f"gemm_nt256={force}: one chain B=256 {one*1e6:.1f} us/step | two concurrent chains of 128 {two*1e6:.1f} us | "
              f"the same two chains on one stream {seq*1e6:.1f} us")
