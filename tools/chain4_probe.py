"""Would FOUR quarter-batch sampling chains fill the chip better than two half-batch ones?  (The pipelined two-chain walk reaches
~520 us per step of 256 sequences; both chains' kernels in one stream take 870 us, the 2048-wide GEMMs alone ~250 us of whole-chip
time.)  n chains of 256 / n sequences, one-step graphs on n streams, started 1/n of a period apart (free-running: the first ~60
steps show the potential before any drift), against the same for n = 2 and n = 1.
    python tools/chain4_probe.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import smd_amd.lib as lib
import smd_amd.ncsn as N
import smd_amd.schedule as S
from smd_amd.engine import Engine, NetConfig

dev = "cuda:0"
B = 256
betas = S.create_noise_schedule(1e-6, 0.01, 1000, "linear")
cfg = NetConfig(architecture="TransformerDDPM", data_channels=512, seq_len=32, num_timesteps=1000, dtype=os.environ.get("DTYPE", "bf16"))
model = N.Model(cfg, dev, seed=0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000)
torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(2_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 2_000_000 / (e0.elapsed_time(e1) * 1e3)


def build(n):
    hB = B // n
    chains = []
    for c in range(n):
        eng = Engine(cfg, dev, share_params_with=model.engine)
        eng.set_option("nt256_min_tiles", max(16, (hB * 32 // 256) * 8))
        eng.set_schedule(betas, with_sampler=True)
        eng.bind(hB, training=False)
        eng.prepare_sampler()
        x = torch.empty(hB, 32, 512, device=dev)
        eng.init_state(x, 4321, c * hB)
        t_ptr = torch.tensor([999], dtype=torch.int32, device=dev)
        mp = torch.zeros(1000, hB, 3, device=dev)
        coll = torch.zeros(41, hB, 32, 512, device=dev)
        io = lib.SampleIO()
        io.x, io.t_ptr = x.data_ptr(), t_ptr.data_ptr()
        io.seed_lo, io.seed_hi, io.sample_offset = 7, 0, c * hB
        io.metrics_partial, io.collection, io.slot_table = mp.data_ptr(), coll.data_ptr(), eng.slot_table.data_ptr()
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            eng.sample_step(io)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            eng.sample_step(io)
        gp = []
        for order in ((2, 1), (1, 2)):                 # pipelined halves, 4 steps per graph
            gg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gg, stream=st):
                for _ in range(4):
                    for part in order:
                        eng.sample_step(io, part)
            gp.append(gg)
        chains.append(dict(eng=eng, io=io, x=x, t=t_ptr, g=g, gp=gp, st=st, keep=(mp, coll)))
    torch.cuda.synchronize()
    return chains


def reset(chains):
    for c, ch in enumerate(chains):
        with torch.cuda.stream(ch["st"]):
            ch["eng"].init_state(ch["x"], 99, c * ch["x"].shape[0])
        lib.check(lib.get_lib().smd_set_timestep(ch["t"].data_ptr(), 999, ch["st"].cuda_stream))
    torch.cuda.synchronize()


def run_free(chains, steps, stagger_us):
    reset(chains)
    n = len(chains)
    for c, ch in enumerate(chains):
        if c and stagger_us:
            with torch.cuda.stream(ch["st"]):
                torch.cuda._sleep(int(c * stagger_us * cyc_per_us))
    marks = []
    t0 = time.perf_counter()
    ev0 = torch.cuda.Event(enable_timing=True); ev0.record(chains[0]["st"])
    for k in range(steps):
        for ch in chains:
            with torch.cuda.stream(ch["st"]):
                ch["g"].replay()
        if (k + 1) % 20 == 0:
            e = torch.cuda.Event(enable_timing=True); e.record(chains[0]["st"]); marks.append(e)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    prev, per = ev0, []
    for e in marks:
        per.append(prev.elapsed_time(e) * 1e3 / 20)
        prev = e
    return steps / wall, per


def run_pipe(chains, steps):
    """ring pipeline: even chains run (output stage, stem) graphs, odd chains (stem, output stage); every replay waits for the
    previous replay of the neighbour chain (c + 1) % n -- with n = 2 this is the shipped arrangement"""
    reset(chains)
    n = len(chains)
    for c, ch in enumerate(chains):
        if c % 2 == 0:
            with torch.cuda.stream(ch["st"]):
                ch["eng"].sample_step(ch["io"], 1)
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event() for _ in range(2)] for _ in range(n)]
    t0 = time.perf_counter()
    for i in range(steps // 4):
        for c, ch in enumerate(chains):
            with torch.cuda.stream(ch["st"]):
                if i > 0:
                    ch["st"].wait_event(ev[(c + 1) % n][(i - 1) & 1])
                ch["gp"][c % 2].replay()
                ev[c][i & 1].record(ch["st"])
    torch.cuda.synchronize()
    return (steps // 4) * 4 / (time.perf_counter() - t0)


for n in (1, 2, 4):
    chains = build(n)
    for rep in range(2):
        if n == 1:
            r, per = run_free(chains, 200, 0)
            print(f"n = 1 chain of 256: {r:7.1f} steps/s; us per step per 20-step window: {' '.join('%.0f' % v for v in per)}", flush=True)
            continue
        for stag in (0, 520 // n, 600 // n):
            r, per = run_free(chains, 200, stag)
            print(f"n = {n} chains of {B // n}, free-running, started {stag:3d} us apart: {r:7.1f} steps/s; us per step per 20-step window: "
                  f"{' '.join('%.0f' % v for v in per)}", flush=True)
        r = run_pipe(chains, 200)
        print(f"n = {n} chains of {B // n}, ring pipeline (4 steps per graph): {r:7.1f} steps/s", flush=True)
    del chains
    torch.cuda.empty_cache()
