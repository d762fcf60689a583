"""Why is the two-chain graph-replayed sampler bimodal (1650 vs 1800+ steps/s with identical code)?

One process, the bench's own chains and captured graphs (bench.Workload), replayed under different arrangements:
  plain      the two torch streams of bench.py
  delay=X    chain B starts X us behind chain A (does the rate depend on the PHASE between the chains?)
  prio       two high-priority streams
  xcd        chain A masked to XCCs 0-3, chain B to 4-7 under the interleaved rule (smd_probe_stream_create_cu_mask) -- which the
             probe showed to leave the OTHER XCCs unmasked: effectively no mask
  xcdblk     mask bits 0..127 / 128..255: half of EVERY XCC's CUs per chain (the only partition a CU mask can express)
  one        both chains on ONE stream (no overlap at all: the sequential reference)
  pipe1      the software pipeline: ONE captured graph per step pair, forked over two streams -- chain A runs (output stage +
             reverse update of step k, stem of step k+1), chain B runs (stem of step k, output stage of step k): one chain's
             encoder kernels always beside the other's 2048-wide GEMMs, the phase re-locked at every replay
  pipe2      the same two halves as two graphs on two streams, each replay of one waiting for the previous replay of the other
  pipe2free  those two graphs free-running
  pipe2:uU:sS:mM   pipe2 with U steps per captured graph (the cross wait and the host launch once per U steps), the first S
             LayerNorm + Dense half-blocks of the output stage moved into the stem pass (engine option sample_split: balances the
             two halves), Dense layers on the 256x256 kernel from M tiles up (nt256_min_tiles: 32 puts out_proj there)
Every run: `steps` replays per chain, an event per chain every `mark` steps -> us/step per interval per chain and the lag of
chain B behind chain A at every mark; a one-wave clock probe on a third stream per mark -> effective shader clock under load.
  python tools/chain_phase.py [--steps 300] [--mark 20] [--tag first]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import smd_amd.lib as lib

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--mark", type=int, default=20)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--tag", default="")
ap.add_argument("--modes", default="plain,pipe2,pipe2:u2,pipe2:u4,pipe2:u8,pipe2:u1:s1,pipe2:u4:s1,pipe2:u4:s2,pipe2:u4:s1:m32,pipe2:u4:s0:m32,pipe2free:u4:s1,plain")
ap.add_argument("--dtype", default="bf16")
args = ap.parse_args()

dev = "cuda:0"
L = lib.get_lib()
a = bench.parse(["--mode", "sample", "--no-cpu-baseline", "--no-extra-configs", "--no-sampler-walk", "--sampler-unroll", "0"])
t_setup = time.perf_counter()
w = bench.Workload(a, "base", args.dtype, 0, 1, dev, None)
w.warm_up(3, False, True)
torch.cuda.synchronize()
print(f"# {args.tag}: setup + warm-up + capture {time.perf_counter() - t_setup:.2f} s; chains {w.nchains}", flush=True)


def masked_stream(mask, layout):
    """layout 0: mask bit i set for the XCCs in `mask` under the interleaved rule (bit i -> XCC i % 8) -- which, as the probe showed,
    leaves the other XCCs UNMASKED; layout 1: bits 32 x .. 32 x + 31 for every x in `mask` (a fraction of every XCC's CUs)"""
    words = (C.c_uint32 * 8)()
    for i in range(256):
        x = (i & 7) if layout == 0 else (i >> 5)
        if (mask >> x) & 1:
            words[i >> 5] |= 1 << (i & 31)
    p = C.c_void_p()
    lib.check(L.smd_probe_stream_create_cu_mask(words, 8, C.byref(p)), "stream_create_cu_mask")
    return torch.cuda.ExternalStream(p.value, device=dev)


# calibrate torch.cuda._sleep: cycles per microsecond
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000)
torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(2_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 2_000_000 / (e0.elapsed_time(e1) * 1e3)

probe_stream = torch.cuda.Stream(device=dev)
plain = [ch["stream"] for ch in w.chains]
STREAMS = {"plain": plain}


def streams_for(mode):
    if mode.startswith("delay") or mode == "plain":
        return plain
    if mode not in STREAMS:
        if mode == "prio":
            STREAMS[mode] = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(2)]
        elif mode == "xcd":
            STREAMS[mode] = [masked_stream(0x0F, 0), masked_stream(0xF0, 0)]
        elif mode == "xcdblk":
            STREAMS[mode] = [masked_stream(0x0F, 1), masked_stream(0xF0, 1)]
        elif mode == "one":
            STREAMS[mode] = [plain[0], plain[0]]
        else:
            raise SystemExit(f"unknown mode {mode}")
    return STREAMS[mode]


PIPE = {}


def parse_pipe(mode):
    """'pipe2:u4:s1:m32' -> (unroll, split, min_tiles)"""
    u, sp, mt = 1, 0, 128
    for tok in mode.split(":")[1:]:
        if tok[0] == "u": u = int(tok[1:])
        elif tok[0] == "s": sp = int(tok[1:])
        elif tok[0] == "m": mt = int(tok[1:])
    return u, sp, mt


def pipe_graphs(kind):
    """capture the pipelined step: chain A = (part 2, part 1), chain B = (part 1, part 2)"""
    if kind in PIPE:
        return PIPE[kind]
    A, B = w.chains
    sA, sB = plain
    torch.cuda.synchronize()
    U, SP, MT = parse_pipe(kind)
    for ch in w.chains:
        ch["eng"].set_option("sample_split", SP)
        ch["eng"].set_option("nt256_min_tiles", MT)
    if kind == "pipe1":
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=sA):
            sB.wait_stream(sA)
            A["eng"].sample_step(A["io"], 2)
            A["eng"].sample_step(A["io"], 1)
            with torch.cuda.stream(sB):
                B["eng"].sample_step(B["io"], 1)
                B["eng"].sample_step(B["io"], 2)
            sA.wait_stream(sB)
        PIPE[kind] = (g,)
    else:
        gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gA, stream=sA):
            for _ in range(U):
                A["eng"].sample_step(A["io"], 2)
                A["eng"].sample_step(A["io"], 1)
        with torch.cuda.graph(gB, stream=sB):
            for _ in range(U):
                B["eng"].sample_step(B["io"], 1)
                B["eng"].sample_step(B["io"], 2)
        PIPE[kind] = (gA, gB)
    torch.cuda.synchronize()
    return PIPE[kind]


def reset(seed):
    """fresh N(0,1) state of both chains (the engine's Philox init keyed by `seed`), t = 999"""
    hB = w.B // w.nchains
    for c, ch in enumerate(w.chains):
        with torch.cuda.stream(ch["stream"]):
            ch["eng"].init_state(ch["x"], seed, c * hB)
        lib.check(L.smd_set_timestep(ch["t_ptr"].data_ptr(), 999, ch["stream"].cuda_stream))
    torch.cuda.synchronize()


def run(mode, seed=4321, steps=None, timing=True):
    steps = args.steps if steps is None else steps
    pipe = mode.startswith("pipe")
    sts = plain if pipe else streams_for(mode)
    delay = float(mode.split("=")[1]) if mode.startswith("delay") else 0.0
    U = 1
    if pipe:
        U, SP, MT = parse_pipe(mode)
        graphs = pipe_graphs("pipe1" if mode == "pipe1" else mode.replace("pipe2free", "pipe2"))
        for ch in w.chains:                        # the options of THIS mode (captured graphs keep what they were captured with)
            ch["eng"].set_option("sample_split", SP)
            ch["eng"].set_option("nt256_min_tiles", MT)
    reset(seed)
    if pipe:                                       # the pipeline's prologue: chain A's first stem
        with torch.cuda.stream(sts[0]):
            w.chains[0]["eng"].sample_step(w.chains[0]["io"], 1)
        torch.cuda.synchronize()
    steps = steps // U
    mark = max(args.mark // U, 1)
    nm = steps // mark
    marks = [[torch.cuda.Event(enable_timing=True) for _ in range(nm + 1)] for _ in sts]
    clk = torch.zeros(nm * 8, dtype=torch.int32, device=dev)
    if delay > 0:
        with torch.cuda.stream(sts[1]):
            torch.cuda._sleep(int(delay * cyc_per_us))
    t0 = time.perf_counter()
    for c, s in enumerate(sts):
        marks[c][0].record(s)
    prev = [None, None]
    for k in range(steps):
        if mode == "pipe1":
            with torch.cuda.stream(sts[0]):
                graphs[0].replay()
        elif pipe:
            cur = [torch.cuda.Event(), torch.cuda.Event()]
            for c in range(2):
                with torch.cuda.stream(sts[c]):
                    if not mode.startswith("pipe2free") and prev[1 - c] is not None:
                        sts[c].wait_event(prev[1 - c])
                    graphs[c].replay()
                    cur[c].record(sts[c])
            prev = cur
        else:
            for c, ch in enumerate(w.chains):
                with torch.cuda.stream(sts[c]):
                    ch["graph"].replay()
        if (k + 1) % mark == 0:
            j = (k + 1) // mark
            for c, s in enumerate(sts):
                marks[c][j].record(s)
            lib.check(L.smd_probe_clock(clk.data_ptr() + 32 * (j - 1), 1, 20, probe_stream.cuda_stream))
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per = [[marks[c][j].elapsed_time(marks[c][j + 1]) * 1e3 / (mark * U) for j in range(nm)] for c in range(len(sts))]
    lag = [marks[0][j].elapsed_time(marks[1][j]) * 1e3 for j in range(nm + 1)] if sts[0] is not sts[1] else [0.0] * (nm + 1)
    ck = clk.cpu().view(nm, 8).numpy().astype("int64") & 0xFFFFFFFF
    mhz = [round(float(r[2]) / max(float(r[4]), 1.0) * 100.0) for r in ck]
    xcc = [int(r[0] & 0xF) for r in ck]
    rate = steps * U / wall
    return dict(mode=mode, steps_per_s=round(rate, 1), us_per_step=round(1e6 / rate, 1), host_issue_frac=round(t_issue / wall, 3),
                chainA_us=[round(v, 1) for v in per[0]], chainB_us=[round(v, 1) for v in per[1]],
                lagB_us=[round(v, 1) for v in lag], clock_mhz=mhz, probe_xcc=xcc)


# the pipelined walks compute what the plain two-chain walk computes: bitwise, after 40 steps from the same state
run("plain", seed=777, steps=40)
want = w.x.clone()
for m in ("pipe1", "pipe2", "pipe2:u4:s1", "pipe2:u2:s1:m32"):
    run(m, seed=777, steps=40)
    print(f"# {m}: state after 40 steps bitwise equal to the plain two-chain walk: {bool(torch.equal(w.x, want))}; "
          f"t = {[int(ch['t_ptr'].item()) for ch in w.chains]}", flush=True)

for rep in range(args.reps):
    for mode in args.modes.split(","):
        r = run(mode)
        r["rep"] = rep
        r["tag"] = args.tag
        print(json.dumps(r), flush=True)
for ss in STREAMS.values():
    for s in ss:
        if isinstance(s, torch.cuda.ExternalStream):
            torch.cuda.synchronize()
