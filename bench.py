"""Benchmark: denoising-steps/sec (train + sample), ddpm-mel-32seq-512, synthetic (B,32,512) latents.

  python bench.py --gpus N --steps K --warmup W

``--gpus N`` with N > 1 and no RANK / WORLD_SIZE in the environment re-launches itself as N ranks under
``torch.distributed.run`` (one process per GPU over RCCL, rendezvous on 127.0.0.1); launched by an external
``torch.distributed.run`` it reads RANK / LOCAL_RANK / WORLD_SIZE as usual.  It refuses to run when the box has fewer
than N GPUs or when WORLD_SIZE disagrees with --gpus: the line it prints always has n_gpus == --gpus.

One "step" = one train_step (q-sample + eps-net forward + backward + clip + Adam, + RCCL gradient
all-reduce when N > 1) followed by one reverse-diffusion step (eps-net forward + fused posterior
update), each on a batch of --batch sequences per GPU.  The reverse steps run as ncsn.diffusion_dynamics runs them: the batch as
two half-batch chains, software-pipelined (chain A: output stage + reverse update of step k and the stem of step k + 1, chain B: stem
and output stage of step k; U steps per captured hipGraph, a cross-chain event per replay; U = the largest divisor of --steps
up to the package default 8: 5 for --steps 20) -- DESIGN.md section 5.  Two denoising evaluations per step, so
``value`` = N * 2K / max-over-ranks(time of K steps): whole-job denoising-steps/sec (weak scaling: the
per-GPU batch is fixed).  Inputs are resident in HBM before the timed region.  The timed region is repeated
``--repeats`` times (blocks of exactly K train steps + K reverse steps, each bracketed by barrier + synchronize);
the MEDIAN block is the headline, every block's value and the relative spread are in the line.

The JSON line also carries
  roofline       the dominant kernel (DenseResBlock GEMM 8192x2048x2048, bf16 MFMA): algorithmic
                 2*M*N*K flops / average launch duration measured with HIP events on the launch
                 stream in this process, against the 2.5 PFLOP/s dense bf16 peak.
  cpu_baseline   the CPU restatement oracle (torch fp32, all host cores) timed on a bounded sample
                 of the same workload on rank 0 when N == 1 (kind "port": JAX is not installable).
  extra_configs  (N == 1) the same step on BASELINE.json's other single-GPU configurations: config 4's network
                 (ddpm-mel-32seq-512-large: L8 H16 K3), config 5's fp8 path (e4m3 GEMMs), and both together.
  sample_T1000_wall_s  (N == 1) wall time of one real ``ncsn.sample()`` call: 1000 reverse steps of B sequences incl. the
                 FiLM tables, graph capture, collection writes and metrics (the bench step replays a captured step only).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

FLOP_FWD_PER_SEQ = {"base": 1_401_159_680, "large": 2_019_426_304}     # BASELINE.md section 2 (C=512)
PEAK_BF16_TFLOPS = 2500.0                                              # MI355X_MICROARCH.md dense bf16
NET_KW = {"base": dict(), "large": dict(num_layers=8, num_heads=16, num_mlp_layers=3)}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps; the median block is reported")
    ap.add_argument("--batch", type=int, default=256, help="sequences per GPU")
    ap.add_argument("--config", choices=["base", "large"], default="base")
    ap.add_argument("--mode", choices=["both", "train", "sample"], default="both")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the large / fp8 / large+fp8 legs (N == 1)")
    ap.add_argument("--no-sampler-walk", action="store_true", help="skip the real 1000-step ncsn.sample() wall time (N == 1)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches for the sample step")
    ap.add_argument("--tr-path", type=int, default=1)
    ap.add_argument("--tuning", action="append", default=[], help="process-wide kernel knob key=value (smd_set_tuning)")
    ap.add_argument("--engine-opt", action="append", default=[], help="extra engine option key=value (A/B runs)")
    ap.add_argument("--group-wgrad", type=int, default=2, help="128-wide wgrads: 2 grouped per encoder layer, 1 grouped at the end, 0 one launch each")
    ap.add_argument("--rng-impl", choices=["philox", "threefry"], default="philox",
                    help="noise streams: in-kernel Philox (default) or jax.random-compatible threefry2x32 draws")
    ap.add_argument("--side-wgrad", type=int, default=1, help="wgrad GEMMs on the engine's side stream (0: single stream)")
    ap.add_argument("--dtype", choices=["bf16", "fp8"], default="bf16",
                    help="fp8: OCP e4m3 operands with per-row E8M0 scales for the DenseResBlock forward and dgrad GEMMs (BASELINE config 5)")
    ap.add_argument("--dp-buckets", type=int, default=1, help="N > 1 GPUs: chunks per gradient all-reduce stage (GradComm)")
    ap.add_argument("--dp-payload", choices=["fp32", "bf16"], default="fp32",
                    help="N > 1 GPUs: wire format of the gradient all-reduce (bf16 halves the xGMI bytes; off by default)")
    ap.add_argument("--dp-algorithm", choices=["all_reduce", "rs_ag"], default="all_reduce",
                    help="N > 1 GPUs: one all_reduce per bucket, or reduce_scatter + all_gather (one direct hop per phase on the xGMI mesh)")
    ap.add_argument("--dp-layer-buckets", type=int, default=0,
                    help="N > 1 GPUs: 1 = the stem slice is reduced per encoder layer behind the engine's per-layer gradient events "
                         "(6 more collectives per step; off until measured to win on RCCL)")
    ap.add_argument("--dp-dry-run", action="store_true",
                    help="one GPU: two ranks share cuda:0 over gloo and walk the multi-GPU code path (CUDA tensors, communication "
                         "stream, gradient events) with kernels of a collective's shape on the communication stream; checks that the "
                         "reduced gradients are bitwise the same with and without that load and on both ranks")
    ap.add_argument("--sampler-chains", type=int, default=2, choices=[1, 2],
                    help="graph-replayed sampling as two concurrent half-batch chains (default) or one chain")
    ap.add_argument("--sampler-unroll", type=int, default=-1,
                    help="two chains: reverse steps per captured graph of the pipelined walk (-1: the largest U <= smd_amd's own default (8) that "
                         "divides --steps; 0: two free-running one-step graphs, the round-4 arrangement)")
    ap.add_argument("--chain-opt", action="append", default=[], help="engine option key=value for the sampler's chain handles (A/B runs)")
    ap.add_argument("--no-roofline-microbench", action="store_true",
                    help="skip the back-to-back launches of the dominant kernel (profiling runs: the trace then holds the timed loops only)")
    return ap.parse_args(argv)


def log(msg: str) -> None:
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def self_launch_if_needed(a) -> None:
    """--gpus N > 1 without a launcher: become the launcher.  Never prints a line for fewer ranks than asked."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={env_world}")
        return
    if a.dp_dry_run and a.gpus <= 1:
        a.gpus = 2
        os.environ["SMD_BENCH_SHARE_DEVICE"] = "1"
        if "--gpus" not in " ".join(sys.argv[1:]):
            sys.argv += ["--gpus", "2"]
        else:
            raise SystemExit("bench.py: --dp-dry-run launches its own two ranks on one GPU; drop --gpus")
    if a.gpus <= 1:
        return
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get("SMD_BENCH_SHARE_DEVICE") == "1" and have >= 1:
        have = a.gpus                          # test hook: every rank on cuda:0 (see main)
    if have < a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} needs {a.gpus} GPUs, this box has {have}")
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log(f"launching {a.gpus} ranks: {' '.join(cmd)}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def dp_dry_run(a, rank, world, dev, dist, comm):
    """--dp-dry-run: the data-parallel train step on CUDA tensors with a communication stream that is really busy.  Three
    trajectories of `steps` train steps from the same state and draws: (1) reduction with emulated load, (2) again (bitwise
    repeatable?), (3) without the load.  All three must leave bitwise the same parameters, on both ranks."""
    import torch
    a.mode = "train"
    results = {}
    for tag, load in (("load_a", True), ("load_b", True), ("no_load", False)):
        torch.manual_seed(0)
        comm.emulate_load = load
        w = Workload(a, a.config, a.dtype, rank, world, dev, comm)
        for _ in range(a.steps):
            w.one_train()
        torch.cuda.synchronize()
        results[tag] = (w.opt.engine.params.clone(), w.opt.engine.grads.clone(), float(w.final_loss()))
        del w
        torch.cuda.empty_cache()
    same = {f"{x}=={y}": bool(torch.equal(results[x][0], results[y][0]) and torch.equal(results[x][1], results[y][1]))
            for x, y in (("load_a", "load_b"), ("load_a", "no_load"))}
    # both ranks hold the same reduced gradient and the same parameters
    digest = torch.stack([results["load_a"][0].double().sum(), results["load_a"][1].double().sum()]).to(dev)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    ranks_agree = all(bool(torch.equal(g, gathered[0])) for g in gathered)
    ok = all(same.values()) and ranks_agree
    if rank == 0:
        print(json.dumps({"dp_dry_run": {"ok": ok, "bitwise": same, "ranks_agree": ranks_agree, "steps": a.steps, "world": world,
                                         "final_loss": results["load_a"][2], "comm": comm.describe(),
                                         "what": "two ranks on one GPU over gloo, CUDA gradient tensors, communication stream running a copy + add "
                                                 "of every chunk beside the backward pass; not a measurement"}}))
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def cpu_baseline(cfg_name: str, batch: int):
    """Oracle on the host cores: 5 train steps + 10 reverse steps at the benchmark batch size (~60 s on
    64 threads, a bounded sample: the batch shrinks on slower hosts)."""
    import torch
    import ddpm_oracle as O
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    ocfg = O.NetConfig(data_channels=512, **NET_KW[cfg_name])
    p = O.init_params(ocfg, 0, torch.float32)
    betas = O.create_noise_schedule(1e-6, 0.01, 1000, "linear")
    g = torch.Generator().manual_seed(1234)
    model = O.make_model(p, ocfg)
    # size the bounded sample: probe 16 sequences, aim at <= ~75 s for 5 train steps (~3 fwd each) + 10 reverse steps
    xp = torch.randn(16, 32, 512, generator=g)
    with torch.no_grad():
        model(xp, torch.ones(16, 1, 1))
        t0 = time.perf_counter()
        model(xp, torch.ones(16, 1, 1))
        per_seq = (time.perf_counter() - t0) / 16
    est = per_seq * batch * 25
    b = batch if est <= 75 else max(16, int(batch * 75 / est) // 16 * 16)
    log(f"cpu_baseline: {per_seq * 1e3:.1f} ms/sequence-forward on {cores} threads -> sample batch {b}")
    x0 = torch.clamp(0.25 * torch.randn(b, 32, 512, generator=g), -1, 1)
    labels = torch.randint(1, 1001, (b,), generator=g).numpy()
    eps = torch.randn(b, 32, 512, generator=g)
    st = O.AdamState()
    n_train, n_rev = 5, 10
    t0 = time.perf_counter()
    for _ in range(n_train):
        p, _m, _g = O.train_step(p, ocfg, st, x0, betas, labels, eps, 1e-3, 1.0)
    t_train = (time.perf_counter() - t0) / n_train
    ts = tuple(range(999, 999 - n_rev, -1))
    zs = {t: torch.randn(b, 32, 512, generator=g) for t in ts}
    model = O.make_model(p, ocfg)
    t0 = time.perf_counter()
    with torch.no_grad():
        O.diffusion_dynamics(model, betas, eps, lambda t: zs[t], t_stop=ts[-1])
    t_sample = (time.perf_counter() - t0) / n_rev
    scale = b / batch       # a step on `batch` sequences costs batch/b times the measured one
    return {"value": round(2.0 / (t_train + t_sample) * scale, 5), "unit": "denoising-steps/sec", "cores": cores,
            "kind": "port",
            "sample": f"oracle/ddpm_oracle.py torch-CPU fp32 ({cores} threads): {n_train} train_steps + {n_rev} reverse steps on "
                      f"{b} of the {batch} sequences ({t_train:.2f} s/train-step, {t_sample:.2f} s/sample-step), "
                      f"rate scaled by {b}/{batch}; a restatement of the reference, not JAX/XLA",
            "train_steps_per_sec": round(scale / t_train, 5), "sample_steps_per_sec": round(scale / t_sample, 5)}


_CHAIN_STREAMS = {}     # device -> the two chain streams, created once per process: HIP maps streams onto a few hardware
                        # queues round-robin, and a LATER pair of fresh streams can land on one queue (the two chains then
                        # serialise: measured 2.2x per sample step on the second workload of a process)


def chain_streams(torch, dev, n):
    if dev not in _CHAIN_STREAMS:
        _CHAIN_STREAMS[dev] = [torch.cuda.Stream(device=dev) for _ in range(2)]
    return _CHAIN_STREAMS[dev][:n]


class Workload:
    """One configuration's resident state: model + optimiser + the sampler chains, and the two timed loops."""

    def __init__(self, a, cfg_name: str, dtype: str, rank: int, world: int, dev: str, comm):
        import numpy as np
        import torch
        import smd_amd.lib as lib
        import smd_amd.ncsn as N
        import smd_amd.schedule as S
        from smd_amd.engine import NetConfig
        from smd_amd.trainer import create_optimizer, train_step
        self.a, self.rank, self.world, self.dev, self.comm = a, rank, world, dev, comm
        self.cfg_name, self.dtype = cfg_name, dtype
        self.N, self.torch = N, torch
        cfg = NetConfig(architecture="TransformerDDPM", data_channels=512, seq_len=32, num_timesteps=1000, dtype=dtype,
                        **NET_KW[cfg_name])
        self.cfg = cfg
        self.model = model = N.Model(cfg, dev, seed=0)
        model._chain_streams = chain_streams(torch, dev, 2)        # every model of this process walks its chains on the same pair
        self.betas = betas = S.create_noise_schedule(1e-6, 0.01, 1000, "linear")
        B = self.B = a.batch
        if comm is not None:
            comm.broadcast_params(model.params)
            model.engine.refresh_weights()
        # ---- resident inputs
        g = torch.Generator().manual_seed(1234 + rank)
        self.x0 = torch.clamp(0.25 * torch.randn(B, 32, 512, generator=g), -1, 1).to(dev)
        self.opt = opt = create_optimizer(model, 1e-3, ema=False)                      # configs/ddpm-base.cfg: --ema=False
        opt.engine.set_option("tr_path", a.tr_path)
        opt.engine.set_option("side_wgrad", a.side_wgrad)
        opt.engine.set_option("group_wgrad", a.group_wgrad)
        for kv in a.engine_opt:
            k, _, v = kv.partition("=")
            opt.engine.set_option(k, int(v))
        self.key = N.make_key(0, a.rng_impl)
        self._train_step = train_step

        # ---- sampler state (replicas: each rank walks its own B sequences).  With graph replay the batch is walked as two
        # concurrent half-batch chains on two streams, exactly as ncsn.diffusion_dynamics does (--sampler-chains 1: one chain).
        # (N.sampler_chain_sizes: uneven splits in multiples of 8 sequences, e.g. 1000 -> 504 + 496; the bench never pads)
        sizes, pad = N.sampler_chain_sizes(model, B, graphed=(a.sampler_chains == 2 and not a.no_graph))
        if pad:
            sizes = [B]
        self.sizes = sizes
        self.nchains = nchains = len(sizes)
        engines = model.chain_engines(2) if nchains == 2 else [model.engine]
        offs = self.offs = [0] + [sum(sizes[:i + 1]) for i in range(len(sizes) - 1)]
        self.x = x = torch.empty(B, 32, 512, device=dev)
        self.chains = []
        nk_d = None
        if a.rng_impl == "threefry":          # the reference's per-iteration noise keys; normals drawn inside the fused step
            import smd_amd.jax_random as J
            _ik, nk = J.sampler_key_tables(N.make_key(7, "threefry"), 1000)
            nk_d = torch.from_numpy(nk.view(np.int32).copy()).to(dev)
        self._nk_d = nk_d
        for c, eng in enumerate(engines):
            for kv in a.chain_opt:
                k, _, v = kv.partition("=")
                eng.set_option(k, int(v))
            eng.set_schedule(betas, with_sampler=True)
            hB = sizes[c]
            eng.bind(hB, training=False)
            eng.prepare_sampler()
            xc = x[offs[c]:offs[c] + hB]
            eng.init_state(xc, 4321, rank * B + offs[c])
            ch = dict(eng=eng, x=xc, t_ptr=torch.tensor([999], dtype=torch.int32, device=dev),
                      metrics=torch.zeros(1000, hB, 3, device=dev), coll=torch.zeros(41, hB, 32, 512, device=dev),
                      graph=None, stream=chain_streams(torch, dev, nchains)[c] if nchains > 1 else None)
            io = lib.SampleIO()
            io.x, io.t_ptr = xc.data_ptr(), ch["t_ptr"].data_ptr()
            io.seed_lo, io.seed_hi, io.sample_offset = 7, 0, rank * B + offs[c]
            io.metrics_partial, io.collection, io.slot_table = ch["metrics"].data_ptr(), ch["coll"].data_ptr(), eng.slot_table.data_ptr()
            if nk_d is not None:
                io.tf_noise_keys, io.tf_n_total, io.tf_t0 = nk_d.data_ptr(), world * B * 32 * 512, 999
            ch["io"] = io
            self.chains.append(ch)
        self.walked = 0
        # the pipelined two-chain walk of ncsn.diffusion_dynamics: chain 0 = (output stage, next stem) x U per graph, chain 1 =
        # (stem, output stage) x U, every replay waiting for the other chain's previous replay; U must divide the block length
        self.unroll = 0
        if nchains == 2:
            want = N._sampler_pipeline_unroll() if a.sampler_unroll < 0 else a.sampler_unroll
            if want > 0:
                self.unroll = max(u for u in range(1, 9) if u <= want and a.steps % u == 0)
        self._ev = [[torch.cuda.Event() for _ in range(2)] for _ in range(2)]
        self._replays = 0

    # ---- the two halves of a step
    def one_train(self):
        self._train_step(self.N.diffusion_loss, self.x0, self.opt, self.betas, self.key, 1e-3, grad_clip=1.0, comm=self.comm,
                         lr_gamma=0.98, lr_interval=10000, sample_offset=self.rank * self.B, global_batch=self.B * self.world)

    def _reset_t(self):
        """Restart the walk: t = 999 AND a fresh N(0,1) state, as a new diffusion_dynamics call would (the step time depends
        on the data: replaying t = 999 ... on a finished sample runs a few per cent faster than a real walk)."""
        torch = self.torch
        import smd_amd.lib as lib
        self.restarts = getattr(self, "restarts", 0) + 1
        for c, ch in enumerate(self.chains):   # the engine's own kernels on the chain's stream (no framework fill on the replay stream)
            st = ch["stream"] if ch["stream"] is not None else torch.cuda.current_stream()
            with torch.cuda.stream(st):
                ch["eng"].init_state(ch["x"], 4321 + self.restarts, self.rank * self.B + self.offs[c])
            lib.check(lib.get_lib().smd_set_timestep(ch["t_ptr"].data_ptr(), 999, st.cuda_stream))
        self.walked = 0
        if self.unroll and self.chains[0]["graph"] is not None:       # the pipeline's prologue: chain 0's first stem
            ch = self.chains[0]
            with torch.cuda.stream(ch["stream"]):
                ch["eng"].sample_step(ch["io"], 1)
            self._replays = 0

    def sample_steps(self, n: int):
        """exactly `n` reverse steps of every chain (n a multiple of the graphs' unroll once they are captured)"""
        torch = self.torch
        per = self.unroll if (self.unroll and self.chains[0]["graph"] is not None) else 1
        assert n % per == 0, (n, per)
        for _ in range(n // per):
            # a reverse walk has T = 1000 iterations: start over before t would pass 0 (the kernels also refuse t < 0)
            if self.walked + per > 990:
                self._reset_t()
            self.walked += per
            for c, ch in enumerate(self.chains):
                if ch["stream"] is None:
                    self._run_chain(ch)
                    continue
                with torch.cuda.stream(ch["stream"]):
                    if per == self.unroll and self.unroll:
                        if self._replays > 0:
                            ch["stream"].wait_event(self._ev[1 - c][(self._replays - 1) & 1])
                        ch["graph"].replay()
                        self._ev[c][self._replays & 1].record(ch["stream"])
                    else:
                        self._run_chain(ch)
            self._replays += 1

    def one_sample(self):
        self.sample_steps(1)

    @staticmethod
    def _run_chain(ch):
        if ch["graph"] is not None:
            ch["graph"].replay()
        else:
            ch["eng"].sample_step(ch["io"])

    def warm_up(self, n: int, do_train: bool, do_sample: bool):
        torch = self.torch
        for _ in range(max(n, 1)):
            if do_train:
                self.one_train()
            if do_sample:
                self.one_sample()
        if do_sample and not self.a.no_graph:
            # weights change under training: tables/operand pack are refreshed per sampling run in the real
            # sampler; here the step content is what is timed, so one captured step per chain is replayed.
            torch.cuda.synchronize()
            U = self.unroll
            for c, ch in enumerate(self.chains):
                s = ch["stream"] if ch["stream"] is not None else torch.cuda.Stream(device=self.dev)
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    ch["eng"].sample_step(ch["io"])
                s.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    if U:
                        for _ in range(U):
                            for part in ((2, 1) if c == 0 else (1, 2)):
                                ch["eng"].sample_step(ch["io"], part)
                    else:
                        ch["eng"].sample_step(ch["io"])
                ch["graph"] = g
            torch.cuda.synchronize()
            self._reset_t()
            self.sample_steps(max(U, 1))
        torch.cuda.synchronize()
        self._reset_t()
        torch.cuda.synchronize()

    def timed_block(self, steps: int, do_train: bool, do_sample: bool, barrier):
        """exactly `steps` train steps, then exactly `steps` reverse steps, each bracketed by barrier + synchronize"""
        barrier()
        t0 = time.perf_counter()
        if do_train:
            for _ in range(steps):
                self.one_train()
        barrier()
        t1 = time.perf_counter()
        if do_sample:
            self.sample_steps(steps)
        barrier()
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1

    def final_loss(self) -> float:
        return float(self.opt.engine.loss_per_sample().mean())

    def sampler_walk(self, num_samples=None):
        """One real 1000-step ``ncsn.sample`` call on ``num_samples`` (default B) sequences (twice: the first call also pays the
        allocations).  Returns (wall times, the arrangement ncsn.diffusion_dynamics chose)."""
        torch, N = self.torch, self.N
        n = self.B if num_samples is None else int(num_samples)
        out = []
        for i in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gen, coll, _m = N.sample(self.model, self.betas, N.make_key(11 + i, self.a.rng_impl), (32, 512), num_samples=n,
                                     sampling="ddpm", sample_offset=self.rank * n)
            torch.cuda.synchronize()
            out.append(time.perf_counter() - t0)
            assert bool(torch.isfinite(gen).all()) and tuple(coll.shape) == (41, n, 32, 512)
            del gen, coll
        arr = dict(getattr(self.model, "sampler_arrangement", {}))
        self.model.drop_sampler_cache()
        return out, arr


def run_blocks(w: Workload, a, dist, do_train, do_sample, steps, warmup, repeats):
    """warm-up, then `repeats` timed blocks; per block the MAX over ranks of the train and the sample time"""
    import torch
    world = w.world

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    w.warm_up(warmup, do_train, do_sample)
    blocks = []
    for _ in range(max(repeats, 1)):
        blocks.append(w.timed_block(steps, do_train, do_sample, barrier))
    tt = torch.tensor(blocks, dtype=torch.float64, device=w.dev)            # [repeats][2]
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return [(float(r[0]), float(r[1])) for r in tt.cpu()]


def summarise(blocks, steps, world, do_train, do_sample, fwd_flops):
    n_eval = (steps if do_train else 0) + (steps if do_sample else 0)
    vals = [world * n_eval / (bt + bs) for bt, bs in blocks]
    order = sorted(range(len(blocks)), key=lambda i: blocks[i][0] + blocks[i][1])
    mid = order[len(order) // 2]                    # the median block (by total time) is the reported one
    t_train, t_sample = blocks[mid]
    med = vals[mid]
    out = {"value": round(med, 3), "t_train": t_train, "t_sample": t_sample,
           "block_values": [round(v, 2) for v in vals],
           "spread": round((max(vals) - min(vals)) / med, 4),
           "train_steps_per_sec": round(world * steps / t_train, 3) if do_train else None,
           "sample_steps_per_sec": round(world * steps / t_sample, 3) if do_sample else None,
           "step_frac_train": round(3 * fwd_flops * steps / t_train / 1e12 / PEAK_BF16_TFLOPS, 4) if do_train else None,
           "step_frac_sample": round(fwd_flops * steps / t_sample / 1e12 / PEAK_BF16_TFLOPS, 4) if do_sample else None}
    if len(vals) > 1:
        out["block_stdev_rel"] = round(statistics.pstdev(vals) / med, 4)
    return out


def roofline_microbench(a, cfg, dev, dtype):
    """DenseResBlock GEMM (R x 2048 x 2048), HIP events on this stream.  The operands rotate through four buffer sets
    (4 x 75 MB > the 32 MiB of L2) so that, as inside the step, a launch does not find its A panel in the XCD's L2 from the
    launch before; both epilogue forms of the block are timed: fc1 (bias -> bf16) is `achieved`, fc2 (bias + fp32 residual
    -> fp32) is reported next to it."""
    import torch
    import smd_amd.lib as lib
    L = lib.get_lib()
    R, M = a.batch * 32, cfg.mlp_dims
    NSET = 4
    As = [torch.randn(R, M, device=dev).to(torch.bfloat16) for _ in range(NSET)]
    Wt = (torch.randn(M, M, device=dev) * 0.02).to(torch.bfloat16)
    bias = torch.zeros(M, device=dev)
    outs = [torch.empty(R, M, dtype=torch.bfloat16, device=dev) for _ in range(NSET)]
    res = [torch.randn(R, M, device=dev) for _ in range(2)]
    outf = [torch.empty(R, M, device=dev) for _ in range(2)]
    st = torch.cuda.current_stream().cuda_stream

    def call_b(i):
        lib.check(L.smd_gemm_bf16_nt(As[i % NSET].data_ptr(), M, Wt.data_ptr(), M, R, M, M, bias.data_ptr(), 0, None, 0,
                                     None, 0, outs[i % NSET].data_ptr(), M, st))

    def call_r(i):
        lib.check(L.smd_gemm_bf16_nt(As[i % NSET].data_ptr(), M, Wt.data_ptr(), M, R, M, M, bias.data_ptr(), 0,
                                     res[i % 2].data_ptr(), M, outf[i % 2].data_ptr(), M, None, 0, st))

    def timed(call, reps=48):
        for i in range(4):
            call(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ms = min(timed(call_b) for _ in range(3))          # three rounds of 48 launches, the best round's average
    ms_r = min(timed(call_r) for _ in range(2))
    ms8 = None
    if dtype == "fp8":
        # the e4m3 form of the same GEMM (v_mfma_scale_f32_32x32x64_f8f6f4, per-row E8M0 scales): what this run's engine
        # launches for the DenseResBlock FORWARD GEMMs; its backward GEMMs are the bf16 kernel timed above
        q8 = [torch.empty(R, M, dtype=torch.uint8, device=dev) for _ in range(NSET)]
        s8 = [torch.empty(R, dtype=torch.int32, device=dev) for _ in range(NSET)]
        w8, ws8 = torch.empty(M, M, dtype=torch.uint8, device=dev), torch.empty(M, dtype=torch.int32, device=dev)
        for i in range(NSET):
            lib.check(L.smd_quantize_rows_e4m3(As[i].data_ptr(), M, R, M, q8[i].data_ptr(), s8[i].data_ptr(), st))
        lib.check(L.smd_quantize_rows_e4m3(Wt.data_ptr(), M, M, M, w8.data_ptr(), ws8.data_ptr(), st))

        def call_8(i):
            lib.check(L.smd_gemm_e4m3_nt(q8[i % NSET].data_ptr(), M, s8[i % NSET].data_ptr(), w8.data_ptr(), M, ws8.data_ptr(),
                                         R, M, M, bias.data_ptr(), None, 0, None, 0, outs[i % NSET].data_ptr(), M, st))
        ms8 = min(timed(call_8) for _ in range(2))
    tf = 2.0 * R * M * M / (ms * 1e-3) / 1e12
    # HBM-side bytes per launch of this kernel: PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs)
    # are collected by tools/collect_profiles.sh and committed as profiles/pmc_gemm_nt256.json (corrected as
    # MI355X_MICROARCH.md prescribes); a profiler cannot wrap the timed run itself.
    traffic = in_step_us = src = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_gemm_nt256.json")))
        if pm.get("shape") == [R, M, M]:
            traffic = pm["traffic_bytes"]
            in_step_us = pm.get("in_step_us")
            src = pm.get("collected")
    except Exception:
        traffic = None
    roof = {"bound": "mfma", "kernel": "gemm_nt256_kernel", "shape": [R, M, M], "achieved": round(tf, 1),
            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4),
            "avg_launch_ms": round(ms, 5), "avg_launch_ms_fp32_residual_form": round(ms_r, 5),
            "operands": f"{NSET} rotating buffer sets (no L2-hot A panel), bias->bf16 epilogue",
            "in_step_us_from_committed_trace": in_step_us,
            "traffic": traffic,
            "traffic_note": "bytes per launch at L2's memory side from committed rocprofv3 PMC passes "
                            "(profiles/pmc_gemm_nt256.json, collected by tools/collect_profiles.sh"
                            + (f": {src}" if src else "") + "); algorithmic bytes = 75.5e6"}
    if ms8 is not None:
        tf8 = 2.0 * R * M * M / (ms8 * 1e-3) / 1e12
        roof["e4m3_form"] = {"kernel": "gemm_nt256_kernel<0, true> (v_mfma_scale_f32_32x32x64_f8f6f4, per-row E8M0 scales)",
                             "achieved": round(tf8, 1), "peak": 5000.0, "unit": "TFLOP/s", "frac": round(tf8 / 5000.0, 4),
                             "avg_launch_ms": round(ms8, 5)}
    # the clock-limited ceiling of this box on the same instruction with random operands and NO data movement
    # (tools/mfma_peak.hip; MI355X_MICROARCH.md "DVFS give-back"): context for `frac`, not a replacement for `peak`
    probe = os.path.join(ROOT, "tools", "mfma_peak")
    if os.path.exists(probe):
        try:
            import subprocess
            txt = subprocess.run([probe], capture_output=True, text=True, timeout=60).stdout
            vals = [float(l.rsplit(":", 1)[1].split()[0]) for l in txt.splitlines() if l.startswith("uniform") and "staggered" in l]
            if vals:
                roof["mfma_only_random_operands_tflops"] = vals[0]
        except Exception:
            pass
    return roof


def main():
    a = parse()
    self_launch_if_needed(a)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, (world, a.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    # test hook (tests/test_gpu_bench_config.py): SMD_BENCH_SHARE_DEVICE=1 puts every rank on cuda:0 with the gloo backend, so the
    # multi-rank code path of this script (sharded inputs, two-stage gradient all-reduce, barriers, max over ranks) can run on a
    # one-GPU box; the line it prints says so and is not a measurement
    share = os.environ.get("SMD_BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = 0
    if torch.cuda.device_count() < (local_rank + 1):
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank}, this box has {torch.cuda.device_count()} GPUs")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))

    import smd_amd.lib as lib
    from smd_amd.trainer import GradComm, assert_distinct_devices, gather_device_identities

    for kv in a.tuning:
        k, _, v = kv.partition("=")
        lib.set_tuning(k, int(v))                  # (bumps lib.tuning_epoch(): cached sampler graphs are keyed on it)
    comm = (GradComm(buckets=a.dp_buckets, payload=a.dp_payload, algorithm=a.dp_algorithm, layer_buckets=bool(a.dp_layer_buckets),
                     measure_exposed=True, emulate_load=a.dp_dry_run)
            if world > 1 else None)
    # which PHYSICAL device every rank sits on (UUID / PCI address), gathered over the job's own process group: N ranks must show
    # N distinct devices or no line is printed (the SMD_BENCH_SHARE_DEVICE test hook is the one exception, and says so in `data`)
    devices = gather_device_identities(dist if world > 1 else None)
    assert_distinct_devices(devices, allow_shared=share)
    if a.dp_dry_run:
        return dp_dry_run(a, rank, world, dev, dist, comm)
    do_train, do_sample = a.mode in ("both", "train"), a.mode in ("both", "sample")

    w = Workload(a, a.config, a.dtype, rank, world, dev, comm)
    log(f"rank {rank}: model + buffers ready, warming up")
    if comm is not None:
        comm.exposed_comm_us()                                     # drop what model set-up recorded
    c0 = comm.collectives if comm is not None else 0
    blocks = run_blocks(w, a, dist, do_train, do_sample, a.steps, a.warmup, a.repeats)
    dp_stats = {}
    if comm is not None and do_train:
        n_train = (max(a.warmup, 1) + a.steps * max(a.repeats, 1))
        ex = comm.exposed_comm_us()
        dp_stats = {"collectives_per_step": round((comm.collectives - c0) / n_train, 2),
                    "exposed_comm_us": None if ex is None else round(ex, 1)}
    sample_desc = (f"hipGraph replay, 2 half-batch chains pipelined (chain A: output stage + next stem, chain B: stem + output stage; "
                   f"{w.unroll} steps per graph, cross-chain event per replay)" if w.unroll
                   else "hipGraph replay, 2 free-running half-batch chains")
    head_nchains = w.nchains
    fwd = FLOP_FWD_PER_SEQ[a.config] * a.batch
    head = summarise(blocks, a.steps, world, do_train, do_sample, fwd)
    loss = w.final_loss() if do_train else float("nan")
    log(f"rank {rank}: {a.repeats} blocks of {a.steps} steps: {head['block_values']} denoising-steps/s")

    roof = None
    step_fracs = {"step_frac_train": head["step_frac_train"], "step_frac_sample": head["step_frac_sample"]}
    if rank == 0 and not a.no_roofline_microbench:
        roof = roofline_microbench(a, w.cfg, dev, a.dtype)
        roof.update(step_fracs)
    elif rank == 0:
        roof = {"bound": "mfma", "kernel": "gemm_nt256_kernel", "achieved": None, "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": None, "traffic": None, **step_fracs}

    # ---- single-GPU extras (rank 0 of a one-rank job; outside the headline's timed region)
    walk = None
    extra = None
    if world == 1 and do_sample and not a.no_sampler_walk:
        ws, arr = w.sampler_walk()
        walk = {"sample_T1000_wall_s": round(ws[-1], 4), "first_call_wall_s": round(ws[0], 4),
                "steps_per_sec": round(1000.0 / ws[-1], 1), "arrangement": arr,
                "what": f"ncsn.sample(sampling='ddpm', num_samples={a.batch}): init draw, FiLM tables for 1000 levels, operand refresh, "
                        "graph capture, 1000 reverse steps with collection + metrics, collate; second call of two"}
        log(f"real 1000-step sampler walk: {ws[-1]:.3f} s ({ws[0]:.3f} s first call)")
        # the reference's own default sampling shape: sample_ncsn.py:54 sample_size = 1000, drawn as ONE batch (train_ncsn.py:540)
        ws2, arr2 = w.sampler_walk(1000)
        walk["num_samples_1000"] = {"sample_T1000_wall_s": round(ws2[-1], 4), "first_call_wall_s": round(ws2[0], 4),
                                    "steps_per_sec": round(1000.0 / ws2[-1], 1),
                                    "seq_steps_per_sec": round(1000.0 * 1000 / ws2[-1], 1), "arrangement": arr2}
        walk["seq_steps_per_sec"] = round(1000.0 * a.batch / ws[-1], 1)
        log(f"real 1000-step sampler walk of 1000 sequences: {ws2[-1]:.3f} s ({arr2.get('chain_sizes')})")
    if world == 1 and not a.no_extra_configs and a.mode == "both":
        extra = {}
        todo = [(c, d) for c in ("base", "large") for d in ("bf16", "fp8") if (c, d) != (a.config, a.dtype)]
        del w
        torch.cuda.empty_cache()
        for c, d in todo:
            we = Workload(a, c, d, rank, world, dev, None)
            b2 = run_blocks(we, a, dist, True, True, a.steps, min(a.warmup, 3), min(a.repeats, 3))
            s = summarise(b2, a.steps, 1, True, True, FLOP_FWD_PER_SEQ[c] * a.batch)
            name = {"base": "base", "large": "large"}[c] + ("_fp8" if d == "fp8" else "")
            extra[name] = {k: s[k] for k in ("value", "train_steps_per_sec", "sample_steps_per_sec", "step_frac_train",
                                             "step_frac_sample", "block_values", "spread")}
            extra[name]["workload"] = (f"ddpm-{'mel' if d == 'bf16' else 'multi'}-32seq-512{'-large' if c == 'large' else ''}.cfg network, "
                                       f"batch={a.batch}, {d}")
            log(f"extra config {name}: {s['value']} denoising-steps/s")
            if c == "large" and not a.no_sampler_walk:
                # BASELINE config 4's own workload: a 1000-step reverse walk of the large net, B sequences per GPU
                wl, arr = we.sampler_walk()
                extra[name]["sampler_walk"] = {"sample_T1000_wall_s": round(wl[-1], 4), "first_call_wall_s": round(wl[0], 4),
                                               "steps_per_sec": round(1000.0 / wl[-1], 1),
                                               "seq_steps_per_sec": round(1000.0 * a.batch / wl[-1], 1), "arrangement": arr}
                log(f"extra config {name}: 1000-step walk {wl[-1]:.3f} s")
            del we
            torch.cuda.empty_cache()
        w = None

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log(f"GPU: train {head['t_train'] / a.steps * 1e3:.3f} ms/step, sample {head['t_sample'] / a.steps * 1e3:.3f} ms/step; CPU baseline next")
        cpu = cpu_baseline(a.config, a.batch)

    if rank == 0:
        n_eval = (a.steps if do_train else 0) + (a.steps if do_sample else 0)
        total = head["t_train"] + head["t_sample"]
        B = a.batch
        nch = head_nchains
        out = {
            "metric": "denoising-steps/sec (train+sample), ddpm-mel-32seq-512",
            "value": head["value"], "unit": "denoising-steps/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * total / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if a.dtype == "bf16" else "fp8 (e4m3 DenseResBlock forward + dgrad GEMMs) + bf16",
            "data": "synthetic" if not share else "synthetic; TEST RUN: all ranks share cuda:0 over gloo (SMD_BENCH_SHARE_DEVICE), not a measurement",
            "config": {"workload": f"ddpm-mel-32seq-512{'-large' if a.config == 'large' else ''}.cfg, batch={B}/GPU synthetic "
                                   f"(32,512) latents, random-init weights; step = 1 train_step + 1 reverse step",
                       "global_batch": B * world, "seq_len": 32, "parallelism": f"dp{world}", "mode": a.mode,
                       "devices": devices,
                       **({"dp": {**comm.describe(), "devices": devices, "distinct_devices": len(set(devices)), "collectives_per_step": dp_stats.get("collectives_per_step"),
                                  "exposed_comm_us": dp_stats.get("exposed_comm_us"),
                                  "what": "exposed_comm_us = mean stall of the compute stream at GradComm.wait() per train step (HIP events)"}}
                          if world > 1 else {}),
                       "sample_step": "eager" if a.no_graph else (sample_desc if nch == 2 else "hipGraph replay"),
                       "rng": a.rng_impl},
            "repeats": a.repeats, "block_values": head["block_values"], "spread": head["spread"],
            "reported_block": "median of the blocks by total time; each block = exactly `steps` train steps + `steps` reverse steps",
            "train_steps_per_sec": head["train_steps_per_sec"], "sample_steps_per_sec": head["sample_steps_per_sec"],
            "seq_steps_per_sec": round(world * n_eval * B / total, 1),
            "train_tflops": round(3 * fwd * a.steps / head["t_train"] / 1e12, 1) if do_train else None,
            "sample_tflops": round(fwd * a.steps / head["t_sample"] / 1e12, 1) if do_sample else None,
            "final_loss": loss,
            "roofline": roof, "cpu_baseline": cpu, "extra_configs": extra, "sampler_walk": walk,
            "sample_T1000_wall_s": walk["sample_T1000_wall_s"] if walk else None,
        }
        assert out["n_gpus"] == a.gpus
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
