"""CPU tests of the host logic: flag surface, configs, data transforms, checkpoint naming and the
data-parallel gradient exchange (gloo, world_size 2)."""
import glob
import os
import pickle

import numpy as np
import pytest
import torch

import ddpm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def test_flag_surface_defaults_match_reference_table():
    import smd_amd.flags as F
    fl = F.make_flags(include_sample=True)
    # SURVEY section 5 "Flag surface to keep"
    expect = dict(seed=0, loss="dsm", continuous_noise=True, learning_rate=3e-4, batch_size=128, epochs=10,
                  max_steps=None, early_stopping=False, grad_clip=1.0, lr_gamma=0.98, lr_schedule_interval=10000,
                  architecture="TransformerDDPM", num_layers=6, num_heads=8, num_mlp_layers=2, mlp_dims=2048,
                  sigma_begin=1.0, sigma_end=1e-2, schedule_type="geometric", num_sigmas=15, ld_steps=100,
                  ld_epsilon=2e-6, sampling="ald", ema=True, mu=0.999, denoise=True, data_shape=["2"], problem="toy",
                  dataset="./output/mix2d", pca_ckpt="", slice_ckpt="", dim_weights_ckpt="", normalize=True,
                  logging_freq=100, snapshot_freq=5000, snapshot_sampling=True, eval_samples=3000,
                  checkpoints_to_keep=50, save_ckpt=True, model_dir="./save/ncsn", verbose=True, sample_seed=1,
                  sampling_dir="samples", sample_size=1000, compute_metrics=False, compute_final_only=False,
                  flush=True, animate=False, infill=False, interpolate=False)
    for k, v in expect.items():
        assert getattr(fl, k) == v, k
    assert len(F.TRAIN_FLAGS) == 41 and len(F.SAMPLE_FLAGS) == 9


def test_flag_syntax_variants(tmp_path):
    import smd_amd.flags as F
    inner = tmp_path / "inner.cfg"
    inner.write_text("--ema=False\n--num_layers=3\n\n# comment\n--data_shape=32,512  \n")
    outer = tmp_path / "outer.cfg"
    outer.write_text(f"--flagfile={inner}\n--num_layers=8\n--nosnapshot_sampling\n--continuous_noise\n")
    fl = F.make_flags(True)
    rest = fl.parse([f"--flagfile={outer}", "--learning_rate", "1e-3", "--noverbose", "--max_steps=7", "pos"])
    assert rest == ["pos"]
    assert fl.ema is False and fl.num_layers == 8 and fl.data_shape == ["32", "512"]      # later wins
    assert fl.snapshot_sampling is False and fl.verbose is False and fl.max_steps == 7 and fl.learning_rate == 1e-3
    assert fl.is_present("ema") and not fl.is_present("mu")
    for bad in (["--nosuchflag=1"], ["--loss=foo"], ["--num_layers=abc"], ["--flagfile=/no/such/file"], ["--noseed"]):
        with pytest.raises(F.FlagError):
            F.make_flags(True).parse(bad)
    assert "--noema" in fl.flags_into_string()


def test_shipped_configs_parse_and_match_reference():
    import smd_amd.flags as F
    ours = sorted(glob.glob(os.path.join(ROOT, "configs", "ddpm-*.cfg")))
    assert len(ours) >= 6
    cwd = os.getcwd()
    for f in ours:
        os.chdir(ROOT)
        a = F.make_flags(True)
        a.parse([f"--flagfile=configs/{os.path.basename(f)}"])
        assert a.loss == "ddpm" and a.sampling == "ddpm" and a.num_sigmas == 1000 and a.schedule_type == "linear"
        ref = os.path.join(REF, "configs", os.path.basename(f))
        if os.path.exists(ref):
            os.chdir(REF)
            b = F.make_flags(True)
            b.parse([f"--flagfile=configs/{os.path.basename(f)}"])
            assert a.flag_values_dict() == b.flag_values_dict(), f
    os.chdir(cwd)
    if os.path.isdir(os.path.join(REF, "configs")):       # every reference flagfile parses (surface test)
        os.chdir(REF)
        for f in glob.glob("configs/**/*.cfg", recursive=True):
            if os.path.basename(f).startswith("mdn"):     # train_mdn.py flagfiles: a different flag set
                continue
            F.make_flags(True).parse([f"--flagfile={f}"])
        os.chdir(cwd)


def test_data_transforms_match_oracle(tmp_path):
    from smd_amd import data
    rng = np.random.default_rng(0)
    x = rng.uniform(-3, 5, size=(4, 32, 512)).astype(np.float32)
    idx = np.sort(rng.choice(512, 42, replace=False))
    s = data.slice_transform(x, idx)
    assert s.shape == (4, 32, 42) and np.array_equal(s, x[..., idx])
    n = data.normalize_dataset(s, -3.0, 5.0)
    assert np.allclose(n, O.normalize_dataset(s, -3.0, 5.0)) and n.min() >= -1 and n.max() <= 1
    np.random.seed(7)
    ours = data.inverse_data_transform(n, True, None, -3.0, 5.0, idx)
    np.random.seed(7)
    filler = np.random.randn(4, 32, 512)
    ref = O.inverse_data_transform(n, True, -3.0, 5.0, idx, filler=filler)
    assert ours.shape == (4, 32, 512) and ours.dtype == np.float64 and np.array_equal(ours, ref)
    assert np.allclose(ours[..., idx], s, atol=1e-5)
    p = tmp_path / "a" / "b.pkl"
    data.save(ours, str(p))
    with open(p, "rb") as f:
        assert np.array_equal(pickle.load(f), ours)
    assert np.array_equal(data.load(str(p)), ours)
    ds = data.SyntheticLatents((32, 512), 64, 16)
    assert ds.examples == 4 and ds.min == -1.0 and ds.max == 1.0
    b = next(iter(ds))
    assert tuple(b.shape) == (16, 32, 512) and float(b.abs().max()) <= 1.0
    a0 = data.SyntheticLatents((32, 8), 64, 16, rank=0, world_size=2)
    a1 = data.SyntheticLatents((32, 8), 64, 16, rank=1, world_size=2)
    assert a0.examples == 2 and not torch.equal(next(iter(a0)), next(iter(a1)))


def test_training_split_is_reshuffled_every_epoch_and_ranks_stay_disjoint():
    """utils/data_utils.py:159-183 (shuffle=True for the training files + an 8*batch buffer, new order every epoch)."""
    from smd_amd import data
    arr = np.arange(96, dtype=np.float32).reshape(96, 1) * np.ones((1, 4), np.float32)
    ds = data.ArrayLatents(arr, 8, shuffle=True, seed=5)
    e0 = torch.cat(list(ds))[:, 0]
    e1 = torch.cat(list(ds))[:, 0]
    assert ds.examples == 12 and len(e0) == 96
    assert sorted(e0.tolist()) == list(range(96)) and sorted(e1.tolist()) == list(range(96))     # a permutation each
    assert not torch.equal(e0, e1) and not torch.equal(e0, torch.arange(96.0))                   # and a new one each epoch
    again = data.ArrayLatents(arr, 8, shuffle=True, seed=5)
    assert torch.equal(torch.cat(list(again))[:, 0], e0)                                         # seeded
    assert not torch.equal(torch.cat(list(data.ArrayLatents(arr, 8, shuffle=True, seed=6)))[:, 0], e0)
    halves = [data.ArrayLatents(arr, 8, rank=r, world_size=2, shuffle=True, seed=5) for r in range(2)]
    for _epoch in range(2):
        a, b = (torch.cat(list(h))[:, 0] for h in halves)
        assert len(a) == len(b) == 48 and sorted(a.tolist() + b.tolist()) == list(range(96))     # global permutation, disjoint
    plain = data.ArrayLatents(arr, 8)
    assert torch.equal(torch.cat(list(plain))[:, 0], torch.arange(96.0))                         # eval split: file order
    assert torch.equal(torch.cat(list(plain))[:, 0], torch.arange(96.0))


class _Rotation:
    """Stand-in for the pickled sklearn PCA of --pca_ckpt: an orthogonal map around a mean."""

    def __init__(self, d, seed):
        q, _ = np.linalg.qr(np.random.default_rng(seed).normal(size=(d, d)))
        self.q, self.mean = q.astype(np.float32), np.float32(0.3)

    def transform(self, x):
        return (x - self.mean) @ self.q

    def inverse_transform(self, x):
        return x @ self.q.T + self.mean


def test_open_dataset_applies_pca_slice_normalise_in_reference_order(tmp_path):
    """input_pipeline.py:144,159-181,189-208: PCA -> dim weights / slice -> per-split min/max -> normalise."""
    from smd_amd import data
    rng = np.random.default_rng(1)
    tr = rng.normal(size=(40, 16)).astype(np.float32)
    ev = rng.normal(size=(24, 16)).astype(np.float32) * 2
    np.save(tmp_path / "train.npy", tr)
    np.save(tmp_path / "eval.npy", ev)
    pca = _Rotation(16, 3)
    data.save(pca, str(tmp_path / "pca.pkl"))
    idx = np.array([1, 4, 5, 9, 14])
    w = rng.uniform(0.5, 2, size=16).astype(np.float32)
    train, valid = data.open_dataset(str(tmp_path), 8, (5,), data_shape=(16,), slice_idx=idx, dim_weights=w,
                                     pca_ckpt=str(tmp_path / "pca.pkl"), cache=False, shuffle=False)
    for ds, raw in ((train, tr), (valid, ev)):
        want = np.take(pca.transform(raw) * w, idx, axis=-1)
        want = want[:(len(want) // 8) * 8]
        assert np.isclose(ds.min, want.min()) and np.isclose(ds.max, want.max())
        got = torch.cat(list(ds)).numpy()
        assert np.allclose(got, O.normalize_dataset(want, want.min(), want.max()), atol=1e-6)
    # and back: un-normalise -> inverse PCA (sample_ncsn.py:456-468 passes the same object)
    full, _ = data.open_dataset(str(tmp_path), 8, (16,), pca_ckpt=str(tmp_path / "pca.pkl"), cache=False, shuffle=False)
    back = data.inverse_data_transform(torch.cat(list(full)).numpy(), True, pca, full.min, full.max)
    assert np.allclose(back, tr, atol=1e-4)
    shuffled, _ = data.open_dataset(str(tmp_path), 8, (16,), cache=False, seed=3)
    assert shuffled.shuffle and not _.shuffle


def test_oracle_label_zero_takes_a_real_uniform_noise_level():
    """utils/losses.py:272-286 with continuous_noise=False: labels in [0, T); label 0 -> uniform in [alphas_prod[T], 1);
    every other label -> its (degenerate) minval alphas_prod'[l - 1]."""
    betas = O.create_noise_schedule(1e-6, 0.01, 1000, "linear")
    ap = np.concatenate([[1.0], np.cumprod(1 - betas.astype(np.float64))])
    key = O.jax_prngkey(11)
    labels, _eps = O.jax_diffusion_loss_draws(key, (4096, 2), 1000, continuous_noise=False)
    assert labels.min() == 0 and labels.max() == 999
    labels[:3] = 0
    u = O.jax_diffusion_loss_u01(key, 4096)
    assert u.dtype == np.float32 and 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.02
    a = O.used_alphas_from_labels(betas, labels, u)
    z = labels == 0
    assert np.allclose(a[~z], ap[labels[~z] - 1], rtol=1e-5)
    assert np.allclose(a[z], ap[1000] + u[z] * (1 - ap[1000]), rtol=1e-5) and (a[z] >= ap[1000] * (1 - 1e-6)).all() and (a[z] < 1).all()
    with pytest.raises(ValueError):
        O.used_alphas_from_labels(betas, labels)
    lab1, _ = O.jax_diffusion_loss_draws(key, (64, 2), 1000, continuous_noise=True)
    assert np.array_equal(O.used_alphas_from_labels(betas, lab1), O.used_alphas_from_labels(betas, lab1, u[:64]))


def test_early_stopping_and_rng_split():
    from smd_amd import ncsn
    from smd_amd.train_utils import EarlyStopping
    es = EarlyStopping(patience=1)
    ok, es = es.update(1.0)
    assert ok and es.best_metric == 1.0
    ok, es = es.update(1.5)
    assert not ok and es.patience_count == 1 and not es.should_stop
    ok, es = es.update(1.5)
    assert not ok and es.should_stop
    a, b, c = ncsn.split(ncsn.PRNGKey(0), num=3)
    assert len({a.seed, b.seed, c.seed}) == 3 and ncsn.split(ncsn.PRNGKey(0), 3)[1] == b
    with pytest.raises(ValueError):
        ncsn.config_from_kwargs("ConvNCSN", (32, 512), {})
    with pytest.raises(ValueError):
        ncsn.config_from_kwargs("TransformerDDPM", (512,), {})
    cfg = ncsn.config_from_kwargs("DenseDDPM", (512,), dict(num_layers=6, num_heads=8, num_mlp_layers=2, mlp_dims=2048))
    assert cfg.sample_shape == (512,)                     # accepts-and-ignores the transformer kwargs


def test_shard_bounds_cover_exactly():
    from smd_amd.trainer import shard_bounds
    for n in (0, 1, 7, 1000, 1001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


# ------------------------------------------------------------------ data-parallel exchange (gloo, 2 ranks)
def _dp_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from smd_amd.trainer import GradComm
    torch.set_num_threads(1)
    comm = GradComm()
    assert comm.world_size == world and comm.rank == rank
    # the trainer's convention: each rank's gradient is already scaled by 1/(GLOBAL batch * S * C), the
    # all-reduce is a SUM over two buckets split at the engine's head offset -> the global-batch gradient
    cfg = O.NetConfig(data_channels=8, num_layers=1, num_heads=4, num_mlp_layers=1, mlp_dims=256)
    p = O.init_params(cfg, 0, torch.float64)
    betas = O.create_noise_schedule(1e-6, 0.01, 1000, "linear")
    g = torch.Generator().manual_seed(5)
    B = 4
    x0 = torch.randn(B, 32, 8, generator=g, dtype=torch.float64)
    labels = torch.randint(1, 1001, (B,), generator=g).numpy()
    eps = torch.randn(B, 32, 8, generator=g, dtype=torch.float64)

    def grads(sl):
        leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        per = O.diffusion_loss(x0[sl], O.make_model(leaf, cfg), betas, labels[sl], eps[sl], "none")
        (per.sum() / B).backward()                      # local sum / GLOBAL batch
        return torch.cat([leaf[k].grad.reshape(-1) for k, _ in O.param_spec(cfg)])

    full = grads(slice(0, B))
    half = B // world
    mine = grads(slice(rank * half, (rank + 1) * half)).clone()
    head = int(mine.numel() * 0.6)                      # any split point: buckets are independent
    comm.reduce_async(mine[head:])
    comm.reduce_async(mine[:head])
    comm.wait()
    assert torch.allclose(mine, full, rtol=1e-10, atol=1e-14), float((mine - full).abs().max())
    # bucketed (3 chunks per stage) and bf16-on-the-wire variants of the same exchange
    again = grads(slice(rank * half, (rank + 1) * half)).float().clone()
    chunked = GradComm(buckets=3)
    assert len(chunked._chunks(again[head:])) == 3 and sum(c.numel() for c in chunked._chunks(again[head:])) == again.numel() - head
    chunked.reduce_async(again[head:])
    chunked.reduce_async(again[:head])
    chunked.wait()
    assert torch.allclose(again.double(), full, rtol=1e-5, atol=1e-9)
    lossy = grads(slice(rank * half, (rank + 1) * half)).float().clone()
    half_wire = GradComm(buckets=2, payload="bf16")
    half_wire.reduce_async(lossy[head:])
    half_wire.reduce_async(lossy[:head])
    half_wire.wait()
    err = float((lossy.double() - full).norm() / full.norm())
    assert lossy.dtype == torch.float32 and 1e-5 < err < 8e-3, err          # bf16 rounding of two addends and of their sum
    with pytest.raises(ValueError):
        GradComm(payload="fp16")
    with pytest.raises(ValueError):
        GradComm(algorithm="tree")
    # reduce_scatter + all_gather instead of all_reduce: the same sums (a tail that does not divide by the world size
    # included), bitwise the same bytes on every rank, also with per-layer buckets (the `after` gate is a no-op on CPU)
    for kw in (dict(), dict(buckets=3), dict(payload="bf16")):
        rs = GradComm(algorithm="rs_ag", **kw)
        v = grads(slice(rank * half, (rank + 1) * half)).float().clone()
        n_odd = v.numel() - (1 - v.numel() % 2)                 # an odd length: one element goes through the tail all_reduce
        v = v[:n_odd].clone()
        gate_calls = []
        rs.reduce_async(v[head:], after=None)
        rs.reduce_async(v[:head], after=lambda s: gate_calls.append(s))
        rs.wait()
        if kw.get("payload") == "bf16":
            assert float((v.double() - full[:n_odd]).norm() / full[:n_odd].norm()) < 8e-3
        else:
            assert torch.allclose(v.double(), full[:n_odd], rtol=1e-5, atol=1e-9), (kw, float((v.double() - full[:n_odd]).abs().max()))
        assert gate_calls == []                                  # CPU tensors: no communication stream to gate
        mine_bytes = v.clone()
        other = [torch.empty_like(v) for _ in range(world)]
        dist.all_gather(other, mine_bytes)
        assert all(torch.equal(o, mine_bytes) for o in other), f"rs_ag {kw}: ranks hold different reduced gradients"
    # what the bench line reports about the exchange: backend, world size, shape, a count of the collectives issued
    d = comm.describe()
    assert d["backend"] == "gloo" and d["world_size"] == world and d["algorithm"] == "all_reduce" and d["layer_buckets"] is False
    c0 = comm.collectives
    v = grads(slice(rank * half, (rank + 1) * half)).float().clone()
    comm.reduce_async(v[head:]); comm.reduce_async(v[:head]); comm.wait()
    assert comm.collectives - c0 == 2 * comm.buckets and comm.exposed_comm_us() is None      # CPU tensors: nothing to time
    flat = torch.full((10,), float(rank))
    comm.broadcast_params(flat)
    assert float(flat.sum()) == 0.0
    # the bench line's `config.dp.devices`: every rank's physical device, gathered over the job's own process group; ranks that
    # share a device are refused (the first SCALE record must prove N distinct GPUs by itself)
    from smd_amd.trainer import assert_distinct_devices, device_identity, gather_device_identities
    ids = gather_device_identities(dist)
    assert len(ids) == world and ids[rank] == device_identity() and ids[rank].startswith("cpu:")
    assert_distinct_devices(ids)                                   # two processes: two pids
    shared = gather_device_identities(dist, mine="uuid:GPU-0")
    assert shared == ["uuid:GPU-0"] * world
    with pytest.raises(RuntimeError, match="share a device"):
        assert_distinct_devices(shared)
    assert_distinct_devices(shared, allow_shared=True)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_data_parallel_gradient_exchange_gloo(tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_custom_ops_are_registered_and_have_no_cpu_path():
    """north_star: 'Python host code calling hand-written HIP kernels through PyTorch-ROCm custom ops (C-ABI)'."""
    import smd_amd.ops as ops  # noqa: F401
    for name in ("eps_forward", "eps_forward_train", "eps_backward", "gemm_bf16_nt", "ddpm_reverse_step_", "q_sample"):
        assert hasattr(torch.ops.smd_amd, name)
    with pytest.raises(RuntimeError, match="GPU only"):
        torch.ops.smd_amd.gemm_bf16_nt(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16),
                                       torch.zeros(64))
    # shape inference without running anything (fake tensors)
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        a = torch.empty(256, 128, dtype=torch.bfloat16)
        out = torch.ops.smd_amd.gemm_bf16_nt(a, torch.empty(512, 128, dtype=torch.bfloat16), torch.empty(512))
        assert tuple(out.shape) == (256, 512) and out.dtype == torch.bfloat16
        e = torch.ops.smd_amd.eps_forward(torch.empty(4, 32, 512), torch.empty(4, 1, 1), 0)
        assert tuple(e.shape) == (4, 32, 512)
        # the differentiable form: the output carries an autograd edge to the flat parameter buffer (registered formula)
        pr = torch.empty(1000, requires_grad=True)
        et = torch.ops.smd_amd.eps_forward_train(torch.empty(4, 32, 512), torch.empty(4, 1, 1), pr, 0)
        assert tuple(et.shape) == (4, 32, 512) and et.requires_grad
        xt, s = torch.ops.smd_amd.q_sample(torch.empty(4, 32, 42), torch.empty(1001), torch.empty(4, dtype=torch.int32),
                                           torch.empty(4, 32, 42))
        assert tuple(xt.shape) == (128, 42) and tuple(s.shape) == (4,)


def test_langevin_host_logic_matches_the_oracle():
    """Host-side pieces of the Langevin samplers: per-update key tables (utils/ebm_utils.py:133,233) and the collection
    slot arithmetic (:149-156) incl. its behaviour when linspace repeats entries."""
    import smd_amd.jax_random as J
    import smd_amd.ncsn as N
    key = O.jax_prngkey(5)
    tk = J.ThreefryKey(int(key[0]), int(key[1]))
    for consistent in (False, True):
        step, infill = J.langevin_key_table(tk, 7, consistent=consistent)
        sk, fk = O.jax_langevin_keys(key, 7, consistent=consistent)
        assert [tuple(int(v) for v in r) for r in step] == [(int(k[0]), int(k[1])) for k in sk]
        if not consistent:
            assert [tuple(int(v) for v in r) for r in infill] == [(int(k[0]), int(k[1])) for k in fk]
    for L, T in ((10, 100), (2, 50), (3, 4), (10, 5)):
        cidx = np.linspace(1, L * T, 100).astype(np.int32)
        for image_idx in range(1, L * T + 1):
            assert N.ald_collection_slot(cidx, image_idx) == O.ald_collection_slot(cidx, image_idx)
    cidx = np.linspace(1, 12, 100).astype(np.int32)                      # 12 updates: every value repeats 8-9 times
    assert N.ald_collection_slot(cidx, 1) == sum(range(9)) + 1 and N.ald_collection_slot(cidx, 13) == -1
    cidx = np.linspace(1, 1000, 100).astype(np.int32)
    slots = [N.ald_collection_slot(cidx, i) for i in range(1, 1001)]
    assert sorted(s for s in slots if s > 0) == list(range(1, 101))      # the usual case: each of the 100 slots once


def test_cli_objective_and_sampler_gates():
    """--loss / --sampling values the engine covers (utils/losses.py: dsm, ddpm; utils/ebm_utils.py: ald, cas, ddpm) are
    accepted by the drivers' gate, everything else exits before any GPU work."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    tr = importlib.import_module("train_ncsn")
    sm = importlib.import_module("sample_ncsn")
    with pytest.raises(SystemExit, match="ssm needs a double backward"):
        tr.main(["train_ncsn.py", "--loss=ssm", "--sampling=ddpm", "--synthetic"])
    from smd_amd.flags import FlagError
    with pytest.raises(FlagError, match="ald.cas.ddpm"):                  # the enum itself rejects anything else (:57-58)
        sm.main(["sample_ncsn.py", "--sampling=hmc", "--synthetic"])
    with pytest.raises(SystemExit, match="DDPM mode"):
        sm.main(["sample_ncsn.py", "--sampling=ald", "--interpolate", "--synthetic"])


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    """`bench.py --gpus N` launches its own N ranks; on a box with fewer GPUs it must fail loudly instead of running one rank and
    printing n_gpus: 1 (the driver invokes it exactly like this).  A WORLD_SIZE that disagrees with --gpus is refused too."""
    import subprocess
    import sys
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have >= 2:
        pytest.skip("a multi-GPU box would really launch the ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "needs 2 GPUs" in (r.stderr + r.stdout) and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="4", RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout) and r.stdout.strip() == ""


def test_shipped_lds_layouts_are_conflict_free_under_the_real_lane_groups():
    """tools/lds_conflicts.py: the LDS bank model of gfx950 with the lane groups ds_read_b128 / ds_write_b64 / the transposing
    read really use (MI355X_MICROARCH.md).  The layouts the fused encoder kernels ship with must have no conflict on any read;
    what remains is the inherent 2-way conflict of the 8-byte writes of four bf16 per lane.  The
    round-3 layouts are kept in the model as the counter-example (they were designed for lane groups of 16 consecutive lanes)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lds_conflicts", os.path.join(root, "tools", "lds_conflicts.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)

    def extra(accesses):
        return {name: m.cycles(kind, fn)[1] for name, kind, _count, fn in accesses}

    shipped = [m.mlp_hs_bwd(lambda x: (0x78 >> (2 * x)) & 3), m.mlp_hs_fwd(), m.attn_block_fwd(True),
               m.attn_block_bwd(lambda r: ((r & 3) << 2) | ((r >> 2) & 3))]
    for acc in shipped:
        for name, e in extra(acc).items():
            kind = next(k for n, k, _c, _f in acc if n == name)
            if kind == "write_b64" and "72-B" not in name and "fp32" not in name:
                assert e == 4, (name, e)                 # 2-way on each of the four 16-lane groups: inherent
            else:
                assert e == 0, (name, e)
    # the layouts they replaced: every one of these reads conflicted
    old = extra(m.mlp_hs_bwd(lambda x: x))
    assert old["S3 A fragments (64-B rows)"] == 4 and old["dz B fragments (64-B rows)"] == 4
    assert extra(m.attn_block_fwd(False))["residual rows, fp32 (epilogue reads)"] == 60
    assert extra(m.attn_block_bwd(lambda r: r))["transposing reads of k, q, dO (first rows)"] == 2


def test_build_identity_is_content_addressed(tmp_path, monkeypatch):
    """build.py: the library's staleness is decided by a hash of the sources / headers / flags it was built from (embedded as
    SMD_BUILD_ID=<16 hex> and returned by smd_build_id()), never by file times."""
    import smd_amd.build as b
    sid = b.source_id()
    assert len(sid) == 16 and int(sid, 16) >= 0 and b.source_id() == sid
    # the id moves with the flags (experiment builds) ...
    monkeypatch.setattr(b, "FLAGS", b.FLAGS + ["-DSMD_SOMETHING"])
    assert b.source_id() != sid
    monkeypatch.undo()
    assert b.source_id() == sid
    # ... and is read back from a binary without loading it
    f = tmp_path / "lib.so"
    f.write_bytes(b"\x7fELF junk SMD_BUILD_ID=0123456789abcdef more junk")
    assert b.built_id(str(f)) == "0123456789abcdef"
    f.write_bytes(b"no id in here")
    assert b.built_id(str(f)) == "" and b.built_id(str(tmp_path / "missing.so")) == ""
    if os.path.exists(b.LIB_PATH):
        import smd_amd.lib as lib
        assert lib.get_lib().smd_build_id().decode() == b.built_id() == sid        # the shipped library is this tree's
        os.utime(b.LIB_PATH, (1, 1))                                               # an ancient file time changes nothing
        assert not b.is_stale()


def test_stable_and_lab_headers_partition_the_exports():
    """include/smd_hip.h is the surface a maintainer binds; tuning knobs, debug views and probes live in smd_hip_lab.h."""
    import smd_amd.lib as lib
    stable, both = set(lib.declared_symbols(lab=False)), set(lib.declared_symbols())
    lab = both - stable
    assert lab == {"smd_set_tuning", "smd_engine_debug_snapshot_bytes", "smd_engine_debug_snapshots", "smd_engine_debug_tensor",
                   "smd_probe_clock", "smd_probe_l2_warm", "smd_probe_tr_read", "smd_probe_stream_create_cu_mask", "smd_probe_stream_destroy"}
    assert {"smd_engine_sample_step_part", "smd_engine_forward_train", "smd_engine_backward_from", "smd_build_id"} <= stable
    assert both == set(lib._SIGS)                                                   # the ctypes table covers exactly the two headers


def test_sampler_pipeline_switches(monkeypatch):
    import smd_amd.ncsn as N
    monkeypatch.delenv("SMD_SAMPLER_PIPELINE", raising=False)
    monkeypatch.delenv("SMD_SAMPLER_UNROLL", raising=False)
    assert N._sampler_pipeline_unroll() == 8
    monkeypatch.setenv("SMD_SAMPLER_UNROLL", "4")
    assert N._sampler_pipeline_unroll() == 4
    monkeypatch.setenv("SMD_SAMPLER_PIPELINE", "0")
    assert N._sampler_pipeline_unroll() == 0


def test_sampler_chain_sizes_cover_the_reference_default_and_ragged_batches(monkeypatch):
    """VERDICT r5 missing #3: sample_ncsn.py:54 sample_size = 1000, drawn as ONE batch (train_ncsn.py:540), must take the pipelined
    two-chain walk: 504 + 496, multiples of the 8-sequence row granule; a ragged batch is padded by < 8 throw-away sequences."""
    import types
    import smd_amd.ncsn as N
    monkeypatch.delenv("SMD_SAMPLER_CHAINS", raising=False)
    tr = types.SimpleNamespace(engine=types.SimpleNamespace(S=32, cfg=types.SimpleNamespace(mlp_dims=2048)))
    assert N.sampler_chain_sizes(tr, 1000, True) == ([504, 496], 0)
    assert N.sampler_chain_sizes(tr, 256, True) == ([128, 128], 0)
    assert N.sampler_chain_sizes(tr, 128, True) == ([64, 64], 0)
    assert N.sampler_chain_sizes(tr, 1001, True) == ([504, 504], 7)
    assert N.sampler_chain_sizes(tr, 999, True) == ([504, 496], 1)
    assert N.sampler_chain_sizes(tr, 1001, True, allow_pad=False) == ([1001], 0)      # jax.random streams: the true array size counts
    assert N.sampler_chain_sizes(tr, 127, True) == ([127], 0)                         # small batches: one chain
    assert N.sampler_chain_sizes(tr, 1000, False) == ([1000], 0)                      # eager / explicit draws: one chain
    for B in range(128, 1200, 37):
        sizes, pad = N.sampler_chain_sizes(tr, B, True)
        assert sum(sizes) == B + pad and 0 <= pad < 8 and all(h % 8 == 0 and h >= 64 for h in sizes) and abs(sizes[0] - sizes[1]) <= 8
    assert N._sampler_chains(tr, 1000, True) == 2
    monkeypatch.setenv("SMD_SAMPLER_CHAINS", "1")
    assert N.sampler_chain_sizes(tr, 1000, True) == ([1000], 0)
    monkeypatch.delenv("SMD_SAMPLER_CHAINS")
    dense = types.SimpleNamespace(engine=types.SimpleNamespace(S=1, cfg=types.SimpleNamespace(mlp_dims=2048)))
    assert N.sampler_chain_sizes(dense, 1024, True) == ([512, 512], 0)                # DenseDDPM: granule 256 vectors, never padded
    assert N.sampler_chain_sizes(dense, 1000, True) == ([1000], 0)
