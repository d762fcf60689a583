"""End-to-end run of the drop-in drivers on the GPU (SURVEY section 8d config-1 style smoke + the section 8f rows):
TFRecord dataset with a slice checkpoint -> train_ncsn.py (jax.random streams, flax-format checkpoints) ->
sample_ncsn.py (restore, generate / infill, the reference's output files)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *flags, timeout=600):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, script), *flags], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, f"{script} failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    return r.stdout + r.stderr


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    import smd_amd.data as D
    import smd_amd.tfrecord as T
    d = tmp_path_factory.mktemp("cli")
    rng = np.random.default_rng(0)
    slice_idx = np.sort(rng.choice(512, 42, replace=False))
    D.save(slice_idx, str(d / "slice.pkl"))
    ds = d / "ds"
    for name, n in (("train-00000-of-00002", 24), ("train-00001-of-00002", 24), ("eval-00000-of-00001", 16)):
        T.write_latents(str(ds / f"{name}.tfrecord"), (rng.standard_normal((n, 32, 512)) * 2).astype(np.float32))
    return d


def common(workdir):
    return ["--flagfile=configs/ddpm-mel-32seq-512.cfg", f"--dataset={workdir / 'ds'}", f"--slice_ckpt={workdir / 'slice.pkl'}",
            f"--model_dir={workdir / 'model'}", "--num_layers=2", "--mlp_dims=256", "--num_mlp_layers=1", "--batch_size=8",
            "--num_sigmas=50", "--rng_impl=threefry"]


def test_train_then_sample_and_infill(workdir):
    import smd_amd.data as D
    import smd_amd.flax_io as FI
    out = run("train_ncsn.py", *common(workdir), "--epochs=2", "--logging_freq=1", "--snapshot_freq=100",
              "--snapshot_sampling=false", "--ckpt_format=flax", "--learning_rate=1e-3")
    ck = sorted(os.listdir(workdir / "model"))
    assert any(f.startswith("checkpoint_") for f in ck), ck
    path = str(workdir / "model" / [f for f in ck if f.startswith("checkpoint_")][-1])
    assert FI.is_flax_file(path)
    sd = FI.read_file(path)
    assert int(np.asarray(sd["0"]["state"]["step"])) == 12                        # 2 epochs x 6 batches of 8
    assert np.isfinite(sd["0"]["target"]["params"]["Dense_1"]["kernel"]).all()
    assert os.path.exists(workdir / "ds" / "cache" / "train_slice_min.pkl")          # the reference's min/max cache
    assert "loss" in out

    samp = workdir / "samples"
    run("sample_ncsn.py", *common(workdir), "--sample_size=4", f"--sampling_dir={samp}", "--sample_seed=3")
    gen = D.load(str(samp / "ncsn" / "generated.pkl"))
    coll = D.load(str(samp / "ncsn" / "collection.pkl"))
    real = D.load(str(samp / "ncsn" / "real.pkl"))
    assert gen.shape == (4, 32, 512) and gen.dtype == np.float64 and np.isfinite(gen).all()     # sample_ncsn.py:452-471
    assert coll.shape == (41, 4, 32, 512) and real.shape == (4, 32, 512)
    # same seed, same checkpoint, jax.random streams -> the same samples again (the unsliced dims are random fill)
    samp2 = workdir / "samples2"
    run("sample_ncsn.py", *common(workdir), "--sample_size=4", f"--sampling_dir={samp2}", "--sample_seed=3")
    gen2 = D.load(str(samp2 / "ncsn" / "generated.pkl"))
    sl = D.load(str(workdir / "slice.pkl"))
    assert np.allclose(gen[..., sl], gen2[..., sl], atol=1e-5)
    other = np.setdiff1d(np.arange(512), sl)
    assert not np.allclose(gen[..., other], gen2[..., other])

    inf = workdir / "infill"
    run("sample_ncsn.py", *common(workdir), "--sample_size=4", f"--sampling_dir={inf}", "--infill=true")
    g = D.load(str(inf / "ncsn" / "generated.pkl"))
    r = D.load(str(inf / "ncsn" / "real.pkl"))
    # the 16 fixed latents (first and last 8 positions) are the real data (sample_ncsn.py:405-423), up to the eval/train
    # min-max round trip; the middle 16 are generated
    import smd_amd.data as D2
    tmin, tmax = D2.load(str(workdir / "ds" / "cache" / "train_slice_min.pkl")), D2.load(str(workdir / "ds" / "cache" / "train_slice_max.pkl"))
    emin, emax = D2.load(str(workdir / "ds" / "cache" / "eval_slice_min.pkl")), D2.load(str(workdir / "ds" / "cache" / "eval_slice_max.pkl"))
    norm = lambda x, lo, hi: 2 * (x - lo) / (hi - lo) - 1
    fixed = list(range(8)) + list(range(24, 32))
    assert np.allclose(norm(g[:, fixed][..., sl], tmin, tmax), norm(r[:, fixed][..., sl], emin, emax), atol=1e-4)
    assert g.shape == (4, 32, 512) and np.isfinite(g).all()

    itp = workdir / "interp"
    run("sample_ncsn.py", *common(workdir), "--sample_size=4", f"--sampling_dir={itp}", "--interpolate=true")
    gi = D.load(str(itp / "ncsn" / "generated.pkl"))
    assert gi.shape == (9, 4, 32, 512) and np.isfinite(gi).all()                 # 9 interpolation points (:425-435)
    assert not os.path.exists(itp / "ncsn" / "collection.pkl")                   # not written when interpolating (:455-456)


def test_config1_dense_ddpm_end_to_end(tmp_path):
    """configs/ddpm-mel-1seq-512.cfg (DenseDDPM on (8, 512) vectors, --ema): a few
    epochs of training on a small fixed set, loss finite and decreasing, then the full 1000-step sampler."""
    import json
    import smd_amd.data as D
    rng = np.random.default_rng(3)
    base = rng.standard_normal((4, 512)).astype(np.float32)
    for split, n in (("train", 64), ("eval", 16)):
        x = base[rng.integers(0, 4, n)] + 0.05 * rng.standard_normal((n, 512)).astype(np.float32)
        D.save(x, str(tmp_path / "ds" / f"{split}.pkl"))
    flags = ["--flagfile=configs/ddpm-mel-1seq-512.cfg", f"--dataset={tmp_path / 'ds'}", f"--model_dir={tmp_path / 'model'}",
             "--batch_size=8", "--num_layers=2", "--mlp_dims=256"]
    run("train_ncsn.py", *flags, "--epochs=12", "--logging_freq=1", "--snapshot_freq=1000", "--learning_rate=2e-3",
        "--ckpt_format=flax")                                               # --ema comes from the config: EMA tree in the file
    rows = [json.loads(l) for l in open(tmp_path / "model" / "train" / "scalars.jsonl")]
    loss = [r["value"] for r in rows if r["tag"] == "loss"]
    assert len(loss) == 96 and np.isfinite(loss).all()                  # 12 epochs x 8 batches
    # eps-prediction MSE starts near 2 (random output + unit-variance target) and falls towards ~1 (most noise levels of
    # the linear schedule leave x_t ~ x_0, where eps is not inferable) within a few dozen steps
    assert np.mean(loss[-16:]) < 0.75 * np.mean(loss[:4]), (np.mean(loss[:4]), np.mean(loss[-16:]))
    lr = [r["value"] for r in rows if r["tag"] == "lr"]
    assert abs(lr[0] - 2e-3) < 1e-9
    out = tmp_path / "samples"
    import smd_amd.flax_io as FI
    ck = [f for f in sorted(os.listdir(tmp_path / "model")) if f.startswith("checkpoint_")]
    sd = FI.read_file(str(tmp_path / "model" / ck[-1]))
    assert list(sd["1"]["params"]) == ["Dense_0", "DenseFiLM_1", "DenseResBlock_2", "DenseFiLM_3", "DenseResBlock_4", "LayerNorm_5", "Dense_6"]
    w, we = sd["0"]["target"]["params"]["Dense_0"]["kernel"], sd["1"]["params"]["Dense_0"]["kernel"]
    assert w.shape == (512, 256) and not np.array_equal(w, we) and abs(float(sd["1"]["mu"]) - 0.999) < 1e-6   # EMA lags the weights
    run("sample_ncsn.py", *flags, "--sample_size=8", f"--sampling_dir={out}", "--sample_ema=true")
    gen = D.load(str(out / "ncsn" / "generated.pkl"))
    coll = D.load(str(out / "ncsn" / "collection.pkl"))
    assert gen.shape == (8, 512) and coll.shape == (41, 8, 512) and np.isfinite(gen).all()
    lo, hi = D.load(str(tmp_path / "ds" / "cache" / "train__min.pkl")), D.load(str(tmp_path / "ds" / "cache" / "train__max.pkl"))
    assert gen.min() >= lo - 1e-4 and gen.max() <= hi + 1e-4              # x0 is clipped to [-1, 1] at t = 0
