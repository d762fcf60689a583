"""CPU tests of the flax-0.3.0 checkpoint reader / writer and of the host-side jax.random key
algebra (8f-1).  No GPU: the engine is replaced by a stub that owns CPU tensors with the engine's tensor table."""
import os

import msgpack
import numpy as np
import pytest
import torch

import ddpm_oracle as O


class StubEngine:
    """The part of smd_amd.engine.Engine the checkpoint code touches, on CPU tensors."""

    def __init__(self, cfg, seed=0, train=True, ema=True):
        from smd_amd.engine import NetConfig
        self.cfg = NetConfig(architecture=cfg.architecture, data_channels=cfg.data_channels, num_layers=cfg.num_layers,
                             num_heads=cfg.num_heads, num_mlp_layers=cfg.num_mlp_layers, mlp_dims=cfg.mlp_dims)
        self.device = torch.device("cpu")
        self.tensor_table, off = [], 0
        for name, shape in O.param_spec(cfg):
            self.tensor_table.append((name, off, tuple(shape)))
            off += int(np.prod(shape))
        g = torch.Generator().manual_seed(seed)
        self.params = torch.randn(off, generator=g)
        self.grads = self.m = self.v = self.ema = None
        if train:
            self.grads = torch.zeros(off)
            self.m, self.v = torch.randn(off, generator=g), torch.rand(off, generator=g)
            self.ema = torch.randn(off, generator=g) if ema else None
        self.step_counter = torch.tensor([1234], dtype=torch.int32)

    def named_views(self, flat=None):
        flat = self.params if flat is None else flat
        return {n: flat[o:o + int(np.prod(s))].view(*s) for n, o, s in self.tensor_table}

    def load_named(self, tensors):
        for k, v in self.named_views().items():
            v.copy_(torch.as_tensor(np.asarray(tensors[k]), dtype=torch.float32))


class Opt:
    def __init__(self, e):
        self.engine = e


class Ema:
    def __init__(self, e, mu=0.999):
        self.params, self.mu = e.ema, mu


CFG = O.NetConfig(data_channels=42, num_layers=2, num_heads=8, num_mlp_layers=2, mlp_dims=256)


def test_wire_format_is_flax_serialization():
    """Arrays are msgpack ExtType 1 = packb((shape, dtype name, bytes)); python scalars stay native."""
    import smd_amd.flax_io as FI
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    raw = FI.to_bytes({"x": a, "n": 3, "f": 0.5, "b": True, "s": np.int32(7)})
    seen = {}

    def hook(code, data):
        seen[code] = seen.get(code, 0) + 1
        return msgpack.unpackb(data, raw=False)

    d = msgpack.unpackb(raw, ext_hook=hook, raw=False)
    assert seen == {1: 2} and d["n"] == 3 and d["f"] == 0.5 and d["b"] is True
    shape, dtype, buf = d["x"]
    assert shape == [2, 3] and dtype == "float32" and buf == a.tobytes()
    assert d["s"][0] == [] and d["s"][1] == "int32"
    back = FI.from_bytes(raw)
    assert np.array_equal(back["x"], a) and back["x"].dtype == np.float32 and int(back["s"]) == 7
    # chunked arrays of later flax versions are reassembled
    chunked = {"__msgpack_chunked_array__": True, "shape": np.asarray([2, 3]), "chunks": {"0": a.ravel()[:4], "1": a.ravel()[4:]}}
    assert np.array_equal(FI.from_bytes(FI.to_bytes({"w": chunked}))["w"], a)


def test_parameter_tree_names_and_layouts():
    import smd_amd.flax_io as FI
    spec = dict(O.param_spec(CFG))
    rng = np.random.default_rng(0)
    named = {k: rng.standard_normal(s).astype(np.float32) for k, s in spec.items()}
    t = FI.params_to_flax(named, CFG)
    # call order of models/ncsn.py:141-179 under one shared child counter (parameter-less modules take a number too)
    assert list(t) == ["Dense_1", "LayerNorm_2", "MultiHeadDotProductAttention_3", "LayerNorm_4", "Dense_5", "Dense_6", "LayerNorm_7",
                       "MultiHeadDotProductAttention_8", "LayerNorm_9", "Dense_10", "Dense_11", "LayerNorm_12", "Dense_13", "DenseFiLM_14",
                       "DenseResBlock_15", "DenseFiLM_16", "DenseResBlock_17", "LayerNorm_18", "Dense_19"]
    att = t["MultiHeadDotProductAttention_3"]
    assert list(att) == ["query", "key", "value", "out"]
    assert att["query"]["kernel"].shape == (128, 8, 16) and att["key"]["bias"].shape == (8, 16)
    assert att["out"]["kernel"].shape == (8, 16, 128) and att["out"]["bias"].shape == (128,)
    E = 128
    qkv = named["enc.0.attn.qkv.kernel"]
    assert np.array_equal(att["key"]["kernel"].reshape(E, E), qkv[:, E:2 * E])
    assert np.array_equal(att["value"]["bias"].ravel(), named["enc.0.attn.qkv.bias"][2 * E:])
    film = t["DenseFiLM_14"]
    assert list(film) == ["Dense_1", "Dense_2", "Dense_3", "Dense_4"]                  # NoiseEncoding_0 has no parameters
    assert np.array_equal(film["Dense_3"]["kernel"], named["film.0.ss.kernel"][:, :256])   # scale (models/ncsn.py:60)
    assert np.array_equal(film["Dense_4"]["kernel"], named["film.0.ss.kernel"][:, 256:])   # shift (:61)
    assert list(t["DenseResBlock_15"]) == ["LayerNorm_0", "Dense_2", "LayerNorm_3", "Dense_5"]
    # every naming rule round-trips and is told apart from the keys alone
    for rule in FI.NAMING_RULES:
        for ac in FI.ATTENTION_CLASS_NAMES:
            tree = FI.params_to_flax(named, CFG, rule, ac)
            assert FI.detect_naming(tree, CFG) == (rule, ac)
            back = FI.params_from_flax(tree, CFG, spec)
            assert all(np.array_equal(back[k], named[k]) for k in named)
    with pytest.raises(KeyError):
        FI.detect_naming({"Conv_0": {}}, CFG)
    dd = O.NetConfig(architecture="DenseDDPM", data_channels=64, num_layers=2, mlp_dims=256)
    sd = dict(O.param_spec(dd))
    nd = {k: rng.standard_normal(s).astype(np.float32) for k, s in sd.items()}
    td = FI.params_to_flax(nd, dd)
    assert list(td) == ["Dense_0", "DenseFiLM_1", "DenseResBlock_2", "DenseFiLM_3", "DenseResBlock_4", "LayerNorm_5", "Dense_6"]
    assert all(np.array_equal(v, nd[k]) for k, v in FI.params_from_flax(td, dd, sd).items())


@pytest.mark.parametrize("fmt", ["flax", "safetensors"])
def test_checkpoint_round_trip_both_formats(tmp_path, fmt):
    import smd_amd.checkpoint as CK
    import smd_amd.flax_io as FI
    from smd_amd.train_utils import EarlyStopping
    src = StubEngine(CFG, seed=1)
    es = EarlyStopping(min_delta=0.5, patience=3, best_metric=1.25, patience_count=2, should_stop=False)
    for step in (0, 1, 2, 3):
        path = CK.save_checkpoint(str(tmp_path), (Opt(src), Ema(src), es), step, keep=2, fmt=fmt)
    assert sorted(os.listdir(tmp_path)) == ["checkpoint_2", "checkpoint_3"]          # flax naming, keep=N rotation
    assert FI.is_flax_file(path) == (fmt == "flax")
    dst = StubEngine(CFG, seed=2)
    found, es2 = CK.restore_checkpoint(str(tmp_path), dst)
    assert found and es2 == es
    for a, b in ((dst.params, src.params), (dst.m, src.m), (dst.v, src.v), (dst.ema, src.ema)):
        assert torch.equal(a, b)
    assert int(dst.step_counter) == 1234
    samp = StubEngine(CFG, seed=3, train=False)
    assert CK.load_ema_params(str(tmp_path), samp) and torch.equal(samp.params, src.ema)
    if fmt == "flax":
        sd = FI.read_file(path)
        assert set(sd) == {"0", "1", "2"} and set(sd["0"]) == {"state", "target"} and set(sd["1"]) == {"mu", "params"}
        assert set(sd["0"]["state"]) == {"step", "param_states"} and list(sd["0"]["target"]) == ["params"]
        leaf = sd["0"]["state"]["param_states"]["Dense_1"]["kernel"]
        assert set(leaf) == {"grad_ema", "grad_sq_ema"} and leaf["grad_ema"].shape == (42, 128)
        assert sd["2"] == es.state_dict() and abs(sd["1"]["mu"] - 0.999) < 1e-12


def test_flax_checkpoint_written_elsewhere_is_read(tmp_path):
    """A file assembled here key by key, the way upstream's state dict looks after a jitted update (sorted keys,
    0-d arrays for step / mu, MultiHeadDotProductAttention naming), restores into the engine."""
    import smd_amd.checkpoint as CK
    import smd_amd.flax_io as FI
    src = StubEngine(CFG, seed=5)
    named = {k: v.numpy() for k, v in src.named_views().items()}
    tree = FI.params_to_flax(named, CFG, "shared", "MultiHeadDotProductAttention")
    srt = lambda d: {k: srt(d[k]) if isinstance(d[k], dict) else d[k] for k in sorted(d)}
    zeros = lambda d: {k: zeros(v) if isinstance(v, dict) else {"grad_ema": np.zeros_like(v), "grad_sq_ema": np.ones_like(v)}
                       for k, v in d.items()}
    sd = {"0": {"state": {"param_states": zeros(srt(tree)), "step": np.asarray(77, np.int32)}, "target": {"params": srt(tree)}},
          "1": {"mu": np.asarray(0.0, np.float32), "params": srt(tree)},
          "2": {"best_metric": 0.5, "min_delta": 0, "patience": 0, "patience_count": 1, "should_stop": False}}
    FI.write_file(str(tmp_path / "checkpoint_9"), sd)
    dst = StubEngine(CFG, seed=6)
    found, es = CK.restore_checkpoint(str(tmp_path), dst)
    assert found and es.best_metric == 0.5 and es.patience_count == 1
    assert torch.equal(dst.params, src.params) and float(dst.m.abs().max()) == 0 and float(dst.v.min()) == 1
    assert int(dst.step_counter) == 77


# engine tensor -> (leaf number in tests/golden/make_flax_fixture.py's LEAVES, column range of the engine tensor)
_FIXTURE_LEAVES = {
    "in_proj.kernel": [(0, None)], "in_proj.bias": [(1, None)],
    "film.0.fc1.kernel": [(2, None)], "film.0.fc1.bias": [(3, None)],
    "film.0.fc2.kernel": [(4, None)], "film.0.fc2.bias": [(5, None)],
    "film.0.ss.kernel": [(6, (0, 8)), (8, (8, 16))], "film.0.ss.bias": [(7, (0, 8)), (9, (8, 16))],      # [scale | shift]
    "res.0.ln1.scale": [(10, None)], "res.0.ln1.bias": [(11, None)],
    "res.0.fc1.kernel": [(12, None)], "res.0.fc1.bias": [(13, None)],
    "res.0.ln2.scale": [(14, None)], "res.0.ln2.bias": [(15, None)],
    "res.0.fc2.kernel": [(16, None)], "res.0.fc2.bias": [(17, None)],
    "ln_o.scale": [(18, None)], "ln_o.bias": [(19, None)],
    "out_proj.kernel": [(20, None)], "out_proj.bias": [(21, None)],
}


def test_hand_assembled_flax_checkpoint_bytes_are_read(tmp_path):
    """tests/golden/flax_dense_ddpm_tiny.msgpack was assembled byte by byte from the msgpack spec and the flax.serialization
    layout (tests/golden/make_flax_fixture.py: no msgpack library, none of this package) -- the file
    flax.training.checkpoints.save_checkpoint writes at /root/reference/train_ncsn.py:395-399 and sample_ncsn.py:331-342
    restores.  Leaf n holds n + i/1024; Adam moments -value / 2*value; EMA value + 100."""
    import shutil
    import smd_amd.checkpoint as CK
    import smd_amd.flax_io as FI
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flax_dense_ddpm_tiny.msgpack")
    cfg = O.NetConfig(architecture="DenseDDPM", data_channels=4, num_layers=1, mlp_dims=8, film_channels=4)
    assert FI.is_flax_file(src)
    sd = FI.read_file(src)
    assert set(sd) == {"0", "1", "2"} and FI.detect_naming(sd["0"]["target"]["params"], cfg)[0] == "shared"
    template = {n: tuple(sh) for n, sh in O.param_spec(cfg)}
    params, m, v, step, ema, mu, es = FI.split_state_dict(sd, cfg, template)
    assert step == 4321 and mu == 0.999
    assert es == dict(min_delta=0, patience=1, best_metric=0.0625, patience_count=1, should_stop=False)
    assert set(params) == set(_FIXTURE_LEAVES) == set(template)
    for name, parts in _FIXTURE_LEAVES.items():
        shape = template[name]
        for leaf, cols in parts:
            sub = shape if cols is None else shape[:-1] + (cols[1] - cols[0],)
            want = (np.float32(leaf) + np.arange(int(np.prod(sub)), dtype=np.float32) / np.float32(1024)).reshape(sub)
            pick = (lambda a: a) if cols is None else (lambda a: a[..., cols[0]:cols[1]])
            assert params[name].dtype == np.float32 and np.array_equal(pick(params[name]), want), name
            assert np.array_equal(pick(m[name]), -want) and np.array_equal(pick(v[name]), 2 * want), name
            assert np.array_equal(pick(ema[name]), want + np.float32(100)), name
    # and through the checkpoint layer: checkpoint_<step> in a model dir restores parameters, moments, step, EMA
    shutil.copy(src, tmp_path / "checkpoint_4321")

    class TinyEngine(StubEngine):
        def __init__(self):
            self.cfg, self.device = cfg, torch.device("cpu")
            self.tensor_table, off = [], 0
            for name, shape in O.param_spec(cfg):
                self.tensor_table.append((name, off, tuple(shape)))
                off += int(np.prod(shape))
            self.params, self.grads = torch.zeros(off), torch.zeros(off)
            self.m, self.v, self.ema = torch.zeros(off), torch.zeros(off), torch.zeros(off)
            self.step_counter = torch.tensor([0], dtype=torch.int32)

    eng = TinyEngine()
    found, stop = CK.restore_checkpoint(str(tmp_path), eng)
    assert found and stop.best_metric == 0.0625 and stop.patience_count == 1 and int(eng.step_counter) == 4321
    views = eng.named_views()
    assert float(views["in_proj.kernel"][0, 1]) == np.float32(1 / 1024) and float(views["out_proj.bias"][3]) == 21 + 3 / 1024
    assert torch.equal(eng.m, -eng.params) and torch.equal(eng.v, 2 * eng.params) and torch.equal(eng.ema, eng.params + 100)
    eng2 = TinyEngine()
    assert CK.load_ema_params(str(tmp_path), eng2) and torch.equal(eng2.params, eng.ema)     # sample_ncsn.py:338-342
    # our writer reproduces the hand-assembled bytes exactly (same map order, same minimal encodings)
    named = lambda flat: {k: t.numpy() for k, t in eng.named_views(flat).items()}
    again = FI.checkpoint_state_dict(cfg, named(eng.params), named(eng.m), named(eng.v), 4321, named(eng.ema), 0.999, es)
    srt = lambda d: {k: srt(d[k]) if isinstance(d[k], dict) else d[k] for k in sorted(d)}
    with open(src, "rb") as f:
        assert FI.to_bytes(srt(again)) == f.read()


def test_host_threefry_key_algebra_matches_the_restatement():
    import smd_amd.jax_random as J
    k = J.PRNGKey(0)
    a, b = J.split(k)
    assert (a.k0, a.k1) == (4146024105, 967050713) and (b.k0, b.k1) == (2718843009, 1272950319)   # JAX documentation
    for seed in (1, 42, 2 ** 31 + 5):
        key = J.PRNGKey(seed)
        assert (key.k0, key.k1) == tuple(int(v) for v in O.jax_prngkey(seed))
        for num in (2, 3, 4):
            assert [(c.k0, c.k1) for c in J.split(key, num)] == [(int(x), int(y)) for x, y in O.jax_split(O.jax_prngkey(seed), num)]
        for n in (1, 2, 5, 8):
            assert J.random_bits_host(key, n) == [int(v) for v in O.jax_random_bits(O.jax_prngkey(seed), n)]
    ik, nk = J.sampler_key_tables(J.PRNGKey(7), 6)
    oi, on = O.jax_sampler_keys(O.jax_prngkey(7), 6)
    assert ik.tolist() == [[int(x) for x in t] for t in oi] and nk.tolist() == [[int(x) for x in t] for t in on]
    import smd_amd.ncsn as N
    assert isinstance(N.make_key(3, "threefry"), J.ThreefryKey) and isinstance(N.make_key(3, "philox"), N.PRNGKey)
    assert N.split(N.make_key(0, "threefry")) == (a, b)
    with pytest.raises(ValueError):
        N.make_key(0, "mt19937")
