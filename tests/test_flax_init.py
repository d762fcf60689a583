"""create_model's initial weights as the reference draws them (smd_amd/flax_init.py, train_ncsn.py:193-203).

No published vectors exist for flax's key folding or jax's truncated normal, so this is pinned as far as it can be without
JAX: (1) the product's host code against the oracle's independent NumPy restatement of every primitive, bit for bit;
(2) the distributional facts the definitions imply; (3) structure (one key per kernel, names folded in, DenseGeneral's flat
draw).  tests/test_jax_goldens.py compares against the reference's own initial parameters once they can be generated."""
import hashlib

import numpy as np
import pytest

import ddpm_oracle as O
import smd_amd.flax_init as FI
import smd_amd.flax_io as FIO
import smd_amd.jax_random as J


def okey(k):
    return (np.uint32(k.k0), np.uint32(k.k1))


def test_primitives_match_the_oracle_restatement_bit_for_bit():
    for seed in (0, 1, 12345):
        k = J.PRNGKey(seed)
        for d in (0, 1, 7, 0xFFFFFFFF, 0x9E3779B9):
            f = FI.fold_in(k, d)
            assert (np.uint32(f.k0), np.uint32(f.k1)) == tuple(O.jax_fold_in(okey(k), d))
        for s in ("kernel", "Dense_1", "MultiHeadDotProductAttention_3", "query"):
            f = FI.fold_in_str(k, s)
            assert (np.uint32(f.k0), np.uint32(f.k1)) == tuple(O.flax_fold_in_str(okey(k), s))
            want = int.from_bytes(hashlib.sha1(s.encode()).digest()[:4], "big")
            assert (np.uint32(f.k0), np.uint32(f.k1)) == tuple(O.jax_fold_in(okey(k), want))
        for n in (1, 2, 7, 1000, 4097):
            assert np.array_equal(FI.random_bits(k, n), O.jax_random_bits(okey(k), n))
            assert np.array_equal(FI.truncated_normal(k, n), O.jax_truncated_normal(okey(k), n))
        assert np.array_equal(FI.lecun_normal(k, (128, 384)), O.jax_lecun_normal(okey(k), (128, 384)))
    # fold_in(key, d) is the Threefry block of the counter pair (0, d): the Random123 vector for key = ctr = 0
    z = FI.fold_in(J.ThreefryKey(0, 0), 0)
    assert (z.k0, z.k1) == (0x6B200159, 0x99BA4EFE)


def test_truncated_normal_is_what_its_definition_says():
    x = FI.truncated_normal(J.PRNGKey(3), 1 << 20).astype(np.float64)
    assert x.min() > -2.0 and x.max() < 2.0                       # open interval
    assert abs(x.mean()) < 5e-3
    assert abs(x.std() - 0.87962566103423978) < 2e-3              # the constant lecun_normal divides by
    # quantiles of a normal truncated to [-2, 2]: P(|x| < 1) = erf(1/sqrt2) / erf(2/sqrt2)
    import math
    assert abs((np.abs(x) < 1).mean() - math.erf(1 / math.sqrt(2)) / math.erf(2 / math.sqrt(2))) < 2e-3
    w = FI.lecun_normal(J.PRNGKey(4), (2048, 512)).astype(np.float64)
    assert abs(w.std() * np.sqrt(2048) - 1.0) < 5e-3


@pytest.mark.parametrize("arch,kw", [("TransformerDDPM", dict(num_layers=2, num_heads=8, num_mlp_layers=1, mlp_dims=256)),
                                      ("DenseDDPM", dict(num_layers=2, mlp_dims=256))])
def test_init_params_structure(arch, kw):
    cfg = O.NetConfig(architecture=arch, data_channels=42, **kw)
    template = dict(O.param_spec(cfg))
    rng = J.split(J.PRNGKey(0), 3)[1]                             # model_rng of train_ncsn.py:318-319
    p = FI.init_params(cfg, rng, template)
    assert set(p) == set(template) and all(p[k].shape == tuple(template[k]) and p[k].dtype == np.float32 for k in p)
    for k, v in p.items():
        if k.endswith(".bias"):
            assert not v.any()
        elif k.endswith(".scale"):
            assert (v == 1).all()
        else:
            assert abs(v.astype(np.float64).std() * np.sqrt(v.shape[0]) - 1.0) < 0.08, k          # variance 1 / fan_in
    # every kernel has its own key: its module path (+ "kernel") folded into model_rng, in the oracle's restatement too
    seen = {}
    for path, our, shape, cols in FIO._walk(FIO.module_tree(cfg), "shared"):
        if path[-1] != "kernel":
            continue
        key = O.flax_param_key(okey(rng), path)
        assert key not in seen.values()
        seen[path] = key
        if len(shape) == 3:      # DenseGeneral draws the flattened kernel: q / k / v (E, H*d), out (H*d, E)
            flat = (shape[0], shape[1] * shape[2]) if path[-2] != "out" else (shape[0] * shape[1], shape[2])
        else:
            flat = tuple(shape)
        want = O.jax_lecun_normal(key, flat)
        got = p[our] if cols is None else p[our][..., cols[0]:cols[1]]
        assert np.array_equal(got.reshape(flat), want), path
    if arch == "TransformerDDPM":
        qkv = p["enc.0.attn.qkv.kernel"]
        assert not np.array_equal(qkv[:, :128], qkv[:, 128:256])    # query / key / value are three draws
    # the same key gives the same weights; another model_rng or another auto-naming rule gives others (the unpinned part)
    assert all(np.array_equal(p[k], FI.init_params(cfg, rng, template)[k]) for k in p)
    assert not np.array_equal(p["in_proj.kernel"], FI.init_params(cfg, J.PRNGKey(1), template)["in_proj.kernel"])
    assert not np.array_equal(p["out_proj.kernel"], FI.init_params(cfg, rng, template, rule="per_class")["out_proj.kernel"])
