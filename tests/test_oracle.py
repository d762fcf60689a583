"""Known-answer / invariant tests that pin the CPU restatement oracle (the reference ships no tests
or golden vectors -- SURVEY section 4 / 8c -- so these closed forms stand in for them), plus checks
that the product's host-side tables are bit-identical to the oracle's."""
import math
import os

import numpy as np
import pytest
import torch

import ddpm_oracle as O

REF = "/root/reference"


def test_param_counts_match_paper_and_survey():
    mk = lambda **kw: O.num_params(O.NetConfig(**kw))
    assert mk(data_channels=42) == 25_579_946              # the paper's 25.58 M
    assert mk(data_channels=512) == 26_603_136
    assert mk(data_channels=146) == 25_806_354
    assert mk(data_channels=512, num_layers=8, num_heads=16, num_mlp_layers=3) == 38_620_032
    assert mk(architecture="DenseDDPM", data_channels=512, num_layers=6) == 67_088_896


def test_schedule_known_answers():
    b = O.create_noise_schedule(1e-6, 0.01, 1000, "linear")
    ap = O.alphas_cumprod(b)
    assert b.dtype == np.float32 and b[0] == np.float32(1e-6) and b[-1] == np.float32(0.01)
    assert abs(ap[0] - 0.999999) < 1e-7
    assert abs(ap[-1] - 0.0066226) < 2e-7
    assert abs(math.sqrt(ap[-1]) - 0.08138) < 1e-5
    g = O.create_noise_schedule(1.0, 0.01, 15, "geometric")
    assert abs(g[0] - 1.0) < 1e-6 and abs(g[-1] - 0.01) < 1e-8 and np.allclose(g[1:] / g[:-1], g[1] / g[0], rtol=1e-5)
    f = O.create_noise_schedule(L=6, schedule="fibonacci")
    assert np.allclose(f, [1e-6, 2e-6, 3e-6, 5e-6, 8e-6, 13e-6])
    with pytest.raises(ValueError):
        O.create_noise_schedule(schedule="cosine")


def test_reverse_step_closed_form_at_t0():
    b = O.create_noise_schedule(1e-6, 0.01, 1000, "linear")
    co = O.reverse_coefficients(b)
    # 1 - ap_0 == beta_0 up to fp32 rounding => mu1 ~ 1, mu2 == 0, sigma == sqrt(1e-20)
    assert co["mu2"][0] == 0.0 and abs(co["mu1"][0] - 1.0) < 2e-2 and abs(co["sigma"][0] - 1e-10) < 1e-15
    x = torch.randn(3, 32, 8, dtype=torch.float64)
    model = lambda s, c: torch.zeros_like(s)
    state, _, _ = O.diffusion_dynamics(model, b, x, lambda t: torch.randn_like(x), t_start=0, t_stop=0)
    expect = float(co["mu1"][0]) * torch.clamp(float(co["sqrt_recip"][0]) * x, -1, 1)
    assert torch.allclose(state, expect, atol=1e-12)


def test_collection_bookkeeping_quirks():
    T = 1000
    tab = O.collection_index_table(T)
    assert tab[0] == 1 and tab[-1] == 1000 and tab[13] == 334 and tab[26] == 667
    hits = [(O.collection_slot_for_t(T, t, tab), t) for t in range(T - 1, -1, -1)]
    hits = [(s, t) for s, t in hits if s >= 0]
    assert hits[0] == (2, 975) and hits[1] == (3, 949) and hits[-1] == (40, 1)
    assert [s for s, _ in hits] == list(range(2, 41))       # slot 1 never written, final state never collected
    x = torch.zeros(2, 32, 4, dtype=torch.float64)
    _, coll, met = O.diffusion_dynamics(lambda s, c: torch.zeros_like(s), O.create_noise_schedule(1e-6, 0.01, T, "linear"),
                                        x + 0.5, lambda t: torch.zeros_like(x))
    assert coll.shape == (41, 2, 32, 4) and met.shape == (4, T, 1)
    assert float(coll[1].abs().max()) == 0.0 and float(coll[40].abs().max()) > 0


def test_encodings_closed_form():
    pe = O.positional_encoding(32, 128)
    assert torch.equal(pe[0, :64], torch.zeros(64, dtype=pe.dtype)) and torch.equal(pe[0, 64:], torch.ones(64, dtype=pe.dtype))
    assert abs(float(pe[3, 0]) - math.sin(3.0)) < 1e-12 and abs(float(pe[3, 64 + 63]) - math.cos(3.0 * 1e-4)) < 1e-12
    s = torch.tensor([[0.25], [1.0]], dtype=torch.float64)
    ne = O.noise_encoding(s, 128)
    for b, sv in enumerate((0.25, 1.0)):
        for i in (0, 17, 63):
            f = 10000.0 ** (-i / 63.0)
            assert abs(float(ne[b, i]) - math.sin(5000 * sv * f)) < 1e-9
            assert abs(float(ne[b, 64 + i]) - math.cos(5000 * sv * f)) < 1e-9


def test_layer_invariants():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 7, 128, generator=g, dtype=torch.float64) * 3 + 1
    p = {"n.scale": torch.ones(128, dtype=torch.float64), "n.bias": torch.zeros(128, dtype=torch.float64)}
    y = O.layer_norm(x, p, "n")
    assert float(y.mean(-1).abs().max()) < 1e-12
    var = x.var(-1, unbiased=False)
    assert torch.allclose(y.var(-1, unbiased=False), var / (var + 1e-6), atol=1e-12)
    # attention with zero q/k kernels = mean over positions of the v projection (then out projection)
    E, H = 128, 8
    ap = {"a.qkv.kernel": torch.randn(E, 3 * E, generator=g, dtype=torch.float64) * 0.1,
          "a.qkv.bias": torch.zeros(3 * E, dtype=torch.float64),
          "a.out.kernel": torch.eye(E, dtype=torch.float64), "a.out.bias": torch.zeros(E, dtype=torch.float64)}
    ap["a.qkv.kernel"][:, :2 * E] = 0
    h = torch.randn(2, 32, E, generator=g, dtype=torch.float64)
    o = O.self_attention(h, ap, "a", H)
    v = h @ ap["a.qkv.kernel"][:, 2 * E:]
    assert torch.allclose(o, v.mean(1, keepdim=True).expand_as(o), atol=1e-12)
    assert abs(float(O.gelu(torch.tensor(1.0, dtype=torch.float64))) - 0.8411919906082768) < 1e-12
    assert abs(float(O.swish(torch.tensor(1.0, dtype=torch.float64))) - 0.7310585786300049) < 1e-12


def test_loss_invariants_and_quirk():
    b = O.create_noise_schedule(1e-6, 0.01, 1000, "linear")
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(4, 32, 6, generator=g, dtype=torch.float64)
    eps = torch.randn(4, 32, 6, generator=g, dtype=torch.float64)
    labels = np.array([1, 2, 500, 1000])
    ua = O.used_alphas_from_labels(b, labels)
    ap = O.alphas_cumprod(b)
    assert ua[0] == 1.0 and ua[1] == ap[0] and ua[3] == ap[998]        # alpha'[l-1]: a_T is never trained
    zero = O.diffusion_loss(x0, lambda xt, s: torch.zeros_like(xt), b, labels, eps, "none")
    assert torch.allclose(zero, (eps ** 2).mean(dim=(1, 2)))
    perfect = O.diffusion_loss(x0, lambda xt, s: eps, b, labels, eps, "mean")
    assert float(perfect) == 0.0
    s = O.diffusion_loss(x0, lambda xt, s: torch.zeros_like(xt), b, labels, eps, "sum")
    assert abs(float(s) - float(zero.sum())) < 1e-12
    seen = {}
    O.diffusion_loss(x0, lambda xt, s: seen.setdefault("s", s) * 0 + xt * 0, b, labels, eps, "mean")
    assert seen["s"].shape == (4, 1, 1) and abs(float(seen["s"][0]) - 1.0) < 1e-12   # conditions on sqrt(alpha)


def test_optimizer_invariants():
    p = {"w": torch.tensor([1.0, -2.0, 3.0], dtype=torch.float64)}
    g = {"w": torch.tensor([0.3, -0.1, 0.2], dtype=torch.float64)}
    c, n = O.clip_grads(g, 1.0)
    assert c is g and abs(float(n) - math.sqrt(0.14)) < 1e-12          # no-op below the threshold
    big = {"w": g["w"] * 100}
    c, n = O.clip_grads(big, 1.0)
    assert abs(float(n) - 1.0) < 1e-12
    st = O.AdamState()
    new = O.adam_update(p, g, st, 1e-3)
    assert torch.allclose(new["w"] - p["w"], -1e-3 * torch.sign(g["w"]), atol=1e-9)   # first step = lr*sign(g)
    assert st.step == 1
    assert O.stepped_lr(1e-3, 0, 10000, 0.98) == 1e-3 and O.stepped_lr(1e-3, 10000, 10000, 0.98) == 1e-3
    assert abs(O.stepped_lr(1e-3, 10001, 10000, 0.98) - 0.98e-3) < 1e-15
    assert abs(O.stepped_lr(1e-3, 30000, 10000, 0.98) - 1e-3 * 0.98 ** 2) < 1e-15
    e = O.ema_update({"w": torch.zeros(3, dtype=torch.float64)}, p, 0.999)
    assert torch.allclose(e["w"], p["w"] * 1e-3)


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32-10
    z = O.philox4x32(np.zeros((1, 4), np.uint32), np.zeros(2, np.uint32))[0]
    assert [int(v) for v in z] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = O.philox4x32(np.full((1, 4), 0xFFFFFFFF, np.uint32), np.full(2, 0xFFFFFFFF, np.uint32))[0]
    assert [int(v) for v in f] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    pi = O.philox4x32(np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], np.uint32),
                      np.array([0xa4093822, 0x299f31d0], np.uint32))[0]
    assert [int(v) for v in pi] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    ctr = np.zeros((20000, 4), np.uint32)
    ctr[:, 0] = np.arange(20000)
    n = O.philox_normal4(ctr, np.array([1, 2], np.uint32)).ravel()
    assert abs(n.mean()) < 0.02 and abs(n.std() - 1.0) < 0.02


def test_product_tables_bit_identical_to_oracle():
    import smd_amd.schedule as S
    for sched, lo, hi, L in (("linear", 1e-6, 0.01, 1000), ("geometric", 1.0, 0.01, 15), ("linear", 1e-4, 0.02, 50)):
        b = S.create_noise_schedule(lo, hi, L, sched)
        assert np.array_equal(b, O.create_noise_schedule(lo, hi, L, sched))
        co = O.reverse_coefficients(b)
        tab = S.reverse_coefficient_table(b)
        for j, k in enumerate(("sqrt_recip", "sqrt_m1", "mu1", "mu2", "sigma", "alpha_prod", "sqrt_alpha_prod",
                               "sqrt_one_minus")):
            assert np.array_equal(tab[:, j], co[k]), k
        if L >= 40:
            slot = S.collection_slot_table(L)
            assert [int(s) for s in slot] == [O.collection_slot_for_t(L, t) for t in range(L)]


def test_engine_parameter_table_matches_oracle_spec():
    """Host-only: smd_engine_create builds the layout without touching a GPU."""
    import ctypes as C
    import smd_amd.lib as lib
    L = lib.get_lib()
    cases = [(0, 512, 32, 6, 8, 2), (0, 42, 32, 8, 16, 3), (1, 512, 1, 6, 8, 2)]
    for arch, Cc, S, nl, nh, nk in cases:
        d = lib.ModelDesc(arch, Cc, S, nl, nh, nk, 2048, 128, 128, 1000)
        h = C.c_void_p()
        lib.check(L.smd_engine_create(C.byref(d), C.byref(h)))
        spec = O.param_spec(O.NetConfig(architecture="DenseDDPM" if arch else "TransformerDDPM", data_channels=Cc,
                                        num_layers=nl, num_heads=nh, num_mlp_layers=nk))
        assert L.smd_engine_num_tensors(h) == len(spec)
        off = 0
        for i, (name, shape) in enumerate(spec):
            nm, o, r, c = C.c_char_p(), C.c_int64(), C.c_int32(), C.c_int32()
            lib.check(L.smd_engine_tensor_info(h, i, C.byref(nm), C.byref(o), C.byref(r), C.byref(c)))
            got_shape = (r.value, c.value) if c.value else (r.value,)
            assert nm.value.decode() == name and got_shape == tuple(shape) and o.value == off
            off += int(np.prod(shape))
        assert L.smd_engine_param_count(h) == off
        assert L.smd_engine_workspace_bytes(h, 256, 1) > L.smd_engine_workspace_bytes(h, 256, 0) > 0
        L.smd_engine_destroy(h)
    bad = lib.ModelDesc(0, 512, 16, 6, 8, 2, 2048, 128, 128, 1000)     # seq_len 16 unsupported -> loud error
    h = C.c_void_p()
    with pytest.raises(ValueError):
        lib.check(L.smd_engine_create(C.byref(bad), C.byref(h)))


def test_library_exports_every_declared_symbol():
    import smd_amd.lib as lib
    L = lib.get_lib()
    names = lib.declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/smd_hip.h but not exported"
    assert set(lib._SIGS) == set(names)


def test_gradient_buckets_cover_the_stem_in_backward_order():
    """Host-only part of the data-parallel interface (include/smd_hip.h smd_engine_grad_bucket): the buckets of the stem
    slice are the encoder layers from the last to the first, contiguous, the last one also holding in_proj, and together they are
    exactly [0, head_param_offset) -- checked against the oracle's parameter layout (no GPU: layout functions only)."""
    import ctypes as C
    import smd_amd.lib as lib
    L = lib.get_lib()
    for kw in (dict(num_layers=6, num_heads=8, num_mlp_layers=2), dict(num_layers=8, num_heads=16, num_mlp_layers=3)):
        cfg = O.NetConfig(data_channels=512, **kw)
        d = lib.ModelDesc(0, 512, 32, kw["num_layers"], kw["num_heads"], kw["num_mlp_layers"], 2048, 128, 128, 1000)
        h = C.c_void_p()
        lib.check(L.smd_engine_create(C.byref(d), C.byref(h)))
        try:
            head = int(L.smd_engine_head_param_offset(h))
            sizes = {}
            for name, shape in O.param_spec(cfg):
                key = name.split(".")[1] if name.startswith("enc.") else name.split(".")[0]
                sizes[key] = sizes.get(key, 0) + int(np.prod(shape))
            nb = int(L.smd_engine_num_grad_buckets(h))
            assert nb == kw["num_layers"]
            end = head
            for b in range(nb):
                off, ln = C.c_int64(), C.c_int64()
                lib.check(L.smd_engine_grad_bucket(h, b, C.byref(off), C.byref(ln)))
                layer = kw["num_layers"] - 1 - b
                want = sizes[str(layer)] + (sizes["in_proj"] if layer == 0 else 0)
                assert off.value + ln.value == end and ln.value == want, (b, off.value, ln.value, want)
                end = off.value
            assert end == 0
            with pytest.raises(ValueError):
                lib.check(L.smd_engine_grad_bucket(h, nb, C.byref(off), C.byref(ln)))
        finally:
            L.smd_engine_destroy(h)


def test_product_never_imports_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "symbolic-music-diffusion_amd")
    import re
    pat = re.compile(r"^\s*(from|import)\s+[\w.]*(oracle)", re.M)
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text) and "sys.path" not in text, f


# ----------------------------------------------------------------------------------------------
# jax.random restatement: pinned by published known answers
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _kat():
    import json
    return json.load(open(os.path.join(GOLDEN, "jax_random_kat.json")))


def test_threefry2x32_random123_vectors():
    for v in _kat()["threefry2x32"]:
        y0, y1 = O.threefry2x32(v["key"], np.uint32([v["ctr"][0]]), np.uint32([v["ctr"][1]]))
        assert [int(y0[0]), int(y1[0])] == v["out"]


def test_jax_random_known_answers():
    """Keys and normals printed in JAX's documentation (PRNGKey(0), three generations of split, six normal draws)."""
    d = _kat()["jax_docs"]
    key = O.jax_prngkey(0)
    assert [int(k) for k in key] == d["key0"]
    n = lambda k: float(O.jax_normal(k, 1)[0])
    assert n(key) == np.float32(d["normal_key0"])
    for gen in ("split1", "split2"):
        key, sub = O.jax_split(key)
        assert [int(k) for k in key] == d[gen]["key"] and [int(k) for k in sub] == d[gen]["subkey"]
        assert n(sub) == np.float32(d[gen]["normal_subkey"])
    key, *subs = O.jax_split(key, 4)
    assert [n(s) for s in subs] == [np.float32(v) for v in d["split3_of_4"]["normals_subkeys"]]


def test_jax_random_bits_layout_and_moments():
    key = O.jax_prngkey(1234)
    # the two counter halves: element i < h comes from word 0 of block (i, i+h), element h+i from word 1
    n = 10
    b = O.jax_random_bits(key, n)
    y0, y1 = O.threefry2x32(key, np.arange(5, dtype=np.uint32), np.arange(5, 10, dtype=np.uint32))
    assert np.array_equal(b, np.concatenate([y0, y1]))
    # odd size: the pad counter is 0 and its output is dropped
    b7 = O.jax_random_bits(key, 7)
    y0, y1 = O.threefry2x32(key, np.uint32([0, 1, 2, 3]), np.uint32([4, 5, 6, 0]))
    assert np.array_equal(b7, np.concatenate([y0, y1])[:7])
    z = O.jax_normal(key, 200_000).astype(np.float64)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01 and abs((z ** 3).mean()) < 0.03
    u = O.jax_uniform(key, 100_000)
    assert u.min() >= 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    r = O.jax_randint(key, 100_000, 1, 1001)
    assert r.min() == 1 and r.max() == 1000 and abs(r.mean() - 500.5) < 3
    assert np.all(O.jax_randint(key, 16, 5, 5) == 5)             # maxval <= minval -> minval


def test_jax_uniform_degenerate_interval_is_minval():
    """utils/losses.py:283-286 draws uniform(minval=abar[l-1], maxval=abar[l]) with minval > maxval: the trailing
    max(minval, .) returns minval exactly."""
    betas = O.create_noise_schedule(1e-6, 0.01, 1000, "linear")
    ap = np.concatenate([np.ones(1, np.float32), O.alphas_cumprod(betas)])
    labels = O.jax_randint(O.jax_prngkey(3), 256, 1, 1001)
    got = O.jax_uniform(O.jax_prngkey(4), 256, ap[labels - 1], ap[labels])
    assert np.array_equal(got, ap[labels - 1])
    assert np.array_equal(got, O.used_alphas_from_labels(betas, labels).astype(np.float32))


def test_jax_erfinv_matches_scipy():
    from scipy.special import erfinv
    x = np.linspace(-0.999999, 0.999999, 20001).astype(np.float32)
    got = O.erfinv_f32(x).astype(np.float64)
    want = erfinv(x.astype(np.float64))
    assert np.max(np.abs(got - want) / (1 + np.abs(want))) < 2e-6


def test_score_matching_and_langevin_closed_forms():
    """Known answers for the NCSN-path restatements (utils/losses.py:129-179, utils/ebm_utils.py:89-271)."""
    sig = O.create_noise_schedule(1.0, 0.01, 10, "geometric")
    assert sig[0] == pytest.approx(1.0) and sig[-1] == pytest.approx(0.01) and sig[3] / sig[4] == pytest.approx(sig[0] / sig[1])
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 4, 6, generator=g, dtype=torch.float64)
    eps = torch.randn(5, 4, 6, generator=g, dtype=torch.float64)
    labels = np.array([0, 3, 9, 5, 1])
    # a model that returns the exact target -eps / sigma has zero loss; the zero model has loss 0.5 sum eps^2 (sigma-free)
    perfect = lambda xt, s: -(xt - x) / s ** 2
    assert float(O.denoising_score_matching_loss(x, perfect, sig, labels, eps, False, "sum")) < 1e-20
    zero = lambda xt, s: torch.zeros_like(xt)
    got = O.denoising_score_matching_loss(x, zero, sig, labels, eps, False, "none")
    assert torch.allclose(got, 0.5 * (eps ** 2).sum(dim=(1, 2)), rtol=1e-10)
    # continuous noise on a decreasing schedule: uniform(minval = sigmas[l-1], maxval = sigmas[l]) is its minval
    lab1 = np.array([1, 4, 9, 2, 7])
    assert np.array_equal(O.dsm_used_sigmas(sig, lab1, True), sig[lab1 - 1])
    assert np.array_equal(O.dsm_used_sigmas(sig, lab1, True, np.full(5, 0.7, np.float32)), sig[lab1 - 1])
    inc = sig[::-1].copy()
    with pytest.raises(ValueError):
        O.dsm_used_sigmas(inc, lab1, True)
    mid = O.dsm_used_sigmas(inc, lab1, True, np.full(5, 0.5, np.float32))
    assert np.allclose(mid, 0.5 * (inc[lab1 - 1] + inc[lab1]), rtol=1e-6)

    # annealed Langevin with a zero score and zero noise leaves the state alone; alpha = eps (sigma / sigma_L)^2
    init = torch.randn(3, 4, 6, generator=g, dtype=torch.float64)
    zeros = lambda *a: torch.zeros(3, 4, 6, dtype=torch.float64)
    s3 = np.array([1.0, 0.1, 0.05], np.float32)
    st, coll, met = O.annealed_langevin_dynamics(zero, s3, init, 2e-5, 4, True, zeros)
    assert torch.equal(st, init) and coll.shape == (102, 3, 4, 6) and met.shape == (4, 3, 4)
    assert torch.allclose(met[2, :, 0], torch.tensor([2e-5 * 400, 2e-5 * 4, 2e-5], dtype=torch.float64), rtol=1e-5)
    assert torch.allclose(met[0], torch.full((3, 4), 1e-5, dtype=torch.float64))          # sqrt(0 + 1e-10)
    # a score pointing at the origin shrinks the state by (1 - alpha) per update; the denoise step by (1 - sigma_L^2)
    pull = lambda xt, s: -xt
    st, _, _ = O.annealed_langevin_dynamics(pull, s3, init, 2e-5, 2, True, zeros)
    a = [2e-5 * (float(s) / 0.05) ** 2 for s in s3]
    want = init * np.prod([(1 - float(np.float32(ai))) ** 2 for ai in a]) * (1 - 0.05 ** 2)
    assert torch.allclose(st, want, rtol=1e-5)
    # infill: the masked part is the template + sigma * noise of the last update, whatever the score does
    mask = torch.zeros(3, 4, 6, dtype=torch.float64)
    mask[:, :2] = 1
    tmpl = torch.ones(3, 4, 6, dtype=torch.float64)
    noise = torch.randn(3, 4, 6, generator=g, dtype=torch.float64)
    st, coll, _ = O.annealed_langevin_dynamics(pull, s3, init, 2e-5, 2, False, zeros, True, tmpl, mask, lambda si, i: noise)
    assert torch.allclose(st[:, :2], (tmpl + 0.05 * noise)[:, :2], rtol=1e-6) and torch.equal(coll[0][:, :2], tmpl[:, :2])
    # consistent sampler: beta = sqrt(1 - (1 - eps / sigma_L^2)^2); the last update adds no noise (next sigma = 0)
    ones = lambda i: torch.ones(3, 4, 6, dtype=torch.float64)
    st, met = O.consistent_langevin_dynamics(zero, s3, init, 5e-4, False, ones)
    beta = (1 - (1 - 5e-4 / 0.05 ** 2) ** 2) ** 0.5
    assert torch.allclose(st, init + beta * (0.1 + 0.05), rtol=1e-5) and met.shape == (4, 3, 1)
    assert float(met[3, 2, 0]) == pytest.approx(1e-5)                                     # noise norm of the last level: 0


def test_e4m3_emulation_quantiser_known_answers():
    """oracle/e4m3_emulation.py restates the ENGINE's e4m3 storage (csrc/smd_common.h e4m3_row_exponent / pack4_e4m3): the OCP
    e4m3 grid (3 mantissa bits, max 448, subnormal step 2^-9), round-to-nearest-even, saturation, and the row-exponent rule."""
    import e4m3_emulation as F8
    r = F8.round_e4m3(torch.tensor([0.0, 1.0, 1.0625, 1.1875, 17.0, 18.0, 19.0, 448.0, 0.001953125, 0.0009765625, -0.40625], dtype=torch.float64))
    # ties to even: 1.0625 -> 1.0 (mantissa 000 vs 001), 1.1875 -> 1.25 (between 1.125 and 1.25: even is 1.25); 17 -> 16, 19 -> 20 (step 2), 18 stays
    assert r.tolist() == [0.0, 1.0, 1.0, 1.25, 16.0, 18.0, 20.0, 448.0, 0.001953125, 0.0, -0.40625]
    e = F8.e4m3_row_exponent(torch.tensor([1.0, 1.75, 1.7500001, 3.9, 448.0, 0.0, 2.0 ** -140]))
    assert e.tolist() == [-8, -8, -7, -6, 0, 0, -126]
    # a row is scaled so that its maximum lands in (224, 448]: never saturated, never more than one binade of head-room wasted
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 256, generator=g, dtype=torch.float64) * torch.logspace(-6, 6, 64, dtype=torch.float64)[:, None]
    q = F8.q8_rows(x)
    amax = x.abs().amax(1)
    s = 2.0 ** F8.e4m3_row_exponent(amax.float()).double()
    assert bool(((amax / s) <= 448.0).all()) and bool(((amax / s) > 224.0 * 0.999).all())
    assert float(((q - x).abs() / amax[:, None]).max()) <= 2.0 ** -4 + 1e-12          # half a step of the top binade (32 / 448 / 2 ... <= 1/16)
    assert float((q - x).norm() / x.norm()) < 4e-2
    # the Dense: forward on quantised operands, dgrad on quantised dY and the per-input-feature quantised kernel, wgrad on bf16 copies
    import bf16_emulation as E
    a = torch.randn(8, 64, generator=g, dtype=torch.float64, requires_grad=True)
    W = E._round(torch.randn(64, 32, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    b = torch.zeros(32, dtype=torch.float64, requires_grad=True)
    out = F8.F8Dense.apply(a, W, b)
    assert torch.equal(out.detach(), F8.q8_rows(a) @ F8.q8_rows(W.detach().t()).t())
    dY = E._round(torch.randn(8, 32, generator=g, dtype=torch.float64))
    out.backward(dY)
    assert torch.equal(a.grad, F8.q8_rows(dY) @ F8.q8_rows(W.detach()).t())
    assert torch.equal(W.grad, E._round(a.detach()).t() @ dY) and torch.equal(b.grad, dY.sum(0))
