"""BASELINE config 5: OCP e4m3 operands with per-row power-of-two (E8M0) scales on v_mfma_scale_f32_32x32x64_f8f6f4 for
the DenseResBlock forward GEMMs (models/shared.py:65,69), bf16 elsewhere.  Tolerance of SURVEY 8c: rel-L2 <= 5e-2 on
eps_hat against the oracle; the single kernels are checked against fp64 arithmetic on the dequantised operands."""
import math

import numpy as np
import pytest
import torch

import ddpm_oracle as O

pytestmark = pytest.mark.gpu
ORACLE_THREADS = 64          # the oracle runs at B = 256 here (tests/conftest.py)

BETAS = O.create_noise_schedule(1e-6, 0.01, 1000, "linear")


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def L():
    import smd_amd.lib as lib
    return lib.get_lib()


def ck(L, rc):
    import smd_amd.lib as lib
    lib.check(rc)


P = lambda t: None if t is None else t.data_ptr()
st = lambda: torch.cuda.current_stream().cuda_stream


def dequant(q8, scale):
    """e4m3 bytes [rows][K] + E8M0 dwords [rows] -> float64"""
    v = q8.view(torch.float8_e4m3fn).float().double().cpu()
    e = (scale.cpu().to(torch.int64) & 0xFF) - 127
    return v * torch.pow(torch.tensor(2.0, dtype=torch.float64), e.double()).unsqueeze(1)


def test_quantize_rows_e4m3(L, dev):
    g = torch.Generator().manual_seed(3)
    rows, K = 512, 2048
    x = (torch.randn(rows, K, generator=g) * torch.logspace(-3, 1, rows).unsqueeze(1)).to(torch.bfloat16)
    x[7] = 0                                               # an all-zero row
    xd = x.to(dev)
    q = torch.zeros(rows, K, dtype=torch.uint8, device=dev)
    s = torch.zeros(rows, dtype=torch.int32, device=dev)
    ck(L, L.smd_quantize_rows_e4m3(P(xd), K, rows, K, P(q), P(s), st()))
    torch.cuda.synchronize()
    amax = x.float().abs().amax(1)
    e_ref = torch.where(amax > 0, torch.ceil(torch.log2(amax.double() / 448.0)).to(torch.int64), torch.zeros(rows, dtype=torch.int64))
    assert torch.equal((s.cpu().to(torch.int64) & 0xFF) - 127, e_ref)
    want = (x.float() * torch.pow(2.0, -e_ref.float()).unsqueeze(1)).clamp(-448, 448).to(torch.float8_e4m3fn)
    same = (want.view(torch.uint8) == q.cpu()).float().mean()
    print(f"quantize_rows_e4m3: {float(same) * 100:.3f} % of the bytes equal torch's float8_e4m3fn cast")
    assert float(same) > 0.999                              # -0 vs +0 and nothing else
    d = dequant(q, s)
    nz = amax > 0
    assert rel(d[nz], x.double()[nz]) < 4e-2                # 3 mantissa bits: ~2.6 % rms per element


@pytest.mark.parametrize("M,N,K,form", [(256, 256, 256, "b"), (512, 2048, 2048, "b"), (8192, 2048, 2048, "r")])
def test_gemm_e4m3_nt(L, dev, M, N, K, form):
    """C = A Bt^T + bias (+ fp32 residual) on e4m3 operands with per-row scales vs fp64 on the dequantised operands."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * (0.5 + 3 * torch.rand(M, 1, generator=g))
    b = torch.randn(N, K, generator=g) * 0.02 * (0.5 + 2 * torch.rand(N, 1, generator=g))
    aD, bD = a.to(torch.bfloat16).to(dev), b.to(torch.bfloat16).to(dev)
    qa, sa = torch.zeros(M, K, dtype=torch.uint8, device=dev), torch.zeros(M, dtype=torch.int32, device=dev)
    qb, sb = torch.zeros(N, K, dtype=torch.uint8, device=dev), torch.zeros(N, dtype=torch.int32, device=dev)
    ck(L, L.smd_quantize_rows_e4m3(P(aD), K, M, K, P(qa), P(sa), st()))
    ck(L, L.smd_quantize_rows_e4m3(P(bD), K, N, K, P(qb), P(sb), st()))
    bias = (0.1 * torch.randn(N, generator=g)).to(dev)
    ref = dequant(qa, sa) @ dequant(qb, sb).t() + bias.double().cpu()
    if form == "b":
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        ck(L, L.smd_gemm_e4m3_nt(P(qa), K, P(sa), P(qb), K, P(sb), M, N, K, P(bias), None, 0, None, 0, P(out), N, st()))
        tol = 4e-3
    else:
        res = torch.randn(M, N, generator=g).to(dev)
        ref = ref + res.double().cpu()
        out = torch.zeros(M, N, device=dev)
        ck(L, L.smd_gemm_e4m3_nt(P(qa), K, P(sa), P(qb), K, P(sb), M, N, K, P(bias), P(res), N, P(out), N, None, 0, st()))
        tol = 2e-4
    torch.cuda.synchronize()
    e = rel(out.float(), ref)
    print(f"gemm_e4m3_nt {M}x{N}x{K} ({form}): rel {e:.2e} vs fp64 on the dequantised operands")
    assert e < tol


def test_layernorm_fwd_e4m3(L, dev):
    g = torch.Generator().manual_seed(11)
    rows, D, S = 512, 2048, 32
    x = torch.randn(rows, D, generator=g) * 2 + 0.5
    gamma, beta = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    scale = 1 + 0.3 * torch.randn(rows // S, D, generator=g)
    shift = 0.3 * torch.randn(rows // S, D, generator=g)
    xd = x.double()
    mu, var = xd.mean(-1, keepdim=True), xd.var(-1, unbiased=False, keepdim=True)
    y = (xd - mu) / torch.sqrt(var + 1e-6) * gamma.double() + beta.double()
    y = O.swish(scale.double().repeat_interleave(S, 0) * y + shift.double().repeat_interleave(S, 0))
    q = torch.zeros(rows, D, dtype=torch.uint8, device=dev)
    s = torch.zeros(rows, dtype=torch.int32, device=dev)
    ob = torch.zeros(rows, D, dtype=torch.bfloat16, device=dev)
    xD, gD, bD, scD, shD = x.to(dev), gamma.to(dev), beta.to(dev), scale.to(dev), shift.to(dev)      # keep them alive
    ck(L, L.smd_layernorm_fwd_e4m3(P(xD), rows, D, P(gD), P(bD), P(scD), P(shD), D, S, 1, P(q), P(s), P(ob), st()))
    torch.cuda.synchronize()
    d = dequant(q, s)
    print(f"layernorm_fwd_e4m3: e4m3 rel {rel(d, y):.2e}, bf16 copy rel {rel(ob.float(), y):.2e}")
    assert rel(d, y) < 4e-2 and rel(ob.float(), y) < 4e-3
    e_ref = torch.ceil(torch.log2(y.abs().amax(1).double() / 448.0)).to(torch.int64)
    assert ((((s.cpu().to(torch.int64) & 0xFF) - 127) - e_ref).abs() <= 1).all()      # bf16-free fp32 row maximum: same binade


GRAD_TOL = 5e-2      # whole-gradient rel-L2 in fp8 mode: the SURVEY 8c eps_hat tolerance carried over to the gradient


@pytest.mark.parametrize("B,L_,H,K_,C", [(8, 6, 8, 2, 512), (256, 6, 8, 2, 512), (16, 8, 16, 3, 512), (256, 8, 16, 3, 512),
                                         (8, 6, 8, 2, 146), (256, 6, 8, 2, 146)])
def test_fp8_forward_and_train_step_parity(B, L_, H, K_, C):
    """--dtype=fp8 engine vs the fp32 oracle: eps_hat <= 5e-2; the training step (e4m3 forward and dgrad GEMMs of
    the DenseResBlocks, bf16 weight gradients on the bf16 copies of the same activations) stays inside GRAD_TOL of the oracle's
    gradient and 2e-2 of its loss.  Cases: the B = 256 steps `extra_configs.base_fp8 / large_fp8` of the bench line time, and
    the multitrack slice C = 146 that BASELINE config 5 names (configs/ddpm-multi-32seq-512.cfg: ragged in_proj / out_proj)."""
    import smd_amd.ncsn as N
    from smd_amd.engine import NetConfig
    ocfg = O.NetConfig(data_channels=C, num_layers=L_, num_heads=H, num_mlp_layers=K_)
    p = O.init_params(ocfg, 0, torch.float32)
    g = torch.Generator().manual_seed(5)
    for k in p:
        if k.endswith(".bias"):
            p[k] = 0.1 * torch.randn(p[k].shape, generator=g)
        elif k.endswith(".scale"):
            p[k] = 1 + 0.1 * torch.randn(p[k].shape, generator=g)
    cfg = NetConfig(architecture="TransformerDDPM", data_channels=C, seq_len=32, num_layers=L_, num_heads=H, num_mlp_layers=K_,
                    num_timesteps=1000, dtype="fp8")
    model = N.Model(cfg, "cuda:0", seed=None)
    model.engine.load_named(p)
    x = torch.clamp(0.25 * torch.randn(B, 32, C, generator=g), -1, 1)
    s = 0.05 + 0.95 * torch.rand(B, generator=g)
    with torch.no_grad():
        ref = O.make_model(p, ocfg)(x, s.view(B, 1, 1))
    out = model(x, s.view(B, 1, 1))
    e = rel(out, ref)
    bf = N.Model(NetConfig(architecture="TransformerDDPM", data_channels=C, seq_len=32, num_layers=L_, num_heads=H,
                           num_mlp_layers=K_, num_timesteps=1000), "cuda:0", seed=None)
    bf.engine.load_named(p)
    e_bf = rel(bf(x, s.view(B, 1, 1)), ref)
    print(f"fp8 forward B={B} L={L_} H={H} K={K_} C={C}: eps_hat rel {e:.3e} (bf16 engine: {e_bf:.3e})")
    assert e < 5e-2
    assert e > e_bf                                         # the e4m3 path really ran (it cannot be as exact as bf16)
    labels = torch.randint(1, 1001, (B,), generator=g)
    eps = torch.randn(B, 32, C, generator=g)
    leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss_ref = O.diffusion_loss(x, O.make_model(leaf, ocfg), BETAS, labels.numpy(), eps, "none")
    loss_ref.mean().backward()
    eng = model.train_engine(ema=False)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(B, training=True)
    eng.loss_backward(x.cuda(), labels.int().cuda(), eps.cuda(), stage=0)
    torch.cuda.synchronize()
    m_eng, m_ref = float(eng.loss_per_sample().mean()), float(loss_ref.detach().mean())
    e_l = rel(eng.loss_per_sample(), loss_ref)

    def grad_err():
        gv = eng.named_views(eng.grads)
        num = den = 0.0
        worst, worst_name = 0.0, ""
        for k, v in leaf.items():
            d2, n2 = float((gv[k].double().cpu() - v.grad.double()).pow(2).sum()), float(v.grad.double().pow(2).sum())
            num, den = num + d2, den + n2
            if n2 > 0 and (d2 / n2) ** 0.5 > worst:
                worst, worst_name = (d2 / n2) ** 0.5, k
        return (num / den) ** 0.5, worst, worst_name

    e_g, w_g, w_name = grad_err()
    # the same step with the DenseResBlock dgrad GEMMs back on bf16 operands (option fp8_dgrad = 0): the e4m3 dgrads are the
    # default in fp8 mode (4 of the step's 14 large GEMMs more on the 5 PF path) and must stay inside the same tolerance
    g8 = eng.grads.clone()
    eng.set_option("fp8_dgrad", 0)
    eng.loss_backward(x.cuda(), labels.int().cuda(), eps.cuda(), stage=0)
    torch.cuda.synchronize()
    e_gb, _, _ = grad_err()
    eng.set_option("fp8_dgrad", 1)
    print(f"fp8 train step B={B} C={C}: loss {m_eng:.6f} vs {m_ref:.6f} (per-sample rel {e_l:.3e}); gradient whole-vector rel "
          f"{e_g:.3e}, worst tensor {w_name} {w_g:.3e} (bf16 dgrads: {e_gb:.3e})")
    assert abs(m_eng - m_ref) / m_ref < 2e-2
    assert e_g < GRAD_TOL and e_gb < GRAD_TOL
    assert w_g < 0.15                                       # measured <= 4.9e-2 (a FiLM LayerNorm scale / bias)
    assert not torch.equal(g8, eng.grads)                   # the e4m3 dgrad path really ran


@pytest.mark.parametrize("L_,H,K_,C", [(6, 8, 2, 512), (2, 16, 3, 146)])
def test_fp8_gradient_against_the_e4m3_emulating_oracle(L_, H, K_, C):
    """VERDICT r5 weak #1b: the fp8 engine's distance from the exact oracle, ATTRIBUTED.  oracle/e4m3_emulation.py evaluates the
    network in float64 with the engine's e4m3 quantisation (row E8M0 scales, saturating round-to-nearest-even, forward A / B
    operands and the dgrad operands of the DenseResBlock layers) on top of bf16_emulation.py's bf16 rounding points.  Against it the
    engine's eps_hat, per-sample loss and gradient must sit well inside their distance from the exact oracle: what the exact-oracle
    figures (2e-2 eps_hat, 1.2e-2 ... 1.8e-2 gradient at random init) measure is the FORMAT, what is left here is the kernels."""
    import e4m3_emulation as F8
    import smd_amd.ncsn as N
    from smd_amd.engine import NetConfig
    import smd_amd.lib as lib
    B = 8
    ocfg = O.NetConfig(data_channels=C, num_layers=L_, num_heads=H, num_mlp_layers=K_)
    p = O.init_params(ocfg, 0, torch.float64)
    g = torch.Generator().manual_seed(5)
    for k in p:
        if k.endswith(".bias"):
            p[k] = 0.1 * torch.randn(p[k].shape, generator=g, dtype=torch.float64)
        elif k.endswith(".scale"):
            p[k] = 1 + 0.1 * torch.randn(p[k].shape, generator=g, dtype=torch.float64)
    p = {k: v.float().double() for k, v in p.items()}          # the engine holds fp32 masters
    cfg = NetConfig(architecture="TransformerDDPM", data_channels=C, seq_len=32, num_layers=L_, num_heads=H, num_mlp_layers=K_,
                    num_timesteps=1000, dtype="fp8")
    model = N.Model(cfg, "cuda:0", seed=None)
    model.engine.load_named(p)
    x0 = torch.clamp(0.25 * torch.randn(B, 32, C, generator=g), -1, 1)
    labels = torch.randint(1, 1001, (B,), generator=g)
    eps = torch.randn(B, 32, C, generator=g)
    eng = model.train_engine(ema=False)
    eng.set_schedule(BETAS, with_sampler=False)
    eng.bind(B, training=True)
    lv = torch.from_numpy(O.used_alphas_from_labels(BETAS, labels.numpy())).sqrt()
    sd = lv.float().reshape(-1).cuda().contiguous()
    emb = torch.zeros(B, 128, dtype=torch.bfloat16, device="cuda")
    lib.check(lib.get_lib().smd_noise_embed(sd.data_ptr(), B, 128, emb.data_ptr(), 128, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    emb = emb.double().cpu()

    def oracle(mk):
        leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        seen = {}
        base = mk(leaf)

        def capturing(x, cond):
            seen["pred"] = base(x, cond)
            return seen["pred"]
        per = O.diffusion_loss(x0.double(), capturing, BETAS, labels.numpy(), eps.double(), "none")
        per.mean().backward()
        return {k: v.grad for k, v in leaf.items()}, per.detach(), seen["pred"].detach()

    exact = oracle(lambda q: O.make_model(q, ocfg))
    res = {}
    for dgrad in (1, 0):
        eng.set_option("fp8_dgrad", dgrad)
        eng.loss_backward(x0.cuda(), labels.int().cuda(), eps.cuda(), stage=0)
        torch.cuda.synchronize()
        gv = {k: v.double().cpu().clone() for k, v in eng.named_views(eng.grads).items()}
        pred = eng.last_pred().double().cpu().clone()
        loss = eng.loss_per_sample().double().cpu().clone()
        emu = oracle(lambda q: F8.make_model(q, ocfg, backward=True, noise_embedding=emb, fp8_dgrad=bool(dgrad)))

        def dist(ref):
            num = sum(float((gv[k] - ref[k]).pow(2).sum()) for k in ref)
            den = sum(float(ref[k].pow(2).sum()) for k in ref)
            worst = max((rel(gv[k], ref[k]), k) for k in ref if float(ref[k].norm()) > 0)
            return (num / den) ** 0.5, worst
        (ge, _), (gm, (wm, wn)) = dist(exact[0]), dist(emu[0])
        res[dgrad] = (ge, gm)
        print(f"fp8 C={C} L={L_} K={K_} fp8_dgrad={dgrad}: eps_hat vs exact {rel(pred, exact[2]):.3e} / vs e4m3 emulation {rel(pred, emu[2]):.3e}; "
              f"per-sample loss {rel(loss, exact[1]):.3e} / {rel(loss, emu[1]):.3e}; gradient {ge:.3e} / {gm:.3e} (worst tensor {wn} {wm:.3e})")
        assert rel(pred, exact[2]) < 5e-2 and ge < GRAD_TOL
        # the attribution: the emulation explains most of the distance (the residue is the bf16 cascade of
        # test_forward_against_the_bf16_emulating_oracle plus e4m3 rounding flips next to ties: a 2^-4 relative step each)
        assert rel(pred, emu[2]) < 0.6 * rel(pred, exact[2])
        assert gm < 0.5 * ge and gm < 1e-2
        assert rel(loss, emu[1]) < 0.5 * rel(loss, exact[1]) + 1e-4
    eng.set_option("fp8_dgrad", 1)
