"""GPU parity tests of the single HIP kernels, called through the C-ABI (include/smd_hip.h).

Each kernel is compared with a plain torch fp32/fp64 CPU computation of the same op on identical
(bf16-rounded where applicable) inputs.  Tolerances are written next to each check.
"""
import math

import numpy as np
import pytest
import torch

import ddpm_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import smd_amd.lib as lib
    return lib.get_lib()


def P(t):
    return None if t is None else t.data_ptr()


def st():
    return torch.cuda.current_stream().cuda_stream


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def ck(lib_mod, rc):
    import smd_amd.lib as lib
    lib.check(rc)


def bf(x):
    return x.to(torch.bfloat16)


# ------------------------------------------------------------------------------------------------
def test_probe_tr_read(L, dev):
    """Documents the lane semantics of ds_read_b64_tr_b16 (what gemm_tn relies on)."""
    img = bf(torch.arange(1024, dtype=torch.float32) % 256).to(dev)
    out = torch.zeros(256, dtype=torch.bfloat16, device=dev)
    ck(L, L.smd_probe_tr_read(P(img), P(out), st()))
    torch.cuda.synchronize()
    got = out.float().cpu().view(64, 4).long()
    lane = torch.arange(64)
    # expected (guide): lane l, elem j reads img[(l&15) + 16*j + 64*(l>>4)]
    exp = torch.stack([(lane & 15) + 16 * j + 64 * (lane >> 4) for j in range(4)], dim=1)
    print("tr_read lanes 0..3:", got[:4].tolist(), "lane 17:", got[17].tolist())
    assert torch.equal(got, exp), f"unexpected ds_read_b64_tr_b16 layout:\n{got[:20]}"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (300, 42, 192), (1000, 4096, 512),
                                   (8192, 2048, 2048), (64, 512, 2048), (512, 128, 2048), (96, 200, 1024),
                                   (8192, 128, 2048)])      # the last three: 32-row tiles with two K-groups per workgroup
def test_gemm_nt_plain(L, dev, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = bf(torch.randn(M, K, generator=g))
    Bt = bf(torch.randn(N, K, generator=g) * 0.5)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ Bt.double().t() + bias.double()
    Ad, Bd, bd = A.to(dev), Bt.to(dev), bias.to(dev)
    out = torch.full((M, N), float("nan"), device=dev)
    outb = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    ck(L, L.smd_gemm_bf16_nt(P(Ad), K, P(Bd), K, M, N, K, P(bd), 0, None, 0, P(out), N, P(outb), N, st()))
    torch.cuda.synchronize()
    e = rel(out, ref)
    eb = rel(outb.float(), ref)
    print(f"gemm_nt {M}x{N}x{K}: rel fp32 {e:.2e} bf16 {eb:.2e}")
    assert e < 2e-5            # fp32 accumulation of exact bf16 products
    assert eb < 4e-3           # one bf16 rounding of the output


def test_gemm_nt_epilogues(L, dev):
    M, N, K = 320, 200, 128
    g = torch.Generator().manual_seed(5)
    A, Bt = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) * 0.2)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    z = A.double() @ Bt.double().t() + bias.double()
    Ad, Bd, bd, rd = A.to(dev), Bt.to(dev), bias.to(dev), res.to(dev)     # keep alive until the sync
    for act, fn in ((1, lambda v: O.gelu(v)), (2, lambda v: O.swish(v))):
        ref = fn(z) + res.double()
        out = torch.empty(M, N, device=dev)
        ck(L, L.smd_gemm_bf16_nt(P(Ad), K, P(Bd), K, M, N, K, P(bd), act, P(rd), N,
                                 P(out), N, None, 0, st()))
        torch.cuda.synchronize()
        e = rel(out, ref)
        print(f"gemm_nt epilogue act={act}: rel {e:.2e}")
        assert e < 1e-4        # fast-math exp in gelu/swish
    # in-place residual stream: out == residual buffer
    buf = res.clone().to(dev)
    ck(L, L.smd_gemm_bf16_nt(P(Ad), K, P(Bd), K, M, N, K, P(bd), 0, P(buf), N, P(buf), N,
                             None, 0, st()))
    torch.cuda.synchronize()
    assert rel(buf, z + res.double()) < 2e-5


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 256), (256, 512, 1024), (1024, 256, 2048)])
def test_gemm_nt256_forced(L, dev, M, N, K, variant):
    """The 256x256 8-phase kernel on small grids (forced), every epilogue input, against fp64."""
    import smd_amd.lib as lib
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = bf(torch.randn(M, K, generator=g))
    Bt = bf(torch.randn(N, K, generator=g) * 0.5)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    z = A.double() @ Bt.double().t() + bias.double()
    Ad, Bd, bd, rd = A.to(dev), Bt.to(dev), bias.to(dev), res.to(dev)
    lib.check(L.smd_set_tuning(b"gemm_nt256", 2))
    lib.check(L.smd_set_tuning(b"gemm_nt256_variant", variant))
    try:
        out = torch.full((M, N), float("nan"), device=dev)
        outb = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        ck(L, L.smd_gemm_bf16_nt(P(Ad), K, P(Bd), K, M, N, K, P(bd), 0, None, 0, P(out), N, P(outb), N, st()))
        out2 = torch.empty(M, N, device=dev)
        ck(L, L.smd_gemm_bf16_nt(P(Ad), K, P(Bd), K, M, N, K, P(bd), 1, P(rd), N, P(out2), N, None, 0, st()))
        torch.cuda.synchronize()
    finally:
        lib.check(L.smd_set_tuning(b"gemm_nt256", 1))
        lib.check(L.smd_set_tuning(b"gemm_nt256_variant", 0))
    e, eb, e2 = rel(out, z), rel(outb.float(), z), rel(out2, O.gelu(z) + res.double())
    print(f"gemm_nt256 v{variant} {M}x{N}x{K}: rel fp32 {e:.2e} bf16 {eb:.2e} gelu+res {e2:.2e}")
    assert e < 2e-5 and eb < 4e-3 and e2 < 1e-4


def test_gemm_nt256_matches_128_tiles_and_is_deterministic(L, dev):
    """Full-size DenseResBlock GEMM: the 8-phase kernel vs the 128-wide kernel (same fp32 accumulation order per
    K-step is not guaranteed -> tolerance), and 6 repeated launches bit-identical (race screen)."""
    import smd_amd.lib as lib
    M, N, K = 8192, 2048, 2048
    g = torch.Generator().manual_seed(11)
    Ad = bf(torch.randn(M, K, generator=g)).to(dev)
    Bd = bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    bd = torch.randn(N, generator=g).to(dev)
    outs = []
    for mode in (0, 1, 1, 1, 1, 1, 1):
        lib.check(L.smd_set_tuning(b"gemm_nt256", mode))
        o = torch.empty(M, N, device=dev)
        ck(L, L.smd_gemm_bf16_nt(P(Ad), K, P(Bd), K, M, N, K, P(bd), 0, None, 0, P(o), N, None, 0, st()))
        outs.append(o)
    torch.cuda.synchronize()
    lib.check(L.smd_set_tuning(b"gemm_nt256", 1))
    e = rel(outs[1], outs[0])
    print(f"gemm_nt256 vs 128-tile kernel at {M}x{N}x{K}: rel {e:.2e}")
    assert e < 1e-5
    for o in outs[2:]:
        assert torch.equal(o, outs[1])


@pytest.mark.parametrize("M,N,K,with_bias", [(256, 256, 128, True), (512, 768, 256, False), (8192, 2048, 2048, True), (8192, 2048, 2048, False)])
def test_gemm_nt256_packed_bf16_epilogue_is_bitwise_the_staged_one(L, dev, M, N, K, with_bias):
    """(bias ->) bf16 outputs of the 256x256 kernel (fc1 of a DenseResBlock, `up`, every dgrad: 7 of the 10 launches of a train
    step) round the accumulators BEFORE the LDS staging and stage row PAIRS as dwords (knob gemm_nt256_pk, default on): half
    the ds_write / ds_read traffic of the fp32 staging.  Same arithmetic (fp32 bias add, one RNE rounding): bit-identical."""
    import smd_amd.lib as lib
    g = torch.Generator().manual_seed(M + N + K + int(with_bias))
    Ad = bf(torch.randn(M, K, generator=g)).to(dev)
    Bd = bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    bd = torch.randn(N, generator=g).to(dev) if with_bias else None
    outs = {}
    lib.check(L.smd_set_tuning(b"gemm_nt256", 2))
    try:
        for pk in (1, 0, 1):
            lib.check(L.smd_set_tuning(b"gemm_nt256_pk", pk))
            o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
            ck(L, L.smd_gemm_bf16_nt(P(Ad), K, P(Bd), K, M, N, K, P(bd), 0, None, 0, None, 0, P(o), N, st()))
            torch.cuda.synchronize()
            outs.setdefault(pk, []).append(o)
    finally:
        lib.check(L.smd_set_tuning(b"gemm_nt256", 1))
        lib.check(L.smd_set_tuning(b"gemm_nt256_pk", 1))
    ref = Ad.double().cpu() @ Bd.double().cpu().t() + (bd.double().cpu() if with_bias else 0.0)
    assert bool(torch.isfinite(outs[1][0].float()).all())
    assert rel(outs[1][0].float(), ref) < 4e-3
    assert torch.equal(outs[1][0], outs[0][0]), "packed-bf16 epilogue differs from the fp32-staged one"
    assert torch.equal(outs[1][0], outs[1][1])


def test_gemm_nt_rejects_bad_k(L, dev):
    import smd_amd.lib as lib
    a = torch.zeros(64, 96, dtype=torch.bfloat16, device=dev)
    o = torch.zeros(64, 64, device=dev)
    with pytest.raises(ValueError):
        lib.check(L.smd_gemm_bf16_nt(P(a), 96, P(a), 96, 64, 64, 96, None, 0, None, 0, P(o), 64, None, 0, st()))


@pytest.mark.parametrize("tr_path", [1, 0])
@pytest.mark.parametrize("M,Kd,N,ldx", [(256, 128, 128, 128), (8192, 42, 128, 64), (96, 512, 4096, 512),
                                         (8192, 128, 2048, 128), (8192, 2048, 2048, 2048), (4096, 2048, 146, 2048)])
def test_gemm_tn(L, dev, tr_path, M, Kd, N, ldx):
    g = torch.Generator().manual_seed(M + Kd + N)
    ldy = (N + 63) // 64 * 64
    X = torch.zeros(M, ldx)
    X[:, :Kd] = torch.randn(M, Kd, generator=g)
    Y = torch.zeros(M, ldy)
    Y[:, :N] = torch.randn(M, N, generator=g) * 0.1
    X, Y = bf(X), bf(Y)
    ref = X[:, :Kd].double().t() @ Y[:, :N].double()
    out = torch.full((Kd, N), float("nan"), device=dev)
    db = torch.full((N,), float("nan"), device=dev)
    Mp = (M + 63) // 64 * 64
    zero = torch.zeros(128, dtype=torch.bfloat16, device=dev)
    slab = torch.full((int(L.smd_gemm_tn_slab_elems()),), float("nan"), device=dev)
    scratch = torch.zeros(max(128, (Kd + N) * Mp if not tr_path else 128), dtype=torch.bfloat16, device=dev)
    Xd, Yd = X.to(dev), Y.to(dev)
    ck(L, L.smd_gemm_bf16_tn(P(Xd), ldx, P(Yd), ldy, M, Kd, N, P(out), N, P(db), P(zero), P(slab), slab.numel(),
                             P(scratch), scratch.numel(), tr_path, st()))
    torch.cuda.synchronize()
    e = rel(out, ref)
    eb = rel(db, Y[:, :N].double().sum(0))
    print(f"gemm_tn tr={tr_path} M={M} Kd={Kd} N={N}: rel dW {e:.2e} db {eb:.2e}")
    assert e < 3e-5            # fp32 accumulation
    assert eb < 3e-5
    if tr_path:                # split-K slabs are reduced in a fixed order: bitwise reproducible
        out2 = torch.empty_like(out)
        ck(L, L.smd_gemm_bf16_tn(P(Xd), ldx, P(Yd), ldy, M, Kd, N, P(out2), N, P(db), P(zero), P(slab), slab.numel(),
                                 P(scratch), scratch.numel(), tr_path, st()))
        torch.cuda.synchronize()
        assert torch.equal(out, out2)


@pytest.mark.parametrize("M,Kd,N,ldx,ldy", [(512, 256, 256, 256, 256), (1000, 512, 256, 512, 320), (4130, 256, 768, 264, 768),
                                             (8192, 2048, 512, 2048, 512), (8192, 2048, 2048, 2048, 2048)])
def test_gemm_tn256(L, dev, M, Kd, N, ldx, ldy):
    """256x256 8-phase wgrad kernel (forced on small grids): ragged M (descriptor zero-fill), padded leading
    dimensions, balanced bias partials, split-K slabs; vs fp64 and vs the 128-wide kernel; bitwise repeatable."""
    import smd_amd.lib as lib
    g = torch.Generator().manual_seed(M + Kd + N)
    X = torch.zeros(M, ldx)
    X[:, :Kd] = torch.randn(M, Kd, generator=g)
    Y = torch.zeros(M, ldy)
    Y[:, :N] = torch.randn(M, N, generator=g) * 0.1 + 0.01
    X, Y = bf(X), bf(Y)
    ref = X[:, :Kd].double().t() @ Y[:, :N].double()
    refb = Y[:, :N].double().sum(0)
    zero = torch.zeros(128, dtype=torch.bfloat16, device=dev)
    slab = torch.full((int(L.smd_gemm_tn_slab_elems()),), float("nan"), device=dev)
    scratch = torch.zeros(128, dtype=torch.bfloat16, device=dev)
    Xd, Yd = X.to(dev), Y.to(dev)
    outs = []
    for mode in (2, 2, 0):
        lib.check(L.smd_set_tuning(b"gemm_tn256", mode))
        out = torch.full((Kd, N), float("nan"), device=dev)
        db = torch.full((N,), float("nan"), device=dev)
        ck(L, L.smd_gemm_bf16_tn(P(Xd), ldx, P(Yd), ldy, M, Kd, N, P(out), N, P(db), P(zero), P(slab), slab.numel(),
                                 P(scratch), scratch.numel(), 1, st()))
        torch.cuda.synchronize()
        outs.append((out, db))
    lib.check(L.smd_set_tuning(b"gemm_tn256", 1))
    e, eb = rel(outs[0][0], ref), rel(outs[0][1], refb)
    e128 = rel(outs[0][0], outs[2][0])
    print(f"gemm_tn256 M={M} Kd={Kd} N={N}: rel dW {e:.2e} db {eb:.2e} vs 128-wide {e128:.2e}")
    assert e < 3e-5 and eb < 3e-5 and e128 < 3e-5
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("rows,M", [(32, 128), (96, 256), (8192, 2048)])
def test_mlp_block_fwd_fused(L, dev, rows, M):
    """Fused LN + fc1 + GELU + fc2 + residual (encoder_fused.hip) vs fp64 with the kernel's two bf16 roundings
    (LN output, GELU output) applied in the reference; saved activations; in-place residual stream."""
    g = torch.Generator().manual_seed(rows + M)
    h = torch.randn(rows, 128, generator=g) * 1.5 + 0.3
    gamma, beta = 1 + 0.2 * torch.randn(128, generator=g), 0.1 * torch.randn(128, generator=g)
    W1 = bf(torch.randn(128, M, generator=g) * 0.09)          # kernel (in, out)
    b1 = 0.1 * torch.randn(M, generator=g)
    W2 = bf(torch.randn(M, 128, generator=g) * (1.0 / math.sqrt(M)))
    b2 = 0.1 * torch.randn(128, generator=g)
    hd = h.double()
    mu, var = hd.mean(-1, keepdim=True), hd.var(-1, unbiased=False, keepdim=True)
    a2 = bf(((hd - mu) / torch.sqrt(var + 1e-6) * gamma.double() + beta.double()).float())
    z = a2.double() @ W1.double() + b1.double()
    u = bf(O.gelu(z).float())
    ref = hd + u.double() @ W2.double() + b2.double()
    hD, gD, bD = h.to(dev), gamma.to(dev), beta.to(dev)
    W1t, W2t = W1.t().contiguous().to(dev), W2.t().contiguous().to(dev)      # [M][128], [128][M]
    b1D, b2D = b1.to(dev), b2.to(dev)
    out = torch.full((rows, 128), float("nan"), device=dev)
    sa = torch.zeros(rows, 128, dtype=torch.bfloat16, device=dev)
    sz = torch.zeros(rows, M, dtype=torch.bfloat16, device=dev)
    su = torch.zeros(rows, M, dtype=torch.bfloat16, device=dev)
    ck(L, L.smd_mlp_block_fwd(P(hD), P(out), rows, P(gD), P(bD), P(W1t), P(b1D), P(W2t), P(b2D), M, P(sa), P(sz), P(su), st()))
    inpl = hD.clone()
    ck(L, L.smd_mlp_block_fwd(P(inpl), P(inpl), rows, P(gD), P(bD), P(W1t), P(b1D), P(W2t), P(b2D), M, None, None, None, st()))
    torch.cuda.synchronize()
    e_out = rel(out.double().cpu() - hd, ref - hd)
    e_a2, e_z, e_u = rel(sa.float(), a2.float()), rel(sz.float(), z), rel(su.float(), u.float())
    print(f"mlp_block_fwd rows={rows} M={M}: delta rel {e_out:.2e} a2 {e_a2:.2e} z1 {e_z:.2e} u {e_u:.2e}")
    assert e_out < 3e-3        # bf16 flips of a2/u at rounding boundaries + fast-math gelu
    assert e_a2 < 2e-3 and e_z < 4e-3 and e_u < 4e-3
    assert torch.equal(inpl, out)


@pytest.mark.parametrize("rows,M", [(32, 512), (192, 1024), (128, 2048), (8192, 2048)])
def test_mlp_block_fwd_hidden_split(L, dev, rows, M):
    """mlp_hs_fwd (4 hidden quarters x groups of 4 / 2 / 1 samples -> four fp32 partial tiles) + ln128_parts (their
    fixed-order sum and the LayerNorm of it) vs fp64 with the kernel's bf16 rounding of the GELU output; repeatable."""
    g = torch.Generator().manual_seed(rows + M + 1)
    h = torch.randn(rows, 128, generator=g) * 1.5 + 0.3
    a2 = bf(torch.randn(rows, 128, generator=g))
    gamma, beta = 1 + 0.2 * torch.randn(128, generator=g), 0.1 * torch.randn(128, generator=g)
    W1 = bf(torch.randn(128, M, generator=g) * 0.09)
    b1 = 0.1 * torch.randn(M, generator=g)
    W2 = bf(torch.randn(M, 128, generator=g) * (1.0 / math.sqrt(M)))
    b2 = 0.1 * torch.randn(128, generator=g)
    hd = h.double()
    z = a2.double() @ W1.double() + b1.double()
    u = bf(O.gelu(z).float())
    ref = hd + u.double() @ W2.double() + b2.double()
    mu, var = ref.mean(-1, keepdim=True), ref.var(-1, unbiased=False, keepdim=True)
    ln_ref = (ref - mu) / torch.sqrt(var + 1e-6) * gamma.double() + beta.double()
    hD, a2D, gD, bD = h.to(dev), a2.to(dev), gamma.to(dev), beta.to(dev)
    W1t, W2t = W1.t().contiguous().to(dev), W2.t().contiguous().to(dev)
    b1D, b2D = b1.to(dev), b2.to(dev)
    outs = []
    for rep in range(3):
        part = torch.full((4, rows, 128), float("nan"), device=dev)
        ck(L, L.smd_mlp_block_fwd_hs(P(a2D), P(hD), rows, P(W1t), P(b1D), P(W2t), P(b2D), M, P(part), st()))
        xo = torch.full((rows, 128), float("nan"), device=dev)
        lo = torch.zeros(rows, 128, dtype=torch.bfloat16, device=dev)
        ck(L, L.smd_ln128_parts(P(part), rows * 128, rows, P(gD), P(bD), P(xo), P(lo), st()))
        outs.append((part, xo, lo))
    torch.cuda.synchronize()
    part, xo, lo = outs[0]
    assert torch.equal(xo, (part[0] + part[1]) + (part[2] + part[3]))
    e_out = rel(xo.double().cpu() - hd, ref - hd)
    e_ln = rel(lo.float(), ln_ref)
    print(f"mlp_block_fwd_hs rows={rows} M={M}: delta rel {e_out:.2e}; ln(x) rel {e_ln:.2e}")
    assert e_out < 3e-3 and e_ln < 4e-3
    for k in (1, 2):
        assert torch.equal(outs[k][0], part) and torch.equal(outs[k][2], lo)


@pytest.mark.parametrize("rows,M", [(128, 512), (256, 2048), (8192, 2048)])
def test_mlp_block_bwd_hidden_split(L, dev, rows, M):
    """mlp_hs_bwd (recompute) vs fp64: u = gelu(a2 W1 + b1), dz = (dh W2^T) gelu'(z) (bf16 outputs) and da2 = dz W1^T as
    the fixed-order sum of the four partial tiles; then ln128_bwd_parts on those tiles vs fp64 autograd of LayerNorm."""
    g = torch.Generator().manual_seed(rows + M + 3)
    a2 = bf(torch.randn(rows, 128, generator=g))
    dh = bf(torch.randn(rows, 128, generator=g) * 0.05)
    W1 = bf(torch.randn(128, M, generator=g) * 0.09)          # fc1 kernel (in 128, out M)
    b1 = 0.1 * torch.randn(M, generator=g)
    W2 = bf(torch.randn(M, 128, generator=g) * (1.0 / math.sqrt(M)))   # fc2 kernel (in M, out 128)
    z = (a2.double() @ W1.double() + b1.double()).requires_grad_(True)
    u_ref = O.gelu(z)
    (gz,) = torch.autograd.grad(u_ref.sum(), z)                # gelu'(z)
    du = dh.double() @ W2.double().t()
    dz_ref = du * gz
    da2_ref = bf(dz_ref.float()).double() @ W1.double().t()    # the kernel contracts the bf16-rounded dz
    a2D, dhD = a2.to(dev), dh.to(dev)
    W1t = W1.t().contiguous().to(dev)                          # [M][128] forward pack of fc1
    W2p = W2.contiguous().to(dev)                              # [M][128] dgrad pack of fc2
    W1p = W1.contiguous().to(dev)                              # [128][M] dgrad pack of fc1
    b1D = b1.to(dev)
    res = []
    for rep in range(2):
        u = torch.zeros(rows, M, dtype=torch.bfloat16, device=dev)
        dz = torch.zeros(rows, M, dtype=torch.bfloat16, device=dev)
        part = torch.full((4, rows, 128), float("nan"), device=dev)
        ck(L, L.smd_mlp_block_bwd_hs(P(a2D), P(dhD), rows, P(W1t), P(W2p), P(W1p), P(b1D), M, P(u), P(dz), P(part), st()))
        res.append((u, dz, part))
    torch.cuda.synchronize()
    u, dz, part = res[0]
    da2 = ((part[0] + part[1]) + (part[2] + part[3])).double().cpu()
    e_u, e_dz, e_da = rel(u.float(), u_ref.detach()), rel(dz.float(), dz_ref.detach()), rel(da2, da2_ref.detach())
    print(f"mlp_block_bwd_hs rows={rows} M={M}: u {e_u:.2e} dz {e_dz:.2e} da2 {e_da:.2e}")
    assert e_u < 4e-3 and e_dz < 6e-3 and e_da < 4e-3
    assert all(torch.equal(a_, b_) for a_, b_ in zip(res[0], res[1]))
    # ---- ln2 backward on the partial tiles
    x = torch.randn(rows, 128, generator=g) * 1.3 + 0.2
    gamma = 1 + 0.2 * torch.randn(128, generator=g)
    dres = 0.05 * torch.randn(rows, 128, generator=g)
    xd = x.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = torch.zeros(128, dtype=torch.float64, requires_grad=True)
    mu, var = xd.mean(-1, keepdim=True), xd.var(-1, unbiased=False, keepdim=True)
    y = (xd - mu) / torch.sqrt(var + 1e-6) * gd + bd
    dout = ((part[0] + part[1]) + (part[2] + part[3])).double().cpu()
    y.backward(dout)
    dx_ref = xd.grad + dres.double()
    xD, gD, rD = x.to(dev), gamma.to(dev), dres.to(dev)
    partial = torch.zeros(rows // 32, 2, 128, device=dev)
    dxb = torch.zeros(rows, 128, dtype=torch.bfloat16, device=dev)
    inpl = rD.clone()
    ck(L, L.smd_ln128_bwd_parts(P(xD), P(part), rows * 128, rows, P(gD), P(inpl), P(inpl), P(dxb), P(partial), st()))
    torch.cuda.synchronize()
    e_dx, e_dxb = rel(inpl, dx_ref), rel(dxb.float(), dx_ref)
    e_g, e_b = rel(partial[:, 0].sum(0), gd.grad), rel(partial[:, 1].sum(0), bd.grad)
    print(f"ln128_bwd_parts rows={rows}: dx {e_dx:.2e} (bf16 {e_dxb:.2e}) dgamma {e_g:.2e} dbeta {e_b:.2e}")
    assert e_dx < 2e-5 and e_dxb < 4e-3 and e_g < 2e-5 and e_b < 2e-5


@pytest.mark.parametrize("rows,H", [(64, 8), (8192, 16)])
def test_attn_block_fwd_partial_sum_input_and_ln2(L, dev, rows, H):
    """smd_attn_block_fwd_ex: the input given as four partial tiles must give bitwise the result of the plain call on
    their fixed-order sum; the emitted a2 must be the LayerNorm of the output rows."""
    g = torch.Generator().manual_seed(rows + H + 7)
    E = 128
    parts = torch.randn(4, rows, E, generator=g) * 0.7
    gamma, beta = 1 + 0.2 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)
    gamma2, beta2 = 1 + 0.2 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)
    Wqkv_t = bf(torch.randn(3 * E, E, generator=g) * 0.12).to(dev)
    bqkv = (0.1 * torch.randn(3 * E, generator=g)).to(dev)
    Wo_t = bf(torch.randn(E, E, generator=g) * 0.09).to(dev)
    bo = (0.1 * torch.randn(E, generator=g)).to(dev)
    pD, gD, bD, g2D, b2D = parts.to(dev), gamma.to(dev), beta.to(dev), gamma2.to(dev), beta2.to(dev)
    x = (pD[0] + pD[1]) + (pD[2] + pD[3])
    plain = torch.empty(rows, E, device=dev)
    ck(L, L.smd_attn_block_fwd(P(x), P(plain), rows, P(gD), P(bD), P(Wqkv_t), P(bqkv), P(Wo_t), P(bo), H, None, None, None, st()))
    out = torch.empty(rows, E, device=dev)
    comb = torch.empty(rows, E, device=dev)
    a2 = torch.zeros(rows, E, dtype=torch.bfloat16, device=dev)
    ck(L, L.smd_attn_block_fwd_ex(None, P(pD), rows * E, P(comb), P(out), rows, P(gD), P(bD), P(Wqkv_t), P(bqkv), P(Wo_t), P(bo), H,
                                  P(g2D), P(b2D), P(a2), None, None, None, st()))
    torch.cuda.synchronize()
    assert torch.equal(comb, x)
    assert torch.equal(out, plain)
    od = out.double().cpu()
    mu, var = od.mean(-1, keepdim=True), od.var(-1, unbiased=False, keepdim=True)
    ln = (od - mu) / torch.sqrt(var + 1e-6) * gamma2.double() + beta2.double()
    e = rel(a2.float(), ln)
    print(f"attn_block_fwd_ex rows={rows} H={H}: a2 rel {e:.2e}")
    assert e < 4e-3


@pytest.mark.parametrize("rows,H", [(32, 8), (96, 16), (64, 4), (8192, 8)])
def test_attn_block_fwd_fused(L, dev, rows, H):
    """Fused LN + QKV + attention + out_proj + residual (encoder_fused.hip) vs the fp64 oracle attention with the
    kernel's bf16 roundings of the LN output, q/k/v and o applied in the reference."""
    g = torch.Generator().manual_seed(rows + H)
    E, d = 128, 128 // H
    h = torch.randn(rows, E, generator=g) * 1.2 - 0.2
    gamma, beta = 1 + 0.2 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)
    Wqkv = bf(torch.randn(E, 3 * E, generator=g) * 0.12)        # kernel (in, out), out = [q | k | v], head h at cols h*d..
    bqkv = 0.1 * torch.randn(3 * E, generator=g)
    Wo = bf(torch.randn(E, E, generator=g) * 0.09)
    bo = 0.1 * torch.randn(E, generator=g)
    hd = h.double()
    mu, var = hd.mean(-1, keepdim=True), hd.var(-1, unbiased=False, keepdim=True)
    a1 = bf(((hd - mu) / torch.sqrt(var + 1e-6) * gamma.double() + beta.double()).float())
    qkv = bf((a1.double() @ Wqkv.double() + bqkv.double()).float())
    B = rows // 32
    q, k, v = [t.double().view(B, 32, H, d).transpose(1, 2) for t in qkv.split(E, dim=-1)]       # (B, H, 32, d)
    qs = bf((q / math.sqrt(d)).float()).double()
    p = torch.softmax(qs @ k.transpose(-1, -2), dim=-1)
    o = bf((p @ v).transpose(1, 2).reshape(rows, E).float())
    ref = hd + o.double() @ Wo.double() + bo.double()
    D = lambda t: t.to(dev)
    hD, gD, bD, WqD, bqD, WoD, boD = D(h), D(gamma), D(beta), D(Wqkv.t().contiguous()), D(bqkv), D(Wo.t().contiguous()), D(bo)
    out = torch.full((rows, E), float("nan"), device=dev)
    sa = torch.zeros(rows, E, dtype=torch.bfloat16, device=dev)
    sq = torch.zeros(rows, 3 * E, dtype=torch.bfloat16, device=dev)
    so = torch.zeros(rows, E, dtype=torch.bfloat16, device=dev)
    ck(L, L.smd_attn_block_fwd(P(hD), P(out), rows, P(gD), P(bD), P(WqD), P(bqD), P(WoD), P(boD), H, P(sa), P(sq), P(so), st()))
    inpl = hD.clone()
    ck(L, L.smd_attn_block_fwd(P(inpl), P(inpl), rows, P(gD), P(bD), P(WqD), P(bqD), P(WoD), P(boD), H, None, None, None, st()))
    torch.cuda.synchronize()
    e_out = rel(out.double().cpu() - hd, ref - hd)
    e_a1, e_qkv, e_o = rel(sa.float(), a1.float()), rel(sq.float(), qkv.float()), rel(so.float(), o.float())
    print(f"attn_block_fwd rows={rows} H={H}: delta rel {e_out:.2e} a1 {e_a1:.2e} qkv {e_qkv:.2e} o {e_o:.2e}")
    assert e_a1 < 2e-3 and e_qkv < 2e-3
    assert e_o < 6e-3          # p is rounded to bf16 before the P V product on the matrix cores
    assert e_out < 6e-3
    assert torch.equal(inpl, out)


@pytest.mark.parametrize("rows,H", [(32, 8), (96, 16), (64, 4), (2048, 8)])
def test_attn_block_bwd_fused(L, dev, rows, H):
    """Fused out_proj dgrad + attention backward + qkv dgrad (encoder_fused.hip) vs fp64 autograd of the attention
    core on the same bf16 q, k, v (dO rounded to bf16 as in the kernel)."""
    g = torch.Generator().manual_seed(7 * rows + H)
    E, d, B = 128, 128 // H, rows // 32
    qkv = bf(torch.randn(rows, 3 * E, generator=g) * 0.8)
    dh = bf(torch.randn(rows, E, generator=g) * 0.05)
    Wo = bf(torch.randn(E, E, generator=g) * 0.09)               # kernel (in, out) == the dgrad pack W [K][N]
    Wqkv = bf(torch.randn(E, 3 * E, generator=g) * 0.09)
    do = bf((dh.double() @ Wo.double().t()).float())             # dO[token][in] = sum_out dh[token][out] W[in][out]
    leaf = qkv.double().clone().requires_grad_(True)
    q, k, v = [t.view(B, 32, H, d).transpose(1, 2) for t in leaf.split(E, dim=-1)]
    p = torch.softmax((q / math.sqrt(d)) @ k.transpose(-1, -2), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(rows, E)
    (o * do.double()).sum().backward()
    dqkv_ref = leaf.grad
    D = lambda t: t.to(dev)
    dhD, qkvD, WoD, WqD = D(dh), D(qkv), D(Wo), D(Wqkv)
    dq = torch.zeros(rows, 3 * E, dtype=torch.bfloat16, device=dev)
    da1 = torch.zeros(rows, E, dtype=torch.bfloat16, device=dev)
    ck(L, L.smd_attn_block_bwd(P(dhD), P(qkvD), P(WoD), P(WqD), P(dq), P(da1), rows, H, st()))
    torch.cuda.synchronize()
    e_q, e_k, e_v = [rel(dq[:, i * E:(i + 1) * E].float(), dqkv_ref[:, i * E:(i + 1) * E]) for i in range(3)]
    da1_ref = dq.double().cpu() @ Wqkv.double().t()
    e_a = rel(da1.float(), da1_ref)
    print(f"attn_block_bwd rows={rows} H={H}: dq {e_q:.2e} dk {e_k:.2e} dv {e_v:.2e} da1 {e_a:.2e}")
    assert max(e_q, e_k, e_v) < 1e-2         # p and ds are rounded to bf16 before the matrix-core products
    assert e_a < 4e-3


@pytest.mark.parametrize("rows,H", [(32, 8), (96, 16), (64, 4), (8192, 8)])
def test_attn_block_bwd_with_both_layernorm_backwards(L, dev, rows, H):
    """smd_attn_block_bwd_ln (round 6): LayerNorm-2 backward on the four da2 partial tiles + residual, the attention half-layer
    backward, LayerNorm-1 backward + residual in ONE launch (models/ncsn.py:159-164 backwards).  Checked (i) against fp64 of its own
    inputs, stage by stage, and (ii) against the three launches it replaces (smd_ln128_bwd_parts -> smd_attn_block_bwd ->
    smd_layernorm_bwd_ex): dh_mid and dqkv to the last bit of fp32 (the same arithmetic on the same layout), the final dh to fp32 rounding (the
    stand-alone LayerNorm-1 kernel sums its rows in another order; bf16 outputs up to rounding flips next to ties)."""
    g = torch.Generator().manual_seed(11 * rows + H)
    E, d, B = 128, 128 // H, rows // 32
    h_mid = torch.randn(rows, E, generator=g) * 1.3 + 0.2
    h = torch.randn(rows, E, generator=g) * 0.9 - 0.1
    parts = torch.randn(4, rows, E, generator=g) * 0.02
    dh = torch.randn(rows, E, generator=g) * 0.05
    g2, g1 = 1 + 0.2 * torch.randn(E, generator=g), 1 + 0.2 * torch.randn(E, generator=g)
    qkv = bf(torch.randn(rows, 3 * E, generator=g) * 0.8)
    Wo = bf(torch.randn(E, E, generator=g) * 0.09)
    Wqkv = bf(torch.randn(E, 3 * E, generator=g) * 0.09)

    def ln_bwd64(x, gamma, dout):
        xr = x.double().requires_grad_(True)
        gr, br = gamma.double().requires_grad_(True), torch.zeros(E, dtype=torch.float64, requires_grad=True)
        O.layer_norm(xr, {"n.scale": gr, "n.bias": br}, "n").backward(dout.double())
        return xr.grad, gr.grad, br.grad

    Dv = lambda t: t.to(dev)
    hmD, hD, pD, g2D, g1D, qkvD, WoD, WqD = Dv(h_mid), Dv(h), Dv(parts.contiguous()), Dv(g2), Dv(g1), Dv(qkv), Dv(Wo), Dv(Wqkv)
    dhD = Dv(dh).clone()
    dq = torch.zeros(rows, 3 * E, dtype=torch.bfloat16, device=dev)
    da1 = torch.zeros(rows, E, dtype=torch.bfloat16, device=dev)
    dh_mid_o = torch.zeros(rows, E, dtype=torch.bfloat16, device=dev)
    dh_o = torch.zeros(rows, E, dtype=torch.bfloat16, device=dev)
    p2 = torch.zeros(rows // 32, 2, E, device=dev)
    p1 = torch.zeros(rows // 32, 2, E, device=dev)
    ck(L, L.smd_attn_block_bwd_ln(P(qkvD), P(WoD), P(WqD), P(dq), P(da1), P(hmD), P(pD), rows * E, P(g2D), P(dhD), P(dh_mid_o), P(p2),
                                  P(hD), P(g1D), P(dh_o), P(p1), rows, H, st()))
    torch.cuda.synchronize()
    # ---- (i) fp64, stage by stage
    da2 = parts.double().sum(0)
    dx2, dg2, db2 = ln_bwd64(h_mid, g2, da2)
    dh_mid_ref = dx2 + dh.double()
    e_mid = rel(dh_mid_o.float(), dh_mid_ref)
    do = bf((dh_mid_o.double().cpu() @ Wo.double().t()).float())
    leaf = qkv.double().clone().requires_grad_(True)
    q, k, v = [t.view(B, 32, H, d).transpose(1, 2) for t in leaf.split(E, dim=-1)]
    pr = torch.softmax((q / math.sqrt(d)) @ k.transpose(-1, -2), dim=-1)
    ((pr @ v).transpose(1, 2).reshape(rows, E) * do.double()).sum().backward()
    e_qkv = rel(dq.float(), leaf.grad)
    da1_ref = bf((dq.double().cpu() @ Wqkv.double().t()).float())
    e_a1 = rel(da1.float(), da1_ref.float())
    dx1, dg1, db1 = ln_bwd64(h, g1, da1.double().cpu())
    dh_ref = dx1 + dh_mid_ref
    e_dh = rel(dhD.double().cpu() - dh_mid_ref, dx1)             # the LayerNorm-1 term alone, the residual taken out
    e_tot = rel(dhD, dh_ref)
    e_g = max(rel(p2[:, 0].sum(0), dg2), rel(p2[:, 1].sum(0), db2), rel(p1[:, 0].sum(0), dg1), rel(p1[:, 1].sum(0), db1))
    print(f"attn_block_bwd_ln rows={rows} H={H}: dh_mid {e_mid:.2e} dqkv {e_qkv:.2e} da1 {e_a1:.2e} LN1 term {e_dh:.2e} dh {e_tot:.2e} "
          f"dgamma/dbeta {e_g:.2e}; dh_out(bf16) {rel(dh_o.float(), dh_ref):.2e}")
    assert e_mid < 4e-3 and e_qkv < 1e-2 and e_a1 < 4e-3
    assert e_dh < 1e-4 and e_tot < 1e-5 and e_g < 1e-4
    assert rel(dh_o.float(), dh_ref) < 4e-3
    # ---- (ii) the three launches it replaces
    dh3 = Dv(dh).clone()
    mid3 = torch.zeros(rows, E, dtype=torch.bfloat16, device=dev)
    q3 = torch.zeros(rows, 3 * E, dtype=torch.bfloat16, device=dev)
    a3 = torch.zeros(rows, E, dtype=torch.bfloat16, device=dev)
    o3 = torch.zeros(rows, E, dtype=torch.bfloat16, device=dev)
    p23 = torch.zeros(rows // 32, 2, E, device=dev)
    part3 = torch.zeros(rows // 32 * 2 * E, device=dev)
    dg3, db3 = torch.zeros(E, device=dev), torch.zeros(E, device=dev)
    b0 = torch.zeros(E, device=dev)
    ck(L, L.smd_ln128_bwd_parts(P(hmD), P(pD), rows * E, rows, P(g2D), P(dh3), P(dh3), P(mid3), P(p23), st()))
    ck(L, L.smd_attn_block_bwd(P(mid3), P(qkvD), P(WoD), P(WqD), P(q3), P(a3), rows, H, st()))
    ck(L, L.smd_layernorm_bwd_ex(P(hD), None, rows, E, P(g1D), P(b0), P(a3), P(dh3), P(dh3), P(o3), P(dg3), P(db3), P(part3), part3.numel(), st()))
    torch.cuda.synchronize()
    # (the two translation units contract the LayerNorm arithmetic differently -- ln128.hip is built without packed-fp32 VALU code --:
    # last-bit differences in fp32, hence a bf16 step in about one element in 10^4 of dh_mid, which the attention backward passes on)
    flips = float((mid3 != dh_mid_o).float().mean())
    print(f"  vs the three launches: dh_mid elements that differ {flips:.1e}; dqkv rel {rel(dq.float(), q3.float()):.1e}; partials rel {rel(p2, p23):.1e}")
    assert flips < 1e-3 and rel(dh_mid_o.float(), mid3.float()) < 2e-4
    assert rel(dq.float(), q3.float()) < 2e-3 and rel(da1.float(), a3.float()) < 2e-3
    assert rel(p2, p23) < 1e-6
    assert rel(dhD, dh3) < 1e-3 and rel(p1[:, 0].sum(0), dg3) < 1e-3 and rel(p1[:, 1].sum(0), db3) < 1e-3      # (inherits those flips)
    assert float((dh_o.float() - o3.float()).abs().max()) <= float(o3.float().abs().max()) * 2 ** -7       # at most one bf16 step apart


@pytest.mark.parametrize("D,film,swish", [(128, False, False), (2048, False, False), (2048, True, True),
                                          (512, True, True), (1024, True, False)])
def test_layernorm_fwd_bwd(L, dev, D, film, swish):
    rows, rps = 128, 32
    g = torch.Generator().manual_seed(D + 3 * film)
    x = torch.randn(rows, D, generator=g) * 1.7 + 0.3
    gamma, beta = 1 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    ns = rows // rps
    ss = torch.cat([1 + 0.3 * torch.randn(ns, D, generator=g), 0.2 * torch.randn(ns, D, generator=g)], dim=1)
    dout = bf(torch.randn(rows, D, generator=g))

    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ssr = ss.double().requires_grad_(True)
    y = O.layer_norm(xr, {"n.scale": gr, "n.bias": br}, "n")
    if film:
        sc = ssr[:, :D].repeat_interleave(rps, 0)
        sh = ssr[:, D:].repeat_interleave(rps, 0)
        y = sc * y + sh
    if swish:
        y = O.swish(y)
    y.backward(dout.double())

    xd, gd, bd, ssd = x.to(dev), gamma.to(dev), beta.to(dev), ss.contiguous().to(dev)
    out = torch.zeros(rows, D, dtype=torch.bfloat16, device=dev)
    fs = ssd if film else None
    fsh = ssd[:, D:] if film else None
    ck(L, L.smd_layernorm_fwd(P(xd), rows, D, P(gd), P(bd), P(fs), P(fsh), 2 * D, rps, int(swish), P(out), st()))
    dx = torch.zeros(rows, D, device=dev)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dss = torch.zeros(ns, 2 * D, device=dev)
    partial = torch.zeros(rows * 2 * D, device=dev)
    doutd = dout.to(dev)
    ck(L, L.smd_layernorm_bwd(P(xd), rows, D, P(gd), P(bd), P(fs), P(fsh), 2 * D, rps, int(swish), P(doutd),
                              P(dx), P(dg), P(db), P(dss) if film else None, P(dss[:, D:]) if film else None,
                              P(partial), partial.numel(), st()))
    torch.cuda.synchronize()
    e_f = rel(out.float(), y)
    print(f"ln D={D} film={film} swish={swish}: fwd {e_f:.2e} dx {rel(dx, xr.grad):.2e} "
          f"dg {rel(dg, gr.grad):.2e} db {rel(db, br.grad):.2e}")
    assert e_f < 4e-3                       # bf16 output rounding
    assert rel(dx, xr.grad) < 1e-4
    assert rel(dg, gr.grad) < 1e-4
    assert rel(db, br.grad) < 1e-4
    if film:
        assert rel(dss, ssr.grad) < 1e-4


@pytest.mark.parametrize("wide", [2, 1])
@pytest.mark.parametrize("fs", [True, False])
@pytest.mark.parametrize("xbf,res,om", [(False, "bf16", 2), (True, None, 2), (False, None, 2), (False, "f32", 3),
                                        (False, None, 1), (True, "f32", 1), (True, "bf16", 3)])
def test_layernorm_bwd_film_forms(L, dev, xbf, res, om, fs, wide):
    """The engine's ResBlock forms of the LayerNorm backward (smd_layernorm_bwd_film): bf16 / fp32 input, residual
    gradient none / fp32 in place / bf16, fp32 and/or bf16 dx, accumulate into dscale/dshift; the 8-wave kernel
    (tuning ln_bwd_wide = 2) and the 4-wave one (1; takes a bf16 residual gradient through the 8-wave kernel)."""
    import smd_amd.lib as lib
    D, rows, rps = 2048, 160, 32                     # 5 row groups of 32 rows
    g = torch.Generator().manual_seed(11 + xbf + 2 * om)
    x = torch.randn(rows, D, generator=g) * 1.3 - 0.2
    if xbf:
        x = bf(x).float()
    gamma, beta = 1 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    ns = rows // rps
    ss = torch.cat([1 + 0.3 * torch.randn(ns, D, generator=g), 0.2 * torch.randn(ns, D, generator=g)], dim=1)
    dout = bf(torch.randn(rows, D, generator=g))
    dres = torch.randn(rows, D, generator=g)
    if res == "bf16":
        dres = bf(dres).float()
    dss0 = torch.randn(ns, 2 * D, generator=g)

    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ssr = ss.double().requires_grad_(True)
    y = O.layer_norm(xr, {"n.scale": gr, "n.bias": br}, "n")
    if fs:
        y = O.swish(ssr[:, :D].repeat_interleave(rps, 0) * y + ssr[:, D:].repeat_interleave(rps, 0))
    y.backward(dout.double())
    want_dx = xr.grad + (dres.double() if res else 0.0)

    lib.check(L.smd_set_tuning(b"ln_bwd_wide", wide))
    try:
        xd = (bf(x) if xbf else x).to(dev)
        gd, bd, ssd = gamma.to(dev), beta.to(dev), ss.contiguous().to(dev)
        dx32 = dres.to(dev).clone() if res == "f32" else torch.zeros(rows, D, device=dev)     # in place when fp32
        dresb = bf(dres).to(dev) if res == "bf16" else None
        dxb = torch.zeros(rows, D, dtype=torch.bfloat16, device=dev)
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        dss = dss0.to(dev).clone()
        partial = torch.zeros(rows * 2 * D, device=dev)
        doutd = dout.to(dev)
        ck(L, L.smd_layernorm_bwd_film(None if xbf else P(xd), P(xd) if xbf else None, rows, D, P(gd), P(bd),
                                       P(ssd) if fs else None, P(ssd[:, D:]) if fs else None, 2 * D, rps, int(fs),
                                       P(doutd), P(dx32) if res == "f32" else None, P(dresb),
                                       P(dx32) if om & 1 else None, P(dxb) if om & 2 else None, P(dg), P(db),
                                       P(dss) if fs else None, P(dss[:, D:]) if fs else None, 1,
                                       P(partial), partial.numel(), st()))
        torch.cuda.synchronize()
    finally:
        lib.check(L.smd_set_tuning(b"ln_bwd_wide", 2))
    if om & 1:
        assert rel(dx32, want_dx) < 1e-4
    if om & 2:
        assert rel(dxb.float(), want_dx) < 4e-3              # bf16 output rounding
    assert rel(dg, gr.grad) < 1e-4
    assert rel(db, br.grad) < 1e-4
    if fs:
        assert rel(dss, dss0.double() + ssr.grad) < 1e-4     # dfilm_accumulate = 1


@pytest.mark.parametrize("H", [8, 16, 4])
def test_attention_fwd_bwd(L, dev, H):
    B, S, E = 5, 32, 128
    d = E // H
    g = torch.Generator().manual_seed(H)
    qkv = bf(torch.randn(B * S, 3 * E, generator=g) * 1.5)
    dout = bf(torch.randn(B * S, E, generator=g))
    qr = qkv.double().requires_grad_(True)
    q, k, v = qr.view(B, S, 3 * E).split(E, dim=-1)
    q = q.reshape(B, S, H, d) / math.sqrt(d)
    k, v = k.reshape(B, S, H, d), v.reshape(B, S, H, d)
    w = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", q, k), dim=-1)
    o = torch.einsum("bhqk,bkhd->bqhd", w, v).reshape(B * S, E)
    o.backward(dout.double())
    out = torch.zeros(B * S, E, dtype=torch.bfloat16, device=dev)
    dq = torch.zeros(B * S, 3 * E, dtype=torch.bfloat16, device=dev)
    qd, dd = qkv.to(dev), dout.to(dev)
    ck(L, L.smd_attention_fwd(P(qd), P(out), B, S, E, H, st()))
    ck(L, L.smd_attention_bwd(P(qd), P(dd), P(dq), B, S, E, H, st()))
    torch.cuda.synchronize()
    print(f"attention H={H}: fwd {rel(out.float(), o):.2e} bwd {rel(dq.float(), qr.grad):.2e}")
    assert rel(out.float(), o) < 4e-3       # bf16 output rounding
    assert rel(dq.float(), qr.grad) < 4e-3


def test_attention_rejects_other_seq_len(L, dev):
    import smd_amd.lib as lib
    t = torch.zeros(16 * 384, dtype=torch.bfloat16, device=dev)
    with pytest.raises(ValueError):
        lib.check(L.smd_attention_fwd(P(t), P(t), 1, 16, 128, 8, st()))


def test_noise_embed_and_rng(L, dev):
    s = torch.tensor([1.0, 0.9999995, 0.5, 0.08137959, 1e-3])
    out = torch.zeros(5, 128, dtype=torch.bfloat16, device=dev)
    sd = s.to(dev)
    ck(L, L.smd_noise_embed(P(sd), 5, 128, P(out), 128, st()))
    ref = O.noise_encoding(s.float()[:, None], 128)      # fp32 like the reference
    torch.cuda.synchronize()
    err = (out.float().cpu() - ref).abs().max()
    print("noise_embed max abs err", float(err))
    assert err < 1.2e-2                                  # bf16 rounding (2^-8) + fp32 range reduction at 5000 rad
    # Philox normals: device vs the NumPy restatement
    Bn, per = 3, 64
    z = torch.zeros(Bn, per, device=dev)
    ck(L, L.smd_rng_normal(P(z), Bn, per, 1234, 77, 3, 10, st()))
    torch.cuda.synchronize()
    ctr = np.zeros((Bn, per // 4, 4), np.uint32)
    ctr[..., 0] = np.arange(per // 4)[None, :]
    ctr[..., 1] = (np.arange(Bn) + 10)[:, None]
    ctr[..., 2] = 3
    ref = O.philox_normal4(ctr, np.array([1234, 77], np.uint32)).reshape(Bn, per)
    err = np.abs(z.cpu().numpy() - ref).max()
    print("philox normal max abs err", err)
    assert err < 5e-5


@pytest.mark.parametrize("C", [512, 42])
def test_reverse_step_kernel(L, dev, C):
    import smd_amd.schedule as S
    B, Sq, T = 3, 32, 1000
    betas = S.create_noise_schedule(1e-6, 0.01, T, "linear")
    coef = torch.from_numpy(S.reverse_coefficient_table(betas))
    slot = torch.from_numpy(S.collection_slot_table(T))
    coefd, slotd = coef.to(dev), slot.to(dev)
    g = torch.Generator().manual_seed(C)
    for t in (999, 975, 1, 0):
        x = torch.randn(B, Sq, C, generator=g)
        eh = torch.randn(B, Sq, C, generator=g)
        z = torch.randn(B, Sq, C, generator=g)
        model = lambda s_, c_: eh.double()
        state, coll, met = O.diffusion_dynamics(model, betas, x.double(), lambda tt: z.double(), t_start=t, t_stop=t)
        xd = x.clone().to(dev)
        tp = torch.tensor([t], dtype=torch.int32, device=dev)
        mp = torch.zeros(T, B, 3, device=dev)
        cl = torch.zeros(41, B, Sq, C, device=dev)
        ehd, zd = eh.to(dev), z.to(dev)
        ck(L, L.smd_ddpm_reverse_step(P(xd), P(ehd), B, Sq, C, P(coefd), T, P(tp), P(zd), 0, 0, 0,
                                      P(mp), P(cl), P(slotd), st()))
        torch.cuda.synchronize()
        assert rel(xd, state) < 1e-5
        m = mp[t].sum(0).cpu().double() / (B * C)
        row = T - 1 - t
        assert abs(m[0] - met[0, row, 0]) / met[0, row, 0] < 1e-5
        assert abs(m[1] - met[1, row, 0]) / met[1, row, 0] < 1e-4
        assert abs(m[2] - met[3, row, 0]) / (met[3, row, 0] + 1e-12) < 1e-4
        s = int(slot[t])
        if s >= 0:
            assert rel(cl[s], coll[s]) < 1e-5
        else:
            assert float(cl.abs().max()) == 0.0
