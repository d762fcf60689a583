"""Parity holes VERDICT r4 named that need no JAX:

(a) a FULL-LENGTH free-running reverse walk (utils/ebm_utils.py:399-401: 1000 iterations of sample_with_beta, :327-394) with
    explicit per-step noise against the fp64 oracle: final state, all 39 written collection slots, the never-written slot 1, and
    the (4, 1000, 1) metrics -- the existing tests compare <= 50-step rollouts only;
(b) parity on TRAINED weights: a few hundred engine train steps move the LayerNorm / FiLM statistics away from the
    random-init ones every other parity test uses (outlier features, grown FiLM scales: what bf16 trunk storage and the e4m3
    row scales are sensitive to); the trained parameters are exported through named_views and forward / loss / gradient /
    three reverse steps are compared again at B = 256, in bf16 and fp8 (utils/losses.py:250-308, train_ncsn.py:260-288).

Tolerances: SURVEY section 8(c).  A free-running walk has no per-step bound there (the bf16 eps_hat error of each step is fed
back into the state); the bound asserted here is the measured divergence with head-room, printed per snapshot.
"""
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


def make(C, L, H, K, seed=0, dtype="bf16"):
    import smd_amd.ncsn as N
    from smd_amd.engine import NetConfig
    ocfg = O.NetConfig(data_channels=C, num_layers=L, num_heads=H, num_mlp_layers=K)
    p = O.init_params(ocfg, seed, torch.float64)
    g = torch.Generator().manual_seed(seed + 1)
    for k in p:
        if k.endswith(".bias"):
            p[k] = 0.1 * torch.randn(p[k].shape, generator=g, dtype=torch.float64)
        elif k.endswith(".scale"):
            p[k] = 1 + 0.1 * torch.randn(p[k].shape, generator=g, dtype=torch.float64)
    cfg = NetConfig(architecture="TransformerDDPM", data_channels=C, seq_len=32, num_layers=L, num_heads=H,
                    num_mlp_layers=K, num_timesteps=1000, dtype=dtype)
    model = N.Model(cfg, "cuda:0", seed=None)
    model.engine.load_named(p)
    return ocfg, p, model


def step_noise(B, C, t):
    g = torch.Generator().manual_seed(100_000 + t)
    return torch.randn(B, 32, C, generator=g)


@pytest.mark.parametrize("name,C,L,H,K,B,odt,tol_state,tol_metric", [
    ("small", 42, 2, 8, 1, 4, torch.float64, 1.5e-2, 2e-3),      # measured (profiles/r5b_full_walk_tests.txt): 5.6e-3 / 2.1e-4
    ("base", 512, 6, 8, 2, 2, torch.float64, 1.5e-2, 2e-3),      # measured: 6.6e-3 / 1.2e-4
])
def test_full_T_walk_against_the_fp64_oracle(name, C, L, H, K, B, odt, tol_state, tol_metric):
    import smd_amd.ncsn as N
    ocfg, p, model = make(C, L, H, K)
    g = torch.Generator().manual_seed(2718)
    init = torch.randn(B, 32, C, generator=g)
    import time
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(8, nthreads))        # 64-row matrices: on a 64-thread host the fork/join of every small op dominates
    t0 = time.perf_counter()
    try:
        with torch.no_grad():
            po = {k: v.to(odt) for k, v in p.items()}
            ref_x, ref_c, ref_m = O.diffusion_dynamics(O.make_model(po, ocfg), BETAS, init.to(odt), lambda t: step_noise(B, C, t).to(odt))
    finally:
        torch.set_num_threads(nthreads)
    print(f"[{name}] oracle walk: {time.perf_counter() - t0:.1f} s of host time")
    x, coll, met = N.diffusion_dynamics(N.PRNGKey(0), model, BETAS, init, noises=lambda t: step_noise(B, C, t))
    torch.cuda.synchronize()
    assert tuple(coll.shape) == (41, B, 32, C) and tuple(met.shape) == (4, 1000, 1)
    # bookkeeping (utils/ebm_utils.py:322-325,387-394): slot 0 = start, slot 1 never written, 39 snapshots, final state absent
    assert torch.equal(coll[0].cpu(), init)
    assert float(coll[1].abs().max()) == 0.0 and float(ref_c[1].abs().max()) == 0.0
    table = O.collection_index_table(1000)
    slot_t = {O.collection_slot_for_t(1000, t, table): t for t in range(1000) if O.collection_slot_for_t(1000, t, table) >= 0}
    assert sorted(slot_t) == list(range(2, 41)) and slot_t[40] == 1 and slot_t[2] == 975
    per_slot = []
    for k in range(2, 41):
        assert float(coll[k].abs().max()) > 0
        per_slot.append(rel(coll[k], ref_c[k]))
    e_final = rel(x, ref_x)
    print(f"[{name}] free-running T = 1000 walk, B = {B}: final state rel {e_final:.3e}; snapshot rel by t: "
          + " ".join(f"{slot_t[k]}:{per_slot[k - 2]:.1e}" for k in (2, 8, 14, 20, 27, 33, 39, 40)) + f"; worst {max(per_slot):.3e}")
    assert e_final < tol_state and max(per_slot) < tol_state
    assert float(x.abs().max()) <= 1.0 + 1e-6                      # the t = 0 step returns the clipped x0
    # metrics rows (grad_norm, step_norm, alpha_prod, noise_norm) for every one of the T iterations
    m, r = met.cpu().double(), ref_m.double()
    e_rows = [rel(m[i], r[i]) for i in range(4)]
    worst_t = float(((m[0, :, 0] - r[0, :, 0]).abs() / r[0, :, 0].abs()).max())
    print(f"[{name}] metrics rel: slope {e_rows[0]:.2e} step {e_rows[1]:.2e} alpha {e_rows[2]:.2e} noise {e_rows[3]:.2e}; "
          f"worst single-t slope error {worst_t:.2e}")
    assert e_rows[2] < 1e-6 and e_rows[3] < 1e-5                   # table data / norms of the explicit draws: fp32 exact-ish
    assert e_rows[0] < tol_metric and e_rows[1] < tol_metric
    assert float(r[3, -1, 0]) == pytest.approx(1e-5, rel=1e-2) and float(m[3, -1, 0]) == pytest.approx(1e-5, rel=1e-2)   # t = 0: z = 0


def test_full_T_walk_graph_replay_equals_eager_bitwise():
    """The captured-graph walk IS the eager walk for all 1000 iterations (one chain; Philox draws keyed by (rng, sample, t))."""
    import smd_amd.ncsn as N
    _, _, model = make(42, 2, 8, 1)
    init = torch.randn(4, 32, 42, generator=torch.Generator().manual_seed(5))
    a = N.diffusion_dynamics(N.PRNGKey(9), model, BETAS, init, use_graph=True)
    b = N.diffusion_dynamics(N.PRNGKey(9), model, BETAS, init, use_graph=False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert float(a[1][1].abs().max()) == 0 and all(float(a[1][k].abs().max()) > 0 for k in range(2, 41))


# ------------------------------------------------------------------ (b) trained weights
def _train(model, steps, B=256, C=512, lr=1e-3, at=None):
    """`steps` engine train steps (configs/ddpm-base.cfg: Adam, lr 1e-3, grad_clip 1) on structured synthetic latents -- a
    low-rank pattern + noise drawn on the device, so the net has something to fit and its FiLM / LayerNorm statistics move --
    with the engine's own Philox label / eps draws."""
    import smd_amd.ncsn as N
    from smd_amd.trainer import create_optimizer, train_step
    g = torch.Generator(device="cuda").manual_seed(99)
    basis = torch.randn(8, 32, C, generator=g, device="cuda")

    def batch():
        coef = torch.randn(B, 8, generator=g, device="cuda")
        return torch.clamp(0.35 * torch.einsum("bk,ksc->bsc", coef, basis) / 8 ** 0.5
                           + 0.05 * torch.randn(B, 32, C, generator=g, device="cuda"), -1, 1)

    opt = create_optimizer(model, lr, ema=False)
    losses = []
    for it in range(steps):
        if at and it in at:
            at[it](it, opt, x0.cpu())       # a look at the state after `it` steps (x0: the batch of the step before)
        x0 = batch()
        _, m = train_step(N.diffusion_loss, x0, opt, BETAS, N.PRNGKey(it), lr, grad_clip=1.0)
        if it % 100 == 0 or it == steps - 1:
            losses.append(float(m["loss"]))
    torch.cuda.synchronize()
    return opt, losses, x0.cpu()


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_parity_on_trained_weights(dtype):
    import smd_amd.lib as lib
    import smd_amd.ncsn as N
    B, C = 256, 512
    ocfg, p0, model = make(C, 6, 8, 2, dtype=dtype)

    def plateau(it, opt, x0):
        """The predict-zero plateau (VERDICT r5 weak #1c).  On the way down from loss 2.0 the net passes through "output (almost)
        nothing": loss ~1.0, |eps_hat| a fraction of |eps| (where a run lingers there depends on the box: round 5's first run of this
        test sat at 1.003 for 300 steps, later runs are through it by step ~50).  The RELATIVE eps_hat error in that state is the bf16
        noise of the trunk -- which does not shrink with the output -- divided by a vanishing output: 4.5e-2 in that run, above
        SURVEY 8c's 1e-2; the quantity the loss and the sampler consume is eps_hat itself next to eps / x: asserted here as an
        ABSOLUTE bound, error / |eps| <= 1e-2, at several points of the descent and at step 300, with the relative figure and
        |eps_hat| / |eps| printed beside it."""
        eng = opt.engine
        q = {k: v.detach().float().cpu().clone() for k, v in model.engine.named_views().items()}
        gp = torch.Generator().manual_seed(300)
        labels, eps = torch.randint(1, 1001, (B,), generator=gp), torch.randn(B, 32, C, generator=gp)
        seen = {}
        base = O.make_model(q, ocfg)

        def capturing(x, cond):
            seen["pred"] = base(x, cond)
            return seen["pred"]
        with torch.no_grad():
            l_ref = float(O.diffusion_loss(x0, capturing, BETAS, labels.numpy(), eps, "mean"))
        eng.loss_backward(x0.cuda(), labels.int().cuda(), eps.cuda(), stage=3)
        torch.cuda.synchronize()
        pred = eng.last_pred().double().cpu()
        err = float((pred - seen["pred"].double()).norm())
        n_pred, n_eps = float(seen["pred"].double().norm()), float(eps.double().norm())
        l_eng = float(eng.loss_per_sample().mean())
        print(f"[plateau {dtype}] after {it} steps: loss {l_eng:.5f} (oracle {l_ref:.5f}); |eps_hat| / |eps| = {n_pred / n_eps:.3f}; eps_hat error: "
              f"relative {err / n_pred:.3e}, ABSOLUTE (error / |eps|) {err / n_eps:.3e}")
        assert err / n_eps < 1e-2
        assert abs(l_eng - l_ref) / l_ref < 5e-3

    opt, losses, x_last = _train(model, 1500, at={10: plateau, 20: plateau, 40: plateau, 300: plateau})
    print(f"[trained {dtype}] loss every 100 of the 1500 steps: " + " ".join(f"{v:.3f}" for v in losses))
    assert losses[-1] < 0.9, "training did not get below the predict-zero plateau (loss 1.0)"
    eng = opt.engine
    p = {k: v.detach().float().cpu().clone() for k, v in model.engine.named_views().items()}     # fp32 oracle at B = 256
    drift = {k: float((p[k].double() - p0[k]).norm() / (p0[k].norm() + 1e-12)) for k in p}
    film = [k for k in p if k.startswith("film") and k.endswith("scale.kernel")]
    print(f"[trained {dtype}] parameter drift: median {np.median(list(drift.values())):.3f}, FiLM scale kernels "
          + " ".join(f"{drift[k]:.3f}" for k in film))
    g = torch.Generator().manual_seed(31)
    tol_fwd = 1e-2 if dtype == "bf16" else 5e-2
    # ---- forward + loss + gradient on the trained parameters, for two batches:
    #   "fresh"      latents the net was NOT trained on (the bench's clip(0.25 N(0,1))): a gradient of ordinary size
    #   "stationary" the training distribution itself: after 1500 steps the mean gradient is what is left of per-sample
    #                gradients that cancel (|g| 0.12 against 0.9 at the start), so the per-sample bf16 error (5e-3 of each
    #                sample's gradient, uncorrelated) is ~7x larger RELATIVE TO THE SUM: measured 4.1e-2, the same factor on every
    #                tensor and under every engine option (profiles/r5d_trained_grad_diag.txt) -- conditioning of the quantity,
    #                not an error of a kernel.  Asserted there: direction (cosine) and a bound with that amplification.
    for tag, x0 in (("fresh", torch.clamp(0.25 * torch.randn(B, 32, C, generator=g), -1, 1)), ("stationary", x_last)):
        labels = torch.randint(1, 1001, (B,), generator=g)
        eps = torch.randn(B, 32, C, generator=g)
        leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        base_model = O.make_model(leaf, ocfg)
        seen = {}

        def capturing(x, cond):
            out = base_model(x, cond)
            seen["pred"] = out.detach()
            return out

        loss_ref = O.diffusion_loss(x0, capturing, BETAS, labels.numpy(), eps, "none")
        loss_ref.mean().backward()
        eng.loss_backward(x0.cuda(), labels.int().cuda(), eps.cuda(), stage=0)
        torch.cuda.synchronize()
        e_pred = rel(eng.last_pred(), seen["pred"])
        m_eng, m_ref = float(eng.loss_per_sample().mean()), float(loss_ref.mean())
        gv = eng.named_views(eng.grads)
        num = sum(float((gv[k].double().cpu() - leaf[k].grad.double()).pow(2).sum()) for k in leaf)
        den = sum(float(leaf[k].grad.double().pow(2).sum()) for k in leaf)
        dot = sum(float((gv[k].double().cpu() * leaf[k].grad.double()).sum()) for k in leaf)
        gg = sum(float(gv[k].double().pow(2).sum()) for k in leaf)
        e_grad, cos = (num / den) ** 0.5, dot / (gg * den) ** 0.5
        yK = eng.debug_tensor("y", ocfg.num_mlp_layers).float()
        print(f"[trained {dtype} / {tag}] trunk before the output LayerNorm: rms {float(yK.pow(2).mean().sqrt()):.3f}, max |x| / rms "
              f"{float(yK.abs().max() / yK.pow(2).mean().sqrt()):.1f}; |eps_hat| rms {float(seen['pred'].pow(2).mean().sqrt()):.3f}")
        print(f"[trained {dtype} / {tag}] eps_hat rel {e_pred:.3e}; loss {m_eng:.6f} vs {m_ref:.6f} ({abs(m_eng - m_ref) / m_ref:.2e}); "
              f"|g| {den ** 0.5:.4f}; gradient whole-vector rel {e_grad:.3e}, cosine {cos:.6f}")
        assert e_pred < tol_fwd
        assert abs(m_eng - m_ref) / m_ref < (5e-3 if dtype == "bf16" else 2.5e-2)
        # bounds = measured x 1.5 (profiles/r5f_trained_tests.txt: bf16 3.9e-2 ... 4.1e-2, fp8 6.3e-2 ... 6.6e-2); what they measure is
        # attributed below (bf16: bf16_emulation.py; fp8: e4m3_emulation.py) and shown harmless by tests/test_gpu_trajectory.py
        assert cos > (0.998 if dtype == "bf16" else 0.996) and e_grad < (6.2e-2 if dtype == "bf16" else 1.0e-1)
    # ---- the sharp gradient check on trained weights: against the oracle WITH THE ENGINE'S ROUNDING POINTS (oracle/bf16_emulation.py:
    # bf16 operand pack, bf16 activations where the engine stores bf16, gradient hooks), B = 8, fp64.  The 4e-2 above is the gradient
    # of a slightly different function -- the loss at the bf16-ROUNDED weights -- divided by a mean gradient that training has
    # driven towards zero; the emulating oracle evaluates that same function, so what is left is the kernels' own error.
    # fp8: the same with oracle/e4m3_emulation.py (e4m3 operands + E8M0 row scales of the DenseResBlock forward and dgrad GEMMs on top).
    if True:
        import bf16_emulation as E
        import e4m3_emulation as F8
        import smd_amd.lib as _lib
        EMU = E if dtype == "bf16" else F8
        B8 = 8
        x8, lab8, eps8 = x_last[:B8], torch.randint(1, 1001, (B8,), generator=g), torch.randn(B8, 32, C, generator=g)
        p64 = {k: v.double() for k, v in p.items()}
        eng.bind(B8, training=True)
        eng.loss_backward(x8.cuda(), lab8.int().cuda(), eps8.cuda(), stage=0)
        torch.cuda.synchronize()
        gv8 = {k: v.double().cpu().clone() for k, v in eng.named_views(eng.grads).items()}
        lv = torch.from_numpy(O.used_alphas_from_labels(BETAS, lab8.numpy())).sqrt()
        sd = lv.float().reshape(-1).cuda().contiguous()
        emb = torch.zeros(B8, 128, dtype=torch.bfloat16, device="cuda")
        _lib.check(_lib.get_lib().smd_noise_embed(sd.data_ptr(), B8, 128, emb.data_ptr(), 128, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()

        def oracle8(mk):
            leaf = {k: v.clone().requires_grad_(True) for k, v in p64.items()}
            O.diffusion_loss(x8.double(), mk(leaf), BETAS, lab8.numpy(), eps8.double(), "none").mean().backward()
            return {k: v.grad for k, v in leaf.items()}

        def dist(ref):
            num = sum(float((gv8[k] - ref[k]).pow(2).sum()) for k in ref)
            den = sum(float(ref[k].pow(2).sum()) for k in ref)
            return (num / den) ** 0.5, den ** 0.5

        (e_exact, gnorm), (e_emu, _) = dist(oracle8(lambda q: O.make_model(q, ocfg))), dist(oracle8(
            lambda q: EMU.make_model(q, ocfg, backward=True, noise_embedding=emb.double().cpu())))
        print(f"[trained {dtype}] B = 8 gradient (|g| {gnorm:.4f}): vs the exact fp64 oracle {e_exact:.3e}; vs the oracle with the engine's "
              f"rounding points {e_emu:.3e}")
        assert e_emu < 1e-2 and e_emu < 0.5 * e_exact
        eng.bind(B, training=True)
    # ---- three reverse steps with explicit draws (utils/ebm_utils.py:327-394) on the trained parameters
    init = torch.randn(B, 32, C, generator=g)
    zs = {t: torch.randn(B, 32, C, generator=g) for t in (999, 998, 997)}
    with torch.no_grad():
        ref, _, mref = O.diffusion_dynamics(O.make_model(p, ocfg), BETAS, init, lambda t: zs[t], t_stop=997)
    got, _, mgot = N.diffusion_dynamics(N.PRNGKey(0), model, BETAS, init, noises=lambda t: zs[t], t_stop=997)
    e_state = rel(got, ref)
    e_slope = rel(mgot[0, :3], mref[0, :3])
    print(f"[trained {dtype}] 3 reverse steps: state rel {e_state:.3e}; slope metric rel {e_slope:.3e}")
    assert e_state < 1e-2
    assert e_slope < tol_fwd
    # ---- and a late, low-noise step where the x0 prediction matters (t = 20 .. 18), teacher-forced from a mid-walk state
    xs = torch.clamp(x_last + 0.05 * torch.randn(B, 32, C, generator=g), -1.5, 1.5)
    z2 = {t: torch.randn(B, 32, C, generator=g) for t in (20, 19, 18)}
    with torch.no_grad():
        ref2, _, _ = O.diffusion_dynamics(O.make_model(p, ocfg), BETAS, xs, lambda t: z2[t], t_start=20, t_stop=18)
    got2, _, _ = N.diffusion_dynamics(N.PRNGKey(0), model, BETAS, xs, noises=lambda t: z2[t], t_start=20, t_stop=18)
    e2 = rel(got2, ref2)
    print(f"[trained {dtype}] reverse steps t = 20..18: state rel {e2:.3e}")
    assert e2 < 1e-2


def philox_step_noise(idx, t, seed, per):
    """z_t of the GLOBAL samples `idx` as the fused reverse step draws it (csrc/rng.h: counter = (element / 4, global sample,
    SMD_STREAM_Z = 2, t), key = (seed_lo, seed_hi)), restated by the oracle's Philox."""
    n4 = per // 4
    ctr = np.zeros((len(idx), n4, 4), np.uint32)
    ctr[..., 0] = np.arange(n4, dtype=np.uint32)[None, :]
    ctr[..., 1] = np.asarray(idx, np.uint32)[:, None]
    ctr[..., 2] = 2
    ctr[..., 3] = t
    z = O.philox_normal4(ctr, np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32))
    return torch.from_numpy(z.reshape(len(idx), per))


@pytest.mark.parametrize("B,sizes,pad", [(1000, [504, 496], 0), (999, [504, 496], 1), (135, [72, 64], 1)])
def test_sampler_at_the_reference_default_batch_and_at_ragged_batches(B, sizes, pad, monkeypatch):
    """VERDICT r5 missing #3 / weak #7: the reference samples ``sample_size = 1000`` sequences as ONE batch
    (sample_ncsn.py:54, train_ncsn.py:536-548).  ncsn.diffusion_dynamics must (i) take the pipelined two-chain walk for it (504 + 496)
    and for batches that are not a multiple of 8 (padding sliced off), and say which arrangement it chose; (ii) give the fp64 oracle's
    result on the same Philox draws, sample by sample (first / last rows of each chain and the rows either side of the split);
    (iii) agree with the one-chain walk of the same batch."""
    import smd_amd.ncsn as N
    C, steps = 512, 12
    ocfg, p, model = make(C, 2, 8, 1)
    seed = 33
    key = N.PRNGKey(seed)
    init = torch.randn(B, 32, C, generator=torch.Generator().manual_seed(B))
    t_stop = 1000 - steps
    monkeypatch.delenv("SMD_SAMPLER_CHAINS", raising=False)
    x2, c2, m2 = N.diffusion_dynamics(key, model, BETAS, init, t_stop=t_stop)
    arr = model.sampler_arrangement
    assert arr["chains"] == 2 and arr["chain_sizes"] == sizes and arr["padded"] == pad and arr["pipelined_unroll"] == 8 and arr["graphed"]
    assert tuple(x2.shape) == (B, 32, C) and tuple(c2.shape) == (41, B, 32, C) and tuple(m2.shape) == (4, 1000, 1)
    assert bool(torch.isfinite(x2).all())
    # (iii) one chain, same draws
    monkeypatch.setenv("SMD_SAMPLER_CHAINS", "1")
    x1, c1, m1 = N.diffusion_dynamics(key, model, BETAS, init, t_stop=t_stop)
    assert model.sampler_arrangement["chains"] == 1 and model.sampler_arrangement["chain_sizes"] == [B]
    monkeypatch.delenv("SMD_SAMPLER_CHAINS")
    assert rel(x2, x1) < 5e-3, rel(x2, x1)
    assert torch.equal(c2[0], c1[0]) and float(c2[1].abs().max()) == 0
    rows = slice(0, steps)
    for r in (0, 1, 3):
        assert rel(m2[r, rows, 0], m1[r, rows, 0]) < 2e-3, (r, rel(m2[r, rows, 0], m1[r, rows, 0]))
    assert torch.equal(m2[2], m1[2])
    # (ii) the fp64 oracle on the rows at the ends of both chains and either side of the split, the same Philox noise
    h0 = sizes[0]
    idx = sorted({0, 1, h0 - 1, h0, h0 + 1, B - 2, B - 1})
    pd = {k: v.double() for k, v in p.items()}
    om = O.make_model(pd, ocfg)
    with torch.no_grad():
        xo, _co, _mo = O.diffusion_dynamics(om, BETAS, init[idx].double(),
                                            lambda t: philox_step_noise(idx, t, seed, 32 * C).view(len(idx), 32, C), t_stop=t_stop)
    errs = [rel(x2[i], xo[j]) for j, i in enumerate(idx)]
    print(f"B={B} as {sizes} (+{pad}): state vs fp64 oracle after {steps} steps, rows {idx}: max {max(errs):.2e}")
    assert max(errs) < 1e-2, errs
    # a second call goes through the cached graphs and is bitwise the first
    xa, ca, ma = N.diffusion_dynamics(key, model, BETAS, init, t_stop=t_stop)
    assert torch.equal(xa, x2) and torch.equal(ca, c2) and torch.equal(ma, m2)
