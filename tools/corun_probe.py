"""Can a MAIN-stream kernel run beside the side stream's 2048 x 2048 weight-gradient GEMM (gemm_tn256: 256 workgroups x 128 KiB of
LDS, 8 waves, one per CU for its whole duration)?  In the two-stream train step the FiLM LayerNorm backward that follows the
four-problem weight-gradient launch takes 150-280 us instead of 36-43 (profiles/r6c_train_main_stream.txt, r6g_train_stream_busy.txt):
it ENDS ~35 us after the GEMM ends, i.e. it does not run beside it at all.  This probe reproduces the pair outside the engine:

  side stream (low priority): N back-to-back smd_gemm_bf16_tn launches (8192 x 2048 x 2048, the 256x256 kernel)
  main stream: a spacer GEMM (so that the side kernel owns the CUs first), then the kernel under test, timed with events

for victims of different LDS footprints: the D = 2048 LayerNorm backward (32 KiB), the LayerNorm forward (32 KiB), mse_loss_grad
(64 B), q_sample (0), and the LayerNorm backward at D = 1024 (16 KiB).  python tools/corun_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import smd_amd.lib as lib
import smd_amd.schedule as S
L = lib.get_lib()
dev = "cuda:0"
P = lambda t: None if t is None else t.data_ptr()
R = 8192
g0 = torch.Generator().manual_seed(0)
side = torch.cuda.Stream(device=dev, priority=0)          # torch knows two levels: the engine's side stream is the LOWER one
main = torch.cuda.Stream(device=dev, priority=-1)

# ---- the side stream's load: dW = X^T dY, 8192 x 2048 x 2048
RS = int(os.environ.get("CORUN_SIDE_ROWS", 8192))        # 32768: one launch = 256 workgroups x the whole 8192-row contraction (~300 us each, the
X = torch.randn(RS, 2048, generator=g0).to(torch.bfloat16).to(dev)    # shape of the engine's four-problem launch); 8192: four m-splits, ~100 us per launch
dY = torch.randn(RS, 2048, generator=g0).to(torch.bfloat16).to(dev)
NSIDE = int(os.environ.get("CORUN_SIDE_LAUNCHES", 5))
dW = torch.zeros(2048, 2048, device=dev)
dbias = torch.zeros(2048, device=dev)
zero_page = torch.zeros(128, dtype=torch.bfloat16, device=dev)
slab_n = int(L.smd_gemm_tn_slab_elems())
slab = torch.zeros(slab_n, device=dev)


def wgrad(stream):
    lib.check(L.smd_gemm_bf16_tn(P(X), 2048, P(dY), 2048, RS, 2048, 2048, P(dW), 2048, P(dbias), P(zero_page), P(slab), slab_n, None, 0, 1,
                                 stream.cuda_stream))


# ---- spacer on the main stream: one ResBlock dgrad (8192 x 2048 x 2048, 256x256 kernel)
Wt = (torch.randn(2048, 2048, generator=g0) * 0.02).to(torch.bfloat16).to(dev)
bias = torch.zeros(2048, device=dev)
sp_out = torch.empty(R, 2048, dtype=torch.bfloat16, device=dev)


def spacer(stream):
    lib.check(L.smd_gemm_bf16_nt(P(X), 2048, P(Wt), 2048, R, 2048, 2048, P(bias), 0, None, 0, None, 0, P(sp_out), 2048, stream.cuda_stream))


# ---- victims
def make_ln_bwd(D):
    x = torch.randn(R, D, generator=g0).to(torch.bfloat16).to(dev)
    g, b = (1 + 0.1 * torch.randn(D, generator=g0)).to(dev), (0.1 * torch.randn(D, generator=g0)).to(dev)
    ss = torch.randn(R // 32, 2 * D, generator=g0).to(dev)
    dout = torch.randn(R, D, generator=g0).to(torch.bfloat16).to(dev)
    dres = torch.randn(R, D, generator=g0).to(torch.bfloat16).to(dev) if D == 2048 else None
    dxb = torch.empty(R, D, dtype=torch.bfloat16, device=dev)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dss = torch.zeros(R // 32, 2 * D, device=dev)
    part = torch.empty(R * 2 * D // 16, device=dev)
    keep = (x, g, b, ss, dout, dres, dxb, dg, db, dss, part)

    def call(stream):
        lib.check(L.smd_layernorm_bwd_film(None, P(x), R, D, P(g), P(b), P(ss), ss[:, D:].data_ptr(), 2 * D, 32, 1, P(dout), None, P(dres), None,
                                           P(dxb), P(dg), P(db), P(dss), dss[:, D:].data_ptr(), 0, P(part), part.numel(), stream.cuda_stream))
    call.keep = keep
    return call


def make_ln_fwd():
    D = 2048
    x = torch.randn(R, D, generator=g0).to(dev)
    g, b = (1 + 0.1 * torch.randn(D, generator=g0)).to(dev), (0.1 * torch.randn(D, generator=g0)).to(dev)
    ss = torch.randn(R // 32, 2 * D, generator=g0).to(dev)
    out = torch.empty(R, D, dtype=torch.bfloat16, device=dev)

    def call(stream):
        lib.check(L.smd_layernorm_fwd(P(x), R, D, P(g), P(b), P(ss), ss[:, D:].data_ptr(), 2 * D, 32, 1, P(out), stream.cuda_stream))
    call.keep = (x, g, b, ss, out)
    return call


def make_mse():
    pred = torch.randn(256, 32, 512, device=dev)
    eps = torch.randn(256, 32, 512, device=dev)
    loss = torch.zeros(256, device=dev)
    dp = torch.zeros(R, 512, dtype=torch.bfloat16, device=dev)

    def call(stream):
        lib.check(L.smd_mse_fwd_bwd(P(pred), P(eps), 256, 32, 512, 512, 1.0 / (256 * 32 * 512), P(loss), P(dp), stream.cuda_stream))
    call.keep = (pred, eps, loss, dp)
    return call


def make_qs():
    betas = S.create_noise_schedule(1e-6, 0.01, 1000, "linear")
    ape = torch.from_numpy(np.concatenate([np.ones(1, np.float32), S.alphas_cumprod(betas)])).to(dev)
    x0 = torch.randn(256, 32, 512, device=dev)
    xt = torch.empty(R, 512, dtype=torch.bfloat16, device=dev)
    eo = torch.empty(256, 32, 512, device=dev)
    s = torch.empty(256, device=dev)

    def call(stream):
        lib.check(L.smd_q_sample(P(x0), 256, 32, 512, 512, 1000, P(ape), None, 1, None, None, 7, 0, None, 0, P(xt), P(eo), P(s), stream.cuda_stream))
    call.keep = (ape, x0, xt, eo, s)
    return call


victims = [("layernorm_bwd D=2048 bf16 (LDS 32 KiB)", make_ln_bwd(2048)), ("layernorm_bwd D=1024 bf16 (LDS 16 KiB)", make_ln_bwd(1024)),
           ("layernorm_fwd D=2048 (LDS 32 KiB)", make_ln_fwd()), ("mse_loss_grad (LDS 64 B)", make_mse()), ("q_sample (no LDS)", make_qs())]


def timed(victim, with_side, n_side=NSIDE):
    torch.cuda.synchronize()
    e_go = torch.cuda.Event()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    es0, es1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e_go.record(torch.cuda.current_stream())
    if with_side:
        side.wait_event(e_go)
        es0.record(side)
        for _ in range(n_side):
            wgrad(side)
        es1.record(side)
    main.wait_event(e_go)
    spacer(main)
    e0.record(main)
    victim(main)
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3, (es0.elapsed_time(es1) * 1e3 if with_side else 0.0)


for name, v in victims:
    for _ in range(3):
        timed(v, True)
    solo = min(timed(v, False)[0] for _ in range(5))
    both = [timed(v, True) for _ in range(5)]
    vb = sorted(b[0] for b in both)[2]
    sb = sorted(b[1] for b in both)[2]
    print(f"{name:42s}: alone {solo:7.1f} us; beside {NSIDE} weight-gradient GEMMs ({RS} rows) on the side stream {vb:7.1f} us (the GEMMs: {sb:7.1f} us)")
