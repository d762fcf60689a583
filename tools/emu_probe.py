"""Probe for tests/test_gpu_engine.py::*emulating*: inference-path and training-path eps_hat against oracle/bf16_emulation.py."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import ddpm_oracle as O, bf16_emulation as E
import test_gpu_engine as T

for arch, C, L, H, K in (("TransformerDDPM", 512, 6, 8, 2), ("TransformerDDPM", 42, 1, 8, 1), ("TransformerDDPM", 42, 0, 8, 1)):
    ocfg, p, model = T.make(arch, C, L, H, K)
    B = 8
    x0, g = T.data(B, (32, C))
    labels = torch.randint(1, 1001, (B,), generator=g)
    eps = torch.randn(B, 32, C, generator=g)
    a = torch.from_numpy(O.used_alphas_from_labels(T.BETAS, labels.numpy()))
    lv = a.sqrt()
    emb = T.device_noise_embedding(lv)
    av = a.double().view(B, 1, 1)
    xt = torch.sqrt(av) * x0.double() + torch.sqrt(1 - av) * eps.double()
    emu = E.make_model(p, ocfg, noise_embedding=emb)(xt, lv.double().view(B, 1, 1))
    exact = O.make_model(p, ocfg)(xt, lv.double().view(B, 1, 1))
    inf = model(xt.float(), lv.view(B, 1, 1))
    eng = model.train_engine(ema=False)
    eng.set_schedule(T.BETAS, with_sampler=False)
    eng.bind(B, training=True)
    eng.loss_backward(x0.cuda(), labels.int().cuda(), eps.cuda(), stage=0)
    torch.cuda.synchronize()
    tr = eng.last_pred().double().cpu()
    r = T.rel
    print(f"emu_probe L={L} C={C}: inference vs emu {r(inf, emu):.3e} vs exact {r(inf, exact):.3e} | training-path pred vs emu {r(tr, emu):.3e} "
          f"vs exact {r(tr, exact):.3e} | inference vs training path {r(inf, tr):.3e} | emu vs exact {r(emu, exact):.3e}")
