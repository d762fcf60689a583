"""Why --dtype=fp8 stops at the DenseResBlock GEMMs (BASELINE config 5): a CPU simulation on the oracle.

The e4m3 operands of the engine (per-row power-of-two scales, round-to-nearest) are emulated inside the oracle's Dense
layers.  With the two Dense layers of every DenseResBlock quantised -- what the engine ships -- eps_hat moves by ~2e-2 from
the fp64 result (the GPU test measures 2.0e-2 on the real kernels, tests/test_gpu_fp8.py).  Quantising the encoder's MLP
products (models/ncsn.py:165-167) as well doubles that to ~4e-2, against the 5e-2 tolerance of SURVEY 8c; the phase stamps of
the fused MLP kernel (profiles/r3s_mlp_hs_phases.txt) put the possible gain at ~1.3 us of its 18 us.  Not worth the margin."""
import torch

import ddpm_oracle as O


def _q_rows(x):
    """OCP e4m3 with one power-of-two scale per row (the smallest e with amax * 2^-e <= 448), as quantize_rows_e4m3 does"""
    shp = x.shape
    x2 = x.reshape(-1, shp[-1]).float()
    amax = x2.abs().amax(1, keepdim=True).clamp_min(1e-30)
    s = torch.pow(2.0, torch.ceil(torch.log2(amax / 448.0)))
    q = (x2 / s).clamp(-448, 448).to(torch.float8_e4m3fn).float() * s
    return q.reshape(shp).to(x.dtype)


def test_e4m3_in_the_encoder_mlp_would_double_the_error(monkeypatch):
    mode = {"res": False, "enc": False}

    def dense(x, p, name):
        W, b = p[name + ".kernel"], p[name + ".bias"]
        res = name.startswith("res.") and name.endswith((".fc1", ".fc2"))
        enc = ".mlp.fc" in name
        if (mode["res"] and res) or (mode["enc"] and enc):
            return _q_rows(x) @ _q_rows(W.t()).t() + b
        return x @ W + b

    monkeypatch.setattr(O, "dense", dense)
    cfg = O.NetConfig(data_channels=512, num_layers=6, num_heads=8, num_mlp_layers=2)
    p = O.init_params(cfg, 0, torch.float64)
    g = torch.Generator().manual_seed(5)
    for k in p:
        if k.endswith(".bias"):
            p[k] = 0.1 * torch.randn(p[k].shape, generator=g, dtype=torch.float64)
        elif k.endswith(".scale"):
            p[k] = 1 + 0.1 * torch.randn(p[k].shape, generator=g, dtype=torch.float64)
    B = 4
    x = torch.clamp(0.25 * torch.randn(B, 32, 512, generator=g, dtype=torch.float64), -1, 1)
    s = (0.05 + 0.95 * torch.rand(B, generator=g, dtype=torch.float64)).view(B, 1, 1)
    model = O.make_model(p, cfg)
    with torch.no_grad():
        ref = model(x, s)
        mode.update(res=True, enc=False)
        e_res = float((model(x, s) - ref).norm() / ref.norm())
        mode.update(res=True, enc=True)
        e_both = float((model(x, s) - ref).norm() / ref.norm())
    print(f"eps_hat rel-L2 vs fp64: DenseResBlock GEMMs on e4m3 {e_res:.3e}; + encoder MLP products on e4m3 {e_both:.3e}")
    assert e_res < 3e-2                       # what ships: comfortably inside the 5e-2 of SURVEY 8c
    assert e_both > 1.6 * e_res               # the encoder products would cost as much again
    assert e_both > 3e-2
