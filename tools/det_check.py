"""Which gradient tensors differ between repeated identical loss_backward calls (race / nondeterminism hunt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.ncsn as N
import smd_amd.schedule as S
from smd_amd.engine import NetConfig

import smd_amd.lib as lib
opts = dict(kv.split("=") for kv in sys.argv[1:] if not kv.startswith("T:"))
for kv in sys.argv[1:]:
    if kv.startswith("T:"):
        k, v = kv[2:].split("=")
        lib.check(lib.get_lib().smd_set_tuning(k.encode(), int(v)))
if "MAINSTREAM" in os.environ:
    _ms = torch.cuda.Stream()
    torch.cuda.set_stream(_ms)
cfg = NetConfig(architecture="TransformerDDPM", data_channels=512, seq_len=32, num_timesteps=1000)
model = N.Model(cfg, "cuda:0", seed=0)
eng = model.train_engine(ema=False)
for k, v in opts.items():
    eng.set_option(k, int(v))
eng.set_schedule(S.create_noise_schedule(1e-6, 0.01, 1000, "linear"), with_sampler=False)
B = 256
eng.bind(B, training=True)
g = torch.Generator().manual_seed(1)
x0 = torch.clamp(0.25 * torch.randn(B, 32, 512, generator=g), -1, 1).cuda()
labels = torch.randint(1, 1001, (B,), generator=g).int().cuda()
eps = torch.randn(B, 32, 512, generator=g).cuda()
ref = None
bad = {}
for it in range(12):
    eng.loss_backward(x0, labels, eps, stage=0)
    torch.cuda.synchronize()
    gv = {k: v.clone() for k, v in eng.named_views(eng.grads).items()}
    if ref is None:
        ref = gv
        continue
    for k in gv:
        if not torch.equal(gv[k], ref[k]):
            d = float((gv[k] - ref[k]).abs().max())
            bad.setdefault(k, []).append(d)
names = list(bad)
print("options", sys.argv[1:], "->", len(names), "tensors differed; first in backward order:", names[-3:] if names else "none",
      {k: len(bad[k]) for k in names[-3:]})

# ---- which workspace regions differ between two identical iterations (forward-saved activations must not)
def plan_offsets(B=256, S=32, C=512, Cp=512, E=128, M=2048, F=128, L=6, K=2):
    R = B * S
    off = 0
    out = []
    def take(name, nbytes):
        nonlocal off
        off = (off + 255) // 256 * 256
        out.append((name, off, nbytes))
        off += nbytes
    take("zero_page", 256); take("step_arrive", 256); take("mlp_part", 4 * R * E * 4); take("x_bf16", R * Cp * 2); take("pe", S * E * 4); take("pred", R * C * 4); take("s", B * 4)
    for l in range(L):
        take(f"h[{l}]", R * E * 4); take(f"h_mid[{l}]", R * E * 4); take(f"a1[{l}]", R * E * 2); take(f"qkv[{l}]", R * 3 * E * 2)
        take(f"o[{l}]", R * E * 2); take(f"a2[{l}]", R * E * 2); take(f"z1[{l}]", R * M * 2); take(f"u[{l}]", R * M * 2)
    take("h_last", R * E * 4); take("af", R * E * 2)
    for k in range(K + 1): take(f"y[{k}]", R * M * 4)
    for k in range(K):
        take(f"ya1[{k}]", R * M * 2); take(f"o1[{k}]", R * M * 2); take(f"ya2[{k}]", R * M * 2); take(f"zf1[{k}]", B * 4 * F * 2)
        take(f"f1[{k}]", B * 4 * F * 2); take(f"p[{k}]", B * 4 * F * 2); take(f"ss[{k}]", B * 2 * M * 4)
    take("ao", R * M * 2); take("emb", B * F * 2)
    take("eps", R * C * 4); take("loss", B * 4); take("dpred", R * Cp * 2); take("dy", R * M * 4)
    for k in range(K + 1): take(f"dyb[{k}]", R * M * 2)
    take("dA_M", R * M * 2)
    for k in range(K): take(f"do1[{k}]", R * M * 2)
    for k in range(K): take(f"dss[{k}]", B * 2 * M * 4)
    for k in range(K):
        take(f"dss_bf16[{k}]", B * 2 * M * 2); take(f"dp[{k}]", B * 4 * F * 2); take(f"df1[{k}]", B * 4 * F * 2)
    take("dh", R * E * 4)
    for i in range(2 * L + 1): take(f"dhb[{i}]", R * E * 2)
    take("dA_E", R * E * 2)
    for l in range(L): take(f"dqkv[{l}]", R * 3 * E * 2)
    take("do_", R * E * 2)
    for l in range(L): take(f"dz1[{l}]", R * M * 2)
    return out

ws = eng.workspace
snaps = []
for it in range(12):
    eng.loss_backward(x0, labels, eps, stage=0)
    torch.cuda.synchronize()
    snaps.append(ws.clone())
regions = plan_offsets()
for name, o, n in regions:
    diffs = sum(int(not torch.equal(snaps[0][o:o + n], s[o:o + n])) for s in snaps[1:])
    if diffs:
        print(f"  region {name:12s} differs in {diffs}/11 repeats")

# ---- shape of the first difference: which rows / how many elements of the earliest differing dhb slot
import numpy as np
reg = {n: (o, sz) for n, o, sz in regions}
cand = [f"dhb[{i}]" for i in range(12, -1, -1)]
for name in cand:
    o, n = reg[name]
    a = snaps[0][o:o + n].view(torch.bfloat16).view(8192, 128).float()
    hit = False
    for s_ in snaps[1:]:
        b = s_[o:o + n].view(torch.bfloat16).view(8192, 128).float()
        ne = (a != b)
        if ne.any():
            rows = ne.any(1).nonzero().flatten()
            print(f"{name}: {int(ne.sum())} elements differ in {rows.numel()} rows; first rows {rows[:12].tolist()} "
                  f"max abs {float((a - b).abs().max()):.3e}; per-row counts {ne.sum(1)[rows[:6]].tolist()}")
            hit = True
            break
    if hit:
        break
