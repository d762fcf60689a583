"""Pairwise bitwise equality of the gradient buffer over repeated identical loss_backward calls (who is the odd one?)."""
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
n = int(os.environ.get("ITERS", "8"))
snaps = []
for it in range(n):
    eng.loss_backward(x0, labels, eps, stage=0)
    torch.cuda.synchronize()
    snaps.append(eng.grads.clone())
print("options", sys.argv[1:])
bad = sum(int(not torch.equal(snaps[0], snaps[j])) for j in range(1, n))
print(f"{bad} of {n - 1} repeats differ from the first")
if n <= 10:
    for i in range(n):
        print(" ".join("=" if torch.equal(snaps[i], snaps[j]) else "x" for j in range(n)))
views = [eng.named_views(s) for s in snaps]
names = list(views[0])
for i in range(1, n):
    d = [k for k in names if not torch.equal(views[i][k], views[i - 1][k])]
    if d:
        print(f"iter {i} vs {i-1}: {len(d)} tensors differ; last in parameter order: {d[-2:]}")
