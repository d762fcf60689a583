"""Interleaved A/B of the hidden-split MLP backward kernel at the bench shape (8192 rows, hidden 2048): a2 / dh fragments re-read
from LDS every chunk (mlp_variant = 9) vs kept in registers (default); outputs compared bit for bit.  python tools/mlp_bwd_ab.py"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.lib as lib
L = lib.get_lib()
dev = "cuda:0"
rows, M = 8192, 2048
g = torch.Generator().manual_seed(0)
a2 = torch.randn(rows, 128, generator=g).to(torch.bfloat16).to(dev)
dh = (torch.randn(rows, 128, generator=g) * 1e-3).to(torch.bfloat16).to(dev)
W1 = (torch.randn(128, M, generator=g) * 0.09)
W2 = (torch.randn(M, 128, generator=g) / math.sqrt(M))
W1t = W1.t().contiguous().to(torch.bfloat16).to(dev)          # [M][128] forward pack of fc1
W1p = W1.contiguous().to(torch.bfloat16).to(dev)              # [128][M] dgrad pack of fc1
W2p = W2.contiguous().to(torch.bfloat16).to(dev)              # [M][128] dgrad pack of fc2
b1 = (0.1 * torch.randn(M, generator=g)).to(dev)
NSET = 3
us = [torch.empty(rows, M, dtype=torch.bfloat16, device=dev) for _ in range(NSET)]
dzs = [torch.empty(rows, M, dtype=torch.bfloat16, device=dev) for _ in range(NSET)]
part = torch.empty(4, rows, 128, device=dev)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()


def call(i):
    lib.check(L.smd_mlp_block_bwd_hs(P(a2), P(dh), rows, P(W1t), P(W2p), P(W1p), P(b1), M, P(us[i % NSET]), P(dzs[i % NSET]), P(part), st))


def timeit(reps=40):
    for i in range(4):
        call(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        call(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


res, outs = {0: [], 9: []}, {}
for rnd in range(5):
    for v in (0, 9):
        lib.check(L.smd_set_tuning(b"mlp_variant", v))
        res[v].append(timeit())
for v in (0, 9):
    lib.check(L.smd_set_tuning(b"mlp_variant", v))
    call(0)
    torch.cuda.synchronize()
    outs[v] = (us[0].clone(), dzs[0].clone(), part.clone())
lib.check(L.smd_set_tuning(b"mlp_variant", 0))
same = all(torch.equal(x, y) for x, y in zip(outs[0], outs[9]))
for v, name in ((9, "fragments re-read per chunk"), (0, "fragments in registers (default)")):
    r = sorted(res[v])
    print(f"mlp_bwd_ab {name}: median {r[len(r) // 2]:.2f} us  min {r[0]:.2f} us   bitwise equal: {same}")
