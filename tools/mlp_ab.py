"""Interleaved A/B of the fused encoder MLP half-layer kernels at the bench shape (8192 rows, hidden 2048)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.lib as lib

L = lib.get_lib()
dev = "cuda:0"
rows, M = int(os.environ.get("ROWS", "8192")), 2048
g = torch.Generator().manual_seed(0)
h = (torch.randn(rows, 128, generator=g) * 1.5 + 0.3).to(dev)
gamma, beta = (1 + 0.2 * torch.randn(128, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
W1t = (torch.randn(M, 128, generator=g) * 0.09).to(torch.bfloat16).to(dev)
W2t = (torch.randn(128, M, generator=g) / math.sqrt(M)).to(torch.bfloat16).to(dev)
b1, b2 = (0.1 * torch.randn(M, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
out = torch.empty(rows, 128, device=dev)
part = torch.empty(4, rows, 128, device=dev)
a2 = torch.randn(rows, 128, device=dev).to(torch.bfloat16)
af = torch.empty(rows, 128, dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()


def old():
    lib.check(L.smd_mlp_block_fwd(P(h), P(out), rows, P(gamma), P(beta), P(W1t), P(b1), P(W2t), P(b2), M, None, None, None, st))


def new():
    lib.check(L.smd_mlp_block_fwd_hs(P(a2), P(h), rows, P(W1t), P(b1), P(W2t), P(b2), M, P(part), st))


def combine():
    lib.check(L.smd_ln128_parts(P(part), rows * 128, rows, P(gamma), P(beta), None, P(af), st))


def timeit(f, reps=40):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


res = {"old": [], "hs": [], "ln128_parts": []}
dbgs = [int(v) for v in os.environ.get("DBG", "").split(",") if v]
for d in dbgs:
    res[f"hs dbg={d}"] = []
for rnd in range(5):
    res["old"].append(timeit(old))
    lib.check(L.smd_set_tuning(b"mlp_hs_dbg", 0))
    res["hs"].append(timeit(new))
    res["ln128_parts"].append(timeit(combine))
    for d in dbgs:
        lib.check(L.smd_set_tuning(b"mlp_hs_dbg", d))
        res[f"hs dbg={d}"].append(timeit(new))
    lib.check(L.smd_set_tuning(b"mlp_hs_dbg", 0))
for k, v in res.items():
    v.sort()
    print(f"mlp_block_fwd {k:10s} rows={rows}: median {v[len(v) // 2]:.1f} us  min {v[0]:.1f} us  (5 rounds x 40 launches)")
