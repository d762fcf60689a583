"""Encoder half-layer kernels isolated vs under in-step conditions (18.2 -> 22 us, 10.3 -> 15 us).
Each kernel (attention forward, hidden-split MLP forward / backward at 8192 rows) is timed
  hot        back to back on the same operands (what tools/kbench.py reports)
  cold       each launch behind a 64 MB device copy (the L2s hold none of its operands and hold dirty lines: what a launch finds
             inside the step behind a 2048-wide GEMM epilogue)
  cold+W     as cold, but the layer's WEIGHTS are read into all eight L2s first (smd_probe_l2_warm) -- what a weight prefetch
             issued under the previous kernel would achieve
  cold+WA    as cold+W with the input activations warmed too (the remaining difference to hot = launch boundary / dirty-line
             write-back)
timing: one HIP-event pair per launch, the thrash / warm kernels outside the pair; median over 60 launches.
    python tools/encoder_cold_ab.py
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import smd_amd.lib as lib

L = lib.get_lib()
dev = "cuda:0"
rows, M, H = int(os.environ.get("ROWS", "8192")), 2048, 8
g = torch.Generator().manual_seed(0)
P = lambda t: None if t is None else t.data_ptr()
st = torch.cuda.current_stream().cuda_stream

# ---- operands
h = (torch.randn(rows, 128, generator=g) * 1.5 + 0.3).to(dev)
parts = (torch.randn(4, rows, 128, generator=g) * 0.7).to(dev)
gamma, beta = (1 + 0.1 * torch.randn(128, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
gamma2, beta2 = (1 + 0.1 * torch.randn(128, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
Wqkv = (torch.randn(384, 128, generator=g) * 0.09).to(torch.bfloat16).to(dev)
Wo = (torch.randn(128, 128, generator=g) * 0.09).to(torch.bfloat16).to(dev)
bqkv, bo = (0.1 * torch.randn(384, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
h_out = torch.empty(rows, 128, device=dev)
a2 = torch.randn(rows, 128, device=dev).to(torch.bfloat16)
W1t = (torch.randn(M, 128, generator=g) * 0.09).to(torch.bfloat16).to(dev)
W2t = (torch.randn(128, M, generator=g) / math.sqrt(M)).to(torch.bfloat16).to(dev)
W2 = W2t.t().contiguous()
W1 = W1t.t().contiguous()
b1, b2 = (0.1 * torch.randn(M, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
part = torch.empty(4, rows, 128, device=dev)
dh = torch.randn(rows, 128, device=dev).to(torch.bfloat16)
u = torch.empty(rows, M, dtype=torch.bfloat16, device=dev)
dz = torch.empty(rows, M, dtype=torch.bfloat16, device=dev)
big = [torch.randn(16 << 20, device=dev) for _ in range(2)]
sink = torch.zeros(1, dtype=torch.int32, device=dev)


def attn():
    lib.check(L.smd_attn_block_fwd_ex(None, P(parts), rows * 128, None, P(h_out), rows, P(gamma), P(beta), P(Wqkv), P(bqkv), P(Wo), P(bo),
                                      H, P(gamma2), P(beta2), P(a2), None, None, None, st))


def mlp_fwd():
    lib.check(L.smd_mlp_block_fwd_hs(P(a2), P(h), rows, P(W1t), P(b1), P(W2t), P(b2), M, P(part), st))


def mlp_bwd():
    lib.check(L.smd_mlp_block_bwd_hs(P(a2), P(dh), rows, P(W1t), P(W2), P(W1), P(b1), M, P(u), P(dz), P(part), st))


KERNELS = {
    "attn_block_fwd (4 partial tiles in, ln2 out)": (attn, [Wqkv, Wo, bqkv, bo, gamma, beta, gamma2, beta2], [parts]),
    "mlp_hs_fwd": (mlp_fwd, [W1t, W2t, b1, b2], [a2, h]),
    "mlp_hs_bwd": (mlp_bwd, [W1t, W2, W1, b1], [a2, dh]),
}


def warm(ts):
    for t in ts:
        n = t.numel() * t.element_size()
        n -= n % 16
        lib.check(L.smd_probe_l2_warm(t.data_ptr(), n, sink.data_ptr(), st))


def timed(f, pre, reps=60):
    out = []
    for _ in range(reps):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    out.sort()
    return out[len(out) // 2], out[0]


def back_to_back(f, reps=60):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def thrash():
    big[1].copy_(big[0])


print(f"rows = {rows}; per-launch event pairs cost ~2-4 us of their own (compare the pair-timed columns with each other)")
for name, (f, ws, acts) in KERNELS.items():
    b2b = back_to_back(f)
    hot = timed(f, lambda: None)
    cold = timed(f, thrash)
    cw = timed(f, lambda: (thrash(), warm(ws)))
    cwa = timed(f, lambda: (thrash(), warm(ws), warm(acts)))
    print(f"{name:46s} back-to-back {b2b:6.1f} us | pair-timed: hot {hot[0]:6.1f} (min {hot[1]:5.1f})  cold {cold[0]:6.1f} (min {cold[1]:5.1f})  "
          f"cold+W {cw[0]:6.1f} (min {cw[1]:5.1f})  cold+WA {cwa[0]:6.1f} (min {cwa[1]:5.1f})")
