"""Where a chunk of mlp_hs_fwd goes: s_memtime stamps of every wave at its phase boundaries (instrumented instantiation of the
kernel, tuning knob mlp_hs_dbg = 64; the stamps replace the result).  8192 rows, hidden 2048 = the bench shape.
    gpurun -- python tools/mlp_hs_phases.py > gpurun_out/<tag>_mlp_hs_phases.txt
The tick of s_memtime differs between boxes of the pool (1.5-3 per ns): compare phases within one run only."""
import os, sys, math, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smd_amd.lib as lib
L = lib.get_lib(); dev = "cuda:0"
rows, M = 8192, 2048
g = torch.Generator().manual_seed(0)
h = (torch.randn(rows, 128, generator=g) * 1.5 + 0.3).to(dev)
W1t = (torch.randn(M, 128, generator=g) * 0.09).to(torch.bfloat16).to(dev)
W2t = (torch.randn(128, M, generator=g) / math.sqrt(M)).to(torch.bfloat16).to(dev)
b1, b2 = (0.1 * torch.randn(M, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
a2 = torch.randn(rows, 128, device=dev).to(torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
part = torch.zeros(4, rows, 128, device=dev)
lib.check(L.smd_set_tuning(b"mlp_hs_dbg", 64))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
COLD = os.environ.get("SMD_COLD") == "1"       # in-step conditions: every launch runs behind a kernel that swept 64 MB through L2
_big = [torch.randn(16 << 20, device=dev) for _ in range(2)] if COLD else None
for it in range(6):
    if COLD:
        _big[1].copy_(_big[0])
    e0.record()
    lib.check(L.smd_mlp_block_fwd_hs(P(a2), P(h), rows, P(W1t), P(b1), P(W2t), P(b2), M, P(part), st))
    e1.record()
    torch.cuda.synchronize()
print(("COLD (behind a 64 MB copy) " if COLD else "") + f"instrumented kernel: {e0.elapsed_time(e1) * 1e3:.1f} us (event pair around one launch; the shipped instantiation: tools/mlp_ab.py)")
raw = part.view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF      # [4][rows][128]
ts = np.zeros((4, 64, 8, 36), np.int64)
for q in range(4):
    for grp in range(64):
        base = raw[q, grp * 128:(grp + 1) * 128].reshape(-1)
        ts[q, grp] = base[:8 * 40].reshape(8, 40)[:, :36]
t0 = ts[..., 0:1]
d = (ts - t0) & 0xFFFFFFFF
names = {1: "prologue done (a2 frags in regs)", 34: "loop done", 35: "partial tile stored"}
for c in range(4):
    for k, n in enumerate(["top", "W1 landed + barrier", "GEMM1 frags read", "GEMM1 MFMAs retired", "GELU + u written", "W2 landed + barrier", "GEMM2 done"]):
        names[2 + c * 8 + k] = f"chunk {c}: {n}"
print("ticks since the wave's first instruction: mean over 256 workgroups x 8 waves   [min .. max]   delta to previous (mean)")
prev = None
for i in sorted(names):
    v = d[..., i].reshape(-1)
    dl = "" if prev is None else f"{(d[..., i] - d[..., prev]).mean():9.0f}"
    print(f"  {names[i]:36s} {v.mean():9.0f}  [{v.min():7d} .. {v.max():7d}]  {dl}")
    prev = i
# per wave group (th = 0: waves 0-3, th = 1: waves 4-7)
for i in (3, 5, 6, 7, 8):
    a_, b_ = d[:, :, :4, i].mean(), d[:, :, 4:, i].mean()
    print(f"  {names[i]:36s} waves 0-3 {a_:9.0f}   waves 4-7 {b_:9.0f}")
# the GELU phase as a window per wave group (one wave of each group on every SIMD: the two share its VALU)
for c in range(4):
    a0, a1 = d[:, :, :4, 5 + c * 8].mean(), d[:, :, :4, 6 + c * 8].mean()
    b0, b1 = d[:, :, 4:, 5 + c * 8].mean(), d[:, :, 4:, 6 + c * 8].mean()
    print(f"  chunk {c} GELU window: waves 0-3 {a0:.0f}..{a1:.0f} ({a1 - a0:.0f})  waves 4-7 {b0:.0f}..{b1:.0f} ({b1 - b0:.0f})  "
          f"both {min(a0, b0):.0f}..{max(a1, b1):.0f} ({max(a1, b1) - min(a0, b0):.0f})")
lib.check(L.smd_set_tuning(b"mlp_hs_dbg", 0))
