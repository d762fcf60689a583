"""Interleaved A/B of the hidden-split encoder MLP forward kernel: 16 waves (four per SIMD, the default) vs 8 waves (two per SIMD,
mlp_variant = 5) at the bench shape (8192 rows; ROWS=4096 = one sampler chain), partial tiles compared bit for bit.
python tools/mlp_fwd16_ab.py"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.lib as lib
L = lib.get_lib()
dev = "cuda:0"
M = 2048
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
for rows in (8192, 4096):
    g = torch.Generator().manual_seed(0)
    h = (torch.randn(rows, 128, generator=g) * 1.5 + 0.3).to(dev)
    W1t = (torch.randn(M, 128, generator=g) * 0.09).to(torch.bfloat16).to(dev)
    W2t = (torch.randn(128, M, generator=g) / math.sqrt(M)).to(torch.bfloat16).to(dev)
    b1, b2 = (0.1 * torch.randn(M, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
    part = torch.empty(4, rows, 128, device=dev)
    a2 = torch.randn(rows, 128, generator=g).to(torch.bfloat16).to(dev)

    def run():
        lib.check(L.smd_mlp_block_fwd_hs(P(a2), P(h), rows, P(W1t), P(b1), P(W2t), P(b2), M, P(part), st))

    def timeit(reps=60):
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    res, outs = {0: [], 5: []}, {}
    for rnd in range(7):
        for v in (0, 5):
            lib.check(L.smd_set_tuning(b"mlp_variant", v))
            res[v].append(timeit())
    for v in (0, 5):
        lib.check(L.smd_set_tuning(b"mlp_variant", v))
        part.zero_()
        run()
        torch.cuda.synchronize()
        outs[v] = part.clone()
    lib.check(L.smd_set_tuning(b"mlp_variant", 0))
    med = {v: sorted(r)[len(r) // 2] for v, r in res.items()}
    print(f"mlp_fwd16_ab rows={rows}: 8 waves {med[5]:.2f} us (min {min(res[5]):.2f})  16 waves {med[0]:.2f} us (min {min(res[0]):.2f})  "
          f"{(med[0] / med[5] - 1) * 100:+.1f} %   bitwise equal: {torch.equal(outs[0], outs[5])}")
