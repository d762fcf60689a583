"""Interleaved A/B of the three encoder kernels whose LDS layouts changed in round 4 (tools/lds_conflicts.py) between two builds of
the library in ONE process: csrc/libsmd_hip_old.so (built from the previous commit) and the shipped csrc/libsmd_hip.so.  Bench
shapes (8192 rows), outputs compared bit for bit.  python tools/lds_swizzle_ab.py [num_heads]"""
import ctypes as C
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.lib as lib
new = lib.get_lib()
old = C.CDLL(os.path.join(ROOT, "symbolic-music-diffusion_amd", "csrc", "libsmd_hip_old.so"))
for name, (res, args) in lib._SIGS.items():
    if hasattr(old, name):
        fn = getattr(old, name)
        fn.restype, fn.argtypes = res, args
H = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
rows, M = 8192, 2048
g = torch.Generator().manual_seed(0)
bf = lambda t: t.to(torch.bfloat16).to(dev)
P = lambda t: None if t is None else t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
a2, dh = bf(torch.randn(rows, 128, generator=g)), bf(torch.randn(rows, 128, generator=g) * 1e-3)
W1 = torch.randn(128, M, generator=g) * 0.09
W2 = torch.randn(M, 128, generator=g) / math.sqrt(M)
W1t, W1p, W2p = bf(W1.t().contiguous()), bf(W1.contiguous()), bf(W2.contiguous())
b1 = (0.1 * torch.randn(M, generator=g)).to(dev)
u, dz = torch.empty(rows, M, dtype=torch.bfloat16, device=dev), torch.empty(rows, M, dtype=torch.bfloat16, device=dev)
part = torch.empty(4, rows, 128, device=dev)
# attention
parts = (0.5 * torch.randn(4, rows, 128, generator=g)).to(dev)
gam, bet = (1 + 0.1 * torch.randn(128, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
Wqkv_t, Wo_t = bf(torch.randn(384, 128, generator=g) * 0.09), bf(torch.randn(128, 128, generator=g) * 0.09)
bqkv, bo = (0.1 * torch.randn(384, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
h_comb, h_out = torch.empty(rows, 128, device=dev), torch.empty(rows, 128, device=dev)
a2o, sa1, so = (torch.empty(rows, 128, dtype=torch.bfloat16, device=dev) for _ in range(3))
sqkv = torch.empty(rows, 384, dtype=torch.bfloat16, device=dev)
Wo_p, Wqkv_p = bf(torch.randn(128, 128, generator=g) * 0.09), bf(torch.randn(128, 384, generator=g) * 0.09)
dqkv, da1 = torch.empty(rows, 384, dtype=torch.bfloat16, device=dev), torch.empty(rows, 128, dtype=torch.bfloat16, device=dev)

KERNELS = {
    "mlp_hs_bwd": (lambda L: L.smd_mlp_block_bwd_hs(P(a2), P(dh), rows, P(W1t), P(W2p), P(W1p), P(b1), M, P(u), P(dz), P(part), st),
                   lambda: (u.clone(), dz.clone(), part.clone())),
    "attn_block_fwd (partial-sum input, training form)": (
        lambda L: L.smd_attn_block_fwd_ex(None, P(parts), rows * 128, P(h_comb), P(h_out), rows, P(gam), P(bet), P(Wqkv_t), P(bqkv), P(Wo_t), P(bo), H,
                                          P(gam), P(bet), P(a2o), P(sa1), P(sqkv), P(so), st),
        lambda: (h_comb.clone(), h_out.clone(), a2o.clone(), sa1.clone(), sqkv.clone(), so.clone())),
    "attn_block_fwd (partial-sum input, sampling form)": (
        lambda L: L.smd_attn_block_fwd_ex(None, P(parts), rows * 128, None, P(h_out), rows, P(gam), P(bet), P(Wqkv_t), P(bqkv), P(Wo_t), P(bo), H,
                                          P(gam), P(bet), P(a2o), None, None, None, st),
        lambda: (h_out.clone(), a2o.clone())),
    "attn_block_bwd": (lambda L: L.smd_attn_block_bwd(P(dh), P(sqkv), P(Wo_p), P(Wqkv_p), P(dqkv), P(da1), rows, H, st),
                       lambda: (dqkv.clone(), da1.clone())),
}


def timeit(f, L, reps=60):
    for _ in range(5):
        assert f(L) == 0, new.smd_last_error()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f(L)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, (f, grab) in KERNELS.items():
    res = {"old": [], "new": []}
    for rnd in range(7):
        for tag, L in (("old", old), ("new", new)):
            res[tag].append(timeit(f, L))
    outs = {}
    for tag, L in (("old", old), ("new", new)):
        f(L)
        torch.cuda.synchronize()
        outs[tag] = grab()
    same = all(torch.equal(x, y) for x, y in zip(outs["old"], outs["new"]))
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"lds_swizzle_ab H={H} {name}: old {med['old']:.2f} us (min {min(res['old']):.2f})  new {med['new']:.2f} us (min {min(res['new']):.2f})  "
          f"{(med['new'] / med['old'] - 1) * 100:+.1f} %   bitwise equal: {same}")
