"""Interleaved A/B of the D = 2048 LayerNorm backward (the engine's all-bf16 forms + the reduce) between csrc/libsmd_hip_old.so
and the shipped library in one process; every output compared bit for bit.  python tools/ln_bwd_ab.py"""
import ctypes as C
import os, sys
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
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
R, D = 8192, 2048
g0 = torch.Generator().manual_seed(0)
x = torch.randn(R, D, generator=g0).to(dev)
xbf = x.to(torch.bfloat16)
g, b = (1 + 0.1 * torch.randn(D, generator=g0)).to(dev), (0.1 * torch.randn(D, generator=g0)).to(dev)
ss = torch.randn(R // 32, 2 * D, generator=g0).to(dev)
dout = torch.randn(R, D, generator=g0).to(torch.bfloat16).to(dev)
dresb = torch.randn(R, D, generator=g0).to(torch.bfloat16).to(dev)
NSET = 3
dxb = [torch.empty(R, D, dtype=torch.bfloat16, device=dev) for _ in range(NSET)]
dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
dss = torch.zeros(R // 32, 2 * D, device=dev)
part = torch.empty(R * 2 * D // 16, device=dev)
P = lambda t: None if t is None else t.data_ptr()
forms = [("res.ln2 (bf16 x, bf16 dx, FiLM + swish)", None, 1), ("res.ln1 (bf16 x, bf16 dres, bf16 dx, FiLM + swish)", dresb, 1),
         ("ln_o (bf16 x, bf16 dx)", None, 0)]
for name, res, fs in forms:
    def call(L, i=0):
        rc = L.smd_layernorm_bwd_film(None, P(xbf), R, D, P(g), P(b), P(ss) if fs else None, ss[:, D:].data_ptr() if fs else None, 2 * D, 32, fs,
                                      P(dout), None, P(res), None, P(dxb[i % NSET]), P(dg), P(db), P(dss) if fs else None,
                                      dss[:, D:].data_ptr() if fs else None, 0, P(part), part.numel(), st)
        assert rc == 0, new.smd_last_error()

    def timeit(L, reps=40):
        for i in range(4):
            call(L, i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            call(L, i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    t = {"old": [], "new": []}
    for rnd in range(7):
        for k, L in (("old", old), ("new", new)):
            t[k].append(timeit(L))
    outs = {}
    for k, L in (("old", old), ("new", new)):
        call(L, 0)
        torch.cuda.synchronize()
        outs[k] = (dxb[0].clone(), dg.clone(), db.clone(), dss.clone())
    same = all(torch.equal(u, v) for u, v in zip(outs["old"], outs["new"]))
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    print(f"ln_bwd_ab {name}: old {med['old']:.2f} us (min {min(t['old']):.2f})  new {med['new']:.2f} us (min {min(t['new']):.2f})  "
          f"{(med['new'] / med['old'] - 1) * 100:+.1f} %   bitwise equal: {same}")
