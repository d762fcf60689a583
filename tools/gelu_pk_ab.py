"""Interleaved two-library A/B of the hidden-split MLP kernels (csrc/libsmd_hip_old.so = the previous commit, built by
tools/build_old_lib.sh, against the shipped library) at the bench shapes: forward and recompute backward, 8192 and 4096 rows;
every output compared bit for bit.  python tools/gelu_pk_ab.py"""
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
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
M = 2048
g = torch.Generator().manual_seed(0)
for rows in (8192, 4096):
    a2 = torch.randn(rows, 128, generator=g).to(torch.bfloat16).to(dev)
    h = (torch.randn(rows, 128, generator=g) * 1.5 + 0.3).to(dev)
    dh = (torch.randn(rows, 128, generator=g) * 1e-3).to(torch.bfloat16).to(dev)
    W1 = (torch.randn(128, M, generator=g) * 0.09)
    W2 = (torch.randn(M, 128, generator=g) / math.sqrt(M))
    W1t = W1.t().contiguous().to(torch.bfloat16).to(dev)
    W1p = W1.contiguous().to(torch.bfloat16).to(dev)
    W2p = W2.contiguous().to(torch.bfloat16).to(dev)
    W2t = W2.t().contiguous().to(torch.bfloat16).to(dev)
    b1, b2 = (0.1 * torch.randn(M, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
    out = {k: dict(part=torch.empty(4, rows, 128, device=dev), u=torch.empty(rows, M, dtype=torch.bfloat16, device=dev),
                   dz=torch.empty(rows, M, dtype=torch.bfloat16, device=dev), dpart=torch.empty(4, rows, 128, device=dev)) for k in ("old", "new")}

    def fwd(L, k):
        rc = L.smd_mlp_block_fwd_hs(P(a2), P(h), rows, P(W1t), P(b1), P(W2t), P(b2), M, P(out[k]["part"]), st)
        assert rc == 0

    def bwd(L, k):
        rc = L.smd_mlp_block_bwd_hs(P(a2), P(dh), rows, P(W1t), P(W2p), P(W1p), P(b1), M, P(out[k]["u"]), P(out[k]["dz"]), P(out[k]["dpart"]), st)
        assert rc == 0

    def timeit(f, L, k, reps=40):
        for _ in range(5):
            f(L, k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f(L, k)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for name, f, keys in (("mlp_hs_fwd", fwd, ("part",)), ("mlp_hs_bwd", bwd, ("u", "dz", "dpart"))):
        res = {"old": [], "new": []}
        for rnd in range(7):
            for k, L in (("old", old), ("new", new)):
                res[k].append(timeit(f, L, k))
        torch.cuda.synchronize()
        same = all(torch.equal(out["old"][q], out["new"][q]) for q in keys)
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        mn = {k: min(v) for k, v in res.items()}
        print(f"gelu_pk_ab {name} rows={rows}: old {med['old']:.2f} us (min {mn['old']:.2f})  new {med['new']:.2f} us (min {mn['new']:.2f})  "
              f"{(med['new'] / med['old'] - 1) * 100:+.1f} %   bitwise equal: {same}")
