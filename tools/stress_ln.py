"""Stress: does layernorm_bwd (D=128 narrow / D=2048 wide) give bitwise identical results while other kernels run
concurrently on a second stream?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.lib as lib
L = lib.get_lib()
dev = "cuda:0"
R = 8192
for D in (128, 2048):
    g = torch.Generator().manual_seed(D)
    x = torch.randn(R, D, generator=g).to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(D, generator=g)).to(dev), torch.zeros(D, device=dev)
    dout = (torch.randn(R, D, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    part = torch.empty(R * 2 * D // 16, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    A = torch.randn(8192, 2048, device=dev).to(torch.bfloat16)
    Wt = torch.randn(2048, 2048, device=dev).to(torch.bfloat16)
    outg = torch.empty(8192, 2048, dtype=torch.bfloat16, device=dev)
    Xw, Yw = A, (torch.randn(8192, 2048, device=dev) * 0.1).to(torch.bfloat16)
    dw = torch.empty(2048, 2048, device=dev); dbw = torch.empty(2048, device=dev)
    zero = torch.zeros(128, dtype=torch.bfloat16, device=dev)
    slab = torch.empty(int(L.smd_gemm_tn_slab_elems()), device=dev)
    scr = torch.zeros(128, dtype=torch.bfloat16, device=dev)
    ref = None
    bad = 0
    torch.cuda.synchronize()
    for it in range(60):
        dx = torch.empty(R, D, device=dev); dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
        with torch.cuda.stream(s2):
            for _ in range(2):
                lib.check(L.smd_gemm_bf16_tn(Xw.data_ptr(), 2048, Yw.data_ptr(), 2048, 8192, 2048, 2048, dw.data_ptr(), 2048, dbw.data_ptr(),
                                             zero.data_ptr(), slab.data_ptr(), slab.numel(), scr.data_ptr(), scr.numel(), 1, s2.cuda_stream))
        with torch.cuda.stream(s1):
            for _ in range(3):
                lib.check(L.smd_layernorm_bwd(x.data_ptr(), R, D, gamma.data_ptr(), beta.data_ptr(), None, None, 0, 1, 0, dout.data_ptr(),
                                              dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, None, part.data_ptr(), part.numel(), s1.cuda_stream))
                dg.zero_(); db.zero_()
            lib.check(L.smd_layernorm_bwd(x.data_ptr(), R, D, gamma.data_ptr(), beta.data_ptr(), None, None, 0, 1, 0, dout.data_ptr(),
                                          dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, None, part.data_ptr(), part.numel(), s1.cuda_stream))
        torch.cuda.synchronize()
        cur = (dx.clone(), dg.clone(), db.clone())
        if ref is None:
            ref = cur
        elif not all(torch.equal(a, b) for a, b in zip(cur, ref)):
            bad += 1
            if bad <= 3:
                print(f"D={D} it={it}: dx equal {torch.equal(cur[0], ref[0])} dg equal {torch.equal(cur[1], ref[1])} db equal {torch.equal(cur[2], ref[2])} "
                      f"ndiff dx {(cur[0] != ref[0]).sum().item()}")
    print(f"D={D}: {bad}/59 runs differed from the first")


# ---- the engine's in-place form (dres aliases dx, bf16 copy), D = 128, many repeats, mixed co-runners
D = 128
g = torch.Generator().manual_seed(5)
x = torch.randn(R, D, generator=g).to(dev)
gamma, beta = (1 + 0.1 * torch.randn(D, generator=g)).to(dev), torch.zeros(D, device=dev)
dout = (torch.randn(R, D, generator=g) * 0.1).to(torch.bfloat16).to(dev)
dh0 = torch.randn(R, D, generator=g).to(dev)
part = torch.empty(R * 2 * D // 16, device=dev)
Xs = torch.randn(8192, 128, device=dev).to(torch.bfloat16)
dws = torch.empty(128, 2048, device=dev); dbs = torch.empty(2048, device=dev)
for inplace in (1, 0):
    ref, bad = None, 0
    NIT = 600
    for it in range(NIT):
        dh = dh0.clone(); dxo = dh if inplace else torch.empty_like(dh0)
        dxb = torch.empty(R, D, dtype=torch.bfloat16, device=dev)
        dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(s2):
            lib.check(L.smd_gemm_bf16_tn(Xw.data_ptr(), 2048, Yw.data_ptr(), 2048, 8192, 2048, 2048, dw.data_ptr(), 2048, dbw.data_ptr(),
                                         zero.data_ptr(), slab.data_ptr(), slab.numel(), scr.data_ptr(), scr.numel(), 1, s2.cuda_stream))
            lib.check(L.smd_gemm_bf16_tn(Xs.data_ptr(), 128, Yw.data_ptr(), 2048, 8192, 128, 2048, dws.data_ptr(), 2048, dbs.data_ptr(),
                                         zero.data_ptr(), slab.data_ptr(), slab.numel(), scr.data_ptr(), scr.numel(), 1, s2.cuda_stream))
        with torch.cuda.stream(s1):
            if it % 3:
                torch.cuda._sleep(20000 * (it % 7))
            lib.check(L.smd_layernorm_bwd_ex(x.data_ptr(), None, R, D, gamma.data_ptr(), beta.data_ptr(), dout.data_ptr(), dh.data_ptr(),
                                             dxo.data_ptr(), dxb.data_ptr(), dg.data_ptr(), db.data_ptr(), part.data_ptr(), part.numel(),
                                             s1.cuda_stream))
        torch.cuda.synchronize()
        cur = (dxo.clone(), dxb.clone(), dg.clone())
        if ref is None:
            ref = cur
        elif not all(torch.equal(a, b) for a, b in zip(cur, ref)):
            bad += 1
            if bad <= 2:
                ne = (cur[0] != ref[0])
                print(f"  inplace={inplace} it={it}: {int(ne.sum())} fp32 elements differ in rows {ne.any(1).nonzero().flatten()[:5].tolist()}")
    print(f"narrow LN backward inplace={inplace}: {bad}/{NIT - 1} runs differed")
