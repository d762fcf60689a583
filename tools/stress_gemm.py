"""Stress: are the NT GEMM kernels bitwise repeatable while other kernels run concurrently on a second stream?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.lib as lib
L = lib.get_lib()
dev = "cuda:0"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
Xw = torch.randn(8192, 2048, device=dev).to(torch.bfloat16)
Yw = (torch.randn(8192, 2048, device=dev) * 0.1).to(torch.bfloat16)
dw = torch.empty(2048, 2048, device=dev); dbw = torch.empty(2048, device=dev)
zero = torch.zeros(128, dtype=torch.bfloat16, device=dev)
slab = torch.empty(int(L.smd_gemm_tn_slab_elems()), device=dev)
scr = torch.zeros(128, dtype=torch.bfloat16, device=dev)
for (M, N, K) in [(8192, 128, 2048), (8192, 2048, 128), (8192, 128, 128), (8192, 2048, 2048), (8192, 384, 128)]:
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    Bt = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ref, bad = None, 0
    for it in range(40):
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        with torch.cuda.stream(s2):
            for _ in range(2):
                lib.check(L.smd_gemm_bf16_tn(Xw.data_ptr(), 2048, Yw.data_ptr(), 2048, 8192, 2048, 2048, dw.data_ptr(), 2048, dbw.data_ptr(),
                                             zero.data_ptr(), slab.data_ptr(), slab.numel(), scr.data_ptr(), scr.numel(), 1, s2.cuda_stream))
        with torch.cuda.stream(s1):
            for _ in range(4):
                lib.check(L.smd_gemm_bf16_nt(A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), 0, None, 0, None, 0,
                                             out.data_ptr(), N, s1.cuda_stream))
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        elif not torch.equal(out, ref):
            bad += 1
            if bad <= 2:
                print(f"  {M}x{N}x{K} it={it}: {(out != ref).sum().item()} elements differ")
    print(f"gemm_nt {M}x{N}x{K}: {bad}/39 runs differed")
