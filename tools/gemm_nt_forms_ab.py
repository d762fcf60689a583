"""Tile forms of gemm_nt_kernel for the GEMMs that are too narrow for the 256 x 256 kernel (knob gemm_nt_form): out_proj
(8192 x 2048 -> 512, fp32 + bias; the same at 4096 rows = one sampler chain), in_proj (8192 x 512 -> 128), the ragged C = 146
out_proj.  Interleaved, rotating operand sets, outputs compared with form 0.  python tools/gemm_nt_forms_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.lib as lib
L = lib.get_lib()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
FORMS = {0: "auto", 1: "<64,2>", 2: "<128,2>", 3: "<128,2,KG2>", 4: "<64,2,KG2>", 5: "<128,3>", 6: "<64,3,KG2>"}
g = torch.Generator().manual_seed(0)
for (M, N, K, name) in [(8192, 512, 2048, "out_proj B=256"), (4096, 512, 2048, "out_proj one chain"), (8192, 128, 512, "in_proj"),
                        (8192, 192, 2048, "out_proj C=146 (padded 192)")]:
    NSET = 4
    As = [torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev) for _ in range(NSET)]
    Bt = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    outs = [torch.empty(M, N, device=dev) for _ in range(NSET)]

    def call(i):
        lib.check(L.smd_gemm_bf16_nt(P(As[i % NSET]), K, P(Bt), K, M, N, K, P(bias), 0, None, 0, P(outs[i % NSET]), N, None, 0, st))

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

    res = {f: [] for f in FORMS}
    for rnd in range(5):
        for f in FORMS:
            lib.check(L.smd_set_tuning(b"gemm_nt_form", f))
            res[f].append(timeit())
    ref = None
    for f in FORMS:
        lib.check(L.smd_set_tuning(b"gemm_nt_form", f))
        call(0)
        torch.cuda.synchronize()
        o = outs[0].clone()
        ref = o if ref is None else ref
        r = sorted(res[f])
        err = float((o - ref).abs().max() / ref.abs().max())
        print(f"gemm_nt_forms {name} [{M} x {K} -> {N}] form {f} {FORMS[f]:12s}: median {r[len(r) // 2]:6.2f} us  min {r[0]:6.2f} us  "
              f"{2.0 * M * N * K / r[len(r) // 2] / 1e6:7.1f} TF   max diff vs auto {err:.1e}")
    lib.check(L.smd_set_tuning(b"gemm_nt_form", 0))
