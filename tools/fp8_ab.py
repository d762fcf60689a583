"""Interleaved timing: DenseResBlock GEMM 8192x2048x2048 in bf16 vs e4m3 (both epilogue forms), and the FiLM LayerNorm
with bf16 / e4m3 / both outputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.lib as lib

L = lib.get_lib()
dev = "cuda:0"
R, M = 8192, 2048
P = lambda t: None if t is None else t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
NSET = 4
As = [torch.randn(R, M, device=dev).to(torch.bfloat16) for _ in range(NSET)]
Wt = (torch.randn(M, M, device=dev) * 0.02).to(torch.bfloat16)
bias = torch.zeros(M, device=dev)
outb = [torch.empty(R, M, dtype=torch.bfloat16, device=dev) for _ in range(NSET)]
res = [torch.randn(R, M, device=dev) for _ in range(2)]
outf = [torch.empty(R, M, device=dev) for _ in range(2)]
q = [torch.zeros(R, M, dtype=torch.uint8, device=dev) for _ in range(NSET)]
s = [torch.zeros(R, dtype=torch.int32, device=dev) for _ in range(NSET)]
qw, sw = torch.zeros(M, M, dtype=torch.uint8, device=dev), torch.zeros(M, dtype=torch.int32, device=dev)
for i in range(NSET):
    lib.check(L.smd_quantize_rows_e4m3(P(As[i]), M, R, M, P(q[i]), P(s[i]), st))
lib.check(L.smd_quantize_rows_e4m3(P(Wt), M, M, M, P(qw), P(sw), st))
x = torch.randn(R, M, device=dev)
gamma, beta = torch.ones(M, device=dev), torch.zeros(M, device=dev)
scale, shift = torch.ones(R // 32, M, device=dev), torch.zeros(R // 32, M, device=dev)

calls = {
    "gemm bf16 (bias->bf16)": lambda i: lib.check(L.smd_gemm_bf16_nt(P(As[i % NSET]), M, P(Wt), M, R, M, M, P(bias), 0, None, 0, None, 0, P(outb[i % NSET]), M, st)),
    "gemm e4m3 (bias->bf16)": lambda i: lib.check(L.smd_gemm_e4m3_nt(P(q[i % NSET]), M, P(s[i % NSET]), P(qw), M, P(sw), R, M, M, P(bias), None, 0, None, 0, P(outb[i % NSET]), M, st)),
    "gemm bf16 (+fp32 res->fp32)": lambda i: lib.check(L.smd_gemm_bf16_nt(P(As[i % NSET]), M, P(Wt), M, R, M, M, P(bias), 0, P(res[i % 2]), M, P(outf[i % 2]), M, None, 0, st)),
    "gemm e4m3 (+fp32 res->fp32)": lambda i: lib.check(L.smd_gemm_e4m3_nt(P(q[i % NSET]), M, P(s[i % NSET]), P(qw), M, P(sw), R, M, M, P(bias), P(res[i % 2]), M, P(outf[i % 2]), M, None, 0, st)),
    "ln film+swish -> bf16": lambda i: lib.check(L.smd_layernorm_fwd(P(x), R, M, P(gamma), P(beta), P(scale), P(shift), M, 32, 1, P(outb[i % NSET]), st)),
    "ln film+swish -> e4m3": lambda i: lib.check(L.smd_layernorm_fwd_e4m3(P(x), R, M, P(gamma), P(beta), P(scale), P(shift), M, 32, 1, P(q[i % NSET]), P(s[i % NSET]), None, st)),
    "ln film+swish -> e4m3 + bf16": lambda i: lib.check(L.smd_layernorm_fwd_e4m3(P(x), R, M, P(gamma), P(beta), P(scale), P(shift), M, 32, 1, P(q[i % NSET]), P(s[i % NSET]), P(outb[i % NSET]), st)),
}


def timeit(f, reps=40):
    for i in range(4):
        f(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        f(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


resu = {k: [] for k in calls}
for rnd in range(5):
    for k, f in calls.items():
        resu[k].append(timeit(f))
for k, v in resu.items():
    v.sort()
    extra = f"  = {2.0 * R * M * M / (v[2] * 1e-6) / 1e12:.0f} TFLOP/s" if k.startswith("gemm") else ""
    print(f"{k:32s} median {v[2]:6.1f} us  min {v[0]:6.1f} us{extra}")
