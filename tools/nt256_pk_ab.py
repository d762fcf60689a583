"""Interleaved A/B of the 256x256 GEMM's packed-bf16 epilogue (knob gemm_nt256_pk) on the DenseResBlock shape, with the
bench's four rotating operand sets; also the e4m3 form.  python tools/nt256_pk_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smd_amd.lib as lib  # noqa: E402

L = lib.get_lib()
dev = "cuda:0"
R, M = 8192, 2048
NSET = 4
As = [torch.randn(R, M, device=dev).to(torch.bfloat16) for _ in range(NSET)]
Wt = (torch.randn(M, M, device=dev) * 0.02).to(torch.bfloat16)
bias = torch.randn(M, device=dev)
outs = [torch.empty(R, M, dtype=torch.bfloat16, device=dev) for _ in range(NSET)]
st = torch.cuda.current_stream().cuda_stream
q8 = [torch.empty(R, M, dtype=torch.uint8, device=dev) for _ in range(NSET)]
s8 = [torch.empty(R, dtype=torch.int32, device=dev) for _ in range(NSET)]
w8, ws8 = torch.empty(M, M, dtype=torch.uint8, device=dev), torch.empty(M, dtype=torch.int32, device=dev)
for i in range(NSET):
    lib.check(L.smd_quantize_rows_e4m3(As[i].data_ptr(), M, R, M, q8[i].data_ptr(), s8[i].data_ptr(), st))
lib.check(L.smd_quantize_rows_e4m3(Wt.data_ptr(), M, M, M, w8.data_ptr(), ws8.data_ptr(), st))


def call_b(i):
    lib.check(L.smd_gemm_bf16_nt(As[i % NSET].data_ptr(), M, Wt.data_ptr(), M, R, M, M, bias.data_ptr(), 0, None, 0, None, 0,
                                 outs[i % NSET].data_ptr(), M, st))


def call_8(i):
    lib.check(L.smd_gemm_e4m3_nt(q8[i % NSET].data_ptr(), M, s8[i % NSET].data_ptr(), w8.data_ptr(), M, ws8.data_ptr(), R, M, M,
                                 bias.data_ptr(), None, 0, None, 0, outs[i % NSET].data_ptr(), M, st))


def timed(call, reps=48):
    for i in range(4):
        call(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        call(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, call in (("bf16", call_b), ("e4m3", call_8)):
    res = {0: [], 1: []}
    for rnd in range(4):
        for pk in (1, 0):
            lib.check(L.smd_set_tuning(b"gemm_nt256_pk", pk))
            res[pk].append(timed(call))
    lib.check(L.smd_set_tuning(b"gemm_nt256_pk", 1))
    fl = 2.0 * R * M * M
    for pk in (0, 1):
        b = min(res[pk])
        print(f"nt256_pk_ab {name} 8192x2048x2048 bias->bf16, pk={pk}: best {b:.2f} us ({fl / b / 1e6:.0f} TF/s), rounds "
              + " ".join(f"{v:.2f}" for v in res[pk]))
