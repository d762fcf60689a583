"""Does the 8192 x 2048 x 2048 ResBlock GEMM run faster as TWO concurrent 4096-row launches on two streams (their HBM bursts --
prologue fetch, 33 MB epilogue store -- no longer coincide) than as one 256-tile launch?  Interleaved rounds, rotating operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.lib as lib
L = lib.get_lib()
dev = "cuda:0"
M, N, K = 8192, 2048, 2048
NSET = 4
As = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(NSET)]
Bt = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
bias = torch.zeros(N, device=dev)
outs = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(NSET)]
lib.check(L.smd_set_tuning(b"gemm_nt256", 2))
s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def gemm(i, r0, rows, st):
    a, o = As[i % NSET], outs[i % NSET]
    lib.check(L.smd_gemm_bf16_nt(a.data_ptr() + r0 * K * 2, K, Bt.data_ptr(), K, rows, N, K, bias.data_ptr(), 0, None, 0, None, 0,
                                 o.data_ptr() + r0 * N * 2, N, st.cuda_stream))


def one(i):
    gemm(i, 0, M, s0)


def split(i, parts, stagger_us=0):
    ev = torch.cuda.Event()
    ev.record(s0)
    streams = [s1, s2, s0, s0][:parts] if parts <= 2 else [s1, s2, s1, s2]
    rows = M // parts
    for p in range(parts):
        st = streams[p]
        st.wait_event(ev)
        gemm(i, p * rows, rows, st)
    for st in {s1, s2}:
        e2 = torch.cuda.Event()
        e2.record(st)
        s0.wait_event(e2)


def timed(f, reps=40):
    for i in range(4):
        f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s0)
    for i in range(reps):
        f(i)
    e1.record(s0)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


res = {"one launch (256 tiles)": [], "two halves, two streams": [], "four quarters, two streams": []}
for _ in range(5):
    res["one launch (256 tiles)"].append(timed(one))
    res["two halves, two streams"].append(timed(lambda i: split(i, 2)))
    res["four quarters, two streams"].append(timed(lambda i: split(i, 4)))
for k, v in res.items():
    v.sort()
    print(f"{k:28s} median {v[2]:.1f} us  min {v[0]:.1f} us  -> {2.0 * M * N * K / v[2] / 1e6:.0f} TF")
