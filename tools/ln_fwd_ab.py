"""Interleaved A/B of the wide LayerNorm forward variants (knob ln_fwd_wide) on the DenseResBlock form: bf16 trunk in, FiLM + swish,
bf16 out, [8192][2048], four rotating buffer sets; also the plain form (ln_o).  python tools/ln_fwd_ab.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smd_amd.lib as lib
L = lib.get_lib()
dev = "cuda:0"
R, D, S = 8192, 2048, 32
NSET = 4
xs = [(torch.randn(R, D, device=dev) * 2).to(torch.bfloat16) for _ in range(NSET)]
outs = [torch.empty(R, D, dtype=torch.bfloat16, device=dev) for _ in range(NSET)]
g, b = 1 + 0.1 * torch.randn(D, device=dev), 0.1 * torch.randn(D, device=dev)
ss = torch.randn(R // S, 2 * D, device=dev)
st = torch.cuda.current_stream().cuda_stream


def call(i, film):
    x, o = xs[i % NSET], outs[i % NSET]
    if film:
        lib.check(L.smd_layernorm_fwd_ex(None, x.data_ptr(), R, D, g.data_ptr(), b.data_ptr(), ss.data_ptr(), ss.data_ptr() + 4 * D, 2 * D, S, 1,
                                         o.data_ptr(), st))
    else:
        lib.check(L.smd_layernorm_fwd_ex(None, x.data_ptr(), R, D, g.data_ptr(), b.data_ptr(), None, None, 0, S, 0, o.data_ptr(), st))


def timed(film, reps=40):
    for i in range(4):
        call(i, film)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        call(i, film)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "2,3,4,5,6".split(","))]
for film in (True, False):
    ref = None
    res = {v: [] for v in variants}
    for rnd in range(4):
        for v in variants:
            lib.check(L.smd_set_tuning(b"ln_fwd_wide", v))
            res[v].append(timed(film))
    for v in variants:
        lib.check(L.smd_set_tuning(b"ln_fwd_wide", v))
        call(0, film)
        torch.cuda.synchronize()
        o = outs[0].clone()
        if ref is None:
            ref = o
        same = torch.equal(o, ref)
        t = min(res[v])
        print(f"ln_fwd_ab film={film} variant {v}: best {t:.2f} us ({R * D * 4 / t / 1e6:.2f} TB/s), rounds " + " ".join(f"{x:.2f}" for x in res[v])
              + f", bitwise equal to variant {variants[0]}: {same}")
lib.check(L.smd_set_tuning(b"ln_fwd_wide", 2))
