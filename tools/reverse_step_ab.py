"""Interleaved A/B of the fused reverse step (smd_ddpm_reverse_step, Philox noise, metrics) between csrc/libsmd_hip_old.so and
the shipped library in one process: B = 256 and 128 sequences of (32, 512), C = 146, and DenseDDPM's (B, 512) states; the
resulting states and metrics compared bit for bit.  python tools/reverse_step_ab.py"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import smd_amd.lib as lib
import smd_amd.schedule as S
new = lib.get_lib()
old = C.CDLL(os.path.join(ROOT, "symbolic-music-diffusion_amd", "csrc", "libsmd_hip_old.so"))
for name, (res, args) in lib._SIGS.items():
    if hasattr(old, name):
        fn = getattr(old, name)
        fn.restype, fn.argtypes = res, args
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
T = 1000
betas = S.create_noise_schedule(1e-6, 0.01, T, "linear")
coef = torch.from_numpy(S.reverse_coefficient_table(np.asarray(betas, dtype=np.float32))).to(dev)
g = torch.Generator().manual_seed(0)
for (B, Sq, Cc) in [(256, 32, 512), (128, 32, 512), (256, 32, 146), (4096, 1, 512)]:
    x0 = torch.randn(B, Sq, Cc, generator=g).to(dev)
    eh = torch.randn(B, Sq, Cc, generator=g).to(dev)
    t = torch.full((1,), 500, dtype=torch.int32, device=dev)
    xs = {k: x0.clone() for k in ("old", "new")}
    met = {k: torch.zeros(T, B, 3, device=dev) for k in ("old", "new")}

    def call(L, k):
        rc = L.smd_ddpm_reverse_step(P(xs[k]), P(eh), B, Sq, Cc, P(coef), T, P(t), None, 11, 22, 0, P(met[k]), None, None, st)
        assert rc == 0, new.smd_last_error()

    def timeit(L, k, reps=50):
        for _ in range(5):
            call(L, k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call(L, k)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    res = {"old": [], "new": []}
    for rnd in range(7):
        for k, L in (("old", old), ("new", new)):
            res[k].append(timeit(L, k))
    for k, L in (("old", old), ("new", new)):
        xs[k].copy_(x0)
        met[k].zero_()
        call(L, k)
    torch.cuda.synchronize()
    same = torch.equal(xs["old"], xs["new"]) and torch.equal(met["old"], met["new"])
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    mb = B * Sq * Cc * 4 * 3 / 1e6
    print(f"reverse_step_ab B={B} S={Sq} C={Cc}: old {med['old']:.2f} us  new {med['new']:.2f} us ({mb / med['new'] / 1e6 * 1e6 / 1e3:.2f} TB/s of {mb:.0f} MB)  "
          f"{(med['new'] / med['old'] - 1) * 100:+.1f} %   bitwise equal: {same}")
