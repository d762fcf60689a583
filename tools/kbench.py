"""Per-kernel micro-benchmark through the C-ABI (HIP events on the launch stream).

  python tools/kbench.py [--reps 30]

Prints one line per kernel/shape: average launch time, algorithmic TFLOP/s or GB/s, fraction of the
MI355X peak (2.5 PFLOP/s dense bf16, 8 TB/s HBM).  Shapes are the ones of ddpm-mel-32seq-512 at
batch 256 (R = 8192 rows).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import smd_amd.lib as lib  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--json", default="")
    ap.add_argument("--variants", type=lambda t: [int(x) for x in t.split(",")], default=[0, 1])
    ap.add_argument("--shapes", default="8192x2048x2048")
    ap.add_argument("--mlp", action="store_true", help="fused encoder MLP half-layer")
    ap.add_argument("--kg-ab", action="store_true", help="A/B of the two-K-group skinny NT GEMM")
    ap.add_argument("--deep-ab", action="store_true", help="A/B of the 2- vs 4-stage skinny NT GEMM")
    ap.add_argument("--ln-ab", action="store_true", help="interleaved A/B of the LayerNorm backward kernels")
    ap.add_argument("--tn128", action="store_true", help="split-K sweep of the 128-wide wgrad kernel")
    ap.add_argument("--tn-ab", action="store_true", help="interleaved A/B of the wgrad kernels")
    ap.add_argument("--gemm-ab", action="store_true", help="only the interleaved A/B of the NT GEMM kernels/variants")
    a = ap.parse_args()
    L = lib.get_lib()
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    rows = []

    def rec(name, shape, ms, flops=None, bytes_=None):
        r = {"kernel": name, "shape": shape, "ms": round(ms, 5)}
        if flops:
            r["tflops"] = round(flops / ms / 1e9, 1)
            r["frac_mfma"] = round(flops / ms / 1e9 / 2500.0, 4)
        if bytes_:
            r["gbps"] = round(bytes_ / ms / 1e6, 1)
            r["frac_hbm"] = round(bytes_ / ms / 1e6 / 8000.0, 4)
        rows.append(r)
        print(json.dumps(r))

    R = 8192
    bf = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
    if a.mlp:
        M = 2048
        h = torch.randn(R, 128, device=dev)
        g, b = torch.ones(128, device=dev), torch.zeros(128, device=dev)
        W1t, W2t = bf(M, 128), bf(128, M)
        b1, b2 = torch.zeros(M, device=dev), torch.zeros(128, device=dev)
        out = torch.empty(R, 128, device=dev)
        sa = torch.empty(R, 128, dtype=torch.bfloat16, device=dev)
        sz, su = torch.empty(R, M, dtype=torch.bfloat16, device=dev), torch.empty(R, M, dtype=torch.bfloat16, device=dev)
        for save, var in ((0, 0), (1, 0), (0, 3), (1, 3)):
            lib.check(L.smd_set_tuning(b"mlp_variant", var))
            f = lambda: lib.check(L.smd_mlp_block_fwd(h.data_ptr(), out.data_ptr(), R, g.data_ptr(), b.data_ptr(), W1t.data_ptr(),
                                                      b1.data_ptr(), W2t.data_ptr(), b2.data_ptr(), M,
                                                      sa.data_ptr() if save else None, sz.data_ptr() if save else None,
                                                      su.data_ptr() if save else None, st))
            rec(f"mlp_block_fwd(save={save},ablate={var})", [R, 128, M], timeit(f, a.reps), flops=4.0 * R * 128 * M)
        lib.check(L.smd_set_tuning(b"mlp_variant", 0))
        return
    if a.kg_ab:
        for (M, N, K) in [(R, 128, 2048), (R, 128, 1024), (R, 42, 2048), (2048, 128, 2048)]:
            A, Bt = bf(M, K), bf(N, K)
            bias = torch.zeros(N, device=dev)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            res32, out32 = torch.randn(M, N, device=dev), torch.empty(M, N, device=dev)
            fb = lambda: lib.check(L.smd_gemm_bf16_nt(A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), 0, None, 0,
                                                      None, 0, out.data_ptr(), N, st))
            fr = lambda: lib.check(L.smd_gemm_bf16_nt(A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), 0,
                                                      res32.data_ptr(), N, out32.data_ptr(), N, None, 0, st))
            for name, f in (("b", fb), ("r", fr)):
                modes = {0: "1 group x 4 stages", 1: "2 K-groups x 3 stages"}
                res = {m: [] for m in modes}
                for _ in range(5):
                    for mode in modes:
                        lib.check(L.smd_set_tuning(b"gemm_nt_kg", mode))
                        res[mode].append(timeit(f, a.reps))
                for mode in modes:
                    rec(f"kg_ab:{modes[mode]}:{name}", [M, N, K], sorted(res[mode])[2],
                        flops=2.0 * M * N * K, bytes_=2.0 * (M * K + N * K + M * N))
        lib.check(L.smd_set_tuning(b"gemm_nt_kg", 1))
        return
    if a.deep_ab:
        for (M, N, K) in [(R, 128, 2048), (R, 128, 512), (R, 384, 128), (R, 128, 128), (R, 512, 2048), (256, 4096, 512)]:
            A, Bt = bf(M, K), bf(N, K)
            bias = torch.zeros(N, device=dev)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            f = lambda: lib.check(L.smd_gemm_bf16_nt(A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), 0, None, 0,
                                                     None, 0, out.data_ptr(), N, st))
            res = {0: [], 1: []}
            for _ in range(5):
                for mode in (0, 1):
                    lib.check(L.smd_set_tuning(b"gemm_nt_deep", mode))
                    res[mode].append(timeit(f, a.reps))
            for mode in (0, 1):
                rec(f"deep_ab:{'4-stage' if mode else '2-stage'}", [M, N, K], sorted(res[mode])[2], flops=2.0 * M * N * K,
                    bytes_=2.0 * (M * K + N * K + M * N))
        lib.check(L.smd_set_tuning(b"gemm_nt_deep", 1))
        return
    if a.ln_ab:
        # the engine's forms of the D = 2048 LayerNorm backward: (name, x bf16, residual gradient, outputs, FiLM+swish)
        D = 2048
        forms = [("res.ln2  (bf16 x, bf16 dx)", 1, None, 2, 1),
                 ("res.ln1  (f32 x, f32 dres in place, f32+bf16 dx)", 0, "f32", 3, 1),
                 ("res.ln1  (f32 x, bf16 dres, bf16 dx)", 0, "bf16", 2, 1),
                 ("ln_o     (f32 x, f32+bf16 dx)", 0, None, 3, 0),
                 ("ln_o     (f32 x, bf16 dx)", 0, None, 2, 0)]
        x = torch.randn(R, D, device=dev)
        xbf = x.to(torch.bfloat16)
        g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
        ss = torch.randn(R // 32, 2 * D, device=dev)
        dout, dresb = bf(R, D), bf(R, D)
        dx, dxb = torch.zeros(R, D, device=dev), torch.empty(R, D, dtype=torch.bfloat16, device=dev)
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        dss = torch.zeros(R // 32, 2 * D, device=dev)
        part = torch.empty(R * 2 * D // 16, device=dev)
        for name, xb, res, om, fs in forms:
            nbytes = R * D * ((2 if xb else 4) + 2 + {None: 0, "f32": 4, "bf16": 2}[res] + (4 if om & 1 else 0) + (2 if om & 2 else 0))
            f = lambda: lib.check(L.smd_layernorm_bwd_film(
                None if xb else x.data_ptr(), xbf.data_ptr() if xb else None, R, D, g.data_ptr(), b.data_ptr(),
                ss.data_ptr() if fs else None, ss[:, D:].data_ptr() if fs else None, 2 * D, 32, fs, dout.data_ptr(),
                dx.data_ptr() if res == "f32" else None, dresb.data_ptr() if res == "bf16" else None,
                dx.data_ptr() if om & 1 else None, dxb.data_ptr() if om & 2 else None, dg.data_ptr(), db.data_ptr(),
                dss.data_ptr() if fs else None, dss[:, D:].data_ptr() if fs else None, 0, part.data_ptr(), part.numel(), st))
            modes = (2,) if res == "bf16" else (1, 2)
            out = {m: [] for m in modes}
            for _ in range(5):
                for m in modes:
                    lib.check(L.smd_set_tuning(b"ln_bwd_wide", m))
                    out[m].append(timeit(f, a.reps))
            for m in modes:
                rec(f"ln_bwd[{'8-wave' if m == 2 else '4-wave'}] {name} +reduce", [R, D], sorted(out[m])[2], bytes_=float(nbytes))
        lib.check(L.smd_set_tuning(b"ln_bwd_wide", 2))
        return
    if a.tn128:
        lib.check(L.smd_set_tuning(b"gemm_tn256", 0))
        for (M, Kd, N) in [(R, 128, 2048), (R, 2048, 128), (R, 128, 384), (R, 128, 128), (R, 512, 128), (R, 2048, 512)]:
            X, Y = bf(M, Kd), bf(M, N)
            out = torch.empty(Kd, N, device=dev)
            db = torch.empty(N, device=dev)
            zero = torch.zeros(128, dtype=torch.bfloat16, device=dev)
            slab = torch.empty(int(L.smd_gemm_tn_slab_elems()), device=dev)
            scratch = torch.zeros(128, dtype=torch.bfloat16, device=dev)
            f = lambda: lib.check(L.smd_gemm_bf16_tn(X.data_ptr(), Kd, Y.data_ptr(), N, M, Kd, N, out.data_ptr(), N,
                                                     db.data_ptr(), zero.data_ptr(), slab.data_ptr(), slab.numel(),
                                                     scratch.data_ptr(), scratch.numel(), 1, st))
            for tgt in (128, 256, 512):
                for deep in (0, 1):
                    lib.check(L.smd_set_tuning(b"tn128_target_wgs", tgt))
                    lib.check(L.smd_set_tuning(b"gemm_nt_deep", deep))
                    rec(f"tn128(target_wgs={tgt},deep={deep}) +reduce", [M, Kd, N], timeit(f, a.reps), flops=2.0 * M * Kd * N)
        lib.check(L.smd_set_tuning(b"gemm_nt_deep", 1))
        lib.check(L.smd_set_tuning(b"gemm_tn256", 1))
        lib.check(L.smd_set_tuning(b"tn128_target_wgs", 512))
        return
    if a.tn_ab:
        for (M, Kd, N) in [(R, 2048, 2048), (R, 2048, 512), (4 * R, 2048, 2048)]:
            X, Y = bf(M, Kd), bf(M, N)
            out = torch.empty(Kd, N, device=dev)
            db = torch.empty(N, device=dev)
            zero = torch.zeros(128, dtype=torch.bfloat16, device=dev)
            slab = torch.empty(int(L.smd_gemm_tn_slab_elems()), device=dev)
            scratch = torch.zeros(128, dtype=torch.bfloat16, device=dev)
            f = lambda: lib.check(L.smd_gemm_bf16_tn(X.data_ptr(), Kd, Y.data_ptr(), N, M, Kd, N, out.data_ptr(), N,
                                                     db.data_ptr(), zero.data_ptr(), slab.data_ptr(), slab.numel(),
                                                     scratch.data_ptr(), scratch.numel(), 1, st))
            res = {0: [], 1: []}
            for _ in range(5):
                for mode in (0, 1):
                    lib.check(L.smd_set_tuning(b"gemm_tn256", mode))
                    res[mode].append(timeit(f, a.reps))
            for mode in (0, 1):
                ms = sorted(res[mode])
                rec(f"tn_ab:{'tn256' if mode else 'tn128'} (+slab reduce)", [M, Kd, N], ms[2], flops=2.0 * M * Kd * N)
        lib.check(L.smd_set_tuning(b"gemm_tn256", 1))
        return
    if a.gemm_ab:
        # within-process interleaved rounds (guide 5.4 rule 24): 128-wide kernel vs the 256^2 8-phase variants
        arms = [("nt128", 0, 0)] + [(f"nt256/v{v}", 1, v) for v in a.variants]
        for spec in a.shapes.split(","):
            dims, _, epi = spec.partition(":")
            epi = epi or "b"          # b: bias -> bf16 ; r: bias + fp32 residual -> fp32 ; g: bias + gelu -> bf16
            M, N, K = (int(v) for v in dims.split("x"))
            A, Bt = bf(M, K), bf(N, K)
            bias = torch.zeros(N, device=dev)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            res32 = torch.randn(M, N, device=dev) if epi == "r" else None
            out32 = torch.empty(M, N, device=dev) if epi == "r" else None
            if epi == "r":
                f = lambda: lib.check(L.smd_gemm_bf16_nt(A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), 0,
                                                         res32.data_ptr(), N, out32.data_ptr(), N, None, 0, st))
            else:
                f = lambda: lib.check(L.smd_gemm_bf16_nt(A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(),
                                                         1 if epi == "g" else 0, None, 0, None, 0, out.data_ptr(), N, st))
            res = {n: [] for n, _, _ in arms}
            for _ in range(5):
                for n, mode, var in arms:
                    lib.check(L.smd_set_tuning(b"gemm_nt256", 2 if mode else 0))
                    lib.check(L.smd_set_tuning(b"gemm_nt256_variant", var))
                    res[n].append(timeit(f, a.reps))
            for n, _, _ in arms:
                ms = sorted(res[n])
                rec(f"gemm_ab:{n}:{epi}", [M, N, K], ms[len(ms) // 2], flops=2.0 * M * N * K)
                rows[-1]["min_ms"] = round(ms[0], 5)
        lib.check(L.smd_set_tuning(b"gemm_nt256", 1))
        lib.check(L.smd_set_tuning(b"gemm_nt256_variant", 0))
        if a.json:
            json.dump(rows, open(a.json, "w"), indent=1)
        return
    # ---- NT GEMMs (forward + dgrad shapes)
    for (M, N, K) in [(R, 2048, 2048), (R, 2048, 128), (R, 128, 2048), (R, 384, 128), (R, 128, 128), (R, 128, 512),
                      (R, 512, 2048), (256, 4096, 512), (1000, 4096, 512)]:
        A, Bt = bf(M, K), bf(N, K)
        bias = torch.zeros(N, device=dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        f = lambda: lib.check(L.smd_gemm_bf16_nt(A.data_ptr(), K, Bt.data_ptr(), K, M, N, K, bias.data_ptr(), 0, None, 0,
                                                 None, 0, out.data_ptr(), N, st))
        rec("gemm_nt", [M, N, K], timeit(f, a.reps), flops=2.0 * M * N * K)
    # ---- TN GEMMs (wgrad shapes), both paths
    for tr in (1, 0):
        for (M, Kd, N) in [(R, 2048, 2048), (R, 128, 2048), (R, 2048, 128), (R, 128, 384), (R, 128, 128), (256, 512, 4096)]:
            X, Y = bf(M, Kd), bf(M, N)
            out = torch.empty(Kd, N, device=dev)
            db = torch.empty(N, device=dev)
            zero = torch.zeros(128, dtype=torch.bfloat16, device=dev)
            slab = torch.empty(int(L.smd_gemm_tn_slab_elems()), device=dev)
            scratch = torch.zeros(max(128, (Kd + N) * M if not tr else 128), dtype=torch.bfloat16, device=dev)
            f = lambda: lib.check(L.smd_gemm_bf16_tn(X.data_ptr(), Kd, Y.data_ptr(), N, M, Kd, N, out.data_ptr(), N,
                                                     db.data_ptr(), zero.data_ptr(), slab.data_ptr(), slab.numel(),
                                                     scratch.data_ptr(), scratch.numel(), tr, st))
            rec(f"gemm_tn(tr={tr})", [M, Kd, N], timeit(f, a.reps), flops=2.0 * M * Kd * N)
    # ---- LayerNorm
    for D, film in ((128, 0), (2048, 0), (2048, 1)):
        x = torch.randn(R, D, device=dev)
        g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
        ss = torch.randn(R // 32, 2 * D, device=dev)
        out = torch.empty(R, D, dtype=torch.bfloat16, device=dev)
        f = lambda: lib.check(L.smd_layernorm_fwd(x.data_ptr(), R, D, g.data_ptr(), b.data_ptr(),
                                                  ss.data_ptr() if film else None, ss[:, D:].data_ptr() if film else None,
                                                  2 * D, 32, film, out.data_ptr(), st))
        rec(f"layernorm_fwd(film={film})", [R, D], timeit(f, a.reps), bytes_=R * D * 6.0)
        dout = bf(R, D)
        dx = torch.empty(R, D, device=dev)
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        dss = torch.zeros(R // 32, 2 * D, device=dev)
        part = torch.empty(R * 2 * D // 16, device=dev)
        f = lambda: lib.check(L.smd_layernorm_bwd(x.data_ptr(), R, D, g.data_ptr(), b.data_ptr(),
                                                  ss.data_ptr() if film else None, ss[:, D:].data_ptr() if film else None,
                                                  2 * D, 32, film, dout.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                                  dss.data_ptr() if film else None, dss[:, D:].data_ptr() if film else None,
                                                  part.data_ptr(), part.numel(), st))
        rec(f"layernorm_bwd(film={film})", [R, D], timeit(f, a.reps), bytes_=R * D * 10.0)
    # ---- attention
    for H in (8, 16):
        qkv = bf(R, 384)
        o = torch.empty(R, 128, dtype=torch.bfloat16, device=dev)
        dq = torch.empty(R, 384, dtype=torch.bfloat16, device=dev)
        f = lambda: lib.check(L.smd_attention_fwd(qkv.data_ptr(), o.data_ptr(), R // 32, 32, 128, H, st))
        rec(f"attention_fwd(H={H})", [R // 32, 32, 128], timeit(f, a.reps), bytes_=R * 512 * 2.0)
        f = lambda: lib.check(L.smd_attention_bwd(qkv.data_ptr(), o.data_ptr(), dq.data_ptr(), R // 32, 32, 128, H, st))
        rec(f"attention_bwd(H={H})", [R // 32, 32, 128], timeit(f, a.reps), bytes_=R * 1280 * 2.0)
    # ---- reverse step
    import smd_amd.schedule as S
    coef = torch.from_numpy(S.reverse_coefficient_table(S.create_noise_schedule(1e-6, 0.01, 1000, "linear"))).to(dev)
    x, eh = torch.randn(256, 32, 512, device=dev), torch.randn(256, 32, 512, device=dev)
    tp = torch.tensor([500], dtype=torch.int32, device=dev)
    f = lambda: lib.check(L.smd_ddpm_reverse_step(x.data_ptr(), eh.data_ptr(), 256, 32, 512, coef.data_ptr(), 1000, tp.data_ptr(),
                                                  None, 1, 2, 0, None, None, None, st))
    rec("ddpm_reverse_step(philox)", [256, 32, 512], timeit(f, a.reps), bytes_=256 * 32 * 512 * 12.0)
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
