"""Which KERNEL of the encoder backward first produces different bits between repeated identical loss_backward calls?

The engine's debug snapshots (smd_engine_debug_snapshots) copy the shared, per-layer overwritten gradient buffers behind
every encoder-layer kernel; this tool runs ITERS identical steps, compares every snapshot segment against iteration 0 in
execution order and reports the first segment that differs (and what the difference looks like).

  python tools/det_first_diff.py [T:knob=value ...] [engine_opt=value ...]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import smd_amd.ncsn as N
import smd_amd.schedule as S
from smd_amd.engine import NetConfig
import smd_amd.lib as lib

opts = dict(kv.split("=") for kv in sys.argv[1:] if not kv.startswith("T:"))
for kv in sys.argv[1:]:
    if kv.startswith("T:"):
        k, v = kv[2:].split("=")
        lib.check(lib.get_lib().smd_set_tuning(k.encode(), int(v)))
cfg = NetConfig(architecture="TransformerDDPM", data_channels=512, seq_len=32, num_timesteps=1000)
model = N.Model(cfg, "cuda:0", seed=0)
eng = model.train_engine(ema=False)
for k, v in opts.items():
    eng.set_option(k, int(v))
eng.set_schedule(S.create_noise_schedule(1e-6, 0.01, 1000, "linear"), with_sampler=False)
B = 256
eng.bind(B, training=True)
L = lib.get_lib()
nbytes = int(L.smd_engine_debug_snapshot_bytes(eng.h))
snap = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
lib.check(L.smd_engine_debug_snapshots(eng.h, snap.data_ptr(), nbytes))
g = torch.Generator().manual_seed(1)
x0 = torch.clamp(0.25 * torch.randn(B, 32, 512, generator=g), -1, 1).cuda()
labels = torch.randint(1, 1001, (B,), generator=g).int().cuda()
eps = torch.randn(B, 32, 512, generator=g).cuda()
n = int(os.environ.get("ITERS", "30"))
RE = B * 32 * 128
U = 34
NL = cfg.num_layers
SEG = [("mlp_hs_bwd -> da2 partial tiles", 0, 16, torch.float32), ("ln128_bwd_parts (ln2) -> dh", 16, 4, torch.float32),
       ("attn_block_bwd -> dA_E", 20, 2, torch.bfloat16), ("layernorm_bwd_narrow128 (ln1) -> dh", 22, 4, torch.float32),
       ("incoming dh", 26, 4, torch.float32), ("saved h_mid", 30, 4, torch.float32)]
ORDER = [4, 5, 0, 1, 2, 3]          # execution order of the segments inside a layer


def seg_view(buf, slot, seg):
    name, off, ln, dt = SEG[seg]
    b = buf[(slot * U + off) * RE:(slot * U + off + ln) * RE]
    return b.view(dt)


ref = None
SAVED = []
first_hist = {}
print("options", sys.argv[1:])
for it in range(n):
    eng.loss_backward(x0, labels, eps, stage=0)
    torch.cuda.synchronize()
    cur = snap.clone()
    if ref is None:
        ref = cur
        continue
    if torch.equal(cur, ref):
        continue
    for slot in range(NL):
        hit = None
        for seg in ORDER:
            a, b = seg_view(cur, slot, seg), seg_view(ref, slot, seg)
            if not torch.equal(a, b):
                hit = seg
                break
        if hit is not None:
            layer = NL - 1 - slot
            key = (layer, SEG[hit][0])
            first_hist[key] = first_hist.get(key, 0) + 1
            if first_hist[key] <= 6:
                a, b = seg_view(cur, slot, hit).float(), seg_view(ref, slot, hit).float()
                if hit == 0:
                    a, b = a.view(4, B * 32, 128), b.view(4, B * 32, 128)
                    d = (a != b)
                    q = d.any(2).any(1).nonzero().flatten().tolist()
                    rows = d.any(2).any(0).nonzero().flatten()
                    cols = d.any(0).any(0).nonzero().flatten()
                    print(f"  it {it}: first diff layer {layer} {SEG[hit][0]}: quarters {q}, {int(d.sum())} elements in {len(rows)} rows "
                          f"(first {rows[:8].tolist()}; row%32 {sorted(set((rows % 32).tolist()))[:16]}), cols {cols[:16].tolist()}..{int(cols[-1])} "
                          f"max abs diff {float((a - b).abs().max()):.3e} vs max abs {float(b.abs().max()):.3e}")
                elif hit == 1:
                    a, b = a.view(B * 32, 128), b.view(B * 32, 128)
                    d = (a != b)
                    rows = d.any(1).nonzero().flatten()
                    print(f"  it {it}: first diff layer {layer} {SEG[hit][0]}: {int(d.sum())} elements in {len(rows)} rows {rows[:8].tolist()} "
                          f"(row%32 {[int(r) % 32 for r in rows[:8]]}) max abs diff {float((a - b).abs().max()):.3e} vs max abs {float(b.abs().max()):.3e}")
                    # which of the two is right?  fp64 recomputation of the ln2 backward of that row from the snapshotted inputs
                    gam = eng.named_views()[f"enc.{layer}.ln2.scale"].double()
                    for r in rows[:3].tolist():
                        def ln_bwd(buf, dres_row):
                            x = seg_view(buf, slot, 5).view(B * 32, 128)[r].double()
                            p = seg_view(buf, slot, 0).view(4, B * 32, 128)[:, r].double()
                            dv = (p[0] + p[1]) + (p[2] + p[3])
                            mean, var = x.mean(), (x * x).mean() - x.mean() ** 2
                            rs = (var + 1e-6).rsqrt()
                            xh = (x - mean) * rs
                            dxh = dv * gam
                            t1, t2 = dxh.mean(), (dxh * xh).mean()
                            return rs * (dxh - t1 - xh * t2) + dres_row.double()
                        SAVED.append(dict(layer=layer, row=r, it=it, x=seg_view(cur, slot, 5).view(B * 32, 128)[r].cpu(),
                                          parts=seg_view(cur, slot, 0).view(4, B * 32, 128)[:, r].cpu(), dres=seg_view(cur, slot, 4).view(B * 32, 128)[r].cpu(),
                                          out_it=a[r].cpu(), out_ref=b[r].cpu(), gamma=gam.float().cpu()))
                        dres_cur = seg_view(cur, slot, 4).view(B * 32, 128)[r]
                        dres_ref = seg_view(ref, slot, 4).view(B * 32, 128)[r]
                        exp_cur, exp_ref = ln_bwd(cur, dres_cur), ln_bwd(ref, dres_ref)
                        e = lambda got, exp: float((got.double() - exp).norm() / exp.norm())
                        print(f"     row {r}: |it{it} - fp64(it{it} inputs)| = {e(a[r], exp_cur):.2e}   |it0 - fp64(it0 inputs)| = {e(b[r], exp_ref):.2e}   "
                              f"inputs equal: dh_in {torch.equal(dres_cur, dres_ref)}")
                        # hypotheses for the wrong one: the incoming dh row was a STALE version (any later state of the previous step)
                        wrong, good_in = (a[r], cur) if e(a[r], exp_cur) > e(b[r], exp_ref) else (b[r], ref)
                        resid = wrong.double() - (ln_bwd(good_in, torch.zeros(128, device="cuda")))      # = the dres it actually used
                        cands = {}
                        for sl2 in range(NL):
                            for sg in (1, 3, 4):
                                cands[f"prev-step layer {NL - 1 - sl2} {SEG[sg][0]}"] = seg_view(ref, sl2, sg).view(B * 32, 128)[r].double()
                        best = sorted(((float((resid - v).norm() / (v.norm() + 1e-30)), k) for k, v in cands.items()))[:3]
                        print(f"     the dres the wrong result used, vs candidate states of that dh row: {[(f'{x:.1e}', k) for x, k in best]}")
                else:
                    a, b = a.view(B * 32, 128), b.view(B * 32, 128)
                    d = (a != b)
                    rows = d.any(1).nonzero().flatten()
                    per = d.sum(1)[rows][:8].tolist()
                    print(f"  it {it}: first diff layer {layer} {SEG[hit][0]}: {int(d.sum())} elements in {len(rows)} rows (first {rows[:8].tolist()}, "
                          f"row%4 {sorted(set((rows % 4).tolist()))}, per-row counts {per}) max abs diff {float((a - b).abs().max()):.3e} "
                          f"vs max abs {float(b.abs().max()):.3e}")
            break
nd = sum(first_hist.values())
print(f"{nd} of {n - 1} repeats differ from the first; first differing kernel histogram:")
for k, v in sorted(first_hist.items(), key=lambda kv: -kv[1]):
    print(f"   layer {k[0]}  {k[1]}: {v}")
if SAVED and os.environ.get("SAVE_ROWS"):
    torch.save(SAVED, os.environ["SAVE_ROWS"])
