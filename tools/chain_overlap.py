"""What are the two pipelined sampling chains doing at the same time?  From a rocprofv3 kernel trace (rocpd sqlite) of
`bench.py --mode sample`: every instant of the timed region is classified by the kernel class running on each chain's stream
(G = 256^2 GEMM, O = out_proj / small GEMMs, E = encoder half-layer kernels, L = wide LayerNorm, R = reverse step, - = nothing) and
the joint classes are summed -- how much of a step has both chains in their 2048-wide GEMMs, both in their encoders, one idle ...
  python tools/chain_overlap.py OUT/t_results.db"""
import sqlite3
import sys
from collections import defaultdict

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end, stream_id from kernels order by start"))
per = defaultdict(list)
for r in rows:
    per[r[3]].append(r)
chains = sorted((k for k, v in per.items() if sum("reverse_step" in r[0] for r in v) > 50), key=lambda k: -len(per[k]))[:2]
if len(chains) < 2:
    raise SystemExit(f"fewer than two streams with reverse steps: { {k: len(v) for k, v in per.items()} }")


def cls(name):
    if "gemm_nt256" in name: return "G"
    if "gemm_nt_kernel" in name: return "O"
    if "attn_block" in name or "mlp_hs" in name or "ln128" in name or "pos_encoding" in name: return "E"
    if "layernorm" in name: return "L"
    if "reverse_step" in name: return "R"
    return "x"


# the steady part: from the 40th to the last-but-40th reverse step of chain 0
ends = [r[2] for r in per[chains[0]] if "reverse_step" in r[0]]
t_lo, t_hi = ends[40], ends[-40]
ev = []
for ci, c in enumerate(chains):
    for name, s, e, _ in per[c]:
        if e <= t_lo or s >= t_hi:
            continue
        k = cls(name)
        ev.append((max(s, t_lo), ci, k))
        ev.append((min(e, t_hi), ci, None))
ev.sort(key=lambda x: (x[0], x[2] is not None))
state = ["-", "-"]
last = t_lo
acc = defaultdict(float)
for t, ci, k in ev:
    acc["".join(state)] += t - last
    last = t
    state[ci] = "-" if k is None else k
tot = sum(acc.values())
nsteps = sum(1 for e in ends if t_lo < e <= t_hi)
print(f"# {nsteps} steps of chain A in the window, {tot / nsteps / 1e3:.1f} us per step; joint kernel classes (chain A, chain B), share of the time and us per step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    if v / tot > 0.004:
        print(f"  {k}   {v / tot * 100:5.1f} %   {v / nsteps / 1e3:7.1f} us")
one_idle = sum(v for k, v in acc.items() if "-" in k and k != "--")
print(f"# both busy {100 - (one_idle + acc['--']) / tot * 100:.1f} %, one chain idle {one_idle / tot * 100:.1f} %, both idle {acc['--'] / tot * 100:.1f} %")
print(f"# both in a 256^2 GEMM {acc['GG'] / tot * 100:.1f} %; both in encoder kernels {acc['EE'] / tot * 100:.1f} %; GEMM beside encoder {(acc['GE'] + acc['EG']) / tot * 100:.1f} %")
