"""Two-chain sampler timeline from a rocprofv3 kernel trace (rocpd sqlite) of `bench.py --mode sample`:
per window of W reverse steps: step period of each chain, lag of chain B's step end behind chain A's (the PHASE between the two
free-running chains), and the average duration of the main kernels inside the window -- which of them stretch in the slow mode?
  python tools/chain_timeline.py OUT/t_results.db [W=20]"""
import re
import sqlite3
import sys
from collections import defaultdict

cur = sqlite3.connect(sys.argv[1]).cursor()
W = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcols = [c for c in ("stream_id", "queue_id") if c in cols]
rows = list(cur.execute(f"select name, start, end, {', '.join(qcols)} from kernels order by start"))
print(f"# {len(rows)} dispatches; id columns {qcols}")


def short(n):
    m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)(<[^()]*>)?\s*\(", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:50]


for qi, qc in enumerate(qcols):
    per = defaultdict(list)
    for r in rows:
        per[r[3 + qi]].append(r)
    print(f"# by {qc}: " + ", ".join(f"{k}: {len(v)}" for k, v in per.items()))
    chains = [k for k, v in per.items() if sum("reverse_step" in r[0] for r in v) > 3 * W]
    if len(chains) < 2:
        print(f"#   fewer than two {qc}s hold reverse steps; skipping")
        continue
    chains = sorted(chains, key=lambda k: -len(per[k]))[:2]
    ends = {c: [r[2] for r in per[c] if "reverse_step" in r[0]] for c in chains}
    A, B = chains
    n = min(len(ends[A]), len(ends[B]))
    print(f"# chains on {qc} {A} / {B}: {len(ends[A])} / {len(ends[B])} reverse steps")
    names = ["gemm_nt256", "mlp_hs_fwd", "attn_block_fwd", "layernorm_fwd_wide", "gemm_nt_kernel", "reverse_step", "ln128"]
    print("# window: period A, period B (us/step) | lag of B's step end behind A's (us, mod period) | avg us of " + ", ".join(names))
    import bisect
    for w0 in range(0, n - W, W):
        ta0, ta1 = ends[A][w0], ends[A][w0 + W]
        pA = (ta1 - ta0) / W / 1e3
        # B's steps inside the window
        i0 = bisect.bisect_left(ends[B], ta0)
        i1 = bisect.bisect_left(ends[B], ta1)
        pB = (ends[B][min(i1, len(ends[B]) - 1)] - ends[B][i0]) / max(i1 - i0, 1) / 1e3 if i1 > i0 else float("nan")
        lags = []
        for i in range(w0, w0 + W):
            j = bisect.bisect_left(ends[B], ends[A][i])
            if j < len(ends[B]):
                lags.append((ends[B][j] - ends[A][i]) / 1e3)
        lag = sum(lags) / max(len(lags), 1)
        durs = defaultdict(list)
        for c in chains:
            for r in per[c]:
                if ta0 <= r[1] < ta1:
                    for nm in names:
                        if nm in r[0]:
                            durs[nm].append((r[2] - r[1]) / 1e3)
                            break
        d = "  ".join(f"{(sum(durs[nm]) / len(durs[nm]) if durs[nm] else 0):6.1f}" for nm in names)
        print(f"  steps {w0:5d}..: A {pA:7.1f}  B {pB:7.1f} | lag {lag:7.1f} | {d}")
    break
