"""Per-stream busy time and gaps of the train steps in a rocprofv3 kernel trace (rocpd sqlite): which queue is the
critical path?  A step = the window from one q_sample dispatch to the next (the optimiser of step i may still be running
on the side stream when step i + 1 starts: opt_overlap).  --timeline K prints every dispatch of window K (stream, start
offset, duration)."""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end, stream_id from kernels order by start"))
starts = [r for r in rows if "q_sample" in r[0]]
print(f"{len(rows)} dispatches, {len(starts)} train steps")
tl = int(sys.argv[sys.argv.index("--timeline") + 1]) if "--timeline" in sys.argv else None
wins = list(zip(starts, starts[1:]))[2:]                                  # skip the warm-up steps
for k, (s, n) in enumerate(wins):
    t0, t1 = s[1], n[1]
    inwin = [r for r in rows if r[1] >= t0 and r[1] < t1]
    per = {}
    for r in inwin:
        per.setdefault(r[3], []).append(r)
    line = [f"step {k}: wall {(t1 - t0) / 1e3:7.1f} us, {len(inwin)} kernels"]
    for q, rs in sorted(per.items()):
        rs.sort(key=lambda r: r[1])
        busy = sum(min(r[2], t1) - r[1] for r in rs) / 1e3
        gaps = sorted(((rs[i + 1][1] - rs[i][2]) / 1e3 for i in range(len(rs) - 1)), reverse=True)
        line.append(f"stream {q}: {len(rs):3d} kernels busy {busy:7.1f} us ({busy / ((t1 - t0) / 1e3) * 100:4.1f} %), "
                    f"largest gaps {' '.join('%.0f' % g for g in gaps[:3])} us")
    print(" | ".join(line))
    if tl is not None and k == tl:
        for r in inwin:
            import re
            m_ = re.search(r"([A-Za-z_][A-Za-z0-9_]*)(<[^()]*>)?\s*\(", r[0]) or re.search(r"([A-Za-z_][A-Za-z0-9_]*_kernel)", r[0])
            nm = (m_.group(1) if m_ else r[0])[-48:]
            print(f"    q{r[3]} +{(r[1] - t0) / 1e3:8.1f} us  {(r[2] - r[1]) / 1e3:7.1f} us  {nm}")
