"""Per-stream busy time and gaps of the train steps in a rocprofv3 kernel trace (rocpd sqlite): which queue is the
critical path?  A step = first q_sample dispatch .. end of the following adam_clip_ema dispatch."""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end, stream_id from kernels order by start"))
starts = [r for r in rows if "q_sample" in r[0]]
ends = [r for r in rows if "adam_clip" in r[0]]
print(f"{len(rows)} dispatches, {len(starts)} train steps")
for k, (s, a) in enumerate(list(zip(starts, ends))[2:]):                 # skip the warm-up steps
    t0, t1 = s[1], a[2]
    inwin = [r for r in rows if r[1] >= t0 and r[2] <= t1]
    per = {}
    for r in inwin:
        per.setdefault(r[3], []).append(r)
    line = [f"step {k}: wall {(t1 - t0) / 1e3:7.1f} us, {len(inwin)} kernels"]
    for q, rs in sorted(per.items()):
        rs.sort(key=lambda r: r[1])
        busy = sum(r[2] - r[1] for r in rs) / 1e3
        gaps = sorted(((rs[i + 1][1] - rs[i][2]) / 1e3 for i in range(len(rs) - 1)), reverse=True)
        line.append(f"stream {q}: {len(rs):3d} kernels busy {busy:7.1f} us ({busy / ((t1 - t0) / 1e3) * 100:4.1f} %), "
                    f"largest gaps {' '.join('%.0f' % g for g in gaps[:3])} us")
    print(" | ".join(line))
