"""Concurrency summary of a rocprofv3 kernel trace (rocpd sqlite): total busy time (union of kernel intervals),
sum of kernel durations, and time during which >= 2 kernels were resident, per stream/queue.
  python tools/overlap.py OUT/NAME_results.db"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute(f"select start, end, name{', ' + qcol if qcol else ''} from kernels order by start"))
ev = []
for r in rows:
    ev.append((r[0], 1)); ev.append((r[1], -1))
ev.sort()
busy = over = 0
depth = 0
last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d
    last = t
tot = sum(r[1] - r[0] for r in rows)
print(f"kernels {len(rows)}  span {(rows[-1][1]-rows[0][0])/1e6:.2f} ms  busy(union) {busy/1e6:.2f} ms  sum {tot/1e6:.2f} ms  >=2 resident {over/1e6:.2f} ms")
if qcol:
    per = {}
    for r in rows:
        per.setdefault(r[3], [0, 0]); per[r[3]][0] += 1; per[r[3]][1] += r[1] - r[0]
    for q, (n, t) in per.items():
        print(f"  {qcol} {q}: {n} kernels, {t/1e6:.2f} ms")
