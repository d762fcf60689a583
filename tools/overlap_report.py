"""Which kernels of the OTHER queue overlap in time with the launches of a given kernel (rocprofv3 rocpd sqlite trace)."""
import sqlite3, sys
db, pat = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute(f"select name, start, end, {qcol or '0'} from kernels order by start"))
t0 = rows[0][1]
targets = [r for r in rows if pat in r[0]]
print(f"{len(rows)} dispatches, {len(targets)} launches of *{pat}*; columns: {cols}")
for k, (name, s, e, q) in enumerate(targets[-int(sys.argv[3]) if len(sys.argv) > 3 else 0:]):
    ov = [(n.replace('(anonymous namespace)::', '').split('(')[0][-44:], max(s, s2), min(e, e2), q2) for n, s2, e2, q2 in rows
          if q2 != q and s2 < e and e2 > s]
    print(f"#{k} {name.split('(')[0][-40:]} q{q} [{(s - t0) / 1e3:.1f} .. {(e - t0) / 1e3:.1f} us] dur {(e - s) / 1e3:.1f} us; overlapping other-queue kernels:")
    for n, a, b, q2 in ov:
        print(f"      q{q2} {n}  overlap {(b - a) / 1e3:.1f} us")
