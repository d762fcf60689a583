"""Per-stream, per-kernel time of the train step from a rocprofv3 kernel trace (rocpd sqlite) of the default TWO-stream step:
the main stream is the critical path (tools/stream_busy.py: ~96 % busy), so its per-kernel microseconds are what a train step
costs; the side stream's kernels (weight gradients, FiLM generators) matter only through the CUs they take.
  python tools/stream_table.py <t_results.db>  ->  profiles/r6_train_main_stream.txt
A step = the window from one q_sample dispatch to the next; the first two windows are warm-up and dropped."""
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end, stream_id from kernels order by start"))
starts = [r for r in rows if "q_sample" in r[0]]
wins = list(zip(starts, starts[1:]))[2:]
nst = len(wins)


def short(n):
    n = n.strip()
    if n.startswith("void "):
        n = n[5:]
    n = n.replace("(anonymous namespace)::", "")
    if n.startswith("_Z"):             # an Itanium-mangled name the trace did not demangle: the <length><identifier> that ends in _kernel
        for m in re.finditer(r"(\d+)", n):
            ln, st = int(m.group(1)), m.end()
            while ln >= 100 and st > m.start():          # "N_122ln128..." reads as 122: drop leading digits that belong to the prefix
                ln = int(str(ln)[1:])
            if n[st:st + ln].endswith("_kernel"):
                return n[st:st + ln]
    depth, out = 0, []
    for ch in n:                       # up to the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)[-72:]


per = {}
wall = 0.0
busy = {}
for (s, n) in wins:
    t0, t1 = s[1], n[1]
    wall += (t1 - t0) / 1e3
    for r in rows:
        if t0 <= r[1] < t1:
            k = (r[3], short(r[0]))
            e = per.setdefault(k, [0.0, 0])
            e[0] += (r[2] - r[1]) / 1e3
            e[1] += 1
            busy[r[3]] = busy.get(r[3], 0.0) + (r[2] - r[1]) / 1e3
streams = sorted(busy, key=lambda q: -busy[q])
print(f"# {sys.argv[1]}: {nst} train steps, wall {wall / nst:.1f} us per step (under rocprofv3)")
for qi, q in enumerate(streams):
    role = "MAIN (critical path)" if qi == 0 else "side"
    print(f"\n## stream {q} -- {role}: {busy[q] / nst:.1f} us of kernels per step = {busy[q] / wall * 100:.1f} % of the wall time")
    print(f"{'us/step':>9} {'%stream':>8} {'calls':>6} {'avg us':>8}  kernel")
    for (qq, name), (us, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        if qq != q:
            continue
        print(f"{us / nst:9.1f} {us / busy[q] * 100:8.1f} {c / nst:6.1f} {us / c:8.1f}  {name}")
