"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a text table.

  rocprofv3 --kernel-trace --stats -d OUT -o NAME -- python bench.py ...
  python tools/prof_summary.py OUT/NAME_results.db [steps] > profiles/<round>_<what>.txt
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    n = sum(r[1] for r in rows)
    print(f"# {path}: {n} dispatches, {tot / 1e3:.2f} ms kernel time, normalised per {steps:g} step(s)")
    print(f"# {'%':>5} {'us/step':>10} {'calls/step':>10} {'avg us':>9} {'min us':>8} {'max us':>9} {'vgpr':>5} {'agpr':>5} {'lds':>7}  kernel")
    for name, c, s, a, mn, mx, vg, ag, lds in rows:
        short = name.replace("(anonymous namespace)::", "").split("(")[0][-60:]
        print(f"  {s / tot * 100:5.1f} {s / steps:10.1f} {c / steps:10.1f} {a:9.1f} {mn:8.1f} {mx:9.1f} {vg or 0:5d} {ag or 0:5d} {lds or 0:7d}  {short}")
    print("\n# by (kernel, grid): the GEMM shapes")
    rows = list(cur.execute(
        "select name, grid_x/workgroup_x, grid_y, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels "
        "where name like '%gemm%' group by name, grid_x, grid_y order by 5 desc"))
    for name, gx, gy, c, s, a in rows[:30]:
        short = name.replace("(anonymous namespace)::", "").split("(")[0][-40:]
        print(f"  {s / steps:10.1f} us/step  calls/step {c / steps:6.1f}  avg {a:8.1f} us  grid ({gx},{gy})  {short}")


if __name__ == "__main__":
    main()
