"""Aggregate rocprofv3 --pmc output (counter_collection CSV files under a directory) per kernel.

  python tools/pmc_summary.py DIR [substring-of-kernel-name]

Prints, per kernel name, the mean per-dispatch value of every collected counter and the number of
dispatches.  FETCH_SIZE is printed raw and doubled (MI355X_MICROARCH.md: on gfx950 rocprofv3 reports
half of the bytes of wide coalesced reads).
"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if pat and pat not in name:
                    continue
                acc[name][row["Counter_Name"]].append((row.get("Dispatch_Id"), float(row["Counter_Value"])))
    for name, ctrs in sorted(acc.items()):
        print(f"== {name[:110]}")
        for c, vals in sorted(ctrs.items()):
            per = collections.defaultdict(float)
            for did, v in vals:
                per[did] += v            # a counter may be reported per XCD / SE instance: sum per dispatch
            m = sum(per.values()) / len(per)
            extra = f"   (x2 = {2 * m:.4g})" if c == "FETCH_SIZE" else ""
            print(f"   {c:34s} mean/dispatch {m:16.6g}   dispatches {len(per)}{extra}")


if __name__ == "__main__":
    main()
