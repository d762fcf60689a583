"""profiles/pmc_gemm_nt256.json from ONE round's own measurements (the file bench.py quotes must not mix rounds):

  python tools/make_pmc_json.py <tag>_pmc_gemm.txt <train_trace.db> <sample_trace.db> <tag> > <tag>_pmc_gemm_nt256.json

* the PMC passes of tools/collect_profiles.sh (rocprofv3 --pmc, one group per pass, on `kbench --gemm-ab --shapes
  8192x2048x2048:b`: the bias -> bf16 form bench.py's roofline times) -> FETCH_SIZE (doubled: MI355X_MICROARCH.md, gfx950
  reports 64 B per 128-B request of wide coalesced reads), WRITE_SIZE, MFMA busy;
* in-step duration of the FULL-SIZE launches of the kernel (> 40 us: the 8192 x 2048 x 2048 ones; the K = 128 / 512 launches of
  the same kernel are left out) from the per-mode kernel traces of the same run.
"""
import json
import re
import sqlite3
import sys


def pmc(path, kernel="gemm_nt256_kernel<0, false>"):
    out, on = {}, False
    for line in open(path):
        if line.startswith("=="):
            on = kernel in line
            continue
        m = re.match(r"\s+(\S+)\s+mean/dispatch\s+(\S+)\s+dispatches\s+(\d+)", line)
        if on and m:
            out[m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return out


def in_step(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select (end-start)/1e3 from kernels where name like '%gemm_nt256_kernel<0, false>%'"))
    full = [r[0] for r in rows if r[0] > 40.0]
    return (round(sum(full) / len(full), 1), len(full), len(rows)) if full else (None, 0, len(rows))


def main():
    pmc_txt, db_train, db_sample, tag = sys.argv[1:5]
    c = pmc(pmc_txt)
    fetch_raw_kib = c["FETCH_SIZE"][0]
    write_kib = c["WRITE_SIZE"][0]
    fetch = int(fetch_raw_kib * 1024 * 2)
    write = int(write_kib * 1024)
    tr, ntr, alltr = in_step(db_train)
    sm, nsm, allsm = in_step(db_sample)
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[0]
    gui = c.get("GRBM_GUI_ACTIVE", (0, 0))[0]
    out = {
        "kernel": "gemm_nt256_kernel<0, false>", "shape": [8192, 2048, 2048], "epilogue": "bias -> bf16",
        "collected": f"{tag}: rocprofv3 --pmc passes of tools/collect_profiles.sh (profiles/{tag}_pmc_gemm.txt), "
                     f"traces profiles/{tag}_train_single_stream_kernel_trace.txt / {tag}_sample_kernel_trace.txt",
        "FETCH_SIZE_KiB_raw": fetch_raw_kib, "WRITE_SIZE_KiB": write_kib,
        "fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes": fetch + write,
        "algorithmic_bytes": 8192 * 2048 * 2 + 2048 * 2048 * 2 + 8192 * 2048 * 2,
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 rocprofv3 FETCH_SIZE counts 64 B per 128-B request of wide "
                      "coalesced reads -> doubled; WRITE_SIZE uncorrected. Infinity-Cache hits are counted: an upper bound of HBM bytes.",
        "dispatches": c["FETCH_SIZE"][1],
        "other_counters_per_dispatch": {k: v[0] for k, v in c.items() if k not in ("FETCH_SIZE", "WRITE_SIZE")},
        "derived": {"mfma_busy_fraction": round(mfma / 4 / 256 / (gui / 8), 3) if gui else None,
                    "note": "MFMA busy cycles per SIMD (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) over kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs)"},
        "in_step_us": {"train_single_stream": tr, "sample_eager": sm,
                       "source": f"average over the full-size (> 40 us) launches of the kernel in the traced steps: {ntr} of {alltr} "
                                 f"launches (train), {nsm} of {allsm} (sample); the K = 128 / 512 launches of the same kernel are left out"},
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
