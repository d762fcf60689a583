#!/bin/bash
# PMC over a whole (train + sample) step, per kernel (two counter groups = two passes): LDS bank conflicts and MFMA busy.
#   tools/pmc_whole_step.sh <tag>  ->  gpurun_out/<tag>_pmc_whole_step.txt
set -u
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --mode both --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extra-configs --no-sampler-walk --no-roofline-microbench --no-graph"
i=0
for P in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $P -d $OUT/${TAG}_pmcw/$i -o p --output-format csv -- $CMD > /dev/null 2> $OUT/${TAG}_pmcw_$i.err
done
python - "$OUT/${TAG}_pmcw" > $OUT/${TAG}_pmc_whole_step.txt <<'PY'
import collections, csv, glob, os, sys
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
print("# rocprofv3 --pmc (two passes) -- python bench.py --mode both --steps 3 --warmup 1 --repeats 1 --no-graph ... ; per-dispatch means")
print("# conflict/active = LDS bank-conflict cycles / LDS-active cycles; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)")
print(f"{'conflict/active':>15} {'mfma_busy':>9} {'lds_active/disp':>15} {'wave_cycles/disp':>16} {'n':>4}  kernel")
rows = []
for k, c in acc.items():
    m = {n: sum(v.values()) / max(len(v), 1) for n, v in c.items()}
    n = max(len(v) for v in c.values())
    la, lc = m.get("SQ_LDS_IDX_ACTIVE", 0), m.get("SQ_LDS_BANK_CONFLICT", 0)
    gui, mf = m.get("GRBM_GUI_ACTIVE", 0), m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    rows.append((m.get("SQ_WAVE_CYCLES", 0) * n, lc / la if la else 0.0, mf / 1024 / (gui / 8) if gui else 0.0, la, m.get("SQ_WAVE_CYCLES", 0), n, k))
for _, ca, mb, la, wc, n, k in sorted(rows, reverse=True)[:28]:
    print(f"{ca:15.3f} {mb:9.3f} {la:15.0f} {wc:16.0f} {n:4d}  {k.replace('(anonymous namespace)::', '')[:90]}")
PY
cat $OUT/${TAG}_pmc_whole_step.txt
