#!/bin/bash
# round 6, GPU call L: HBM-side bytes of the two north_star elementwise kernels (rocprofv3 PMC FETCH_SIZE / WRITE_SIZE, separate passes,
# corrected as MI355X_MICROARCH.md prescribes) next to their event-timed durations -> GB/s
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for P in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $P -d $OUT/r6l_pmc/$P -o p --output-format csv -- python $R/tools/reverse_step_probe.py > /dev/null 2> $OUT/r6l_pmc_$P.err
done
{ echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB per dispatch; FETCH_SIZE doubled per MI355X_MICROARCH.md) of tools/reverse_step_probe.py: B = 256 and 128,"
  echo "# Philox / explicit draws, with / without metrics averaged together per kernel; durations: profiles/r6e_reverse_step_probe.txt (HIP events)"
  python $R/tools/pmc_summary.py $OUT/r6l_pmc reverse_step
  python $R/tools/pmc_summary.py $OUT/r6l_pmc q_sample
} > $OUT/r6l_hbm_kernels_pmc.txt
rm -rf $OUT/r6l_pmc
cat $OUT/r6l_hbm_kernels_pmc.txt | cut -c1-200
python $R/tools/reverse_step_probe.py 2>&1 | grep -v amdgpu.ids
