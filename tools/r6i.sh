#!/bin/bash
# round 6, GPU call I: the whole GPU suite on the final kernels + the round's evidence set (bench, traces, PMC passes)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 2700 python -m pytest tests/ -q -m gpu -s > $OUT/r6i_full_gpu_suite.txt 2>&1
grep "passed\|failed\|^FAILED\|^ERROR" $OUT/r6i_full_gpu_suite.txt | tail -8 | cut -c1-300
bash tools/collect_profiles.sh r6i > $OUT/r6i_collect.log 2>&1
tail -2 $OUT/r6i_collect.log | cut -c1-1200
