#!/bin/bash
# round 5, GPU call F: trained-weights parity incl. the emulating oracle; the round's evidence set
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_full_walk.py -x -q -s -k trained > $OUT/r5f_trained_tests.txt 2>&1
grep "trained\|passed\|failed" $OUT/r5f_trained_tests.txt | cut -c1-300 | head -30
bash tools/collect_profiles.sh r5f > $OUT/r5f_collect.log 2>&1
tail -2 $OUT/r5f_collect.log | cut -c1-600
