#!/bin/bash
# round 6, GPU call E: reverse-step / q_sample probe; the whole GPU suite (no -x)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
python tools/reverse_step_probe.py 2>&1 | tee $OUT/r6e_reverse_step_probe.txt
timeout 2700 python -m pytest tests/ -q -m gpu -s > $OUT/r6e_full_gpu_suite.txt 2>&1
grep "passed\|failed\|^FAILED\|^ERROR" $OUT/r6e_full_gpu_suite.txt | tail -15 | cut -c1-300
grep "philox normal\|\[trajectory\|per-step\|window means\|parameter distance\|held-out" $OUT/r6e_full_gpu_suite.txt | cut -c1-600
