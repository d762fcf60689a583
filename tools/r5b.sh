#!/bin/bash
# round 5, GPU call B: pipelined two-chain sampler + the new full-walk / trained-weights parity tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python $R/tools/chain_phase.py --tag run1 --reps 2 > $OUT/r5b_chain_phase_1.txt 2>&1
python $R/tools/chain_phase.py --tag run2 --reps 1 --dtype fp8 --modes plain,pipe1,pipe2,plain > $OUT/r5b_chain_phase_2_fp8.txt 2>&1
cd $R
timeout 900 python -m pytest tests/test_gpu_full_walk.py -x -q -s > $OUT/r5b_full_walk_tests.txt 2>&1
tail -5 $OUT/r5b_full_walk_tests.txt
grep "^#" $OUT/r5b_chain_phase_1.txt
