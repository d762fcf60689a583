#!/bin/bash
# round 5, GPU call C: unrolled / re-balanced pipelined sampler; trained-weights parity; autograd objective; DP dry run
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python $R/tools/chain_phase.py --tag c1 --reps 2 > $OUT/r5c_chain_phase_1.txt 2>&1
python $R/tools/chain_phase.py --tag c2 --reps 1 --dtype fp8 --modes plain,pipe2,pipe2:u4,pipe2:u4:s1,pipe2:u4:s1:m32,plain > $OUT/r5c_chain_phase_2_fp8.txt 2>&1
cd $R
timeout 1200 python -m pytest tests/test_gpu_full_walk.py -x -q -s -k trained > $OUT/r5c_trained_tests.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_bench_config.py -x -q -s -k "arbitrary_objective or dry_run or two_ranks or sampler_graphs or two_chain" > $OUT/r5c_new_tests.txt 2>&1
tail -3 $OUT/r5c_trained_tests.txt; tail -3 $OUT/r5c_new_tests.txt
