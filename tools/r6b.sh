#!/bin/bash
# round 6, GPU call B: the parity additions (training trajectory, e4m3-emulating oracle, plateau state) + the main-stream table
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_trajectory.py -x -q -s > $OUT/r6b_trajectory_tests.txt 2>&1
grep -v Warning $OUT/r6b_trajectory_tests.txt | tail -60 | cut -c1-260
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -s -k "emulating" > $OUT/r6b_fp8_emulation_tests.txt 2>&1
grep "fp8 C=\|passed\|failed\|Error\|assert" $OUT/r6b_fp8_emulation_tests.txt | cut -c1-400
timeout 1500 python -m pytest tests/test_gpu_full_walk.py -x -q -s -k "trained" > $OUT/r6b_trained_tests.txt 2>&1
grep "plateau\|trained\|passed\|failed\|Error\|assert" $OUT/r6b_trained_tests.txt | cut -c1-400
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/r6b_kt_train2 -o t -- python $R/bench.py --mode train --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-graph --no-roofline-microbench > /dev/null 2> $OUT/r6b_kt_train2.err
python $R/tools/stream_table.py $OUT/r6b_kt_train2/t_results.db > $OUT/r6b_train_main_stream.txt
rm -rf $OUT/r6b_kt_train2
head -60 $OUT/r6b_train_main_stream.txt
