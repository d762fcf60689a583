#!/bin/bash
# round 6, GPU call A: the new host-side paths (B = 1000 / ragged sampler batches, aliasing autograd gradient), a baseline bench
# of the unchanged kernels, and the two-stream train trace for the main-stream critical-path table
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_full_walk.py -x -q -s -k "reference_default_batch" > $OUT/r6a_b1000_tests.txt 2>&1
tail -5 $OUT/r6a_b1000_tests.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -k "two_chain or pipelined or arbitrary_objective or cached_sampler or sample_api" > $OUT/r6a_engine_tests.txt 2>&1
tail -3 $OUT/r6a_engine_tests.txt | cut -c1-300
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $OUT/r6a_bench.json 2> $OUT/r6a_bench.err
cat $OUT/r6a_bench.json | cut -c1-1500
rocprofv3 --kernel-trace --stats -d $OUT/r6a_kt_train2 -o t -- python $R/bench.py --mode train --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-graph --no-roofline-microbench > /dev/null 2> $OUT/r6a_kt_train2.err
python $R/tools/stream_table.py $OUT/r6a_kt_train2/t_results.db > $OUT/r6a_train_main_stream.txt
python $R/tools/stream_busy.py $OUT/r6a_kt_train2/t_results.db --timeline 3 > $OUT/r6a_train_stream_busy.txt
rm -rf $OUT/r6a_kt_train2
head -50 $OUT/r6a_train_main_stream.txt
