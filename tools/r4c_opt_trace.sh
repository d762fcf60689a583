#!/bin/bash
# traces of the train step with the one-sweep optimiser: single-stream (kernel costs), two-stream with opt_overlap 0 and 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/r4c_kt1 -o t -- python $R/bench.py --mode train --side-wgrad 0 --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-graph --no-roofline-microbench > /dev/null 2> $OUT/r4c_kt1.err
python $R/tools/prof_summary.py $OUT/r4c_kt1/t_results.db 12 > $OUT/r4c_train_single_stream_kernel_trace.txt
for m in 0 3; do
SMD_OPT_OVERLAP=$m rocprofv3 --kernel-trace --stats -d $OUT/r4c_kt2_$m -o t -- python $R/bench.py --mode train --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-graph --no-roofline-microbench > /dev/null 2> $OUT/r4c_kt2_$m.err
python $R/tools/prof_summary.py $OUT/r4c_kt2_$m/t_results.db 12 > $OUT/r4c_train_two_stream_overlap${m}_kernel_trace.txt
python $R/tools/stream_busy.py $OUT/r4c_kt2_$m/t_results.db > $OUT/r4c_train_stream_busy_overlap${m}.txt
done
rm -rf $OUT/r4c_kt1 $OUT/r4c_kt2_0 $OUT/r4c_kt2_3
grep -E "adam|sumsq|recast|opt_prepare|dispatches" $OUT/r4c_train_*kernel_trace.txt
head -30 $OUT/r4c_train_stream_busy_overlap3.txt
