#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for sfx in "" "_slp"; do
SMD_LIB_SUFFIX=$sfx rocprofv3 --kernel-trace --stats -d $OUT/r4k_kt$sfx -o t -- python $R/bench.py --mode sample --steps 10 --warmup 2 --repeats 1 --no-sampler-walk --no-cpu-baseline --no-graph --no-roofline-microbench > /dev/null 2> $OUT/r4k_kt$sfx.err
python $R/tools/prof_summary.py $OUT/r4k_kt$sfx/t_results.db 12 > $OUT/r4k_sample_trace$sfx.txt
SMD_LIB_SUFFIX=$sfx rocprofv3 --kernel-trace --stats -d $OUT/r4k_ktt$sfx -o t -- python $R/bench.py --mode train --side-wgrad 0 --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-graph --no-roofline-microbench > /dev/null 2> $OUT/r4k_ktt$sfx.err
python $R/tools/prof_summary.py $OUT/r4k_ktt$sfx/t_results.db 12 > $OUT/r4k_train_trace$sfx.txt
rm -rf $OUT/r4k_kt$sfx $OUT/r4k_ktt$sfx
done
