#!/bin/bash
# round 5, GPU call A: the sampler's bimodality.  tools/r5a_bimodal.sh  -> gpurun_out/r5a_*
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python $R/tools/chain_phase.py --tag first --reps 1 > $OUT/r5a_chain_phase_1.txt 2>&1
$R/tools/cumask_probe > $OUT/r5a_cumask_probe.txt 2>&1
python $R/tools/chain_phase.py --tag second --reps 2 > $OUT/r5a_chain_phase_2.txt 2>&1
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $OUT/r5a_bench_default.json 2> $OUT/r5a_bench_default.err
rocprofv3 --kernel-trace -d $OUT/r5a_kt -o t -- python $R/bench.py --mode sample --steps 100 --warmup 3 --repeats 5 --no-cpu-baseline --no-extra-configs --no-sampler-walk --no-roofline-microbench > $OUT/r5a_kt_bench.json 2> $OUT/r5a_kt.err
python $R/tools/chain_timeline.py $OUT/r5a_kt/t_results.db 20 > $OUT/r5a_chain_timeline.txt 2>&1
rm -rf $OUT/r5a_kt
tail -3 $OUT/r5a_chain_phase_1.txt | cut -c1-300
