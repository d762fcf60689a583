#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --mode train --steps 30 --warmup 5 --repeats 2 --no-cpu-baseline --no-extra-configs --no-sampler-walk --no-roofline-microbench"
for cfg in "large bf16" "large fp8" "base bf16"; do set -- $cfg
 for ov in 3 0; do
  SMD_OPT_OVERLAP=$ov $B --config $1 --dtype $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 overlap $ov train', d['train_steps_per_sec'], d['block_values'])"
 done
done
rocprofv3 --kernel-trace --stats -d $OUT/r4p_kt -o t -- python $R/bench.py --config large --mode train --side-wgrad 0 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-graph --no-roofline-microbench > /dev/null 2> $OUT/r4p_kt.err
python $R/tools/prof_summary.py $OUT/r4p_kt/t_results.db 8 > $OUT/r4p_large_train_trace.txt
rm -rf $OUT/r4p_kt
head -30 $OUT/r4p_large_train_trace.txt
