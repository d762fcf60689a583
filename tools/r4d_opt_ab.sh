#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --mode train --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra-configs --no-sampler-walk --no-roofline-microbench"
run() { # label, env overlap, extra args
  SMD_OPT_OVERLAP=$2 $B $3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['train_steps_per_sec'], d['block_values'])" >> $OUT/r4d_ab.txt
}
rm -f $OUT/r4d_ab.txt
run "overlap0" 0 ""
for g in 0 32 64 128 256 512; do run "overlap3 side_blocks=$g" 3 "--engine-opt opt_side_blocks=$g"; done
run "overlap0" 0 ""
run "overlap2 (early norm only)" 2 ""
for g in 64 128; do run "overlap1 side_blocks=$g" 1 "--engine-opt opt_side_blocks=$g"; done
for g in 0 64; do
SMD_OPT_OVERLAP=3 rocprofv3 --kernel-trace --stats -d $OUT/r4d_kt_$g -o t -- python $R/bench.py --mode train --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-graph --no-roofline-microbench --engine-opt opt_side_blocks=$g > /dev/null 2> $OUT/r4d_kt_$g.err
python $R/tools/stream_busy.py $OUT/r4d_kt_$g/t_results.db --timeline 5 > $OUT/r4d_stream_busy_side$g.txt
rm -rf $OUT/r4d_kt_$g
done
SMD_OPT_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $OUT/r4d_kt_o0 -o t -- python $R/bench.py --mode train --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-graph --no-roofline-microbench > /dev/null 2> $OUT/r4d_kt_o0.err
python $R/tools/stream_busy.py $OUT/r4d_kt_o0/t_results.db --timeline 5 > $OUT/r4d_stream_busy_overlap0.txt
rm -rf $OUT/r4d_kt_o0
cat $OUT/r4d_ab.txt
