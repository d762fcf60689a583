#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for f in 3 4; do timeout 300 $R/tools/rsq_repro 5000 $f 2>&1 | grep -E "victim \["; done > $OUT/r4j_rsq_repro_noslp.txt
B="python $R/bench.py --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra-configs --no-sampler-walk --no-roofline-microbench"
rm -f $OUT/r4j_slp_ab.txt
for rep in 1 2; do for sfx in "" "_slp"; do
  SMD_LIB_SUFFIX=$sfx $B 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib$sfx', d['value'], 'train', d['train_steps_per_sec'], 'sample', d['sample_steps_per_sec'])" >> $OUT/r4j_slp_ab.txt
done; done
cat $OUT/r4j_rsq_repro_noslp.txt $OUT/r4j_slp_ab.txt
