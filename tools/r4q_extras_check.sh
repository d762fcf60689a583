#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
for ov in 1 2 3 0; do
SMD_OPT_OVERLAP=$ov python $R/bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline --no-sampler-walk --no-roofline-microbench 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap $ov base', d['train_steps_per_sec'], d['sample_steps_per_sec'], {k:(v['train_steps_per_sec'],v['sample_steps_per_sec']) for k,v in d['extra_configs'].items()})"
done
