#!/bin/bash
# round 6, GPU call J: what the driver runs at round end -- smoke(), the default bench -- and the two-rank share-device bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r6j_bench.json 2> $OUT/r6j_bench.err
python -c "
import json; d=json.load(open('$OUT/r6j_bench.json'))
print({k: d[k] for k in ('value','train_steps_per_sec','sample_steps_per_sec','sample_T1000_wall_s','ms_per_step')}, d['roofline']['frac'], d['cpu_baseline']['value'])"
SMD_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --repeats 1 --no-extra-configs --no-sampler-walk --no-cpu-baseline --no-roofline-microbench 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 ranks on one device (test hook):', d['n_gpus'], d['config']['devices'], d['config']['dp'].get('distinct_devices'), d['data'][:40])"
python bench.py --gpus 2 --steps 5 2>&1 | tail -2 | cut -c1-200
