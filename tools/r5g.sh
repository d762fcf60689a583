#!/bin/bash
# round 5, GPU call G: sampler defaults A/B in the bench itself (driver-like runs, interleaved, one box)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --mode sample --steps 20 --warmup 5 --repeats 7 --no-cpu-baseline --no-extra-configs --no-roofline-microbench"
for rep in 1 2; do
  for cfg in "s0m128:" "s1m32:--chain-opt sample_split=1 --chain-opt nt256_min_tiles=32" "s1m128:--chain-opt sample_split=1" "s0m32:--chain-opt nt256_min_tiles=32" "free:--sampler-unroll 0"; do
    name=${cfg%%:*}; opts=${cfg#*:}
    $B $opts 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name rep$rep', d['sample_steps_per_sec'], d['block_values'], 'walk', d['sampler_walk']['steps_per_sec'] if d.get('sampler_walk') else None)" >> $OUT/r5g_sampler_defaults_ab.txt
  done
done
cat $OUT/r5g_sampler_defaults_ab.txt
