#!/bin/bash
# round 6, GPU call F: reverse-step variants (prefetch, 8 row groups), the tests the last call failed, sampler tests, bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
python tools/reverse_step_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r6f_reverse_step_probe.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trajectory.py tests/test_gpu_bench_config.py tests/test_gpu_langevin.py -q -m gpu -s -k "attn_block_bwd or trajectory or two_shards or reverse_step or langevin or sampler" > $OUT/r6f_tests.txt 2>&1
grep "passed\|failed\|^FAILED\|^ERROR" $OUT/r6f_tests.txt | tail -8 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_full_walk.py -q -m gpu -k "sampler or walk or pipelined or two_chain or reference_default" > $OUT/r6f_sampler_tests.txt 2>&1
tail -3 $OUT/r6f_sampler_tests.txt | cut -c1-300
cd /tmp; export TMPDIR=/tmp
for rg in 8 4 8 4; do
python $R/bench.py --mode sample --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-roofline-microbench --no-extra-configs --no-sampler-walk --tuning reverse_rg=$rg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('reverse_rg=$rg sample', d['sample_steps_per_sec'], d['block_values'])"
done | tee $OUT/r6f_reverse_rg_ab.txt
python $R/bench.py --steps 20 --warmup 5 > $OUT/r6f_bench.json 2> $OUT/r6f_bench.err
python -c "
import json; d=json.load(open('$OUT/r6f_bench.json'))
print({k: d[k] for k in ('value','train_steps_per_sec','sample_steps_per_sec','sample_T1000_wall_s')}, d['roofline']['frac'])
print({k: v['value'] for k, v in d['extra_configs'].items()})"
