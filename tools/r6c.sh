#!/bin/bash
# round 6, GPU call C: LayerNorm backwards inside the attention backward launch (kernel + engine tests, A/B bench), trajectory v2, plateau probe
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -s -k "attn_block_bwd" > $OUT/r6c_attn_bwd_ln_kernel_tests.txt 2>&1
grep "attn_block_bwd\|passed\|failed\|Error\|assert" $OUT/r6c_attn_bwd_ln_kernel_tests.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -s -k "layernorm_backwards_inside or gradient_parity or snapshots or emulating" > $OUT/r6c_engine_tests.txt 2>&1
grep "fused LayerNorm\|passed\|failed\|Error\|assert" $OUT/r6c_engine_tests.txt | cut -c1-300
cd /tmp; export TMPDIR=/tmp
for i in 1 2; do
for m in 2 1; do
  python $R/bench.py --mode train --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-roofline-microbench --engine-opt fused_attn_bwd=$m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused_attn_bwd=$m train steps/s', d['train_steps_per_sec'], d['block_values'])"
done; done | tee $OUT/r6c_fused_ln_ab.txt
rocprofv3 --kernel-trace --stats -d $OUT/r6c_kt_train2 -o t -- python $R/bench.py --mode train --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-graph --no-roofline-microbench > /dev/null 2> $OUT/r6c_kt_train2.err
python $R/tools/stream_table.py $OUT/r6c_kt_train2/t_results.db > $OUT/r6c_train_main_stream.txt
rm -rf $OUT/r6c_kt_train2
head -70 $OUT/r6c_train_main_stream.txt
cd $R
timeout 1800 python -m pytest tests/test_gpu_trajectory.py -x -q -s > $OUT/r6c_trajectory_tests.txt 2>&1
grep -v "Warning\|warn" $OUT/r6c_trajectory_tests.txt | grep "trajectory\|^  \|passed\|failed\|Error\|assert" | cut -c1-330 | head -80
timeout 1500 python -m pytest tests/test_gpu_full_walk.py -x -q -s -k "trained" > $OUT/r6c_trained_tests.txt 2>&1
grep "plateau\|passed\|failed\|Error\|assert" $OUT/r6c_trained_tests.txt | cut -c1-300
