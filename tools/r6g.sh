#!/bin/bash
# round 6, GPU call G: static wave priority of the main stream's small-LDS kernels (SMD_MAIN_PRIO 3 vs the -DSMD_MAIN_PRIO=0 build),
# interleaved in one box; the two-stream train trace with it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
ab() {
  env $3 python $R/bench.py --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-roofline-microbench --no-extra-configs --no-sampler-walk $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'value', d['value'], 'train', d['train_steps_per_sec'], 'sample', d['sample_steps_per_sec'])"
}
for i in 1 2 3; do
  ab prio3 "" "SMD_X=1"
  ab prio0 "" "SMD_LIB_SUFFIX=_noprio"
done | tee $OUT/r6g_prio_ab.txt
for c in large; do
  ab prio3_large "--config large" "SMD_X=1"
  ab prio0_large "--config large" "SMD_LIB_SUFFIX=_noprio"
done | tee -a $OUT/r6g_prio_ab.txt
rocprofv3 --kernel-trace --stats -d $OUT/r6g_kt_train2 -o t -- python $R/bench.py --mode train --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-graph --no-roofline-microbench > /dev/null 2> $OUT/r6g_kt_train2.err
python $R/tools/stream_table.py $OUT/r6g_kt_train2/t_results.db > $OUT/r6g_train_main_stream.txt
python $R/tools/stream_busy.py $OUT/r6g_kt_train2/t_results.db --timeline 3 > $OUT/r6g_train_stream_busy.txt
rm -rf $OUT/r6g_kt_train2
head -40 $OUT/r6g_train_main_stream.txt
cd $R
timeout 900 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_kernels.py -q -m gpu -k "layernorm or repeat or determin or bitwise" 2>&1 | tail -3
