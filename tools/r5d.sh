#!/bin/bash
# round 5, GPU call D: pipelined sampler in the product path (tests + bench A/B), trained-gradient diagnosis, encoder kernels cold vs hot
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_full_walk.py -x -q -s -k "pipelined or two_chain or sampler_graphs or arbitrary or graph_replay or full_walk_small" > $OUT/r5d_tests.txt 2>&1
tail -3 $OUT/r5d_tests.txt
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r5d_bench_pipelined.json 2> $OUT/r5d_bench_pipelined.err
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sampler-unroll 0 > $OUT/r5d_bench_free_running.json 2> $OUT/r5d_bench_free_running.err
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r5d_bench_pipelined_b.json 2> $OUT/r5d_bench_pipelined_b.err
python $R/tools/trained_grad_diag.py > $OUT/r5d_trained_grad_diag.txt 2>&1
python $R/tools/encoder_cold_ab.py > $OUT/r5d_encoder_cold_ab.txt 2>&1
python $R/tools/attn_phases.py > $OUT/r5d_attn_phases_hot.txt 2>&1
SMD_COLD=1 python $R/tools/attn_phases.py > $OUT/r5d_attn_phases_cold.txt 2>&1
python $R/tools/mlp_hs_phases.py > $OUT/r5d_mlp_phases_hot.txt 2>&1
SMD_COLD=1 python $R/tools/mlp_hs_phases.py > $OUT/r5d_mlp_phases_cold.txt 2>&1
cat $OUT/r5d_encoder_cold_ab.txt
