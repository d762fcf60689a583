#!/bin/bash
# round 5, GPU call E: MFMA filler probe; trained-weights parity (fresh / stationary batches); DP tests; the whole GPU suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
tools/mfma_filler_probe > $OUT/r5e_mfma_filler_probe.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_full_walk.py -x -q -s -k trained > $OUT/r5e_trained_tests.txt 2>&1
tail -3 $OUT/r5e_trained_tests.txt
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_full_walk.py::test_parity_on_trained_weights > $OUT/r5e_full_gpu_suite.txt 2>&1
tail -5 $OUT/r5e_full_gpu_suite.txt
