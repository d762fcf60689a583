#!/bin/bash
# round 5, GPU call I: final validation of the round's library -- the whole GPU suite, smoke(), the bench as the driver runs it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/r5i_build.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu -x --durations=15 > $OUT/r5i_full_gpu_suite.txt 2>&1
tail -22 $OUT/r5i_full_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r5i_smoke.txt 2>&1; tail -3 $OUT/r5i_smoke.txt
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r5i_bench.json 2> $OUT/r5i_bench.err
tail -c 600 $OUT/r5i_bench.err
