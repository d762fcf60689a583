#!/bin/bash
# Collects the evidence files of one round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*  (copy what should be judged into profiles/)
# 1. bench.py JSON line (default run incl. cpu_baseline)     2. rocprofv3 kernel-trace summary of the same command
# 3. PMC passes on the dominant kernel (separate runs: SQ cycles, MFMA busy, FETCH_SIZE, WRITE_SIZE)  4. kbench table
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 50 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o t -- python $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-extra-configs --no-sampler-walk --no-cpu-baseline --no-graph > /dev/null 2> $OUT/${TAG}_kt.err
python $R/tools/prof_summary.py $OUT/${TAG}_kt/t_results.db 12 > $OUT/${TAG}_bench_kernel_trace.txt
# per-mode traces with undistorted per-kernel times: train on ONE stream, sample step eager; the roofline microbench
# (104 extra launches of the dominant kernel) is left out of these so that calls/step and the shares are the step's own
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_train -o t -- python $R/bench.py --mode train --side-wgrad 0 --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-graph --no-roofline-microbench > /dev/null 2> $OUT/${TAG}_kt_train.err
python $R/tools/prof_summary.py $OUT/${TAG}_kt_train/t_results.db 12 > $OUT/${TAG}_train_single_stream_kernel_trace.txt
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_sample -o t -- python $R/bench.py --mode sample --steps 10 --warmup 2 --repeats 1 --no-sampler-walk --no-cpu-baseline --no-graph --no-roofline-microbench > /dev/null 2> $OUT/${TAG}_kt_sample.err
python $R/tools/prof_summary.py $OUT/${TAG}_kt_sample/t_results.db 12 > $OUT/${TAG}_sample_kernel_trace.txt
# the default two-stream train step without the microbench (critical-path analysis: tools/stream_busy.py)
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_train2 -o t -- python $R/bench.py --mode train --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-graph --no-roofline-microbench > /dev/null 2> $OUT/${TAG}_kt_train2.err
python $R/tools/prof_summary.py $OUT/${TAG}_kt_train2/t_results.db 12 > $OUT/${TAG}_train_two_stream_kernel_trace.txt
python $R/tools/stream_busy.py $OUT/${TAG}_kt_train2/t_results.db > $OUT/${TAG}_train_stream_busy.txt
cp $OUT/${TAG}_kt_train/t_results.db $OUT/${TAG}_train.db; cp $OUT/${TAG}_kt_sample/t_results.db $OUT/${TAG}_sample.db
rm -rf $OUT/${TAG}_kt_train $OUT/${TAG}_kt_sample $OUT/${TAG}_kt_train2
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $P | cut -d" " -f1)
  rocprofv3 --pmc $P -d $OUT/${TAG}_pmc/$N -o p --output-format csv -- python $R/tools/kbench.py --gemm-ab --variants 0 --reps 5 \
      --shapes 8192x2048x2048:b > /dev/null 2> $OUT/${TAG}_pmc_$N.err
done
python $R/tools/pmc_summary.py $OUT/${TAG}_pmc gemm_nt > $OUT/${TAG}_pmc_gemm.txt
# the file bench.py quotes (roofline.traffic / in_step_us): regenerated from THIS run's PMC passes and traces
python $R/tools/make_pmc_json.py $OUT/${TAG}_pmc_gemm.txt $OUT/${TAG}_train.db $OUT/${TAG}_sample.db $TAG > $OUT/${TAG}_pmc_gemm_nt256.json
rm -f $OUT/${TAG}_train.db $OUT/${TAG}_sample.db
python $R/tools/kbench.py --json $OUT/${TAG}_kbench.json > /dev/null 2>&1
python $R/tools/kbench.py --gemm-ab --variants 0 --shapes 8192x2048x2048:b,8192x2048x2048:r,8192x2048x128:g 2>/dev/null | grep gemm_ab > $OUT/${TAG}_gemm_ab.txt
python $R/tools/kbench.py --tn-ab 2>/dev/null | grep tn_ab >> $OUT/${TAG}_gemm_ab.txt
rm -rf $OUT/${TAG}_kt/*.db.tmp
cat $OUT/${TAG}_bench.json
