#!/bin/bash
# round 6, GPU call D: full GPU suite on the current tree (fused LayerNorm backwards, hardware sin/cos Box-Muller, quad q_sample,
# trajectory v3) + train-step A/Bs of scheduling options and the out_proj tile form
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests/ -x -q -m gpu -s > $OUT/r6d_full_gpu_suite.txt 2>&1
grep "passed\|failed\|Error" $OUT/r6d_full_gpu_suite.txt | tail -5 | cut -c1-300
grep "philox normal\|trajectory\|per-step\|window means\|parameter distance\|held-out" $OUT/r6d_full_gpu_suite.txt | cut -c1-420
cd /tmp; export TMPDIR=/tmp
ab() {  # name, extra args
  python $R/bench.py --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-roofline-microbench --no-extra-configs --no-sampler-walk $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'value', d['value'], 'train', d['train_steps_per_sec'], 'sample', d['sample_steps_per_sec'])"
}
for i in 1 2; do
  ab default ""
  ab wgrad256_group2 "--engine-opt wgrad256_group=2"
  ab pair_wgrad0 "--engine-opt pair_wgrad=0"
  ab form_wk2 "--tuning gemm_nt_form_wk=2"
  ab form_wk3 "--tuning gemm_nt_form_wk=3"
  ab form_wk5 "--tuning gemm_nt_form_wk=5"
  ab form_wk4 "--tuning gemm_nt_form_wk=4"
done | tee $OUT/r6d_ab.txt
python $R/tools/kbench.py --json $OUT/r6d_kbench.json > /dev/null 2>&1
python -c "
import json; d=json.load(open('$OUT/r6d_kbench.json'))
for k,v in d.items():
    if any(s in k for s in ('q_sample','reverse','mse','out_proj','attn','mlp_hs','ln_bwd','adam')): print(k, v)
" | cut -c1-200 | head -40
