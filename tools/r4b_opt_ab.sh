set -x
python -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "optimizer or adam or one_sweep" -s 2>&1 | tail -25 > gpurun_out/r4b_tests.txt
for m in 0 1 3 0 3; do
  SMD_OPT_OVERLAP=$m python bench.py --mode train --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra-configs --no-sampler-walk --no-roofline-microbench 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap $m train', d['train_steps_per_sec'], d['block_values'], d['final_loss'])" >> gpurun_out/r4b_ab.txt
done
cat gpurun_out/r4b_tests.txt gpurun_out/r4b_ab.txt
