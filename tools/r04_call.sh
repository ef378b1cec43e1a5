#!/bin/bash
. tools/gpu_steps.sh
O=gpurun_out/r04a; mkdir -p $O
step 900 $O/gpu_tests.log python -m pytest tests -x -q -m gpu
tail -3 $O/gpu_tests.log
step 300 $O/bench.log python bench.py
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04a/bench.log').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'])
print({k: v['us_per_step'] for k, v in d['kernels_us_per_step'].items()})
PY
cat gpurun_out/parity_relu_flips.json
